"""BASELINE configs[4]: the batched 512-frame sequence (ORB extraction + consecutive-frame matching + 25 local-BA windows)
on ONE GPU, run as the eight 64-frame shards an 8-GPU job would own -- through the same shard code (eao_fusion_amd/shard.py,
sequence.py), with the halo frame handed from shard to shard -- and as one unsharded run.  Sharding must not change a bit;
frames {0, 63, 64, 511}, the shard-boundary pair (63, 64) and windows {0, 24} are checked against the CPU oracle."""
import numpy as np
import pytest
import torch

from eao_fusion_amd import shard

pytestmark = pytest.mark.gpu
N_FRAMES, WORLD = 512, 8


@pytest.fixture(scope="module")
def sequence():
    import eao_fusion_amd as E
    assert E.load().eao_device_check() == 0, E.load().eao_last_error()
    from eao_fusion_amd import sequence as S
    frames = np.stack([S.sequence_frame(f) for f in range(N_FRAMES)])
    # ---- eight shards, one after the other on this GPU
    shards, halo = [], None
    for r in range(WORLD):
        lo, hi = shard.frame_shard(N_FRAMES, r, WORLD)
        assert shard.halo_frame(r, N_FRAMES, WORLD) == (lo - 1 if r else None)
        out = S.run_shard(frames[lo:hi], lo, r, WORLD, halo=halo)
        k, d, n = out["seq"].last_frame()
        halo = (d.clone(), n)
        shards.append({k2: (v.cpu().numpy() if torch.is_tensor(v) else v) for k2, v in out.items() if k2 != "seq"})
        del out
    # ---- the whole sequence at once
    whole = S.run_shard(frames, 0, 0, 1)
    whole = {k2: (v.cpu().numpy() if torch.is_tensor(v) else v) for k2, v in whole.items() if k2 != "seq"}
    return frames, shards, whole, S


def test_sharding_changes_nothing(sequence):
    frames, shards, whole, S = sequence
    for r, sh in enumerate(shards):
        lo, hi = shard.frame_shard(N_FRAMES, r, WORLD)
        assert np.array_equal(sh["n"], whole["n"][lo:hi])
        assert np.array_equal(sh["kps"], whole["kps"][lo:hi]) and np.array_equal(sh["desc"], whole["desc"][lo:hi])
        # pair (lo - 1, lo) went through the halo: identical to the unsharded pair; shard 0 has no pair 0
        first = 0 if r else 1
        assert np.array_equal(sh["match"][first:], whole["match"][lo + first:hi]), "match tables of shard %d" % r
        if r == 0:
            assert (sh["match"][0] == -1).all()
        ws = shard.window_shard(S.N_WINDOWS, r, WORLD)
        assert np.array_equal(sh["ba_cams"], whole["ba_cams"][ws]) and np.array_equal(sh["ba_points"], whole["ba_points"][ws])
    assert whole["n"].min() >= 900 and int(whole["n"].sum()) > 500 * N_FRAMES


def test_sequence_is_deterministic(sequence):
    frames, shards, whole, S = sequence
    lo, hi = shard.frame_shard(N_FRAMES, 3, WORLD)
    halo_src = S.run_shard(frames[lo - 64:lo], lo - 64, 2, WORLD)
    _, d, n = halo_src["seq"].last_frame()
    again = S.run_shard(frames[lo:hi], lo, 3, WORLD, halo=(d.clone(), n))
    for k in ("n", "kps", "desc", "match"):
        assert np.array_equal(again[k].cpu().numpy(), shards[3][k]), k
    assert np.array_equal(again["ba_cams"], shards[3]["ba_cams"]) and np.array_equal(again["ba_points"], shards[3]["ba_points"])


def test_spot_parity_against_the_oracle(sequence, oracle):
    frames, shards, whole, S = sequence
    from eao_fusion_amd.orb import KP_DTYPE
    orc = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
    ref = {}
    for f in (0, 63, 64, 510, 511):
        ref[f] = orc.extract(frames[f])
    for f in (0, 63, 64, 511):
        r, i = divmod(f, 64)
        n = int(shards[r]["n"][i])
        okps, odesc = ref[f]
        assert n == len(okps)
        assert np.array_equal(shards[r]["kps"][i, :n].copy().view(KP_DTYPE).reshape(-1), okps), "keypoints of frame %d" % f
        assert np.array_equal(shards[r]["desc"][i, :n], odesc), "descriptors of frame %d" % f
    # pairs (63, 64) -- across the shard boundary, through the halo -- and (510, 511)
    for f in (64, 511):
        r, i = divmod(f, 64)
        want = oracle.hamming_best2(ref[f - 1][1], ref[f][1])
        na = len(ref[f - 1][1])
        got = shards[r]["match"][i, :na]
        assert np.array_equal(got, want), "pair (%d, %d)" % (f - 1, f)
        assert (shards[r]["match"][i, na:] == -1).all()
    # windows 0 and 24 within 1e-4 of the oracle's update (BASELINE north_star)
    for w in (0, 24):
        p = S.window_problem(w)
        o = oracle.local_ba(p)
        r = shard.window_owner(w, WORLD)
        k = shard.window_shard(S.N_WINDOWS, r, WORLD).index(w)
        res = shards[r]["ba_results"][k]
        assert list(res["iters"]) == list(o["iters"]) and np.array_equal(res["edge_outlier"], o["edge_outlier"])
        for name, new_g, new_c, old in (("poses", res["poses"], o["poses"], p["poses"]), ("points", res["points"], o["points"], p["points"])):
            upd = np.abs(new_c.astype(np.float64) - old.astype(np.float64)).max()
            err = np.abs(new_g.astype(np.float64) - new_c.astype(np.float64)).max()
            assert err <= 1e-4 * max(upd, 1e-6) + 2 * np.spacing(np.abs(new_c).max().astype(np.float32)), "%s of window %d" % (name, w)
