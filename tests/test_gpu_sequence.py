"""BASELINE configs[4]: the batched 512-frame sequence (ORB extraction + consecutive-frame matching + 25 local-BA windows)
on ONE GPU, run as the eight 64-frame shards an 8-GPU job would own -- through the same shard code (eao_fusion_amd/shard.py,
sequence.py), with the halo frame handed from shard to shard -- and as one unsharded run.  Sharding must not change a bit;
ALL 512 frames, ALL 511 pairs (the seven shard-boundary pairs included) and ALL 25 windows are checked against the CPU oracle."""
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


def _oracle_extract_all(oracle, frames, workers=8):
    """The CPU oracle over every frame (21 ms each, one OrbOracle per worker thread; ctypes calls release the GIL)."""
    import threading
    from concurrent.futures import ThreadPoolExecutor
    tl = threading.local()

    def one(f):
        if not hasattr(tl, "orc"):
            tl.orc = oracle.OrbOracle(1000, 1.2, 8, 20, 7)
        return tl.orc.extract(frames[f])
    with ThreadPoolExecutor(workers) as ex:
        return list(ex.map(one, range(len(frames))))


def test_full_parity_against_the_oracle(sequence, oracle):
    """configs[4] in full, not as spots -- ALL 512 frames (keypoints + descriptors bit for bit), ALL 511
    consecutive pairs (best / second-best tables, the seven shard-boundary pairs through the halo among them) and ALL 25
    local-BA windows (iteration counts, outlier tables, updates within 1e-4) against the CPU oracle."""
    frames, shards, whole, S = sequence
    from concurrent.futures import ThreadPoolExecutor
    from eao_fusion_amd.orb import KP_DTYPE
    ref = _oracle_extract_all(oracle, frames)
    for f in range(N_FRAMES):
        r, i = divmod(f, 64)
        n = int(shards[r]["n"][i])
        okps, odesc = ref[f]
        assert n == len(okps), "keypoint count of frame %d" % f
        assert np.array_equal(shards[r]["kps"][i, :n].copy().view(KP_DTYPE).reshape(-1), okps), "keypoints of frame %d" % f
        assert np.array_equal(shards[r]["desc"][i, :n], odesc), "descriptors of frame %d" % f
    with ThreadPoolExecutor(8) as ex:
        wants = list(ex.map(lambda f: oracle.hamming_best2(ref[f - 1][1], ref[f][1]), range(1, N_FRAMES)))
    for f in range(1, N_FRAMES):
        r, i = divmod(f, 64)
        na = len(ref[f - 1][1])
        assert np.array_equal(shards[r]["match"][i, :na], wants[f - 1]), "pair (%d, %d)%s" % (f - 1, f, " -- across a shard boundary" if i == 0 else "")
        assert (shards[r]["match"][i, na:] == -1).all()
    probs = [S.window_problem(w) for w in range(S.N_WINDOWS)]
    with ThreadPoolExecutor(8) as ex:
        orcs = list(ex.map(oracle.local_ba, probs))
    for w in range(S.N_WINDOWS):
        p, o = probs[w], orcs[w]
        r = shard.window_owner(w, WORLD)
        k = shard.window_shard(S.N_WINDOWS, r, WORLD).index(w)
        res = shards[r]["ba_results"][k]
        assert list(res["iters"]) == list(o["iters"]) and np.array_equal(res["edge_outlier"], o["edge_outlier"]), "window %d" % w
        for name, new_g, new_c, old in (("poses", res["poses"], o["poses"], p["poses"]), ("points", res["points"], o["points"], p["points"])):
            upd = np.abs(new_c.astype(np.float64) - old.astype(np.float64)).max()
            err = np.abs(new_g.astype(np.float64) - new_c.astype(np.float64)).max()
            assert err <= 1e-4 * max(upd, 1e-6) + 2 * np.spacing(np.abs(new_c).max().astype(np.float32)), "%s of window %d" % (name, w)
