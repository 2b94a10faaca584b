"""N > 1 path on CPU: world_size-2 gloo processes exercise the frame / window sharding and the collectives bench.py
uses (no GPU needed)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from eao_fusion_amd import shard


def test_shards_partition_frames():
    for n, world in [(512, 8), (64, 2), (10, 4), (3, 8), (1, 1)]:
        seen = []
        for r in range(world):
            lo, hi = shard.frame_shard(n, r, world)
            seen += list(range(lo, hi))
            for f in range(lo, hi):
                assert shard.frame_owner(f, n, world) == r
        assert seen == list(range(n))
    assert shard.halo_frame(0, 512, 8) is None and shard.halo_frame(3, 512, 8) == 191
    assert [shard.window_owner(w, 8) for w in range(10)] == [0, 1, 2, 3, 4, 5, 6, 7, 0, 1]


def _worker(rank, world, port, n_frames, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard.frame_shard(n_frames, rank, world)
    local = [1000 + (f * 7) % 13 for f in range(lo, hi)]          # stand-in for per-frame keypoint counts
    allc = shard.gather_frame_counts(local, n_frames)
    thr, units, secs = shard.aggregate_throughput(sum(local), 0.5 + 0.25 * rank)
    # halo exchange of the consecutive-frame matcher: each rank's last frame goes to the next rank
    last = torch.full((5, 32), 10 * rank + 1, dtype=torch.uint8)
    hd, hn = shard.exchange_halo(last, 3 + rank)
    halo = None if hd is None else (int(hd[0, 0]), hn)
    dist.barrier()
    q.put((rank, allc.tolist(), thr, units, secs, halo, shard.window_shard(25, rank, world)))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [9, 64])
def test_two_rank_gather_and_throughput(n_frames):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = [1000 + (f * 7) % 13 for f in range(n_frames)]
    for rank, allc, thr, units, secs, halo, windows in res:
        assert allc == expect                      # every rank sees every frame, in global order
        assert units == sum(expect) and secs == 0.75   # SUM of units, MAX of time
        assert thr == pytest.approx(sum(expect) / 0.75)
        assert halo == (None if rank == 0 else (1, 3))  # rank 1 receives rank 0's last frame (value 1, 3 keypoints)
        assert windows == list(range(rank, 25, 2))


def _payload_worker(rank, world, port, n_frames, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cap = 11
    lo, hi = shard.frame_shard(n_frames, rank, world)
    kps = torch.stack([_rec(f, cap)[0] for f in range(lo, hi)])
    desc = torch.stack([_rec(f, cap)[1] for f in range(lo, hi)])
    cnt = torch.tensor([_rec(f, cap)[2] for f in range(lo, hi)], dtype=torch.int32)
    K, D, Cn = shard.gather_frame_results(kps, desc, cnt, n_frames)          # (keypoints, descriptors) of every frame
    hk, hd, hn = shard.exchange_halo_frame(kps[-1], desc[-1], int(cnt[-1]))
    ws = shard.window_shard(7, rank, world)                                   # per-window BA results
    cams = torch.stack([_win(w)[0] for w in ws])
    pts = torch.stack([_win(w)[1] for w in ws])
    AC, AP = shard.gather_window_results(cams, pts, 7)
    dist.barrier()
    q.put((rank, K.numpy(), D.numpy(), Cn.numpy(), None if hk is None else (hk.numpy(), hd.numpy(), hn), AC.numpy(), AP.numpy()))
    dist.destroy_process_group()


def _rec(f, cap):
    g = torch.Generator().manual_seed(100 + f)
    return (torch.randint(0, 256, (cap, 28), dtype=torch.uint8, generator=g), torch.randint(0, 256, (cap, 32), dtype=torch.uint8, generator=g), 1 + f % cap)


def _win(w):
    g = torch.Generator().manual_seed(900 + w)
    return torch.rand((5, 16), generator=g), torch.rand((13, 3), generator=g)


@pytest.mark.parametrize("n_frames", [7, 16])
def test_two_rank_payload_allgather(n_frames):
    """Every rank ends with every frame's keypoints + descriptors and every window's poses + points, byte for byte."""
    import numpy as np
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_payload_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cap = 11
    for rank, K, D, Cn, halo, AC, AP in res:
        assert K.shape == (n_frames, cap, 28) and D.shape == (n_frames, cap, 32)
        for f in range(n_frames):
            ek, ed, en = _rec(f, cap)
            assert np.array_equal(K[f], ek.numpy()) and np.array_equal(D[f], ed.numpy()) and Cn[f] == en
        if rank == 0:
            assert halo is None
        else:
            lo, _ = shard.frame_shard(n_frames, rank, 2)
            ek, ed, en = _rec(lo - 1, cap)
            assert np.array_equal(halo[0], ek.numpy()) and np.array_equal(halo[1], ed.numpy()) and halo[2] == en
        for w in range(7):
            ec, ep = _win(w)
            assert np.array_equal(AC[w], ec.numpy()) and np.array_equal(AP[w], ep.numpy())


def test_bench_entry_command_two_ranks_dry_run():
    """`python bench.py --gpus 2` as the driver types it: the parent starts torch.distributed.run as a child (it never
    touches HIP itself), both ranks come up, run the collectives of the batched-sequence config on gloo and rank 0 prints
    ONE JSON line.  --dry-run keeps it on the CPU (stand-in payloads, nothing measured)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--batch", "4"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dry_run"] is True and out["collectives_ok"] is True and out["frames"] == 8
    # a world size that does not match --gpus is refused with a message, not an assert
    env = dict(os.environ, RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr
