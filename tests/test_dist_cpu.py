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
