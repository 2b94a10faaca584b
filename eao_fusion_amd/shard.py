"""Frame / BA-window sharding across the GPUs of one node.

The hot path has no data-path collective: frames are independent units (contiguous shards, so the consecutive-frame
matcher needs one halo frame per shard boundary) and BA windows are independent problems (round robin).  The only
collectives are the timing barrier / max-reduce and one all_gather of per-frame results at the end, over
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests)."""
import torch
import torch.distributed as dist


def frame_shard(n_frames, rank, world):
    """Contiguous shard [lo, hi) of frame indices owned by `rank`; the first n_frames % world ranks get one more."""
    base, rem = divmod(n_frames, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def frame_owner(f, n_frames, world):
    for r in range(world):
        lo, hi = frame_shard(n_frames, r, world)
        if lo <= f < hi:
            return r
    raise IndexError(f)


def halo_frame(rank, n_frames, world):
    """Index of the frame the matcher of `rank` needs from its left neighbour (pair (f-1, f) is owned by f's shard)."""
    lo, hi = frame_shard(n_frames, rank, world)
    return lo - 1 if lo > 0 and hi > lo else None


def window_owner(w, world):
    return w % world


def gather_frame_counts(local_counts, n_frames, device=None):
    """all_gather of the per-frame keypoint counts of every shard (shards may differ by one frame => padded).
    Returns a tensor of n_frames counts in global frame order, identical on every rank."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    local_counts = torch.as_tensor(local_counts, dtype=torch.int32, device=device)
    if world == 1:
        return local_counts.clone()
    cap = (n_frames + world - 1) // world
    pad = torch.full((cap,), -1, dtype=torch.int32, device=local_counts.device)
    pad[:local_counts.numel()] = local_counts
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    out = []
    for r in range(world):
        lo, hi = frame_shard(n_frames, r, world)
        out.append(parts[r][:hi - lo])
    assert rank < world
    return torch.cat(out)


def aggregate_throughput(units_local, seconds_local, device=None):
    """Whole-job throughput: all ranks' units / the slowest rank's time (bench.py contract)."""
    u = torch.tensor([float(units_local)], dtype=torch.float64, device=device)
    t = torch.tensor([float(seconds_local)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(u.item()) / float(t.item()), float(u.item()), float(t.item())


def window_shard(n_windows, rank, world):
    """BA windows are independent problems: window w belongs to rank w mod world."""
    return list(range(rank, n_windows, world))


def exchange_halo(last_desc, last_count, device=None):
    """The consecutive-frame matcher of a shard needs the last frame of the previous shard: every rank contributes the
    descriptors (cap, 32) u8 and keypoint count of its LAST frame to one all_gather and picks the entry of rank - 1.
    Returns (desc, count) of the halo frame, or (None, 0) on rank 0 / a single rank."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if world == 1:
        return None, 0
    cnt = torch.tensor([int(last_count)], dtype=torch.int32, device=device)
    descs = [torch.empty_like(last_desc) for _ in range(world)]
    cnts = [torch.empty_like(cnt) for _ in range(world)]
    dist.all_gather(descs, last_desc.contiguous())
    dist.all_gather(cnts, cnt)
    if rank == 0:
        return None, 0
    return descs[rank - 1], int(cnts[rank - 1].item())
