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


def _world():
    if dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


def _all_gather_rows(local, rows_cap):
    """all_gather of a [rows, ...] tensor whose row count may differ by one between ranks: rows are padded to `rows_cap`,
    ONE collective moves everything (all_gather_into_tensor: RCCL runs it as a ring over the xGMI links), and the
    caller cuts the padding off again.  Returns a [world, rows_cap, ...] tensor."""
    world, _ = _world()
    pad = torch.zeros((rows_cap,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    out = torch.empty((world,) + tuple(pad.shape), dtype=local.dtype, device=local.device)
    if world == 1:
        out[0] = pad
    else:
        dist.all_gather_into_tensor(out.view(-1), pad.view(-1))
    return out


def gather_frame_results(kps, desc, counts, n_frames):
    """at the end of the batched sequence every rank holds every frame's result.  kps [B, cap, 28] u8
    (cv::KeyPoint records), desc [B, cap, 32] u8, counts [B] i32 of THIS rank's contiguous frame shard.  One all_gather of
    the 60-byte (keypoint, descriptor) records (3.9 MB per rank for 64 frames x 1011 slots) and one of the counts.
    Returns (kps [n_frames, cap, 28], desc [n_frames, cap, 32], counts [n_frames]) in global frame order, identical on
    every rank."""
    world, _ = _world()
    B, cap = int(kps.shape[0]), int(kps.shape[1])
    rows = (n_frames + world - 1) // world
    rec = torch.cat([kps.reshape(B, cap, 28), desc.reshape(B, cap, 32)], dim=2)      # [B, cap, 60]
    allrec = _all_gather_rows(rec, rows)
    allcnt = _all_gather_rows(counts.to(torch.int32).reshape(B, 1), rows)
    ks, ds, cs = [], [], []
    for r in range(world):
        lo, hi = frame_shard(n_frames, r, world)
        ks.append(allrec[r, :hi - lo, :, :28])
        ds.append(allrec[r, :hi - lo, :, 28:])
        cs.append(allcnt[r, :hi - lo, 0])
    return torch.cat(ks).contiguous(), torch.cat(ds).contiguous(), torch.cat(cs).contiguous()


def gather_window_results(cams, points, n_windows):
    """the optimised keyframe poses and map points of every local-BA window on every rank.  cams
    [Wl, n_cams, 16] f32 and points [Wl, n_points, 3] f32 are the results of THIS rank's windows (window w belongs to rank
    w mod world, in ascending order).  Returns (cams [n_windows, n_cams, 16], points [n_windows, n_points, 3]) in window
    order, identical on every rank."""
    world, _ = _world()
    rows = (n_windows + world - 1) // world
    Wl = int(cams.shape[0])
    nc, npt = int(cams.shape[1]) * 16, int(points.shape[1]) * 3
    flat = torch.cat([cams.reshape(Wl, nc), points.reshape(Wl, npt)], dim=1)
    allw = _all_gather_rows(flat, rows)
    oc = torch.empty((n_windows, cams.shape[1], 16), dtype=cams.dtype, device=cams.device)
    op = torch.empty((n_windows, points.shape[1], 3), dtype=points.dtype, device=points.device)
    for r in range(world):
        ws = window_shard(n_windows, r, world)
        for i, w in enumerate(ws):
            oc[w] = allw[r, i, :nc].reshape(-1, 16)
            op[w] = allw[r, i, nc:].reshape(-1, 3)
    return oc, op


def exchange_halo_frame(last_kps, last_desc, last_count):
    """Halo of the consecutive-frame matcher with the keypoints as well (the guided matcher needs positions): every
    rank contributes its LAST frame's (cap, 28) + (cap, 32) records; rank r > 0 keeps those of rank r - 1."""
    world, rank = _world()
    if world == 1:
        return None, None, 0
    cap = int(last_kps.shape[0])
    rec = torch.cat([last_kps.reshape(1, cap, 28), last_desc.reshape(1, cap, 32)], dim=2)
    allrec = _all_gather_rows(rec, 1)
    cnt = torch.tensor([[int(last_count)]], dtype=torch.int32, device=last_kps.device)
    allcnt = _all_gather_rows(cnt, 1)
    if rank == 0:
        return None, None, 0
    return allrec[rank - 1, 0, :, :28].contiguous(), allrec[rank - 1, 0, :, 28:].contiguous(), int(allcnt[rank - 1, 0, 0].item())
