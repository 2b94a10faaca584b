#!/usr/bin/env python3
"""bench.py -- headline benchmark of the ORB front-end + local-BA hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A "step" = one pass of the hot path over one batch of synthetic input, inputs already resident in HBM:
  * ORB extraction of BASELINE.json configs[1]: 64 synthetic 640x480 frames, nFeatures 1000, 8 levels   -> `value`
  * inside the same timed step, on the same stream: Hamming best-2 matching of consecutive frames'
    descriptor sets is NOT included in `value` (reported under "extra").
Local BA (configs[3], 20 KF x 3000 MP) and Hamming (configs[2]) are measured after the timed region and reported
under "extra" -- the combined BASELINE metric has two halves; `value` is its first half (ORB kpts/s), the BA half
is extra.ba (one window, latency at the C-ABI) and extra.ba_batch (25 windows per call, residual blocks/s).
`python bench.py --gpus N` (N > 1) starts its own torch.distributed.run child; extra.sequence is BASELINE configs[4].

The default batch is the 64 frames configs[1] names; the pyramid chain and the quad-tree are latency-bound, so larger batches
amortise them -- `extra.orb_batch256` carries the 256-frame figure of the same call.

Multi-GPU: frames are independent units => each rank extracts its own B-frame shard (--batch, default 64) (weak scaling, no data-path
collective); the only collectives are the timing barrier/max and one all_gather of the per-frame keypoint counts
(RCCL), which is outside the timed region.

One JSON line is printed by rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SETTLE_STEPS = 50       # untimed steps before the W warm-up steps (clock settling; see main())
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
VALU_PEAK_WAVE_INSTS = 1024 * 2.4e9 / 4   # wave instructions / s: 256 CUs x 4 SIMDs, one (integer / packed) VALU instruction per 4 cycles
                                          # (tools/ubench/intops.hip; its output at 1 / 2 / 4 / 8 waves per SIMD: profiles/r03_ubench_intops.txt)


def _source_sha(rel):
    """sha256 (first 16 hex digits) of a kernel source file: what a committed counter file was captured from (a kernel edited after the
    PMC pass would otherwise keep feeding stale instruction / byte counts into fractions that look live)."""
    import hashlib
    try:
        return hashlib.sha256(open(os.path.join(ROOT, rel), "rb").read()).hexdigest()[:16]
    except OSError:
        return None


LM_SOURCES = ("eao_fusion_amd/csrc/lm_internal.h", "eao_fusion_amd/csrc/lba.hip", "eao_fusion_amd/csrc/gba.hip")      # (round 6: csrc/lm.hip split; the KERNEL sources -- lm_host.hip is host code + the PCIe copy kernel)


ORB_SOURCES = ("eao_fusion_amd/csrc/orb.hip", "eao_fusion_amd/csrc/orb_internal.h")      # (round 6: csrc/orb.hip split; the pyramid / FAST / blur / description kernels and the records they read)


def _fresh(pmc, rel):
    """True when the committed counter file names the hash of today's kernel source (files older than round 4 carry none: stale by definition here)."""
    return bool(pmc) and pmc.get("source_sha16", {}).get(rel) == _source_sha(rel)


def _profile(name):
    """The newest committed profile file of that name (profiles/r05_<name>, else r04_ / r03_ / r02_<name>): recorded figures the line quotes."""
    for tag in ("r06", "r05", "r04", "r03", "r02"):
        f = os.path.join(ROOT, "profiles", "%s_%s" % (tag, name))
        if os.path.exists(f):
            return f
    return None


def _kernel_avg_us(csv_name, kernel_prefix):
    """avg_us of the first row whose kernel starts with kernel_prefix in a committed rocprofv3 summary (tools/summarize_rocprof.py)."""
    f = _profile(csv_name)
    if not f:
        return None, None
    for ln in open(f):
        if ln.startswith(kernel_prefix):
            c = ln.strip().split(",")
            # (template arguments contain commas: the numeric columns are the last six)
            return float(c[-5]), os.path.relpath(f, ROOT)
    return None, os.path.relpath(f, ROOT)

# algorithmic bytes per frame of each stage at 640x480 / 8 levels (DESIGN.md "roofline accounting")
LEVELS = [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)]
PX = [w * h for w, h in LEVELS]
BYTES_PYRAMID = sum(PX[:-1]) + sum(PX[1:])      # read levels 0..6, write levels 1..7
BYTES_FAST = sum(PX)                            # every level read once (+ 4 B per candidate, added at run time)
BYTES_BLUR = 2 * sum(PX)                        # read + write every level
BYTES_PER_KP = 749 + 512 + 32 + 28              # IC disc + BRIEF samples + descriptor + keypoint record


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)     # (a step is 0.28 ms: the GPU's clocks need ~15 ms of work to settle --
    ap.add_argument("--warmup", type=int, default=50)     #  3 warm-up steps read 0.299 ms per step, 10 read 0.290, 50 read 0.275)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--dry-run", action="store_true",
                    help="rehearse the N-rank launch and the collectives on the CPU (gloo, stand-in payloads, no HIP call, no timing)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` as typed: start one rank per GPU under torch.distributed.run as a CHILD process and pass its
        # output and return code through.  Nothing in this process has touched HIP or torch.cuda yet (and it never execs).
        sys.exit(self_launch(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (start it as `python bench.py --gpus N`, or under torch.distributed.run "
                 "--nproc-per-node N)" % (args.gpus, world))
    if args.dry_run:
        return dry_run(args, rank, world)

    import torch
    import torch.distributed as dist

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=300))      # a rank that never arrives fails the job instead of hanging it
        # one line per rank on stderr: a first real N-rank run that dies in set-up says where
        try:
            rccl = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception:  # noqa: BLE001
            rccl = "?"
        print("[bench rank %d/%d] local_rank %d device %s (%s), RCCL %s, MASTER %s:%s" % (rank, world, local_rank, dev, torch.cuda.get_device_name(dev), rccl,
              os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT")), file=sys.stderr, flush=True)

    import eao_fusion_amd as E  # after torch: one libamdhip64 in the process
    from eao_fusion_amd import sequence, shard, synth

    B, W, H = args.batch, 640, 480
    n_frames = B * world                                   # weak scaling: B frames per GPU
    lo, hi = shard.frame_shard(n_frames, rank, world)      # contiguous shard of the sequence owned by this rank
    assert hi - lo == B
    frames = np.stack([synth.synth_frame(1000 + f, W, H) for f in range(lo, hi)])
    d_img = torch.from_numpy(frames).to(dev)
    seq = sequence.SequenceShard(B, W, H, dev)             # the extractor + its device-resident outputs (also used by tests/test_gpu_sequence.py)
    ext, cap, d_kps, d_desc, d_n = seq.ext, seq.cap, seq.d_kps, seq.d_desc, seq.d_n

    def step():
        seq.extract(d_img)

    def barrier():
        if world > 1:
            dist.barrier()

    # (part of the set-up, disclosed as "settle_steps": the GPU's clocks need ~15 ms of work to settle, whatever W the caller
    #  asks for -- with 3 warm-up steps alone a step reads 0.299 ms, with 50 it reads 0.275)
    for _ in range(SETTLE_STEPS):
        step()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    # ---- `value_cold` (VERDICT r3 next #8a): the timed loop above replays the SAME 64 frames -- 19.7 MB of input and the 60 MB pyramid stay in the
    #      256 MB Infinity Cache, so none of its traffic is HBM traffic.  Here sixteen DISTINCT batches (the frames shifted by another offset each:
    #      other corners, other keypoints; 315 MB of input) rotate through the same extractor, K steps, same barriers: every step's input comes from HBM.
    cold = {}
    try:
        NB = 16
        rot = [torch.roll(d_img, shifts=(5 * k, 9 * k), dims=(1, 2)).contiguous() for k in range(NB)]
        for k in range(NB):
            seq.extract(rot[k])
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        tc0 = time.perf_counter()
        kp_cold = 0
        for k in range(args.steps):
            seq.extract(rot[k % NB])
        torch.cuda.synchronize()
        barrier()
        tc1 = time.perf_counter()
        for k in range(NB):      # (keypoints per step: one pass over the sixteen batches, outside the timed region)
            seq.extract(rot[k])
            kp_cold += int(d_n.to(torch.int64).sum().item())
        v_cold, _kpc, el_cold = shard.aggregate_throughput(int(round(kp_cold / NB)) * args.steps, tc1 - tc0, device=dev)
        cold = {"value_cold": round(v_cold, 1), "ms_per_step_cold": round(el_cold / args.steps * 1e3, 4), "distinct_batches": NB,
                "input_MB": round(NB * B * W * H / 1e6, 1),
                "note": "same extractor, K steps over 16 distinct 64-frame batches in rotation (the frames rolled by (5 k, 9 k) pixels): the input of a step is "
                        "not in the 256 MB Infinity Cache when the step starts"}
        del rot
        seq.extract(d_img)
        torch.cuda.synchronize()
    except Exception as ex:  # noqa: BLE001
        cold = {"value_cold_error": repr(ex)}
    # per-stage HIP-event timing: the same K steps again with events recorded between the kernels on the streams they
    # run on, the stages one after the other (the timed steps above overlap FAST with the pyramid and the blur with the quad-tree)
    ext.set_profiling(True)
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    stage_ms = ext.last_timing()
    ext.set_profiling(False)
    # whole-job throughput: SUM of the units of all ranks / MAX of the elapsed times (RCCL all_reduce when N > 1)
    value, kp_all_steps, elapsed = shard.aggregate_throughput(int(d_n.to(torch.int64).sum().item()) * args.steps, t1 - t0, device=dev)
    kp_total_step = int(round(kp_all_steps / args.steps))

    # ---- roofline of the dominant kernel (largest average HIP-event duration over the timed steps)
    n_host = d_n.cpu().numpy()
    kp_rank = int(n_host.sum())
    n_cand = 0
    try:
        n_cand = sum(len(ext.level_candidates(l, 0)) for l in range(8)) * B  # frame 0 as the estimate
    except Exception:
        pass
    stage_bytes = {
        "pyramid": B * BYTES_PYRAMID,
        "fast": B * BYTES_FAST + 4 * n_cand,
        "quadtree": 8 * n_cand + 4 * kp_rank,     # candidate words gathered (read+write) + selected keypoints
        "blur": B * BYTES_BLUR,
        "orient_describe": kp_rank * BYTES_PER_KP,
    }
    dom = max(stage_bytes, key=lambda k: stage_ms[k])
    achieved = stage_bytes[dom] / (stage_ms[dom] * 1e-3) / 1e9
    # HBM traffic of the dominant kernel: FETCH_SIZE + WRITE_SIZE from the COMMITTED PMC passes of this same command (two
    # separate rocprofv3 --pmc runs, profiles/r02_pmc_traffic.json) -- a recorded figure, not measured in this run.  FETCH_SIZE
    # carries the guide's x2 (MI355X_MICROARCH.md, HBM): the blur's own byte count confirms it for these 4 / 12-byte-per-lane
    # reads (22/16 halo rows x 60.8 MB = 83.6 MB expected, 2 x 42.9 MB counted).
    traffic, traffic_note = None, None
    try:
        pmc_file = _profile("pmc_traffic.json")
        pmc = json.load(open(pmc_file))
        if pmc.get("batch") == B and dom in pmc["kernels"] and all(_fresh(pmc, rel) for rel in ORB_SOURCES):
            k = pmc["kernels"][dom]
            traffic = int(2 * k["fetch_bytes_per_step"] + k["write_bytes_per_step"])   # all launches of the stage in one step
            traffic_note = "from the committed PMC passes (%s): 2 x FETCH_SIZE + WRITE_SIZE per step" % os.path.relpath(pmc_file, ROOT)
    except Exception:
        pass
    # VALU roofline: wave instructions per launch (SQ_INSTS_VALU of the committed PMC pass, a property of the kernel and its
    # input) / the launch duration measured in THIS run / the chip's issue rate (1024 SIMDs x one wave instruction per 4 cycles)
    valu, valu_insts, sq_file = {}, {}, _profile("pmc_sq.json")
    try:
        sq = json.load(open(sq_file))
        kname = {"pyramid": "k_resize", "fast": "k_fast_cells", "quadtree": "k_quadtree", "blur": "k_blur7", "orient_describe": "k_orient_describe"}
        if sq.get("batch") == B and all(_fresh(sq, rel) for rel in ORB_SOURCES):
            for st_, kn in kname.items():
                if kn in sq["kernels"] and stage_ms.get(st_, 0) > 0:
                    insts = sq["kernels"][kn]["SQ_INSTS_VALU"] * sq["kernels"][kn].get("launches_per_step", 1)
                    valu_insts[st_] = insts
                    valu[st_] = round(insts / (stage_ms[st_] * 1e-3) / VALU_PEAK_WAVE_INSTS, 4)
    except Exception:
        pass
    hbm = {"achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5)}
    common = {"kernel": dom, "traffic": traffic, "traffic_source": traffic_note,
              "algorithmic_bytes_per_launch": int(stage_bytes[dom]), "avg_launch_ms": round(stage_ms[dom], 4),
              "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
              "valu_frac": valu.get(dom), "valu_frac_by_stage": valu,
              "pipeline_GBps": round(sum(stage_bytes.values()) / (stage_ms["total"] * 1e-3) / 1e9, 2)}
    # roofline.frac is the HBM fraction SURVEY.md s8(d) defines (algorithmic bytes / launch time / 8 TB/s).  The kernel itself is bound by VALU issue
    # (integer / packed byte arithmetic, no MFMA shape): that view -- wave instructions per launch from the committed SQ pass over the chip's
    # measured issue rate -- sits beside it under roofline.valu, and is null when the committed pass was not captured from today's orb.hip.
    roofline = dict(dict({"bound": "hbm"}, **hbm), **common)
    roofline["limiter"] = "VALU issue (see roofline.valu): ~54 lane-instructions per pixel put FAST at the integer-issue ceiling, 12x above its HBM time"
    if valu.get(dom) is not None:
        ginst = valu_insts[dom] / (stage_ms[dom] * 1e-3) / 1e9
        roofline["valu"] = {"achieved": round(ginst, 2), "peak": round(VALU_PEAK_WAVE_INSTS / 1e9, 1), "unit": "Gwave-inst/s", "frac": valu[dom],
                            "note": "SQ_INSTS_VALU per launch (committed PMC pass, %s) / this run's launch time (HIP events on the kernel's stream) / 6.14e11 "
                                    "wave-instructions/s (1024 SIMDs x 2.4 GHz / 4 cycles: profiles/r04_ubench_intops.txt, with the fp32 rows beside the integer ones)"
                                    % os.path.relpath(sq_file, ROOT)}
    else:
        roofline["valu"] = None
        roofline["valu_note"] = "no committed SQ pass captured from the current eao_fusion_amd/csrc/orb.hip (source hash differs): derived fields left out"
    roofline["cache_note"] = ("the timed steps replay the same 64 frames: input + pyramid (80 MB) live in the 256 MB Infinity Cache and FETCH_SIZE counts those hits, "
                              "so `traffic` is cache-side traffic, not HBM traffic; `value_cold` rotates 16 distinct batches")

    # every collective first (all ranks), the rank-0-only measurements afterwards: no rank waits inside RCCL for minutes
    gathered = {}
    # ---- BASELINE configs[4] (batched sequence), outside the timed region: consecutive-frame matching inside the shard (one
    #      halo frame from the previous rank), this rank's share of the 25 local-BA windows (window w -> rank w mod N) in ONE
    #      eao_local_ba_batch call, and the s8(e) all-gather of every frame's (keypoints, descriptors) and every window's result
    seqr = {}
    if not args.no_extra:
        try:
            seqr = measure_sequence(E, sequence, shard, torch, dist, dev, rank, world, seq, n_frames)
        except Exception as ex:  # noqa: BLE001
            seqr = {"error_rank%d" % rank: repr(ex)}
            if world > 1:
                raise
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    seq_out = seqr
    extra = {}
    cpu_baseline = None
    if rank == 0:
        if not args.no_extra:
            extra = measure_extra(E, synth, torch, dev)
            try:
                extra["class_surface"] = measure_class_surface(synth)
            except Exception as ex:  # noqa: BLE001
                extra["class_surface"] = {"error": repr(ex)}
            try:
                extra["mixed_load"] = measure_mixed_load(synth)
            except Exception as ex:  # noqa: BLE001
                extra["mixed_load"] = {"error": repr(ex)}
        if not args.no_cpu_baseline:
            cpu_baseline, cpu_extra = measure_cpu(frames, synth, extra)
            extra.update(cpu_extra)
    if seq_out:
        seq_out["ranks"] = world                 # the world size RCCL saw: an N-GPU run describes itself
        extra["sequence"] = seq_out
    if rank == 0:
        roofline["ba"] = ba_roofline(extra)
        roofline = ordered_roofline(roofline, flat_scalars(roofline, extra), extra)
        if cpu_baseline is not None:      # (CPU-side figures of the other half of the metric, beside the headline one: the driver's record keeps this object's scalars)
            for k_, fn_ in (("ba_ms_per_lba", lambda: extra["cpu_ba"]["ms_per_lba"]), ("pose_opt_us", lambda: round(extra["cpu_pose_optimization_ms"] * 1e3, 1))):
                try:
                    cpu_baseline[k_] = fn_()
                except Exception:  # noqa: BLE001
                    pass
        out = {
            "metric": "ORB kpts/s (640x480, 1k feat) + local-BA residuals/s (20 KF x 3k pts)",
            "value": round(value, 1), "unit": "kpts/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "settle_steps": SETTLE_STEPS,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "ORBextractor FAST+rBRIEF, 640x480, 8-level pyramid, nFeatures 1000, batch=%d synthetic frames per GPU (BASELINE configs[1])" % B,
                       "frames_per_step": B * world, "keypoints_per_step": kp_total_step, "parallelism": "frames sharded per GPU, no data-path collective"},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "extra": extra,
        }
        out.update(cold)
        print(json.dumps(out))
    if seq_out and seq_out.get("allgather_ok") is False:
        print("[bench rank %d] the s8(e) all-gather returned payloads that differ from what the ranks sent" % rank, file=sys.stderr)
        return 3
    return 0


ROOFLINE_HEAD = ("bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "valu_frac", "ba_single_ms", "ba_single_frac", "ba_batched_ms", "ba_batched_frac",
                 "ba_traffic_over_algorithmic", "hamming_matrix_frac", "pose_opt_us", "track_frame_ms", "ba_map_scale_banded_ms", "ba_map_scale_banded_frac", "mixed_p99_ms",
                 "mixed_idle_p50_ms")
ROOFLINE_NEXT = ("algorithmic_bytes_per_launch", "ba_cpu_ms", "cs_lba_ms", "cs_lba_observations_ref_ms", "ba_map_scale_ms", "ba_map_scale_frac", "mixed_beside_batch_p99_ms",
                 "mixed_beside_lba_and_map_ba_p99_ms", "mixed_no_priorities_p99_ms", "kernel", "pipeline_GBps")


def ordered_roofline(roofline, flats, extra):
    """the driver's record keeps the FIRST 20 scalars of `roofline` (strings count, nested objects are dropped).  So: the contract's members and the
    numbers every fraction of the verdict is recomputed from come first, in ROOFLINE_HEAD's order (a figure that was not measured stays in its place as null); the second
    tier follows; every other flat figure moves to extra.flat, the prose to extra.roofline_notes; the nested objects close the record."""
    src = dict(roofline, **flats)
    out = {}
    for k in ROOFLINE_HEAD + ROOFLINE_NEXT:
        if k in ROOFLINE_HEAD or src.get(k) is not None:
            out[k] = src.get(k)
    notes = {k: src[k] for k in ("traffic_source", "limiter", "cache_note", "valu_note") if src.get(k) is not None}
    extra["roofline_notes"] = notes
    extra["flat"] = {k: v for k, v in flats.items() if k not in out}
    for k, v in roofline.items():      # nested objects last (stage_ms, valu_frac_by_stage, valu, ba)
        if isinstance(v, (dict, list)):
            out[k] = v
    return out


def flat_scalars(roofline, extra):
    """the driver's record keeps the SCALAR members of `roofline` only (nested objects and `extra` are dropped from BENCH_rNN.json.parsed), so the
    other half of the two-part metric -- local BA -- and the figures every fraction of the verdict is recomputed from are repeated here as flat numbers.  Each is a copy
    of a value measured in this run (its long form stays where it was: roofline.ba.*, extra.*); a figure that was not measured is left out, never guessed."""
    out = {}

    def put(key, fn):
        try:
            v = fn()
            if v is not None:
                out[key] = v
        except Exception:  # noqa: BLE001
            pass
    rb = roofline.get("ba", {})
    put("ba_single_ms", lambda: extra["ba"]["ms_per_lba_wall"])
    put("ba_single_device_ms", lambda: extra["ba"]["ms_per_lba_device"])
    put("ba_single_frac", lambda: rb["single_window"]["frac"])
    put("ba_single_iterations", lambda: rb["single_window"]["iterations"])
    put("ba_single_bytes_per_iteration", lambda: rb["single_window"]["algorithmic_bytes_per_iteration"])
    put("ba_single_residual_blocks_per_s", lambda: extra["ba"]["ba_residual_blocks_per_s"])
    put("ba_batched_ms", lambda: extra["ba_batch"]["ms_per_call"])
    put("ba_batched_device_ms", lambda: extra["ba_batch"]["device_ms"])
    put("ba_batched_frac", lambda: rb["batched"]["frac"])
    put("ba_batched_iterations", lambda: rb["batched"]["iterations"])
    put("ba_batched_bytes_per_iteration", lambda: rb["batched"]["algorithmic_bytes_per_iteration"])
    put("ba_batched_residual_blocks_per_s", lambda: extra["ba_batch"]["ba_residual_blocks_per_s"])
    put("ba_traffic_over_algorithmic", lambda: rb["batched"]["traffic_over_algorithmic"])
    put("ba_single_traffic_over_algorithmic", lambda: rb["single_window"]["traffic_over_algorithmic"])
    put("ba_cpu_ms", lambda: extra["cpu_ba"]["ms_per_lba"])
    put("ba_gpu_over_cpu", lambda: round(extra["cpu_ba"]["ms_per_lba"] / extra["ba"]["ms_per_lba_wall"], 2))
    put("ba_map_scale_ms", lambda: extra["bundle_adjustment_map_scale"]["ms_per_call"])
    put("ba_map_scale_frac", lambda: extra["bundle_adjustment_map_scale"]["frac_hbm"])
    put("ba_map_scale_banded_ms", lambda: extra["bundle_adjustment_map_scale_banded"]["ms_per_call"])
    put("ba_map_scale_banded_frac", lambda: extra["bundle_adjustment_map_scale_banded"]["frac_hbm"])
    put("hamming_matrix_us", lambda: round(extra["hamming_matrix"]["ms_per_launch"] * 1e3, 2))
    put("hamming_matrix_frac", lambda: extra["hamming_matrix"]["frac_hbm"])
    put("hamming_best2_us", lambda: round(extra["hamming_best2"]["ms_per_launch"] * 1e3, 2))
    put("pose_opt_us", lambda: round(extra["pose_optimization_c_abi_ms"] * 1e3, 1))
    put("pose_opt_cpu_us", lambda: round(extra["cpu_pose_optimization_ms"] * 1e3, 1))
    put("track_frame_ms", lambda: extra["tracking_motion_model_device"]["ms_motion_model_plus_local_map"])
    put("track_local_map_ms", lambda: extra["tracking_frame_device_ms"])
    put("track_motion_model_ms", lambda: extra["tracking_motion_model_device"]["ms_per_call"])
    put("track_reference_keyframe_ms", lambda: extra["tracking_reference_keyframe_device"]["ms_per_call"])
    ml = extra.get("mixed_load", {})
    put("mixed_p99_ms", lambda: ml["priorities"]["device_chain"]["beside_lba"]["frame_ms"]["p99"])
    put("mixed_idle_p50_ms", lambda: ml["priorities"]["device_chain"]["idle"]["frame_ms"]["p50"])
    put("mixed_beside_batch_p99_ms", lambda: ml["priorities"]["device_chain"]["beside_lba_batch25"]["frame_ms"]["p99"])
    put("mixed_beside_lba_and_map_ba_p99_ms", lambda: ml["priorities"]["device_chain"]["beside_lba_and_map_ba"]["frame_ms"]["p99"])
    put("mixed_no_priorities_p99_ms", lambda: ml["no_priorities"]["device_chain"]["beside_lba"]["frame_ms"]["p99"])
    put("mixed_class_surface_p99_ms", lambda: ml["priorities"]["class_surface"]["beside_lba"]["frame_ms"]["p99"])
    put("mixed_class_surface_idle_p50_ms", lambda: ml["priorities"]["class_surface"]["idle"]["frame_ms"]["p50"])
    put("mixed_results_identical", lambda: bool(ml["priorities"]["results_identical"]))
    put("orb_single_frame_c_abi_ms", lambda: extra["orb_single_frame_host_api"]["ms_per_frame"])
    put("stream_copy_GBps", lambda: extra["stream_copy_GBps"])
    cs = extra.get("class_surface", {})
    for call, key in (("orb_extractor_call", "cs_orb_call_ms"), ("orb_extractor_call_with_pyramid", "cs_orb_call_with_pyramid_ms"), ("pose_optimization", "cs_pose_opt_ms"),
                      ("local_bundle_adjustment", "cs_lba_ms"), ("search_by_projection_local_map", "cs_sbp_local_map_ms"),
                      ("search_by_projection_last_frame", "cs_sbp_last_frame_ms"), ("search_by_bow_kf_frame", "cs_sbow_ms")):
        put(key, lambda c=call: cs[c]["call_ms"])
        put(key.replace("_ms", "_c_abi_ms"), lambda c=call: cs[c]["c_abi_ms"])
    put("cs_lba_overhead_frac", lambda: cs["local_bundle_adjustment"]["adapter_overhead_frac_of_c_abi"])
    put("cs_lba_observations_ref_ms", lambda: cs["local_bundle_adjustment_with_accessors"]["call_ms"])
    put("cs_map_ba_ms", lambda: cs["bundle_adjustment_map"]["call_ms"])
    put("cs_map_ba_c_abi_ms", lambda: cs["bundle_adjustment_map"]["c_abi_ms"])
    put("cs_map_ba_observations_ref_ms", lambda: cs["bundle_adjustment_map_with_accessors"]["call_ms"])
    gs, cg = extra.get("guided_searches", {}), extra.get("cpu_guided_searches", {})
    for name in ("search_by_bow_kf_frame", "search_by_bow_kf_kf", "search_for_triangulation", "fuse_search_pose"):
        put("gs_%s_ms" % name, lambda n=name: gs[n]["ms_per_call"])
        put("gs_%s_over_cpu" % name, lambda n=name: cg[n]["gpu_over_cpu_time"])
        put("gs_%s_handles_ms" % name, lambda n=name: gs[n]["ms_per_call_handles"])
        put("gs_%s_handles_over_cpu" % name, lambda n=name: cg[n]["gpu_handles_over_cpu_time"])
    put("gs_triangulation_batch10_ms", lambda: gs["search_for_triangulation_batch"]["ms_per_call"])
    put("gs_fuse_batch10_ms", lambda: gs["fuse_search_batch"]["ms_per_call"])
    put("gs_triangulation_batch10_handles_ms", lambda: gs["search_for_triangulation_batch"]["ms_per_call_handles"])
    put("gs_fuse_batch10_handles_ms", lambda: gs["fuse_search_batch"]["ms_per_call_handles"])
    return out


def ba_roofline(extra):
    """The BA half of the metric against the HBM roofline.  Unit = one LM iteration of one window: E x 520 + P x 360 algorithmic bytes (SURVEY.md s8d:
    linearisation + Schur assembly + solve + back substitution and the error pass).  `achieved` is measured in THIS run: bytes of all iterations of the
    call / the call's device span (HIP events on the library's stream).  `traffic` = HBM-side bytes per LM iteration from the committed FETCH_SIZE /
    WRITE_SIZE passes of the same calls (tools/prof_round.sh -> profiles/r04_ba_pmc_traffic.json; 2 x FETCH_SIZE + WRITE_SIZE as MI355X_MICROARCH.md
    prescribes for gfx950), null when that file was not captured from today's LM sources (lba.hip, gba.hip, lm_host.hip, lm_internal.h).  The dominant launch is named with its recorded average duration."""
    out = {}
    P_, per_edge, per_point = 3000, 520, 360
    tr = None
    try:
        f = _profile("ba_pmc_traffic.json")
        tr = json.load(open(f)) if f else None
        if not all(_fresh(tr, rel) for rel in LM_SOURCES):
            tr = None
    except Exception:  # noqa: BLE001
        tr = None

    def dominant(csv_name, part):
        """(kernel, avg_us) of the k_ba_* launch with the largest total time in a committed rocprofv3 summary"""
        f = _profile(csv_name)
        best = (None, None, 0.0)
        if f:
            for ln in open(f):
                if ln.startswith("k_ba_") and not ln.startswith("k_ba_upload"):
                    c = ln.strip().split(",")
                    tot = float(c[-2])
                    if tot > best[2]:
                        best = (",".join(c[:-6]), float(c[-5]), tot)
        return best[0], best[1], (os.path.relpath(f, ROOT) if f else None)
    try:
        b = extra.get("ba_batch")
        if b:
            E_ = b["ba_residual_blocks_per_s"] * b["ms_per_call"] * 1e-3 / max(b["linearizations"], 1)       # edges per window (average)
            nwin = 25
            rounds = max(b["linearizations"] // nwin, 1)          # `linearizations` counts every window's; a launch advances all 25
            by_iter = nwin * (E_ * per_edge + P_ * per_point)
            it_ms = b["device_ms"] / rounds
            ach = by_iter / (it_ms * 1e-3) / 1e9
            kn, dom_us, src = dominant("ba_batch_kernel_stats.csv", "batched")
            t = (tr or {}).get("batched")
            out["batched"] = {"workload": "25 windows x (20 + 4 KF, 3000 MP) in ONE eao_local_ba_batch call", "bound": "hbm", "unit": "GB/s",
                              "algorithmic_bytes_per_iteration": int(by_iter), "iterations": rounds, "avg_iteration_ms": round(it_ms, 4),
                              "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "frac": round(ach / HBM_PEAK_GBS, 4),
                              "traffic": None if not t else int(t["hbm_bytes_per_iteration"]),
                              "traffic_over_algorithmic": None if not t else round(t["hbm_bytes_per_iteration"] / by_iter, 3),
                              "traffic_source": None if not t else "profiles/r04_ba_pmc_traffic.json (2 x FETCH_SIZE + WRITE_SIZE of the launches of one LM iteration)",
                              "kernel": kn, "avg_launch_ms": None if dom_us is None else round(dom_us * 1e-3, 4), "kernel_source": src,
                              "note": "achieved / frac: all launches of an LM iteration, measured in this run.  The launches are bound by the CUs' texture-address / L1 path, LDS and "
                                      "dependent-launch latency (profiles/r04_ba_pair_ablation.txt), not by streaming"}
        s1 = extra.get("ba")
        if s1:
            E_ = s1["ba_residual_blocks_per_s"] * s1["ms_per_lba_wall"] * 1e-3 / max(s1["linearizations_per_lba"], 1)
            by_iter = E_ * per_edge + P_ * per_point
            it_ms = s1["ms_per_lba_device"] / max(s1["linearizations_per_lba"], 1)
            ach = by_iter / (it_ms * 1e-3) / 1e9
            kn, dom_us, src = dominant("ba_single_kernel_stats.csv", "single")
            t = (tr or {}).get("single_window")
            out["single_window"] = {"workload": "one LocalBundleAdjustment window (BASELINE configs[3])", "bound": "latency (one window is 9 MB and ~60 dependent launches)",
                                    "unit": "GB/s", "algorithmic_bytes_per_iteration": int(by_iter), "iterations": s1["linearizations_per_lba"],
                                    "avg_iteration_ms": round(it_ms, 4), "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "frac": round(ach / HBM_PEAK_GBS, 4),
                                    "traffic": None if not t else int(t["hbm_bytes_per_iteration"]),
                                    "traffic_over_algorithmic": None if not t else round(t["hbm_bytes_per_iteration"] / by_iter, 3),
                                    "kernel": kn, "avg_launch_ms": None if dom_us is None else round(dom_us * 1e-3, 4), "kernel_source": src}
    except Exception as ex:  # noqa: BLE001
        out["error"] = repr(ex)
    return out


def self_launch(n):
    import socket
    import subprocess
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL needs it on this host driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def dry_run(args, rank, world):
    """CPU rehearsal of the multi-rank plumbing (gloo): shard arithmetic, halo exchange, the s8(e) all-gathers of
    (keypoints, descriptors) and of per-window BA results, the throughput reduction.  Payloads are stand-ins derived from the
    frame / window index (so that every rank can check what it received); no extraction, no timing, `value` is null."""
    import torch
    import torch.distributed as dist
    from eao_fusion_amd import shard
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    B, cap = args.batch, 16
    n_frames = B * world
    lo, hi = shard.frame_shard(n_frames, rank, world)

    def rec(f):      # stand-in (keypoints, descriptors, count) of frame f
        g = torch.Generator().manual_seed(7000 + f)
        return (torch.randint(0, 256, (cap, 28), dtype=torch.uint8, generator=g), torch.randint(0, 256, (cap, 32), dtype=torch.uint8, generator=g),
                3 + f % (cap - 3))
    mine = [rec(f) for f in range(lo, hi)]
    kps, desc = torch.stack([m[0] for m in mine]), torch.stack([m[1] for m in mine])
    cnt = torch.tensor([m[2] for m in mine], dtype=torch.int32)
    hk, hd, hn = shard.exchange_halo_frame(kps[-1], desc[-1], int(cnt[-1]))
    ok = True
    if rank > 0:
        ek, ed, en = rec(lo - 1)
        ok &= bool(torch.equal(hk, ek) and torch.equal(hd, ed) and hn == en)
    K, D, Cn = shard.gather_frame_results(kps, desc, cnt, n_frames)
    for f in (0, n_frames // 2, n_frames - 1):
        ek, ed, en = rec(f)
        ok &= bool(torch.equal(K[f], ek) and torch.equal(D[f], ed) and int(Cn[f]) == en)
    nwin = 25
    ws = shard.window_shard(nwin, rank, world)
    cams = torch.stack([torch.full((24, 16), float(w)) for w in ws]) if ws else torch.zeros((0, 24, 16))
    pts = torch.stack([torch.full((30, 3), float(-w)) for w in ws]) if ws else torch.zeros((0, 30, 3))
    AC, AP = shard.gather_window_results(cams, pts, nwin)
    ok &= all(float(AC[w, 0, 0]) == w and float(AP[w, 0, 0]) == -w for w in range(nwin))
    thr, units, secs = shard.aggregate_throughput(int(cnt.sum()), 1.0 + rank)
    ok &= secs == float(world) and units == float(Cn.sum())
    flag = torch.tensor([1 if ok else 0])
    if world > 1:
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "ORB kpts/s (640x480, 1k feat) + local-BA residuals/s (20 KF x 3k pts)", "value": None, "unit": "kpts/s",
                          "n_gpus": world, "dry_run": True, "collectives_ok": bool(flag.item()), "frames": n_frames, "ba_windows": nwin,
                          "note": "CPU rehearsal of the launcher and the collectives (gloo); nothing was measured"}))
    return 0 if flag.item() else 1


def measure_sequence(E, sequence, shard, torch, dist, dev, rank, world, seq, n_frames):
    """This rank's share of the batched-sequence work and the collectives around it.  Every rank
    calls this (the collectives are symmetric); the returned dict is the same on every rank."""
    B = seq.B
    ev = lambda: torch.cuda.Event(enable_timing=True)
    # halo: the last frame of the previous shard (RCCL all_gather of one frame's records per rank)
    kl, dl, nl = seq.last_frame()
    hk, hd, hn = shard.exchange_halo_frame(kl, dl, nl)
    halo = (hd, hn) if hd is not None else (None, 0)
    seq.match(*halo)
    torch.cuda.synchronize()
    e0, e1 = ev(), ev()
    reps = 10
    e0.record()
    for _ in range(reps):
        seq.match(*halo)
    e1.record()
    torch.cuda.synchronize()
    t_match = e0.elapsed_time(e1) / reps * 1e-3
    pairs = B - (0 if halo[0] is not None else 1)
    # local BA: this rank's windows, packed once, ONE C-ABI call for all of them
    ws = shard.window_shard(sequence.N_WINDOWS, rank, world)
    probs = [sequence.window_problem(w) for w in ws]
    blocks, t_ba, cams, pts = 0.0, 0.0, None, None
    if probs:
        pk = E.Optimizer.pack_batch(probs)
        for _ in range(3):                                                # warm-up (contexts, arenas, pinned mirrors, code objects)
            res = E.Optimizer.LocalBundleAdjustmentBatch(None, packed=pk)
        Lc = E.load()
        tt = []
        for _ in range(7):                                                # timed at the C-ABI, like extra.ba_batch: median of 7 calls
            t0 = time.perf_counter()
            E._lib.check(Lc.eao_local_ba_batch(pk["P"], pk["n"], None, pk["R"]))
            tt.append(time.perf_counter() - t0)
        t_ba = float(np.median(tt))
        blocks = float(np.mean([len(p["edge_cam"]) for p in probs])) * res[0]["timing"]["linearizations"]
        cams = torch.from_numpy(np.stack([r["poses"].reshape(-1, 16) for r in res])).to(dev)
        pts = torch.from_numpy(np.stack([r["points"] for r in res])).to(dev)
    else:
        cams, pts = torch.zeros((0, 24, 16), device=dev), torch.zeros((0, 3000, 3), device=dev)
    # s8(e): every rank ends with every frame's keypoints + descriptors and every window's poses + points
    if world > 1:
        dist.barrier()
    shard.gather_frame_results(seq.d_kps, seq.d_desc, seq.d_n, n_frames)       # warm-up (allocations, RCCL channel set-up)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    K, D, Cn = shard.gather_frame_results(seq.d_kps, seq.d_desc, seq.d_n, n_frames)
    AC, AP = shard.gather_window_results(cams, pts, sequence.N_WINDOWS)
    torch.cuda.synchronize()
    t_gather = time.perf_counter() - t0
    ok = bool(torch.equal(K[rank * B:(rank + 1) * B], seq.d_kps) and torch.equal(Cn[rank * B:(rank + 1) * B], seq.d_n) and AC.shape[0] == sequence.N_WINDOWS)
    tsr = torch.tensor([float(pairs), blocks, 1.0 if ok else 0.0], dtype=torch.float64, device=dev)
    tmax = torch.tensor([t_match, t_ba, t_gather], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tsr, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    pairs_all, blocks_all, ok_all = float(tsr[0]), float(tsr[1]), float(tsr[2]) == world
    tm, tb, tg = float(tmax[0]), float(tmax[1]), float(tmax[2])
    gbytes = int(K.numel() + D.numel() + 4 * Cn.numel() + 4 * AC.numel() + 4 * AP.numel())
    return {"frames": n_frames, "frame_pairs": int(pairs_all), "frame_pairs_per_s": round(pairs_all / tm, 1) if tm > 0 else None,
            "match_ms_per_shard": round(tm * 1e3, 4), "ba_windows": sequence.N_WINDOWS,
            "ba_residual_blocks_per_s": round(blocks_all / tb, 1) if tb > 0 else None, "ba_ms_slowest_rank": round(tb * 1e3, 3),
            "allgather_ms": round(tg * 1e3, 3), "allgather_bytes_per_rank_result": gbytes, "allgather_ok": ok_all,
            "allgather_keypoints": int(Cn.sum().item()),
            "note": "per shard: device-resident best-2 matching of consecutive frames (eao_hamming_best2_sequence_device, halo frame from the "
                    "previous rank) and ONE eao_local_ba_batch call for the rank's windows (window w -> rank w mod N); then the s8(e) "
                    "all-gather of (keypoints, descriptors) of every frame and of every window's poses + points"}


def search_cases(synth, n=1000):
    """The nine guided searches of csrc/search.hip (rows a13-a15) on one synthetic two-keyframe scene, as (name, call(binding)) pairs -- the same calls time the
    product (search.product()) in measure_extra and the oracle (oracle.search_binding()) in measure_cpu."""
    sc = synth.synth_search_scene(n=n, seed=8300)
    P = sc["points"]
    ang = ((np.arange(len(P["active"])) * 37) % 360).astype(np.float32)
    s1 = dict(descriptors=sc["K1"]["descriptors"], angle=sc["K1"]["kp_angle"], valid=(sc["mp1"] >= 0).astype(np.uint8), fv=sc["fv1"])
    s2 = dict(descriptors=sc["K2"]["descriptors"], angle=sc["K2"]["kp_angle"], valid=(sc["mp2"] >= 0).astype(np.uint8), fv=sc["fv2"])
    T = sc["T2w"].astype(np.float64)
    pose15 = np.concatenate([T[:3, :3].ravel(), T[:3, 3], -T[:3, :3].T @ T[:3, 3]]).astype(np.float32)
    pm = np.stack([sc["K1"]["kp_x"], sc["K1"]["kp_y"]], 1)

    def pts_of(mp):
        idx = np.maximum(mp, 0)
        d = {k: np.ascontiguousarray(P[k][idx]) for k in ("Xw", "normal", "min_dist_inv", "max_dist_inv", "max_dist", "descriptors")}
        d["active"] = ((mp >= 0) & (P["active"][idx] > 0)).astype(np.uint8)
        return d
    P1, P2 = pts_of(sc["mp1"]), pts_of(sc["mp2"])
    cases = [
        ("search_by_projection_sim3", lambda g: g.search_by_projection_sim3(sc["K2"], sc["Scw"], sc["K"], P, 10)),
        ("search_by_projection_kf", lambda g: g.search_by_projection_kf(sc["K2"], sc["T2w"], sc["K"], P, ang, 15, 100, True)),
        ("search_by_bow_kf_frame", lambda g: g.search_by_bow(0, s1, s2, 0.75, True)),
        ("search_by_bow_kf_kf", lambda g: g.search_by_bow(1, s1, s2, 0.8, True)),
        ("search_for_triangulation", lambda g: g.search_for_triangulation(sc["K1"], sc["fv1"], sc["K2"], sc["fv2"], sc["F12"], sc["ex"], sc["ey"], 0, True)),
        ("search_for_initialization", lambda g: g.search_for_initialization(sc["K1"], sc["K2"], pm, 100, 0.9, True)),
        ("fuse_search_pose", lambda g: g.fuse_search(sc["K2"], 0, pose15, sc["K"], sc["bf"], P, 3.0)),
        ("fuse_search_sim3", lambda g: g.fuse_search(sc["K2"], 1, sc["Scw"], sc["K"], sc["bf"], P, 3.0)),
        ("search_by_sim3", lambda g: g.search_by_sim3(sc["K1"], sc["T1w"], P1, sc["K2"], sc["T2w"], P2, sc["K"], 1.0, sc["R12"], sc["t12"], 7.5)),
    ]
    return sc, pose15, cases


def time_calls(fn, reps=12, warm=2):
    for _ in range(warm):
        r = fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3, r


def measure_extra(E, synth, torch, dev):
    """Hamming (configs[2]) and local BA (configs[3]) on GPU 0, outside the timed region."""
    extra = {}
    L = E.load()
    try:
        a, b = synth.synth_descriptors(1000, 2000)
        pairs = 64
        dA = torch.from_numpy(np.tile(a, (pairs, 1, 1))).to(dev)
        dB = torch.from_numpy(np.tile(b, (pairs, 1, 1))).to(dev)
        dD = torch.zeros((pairs, 1000, 1000), dtype=torch.int16, device=dev)
        dO = torch.zeros((pairs, 1000, 4), dtype=torch.int32, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        for mode in ("matrix", "best2"):
            def run():
                if mode == "matrix":
                    E._lib.check(L.eao_hamming_matrix_device(dA.data_ptr(), 1000, dB.data_ptr(), 1000, pairs, dD.data_ptr(), st))
                else:
                    E._lib.check(L.eao_hamming_best2_device(dA.data_ptr(), 1000, dB.data_ptr(), 1000, pairs, None, dO.data_ptr(), st))
            for _ in range(3):
                run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            reps = 20
            for _ in range(reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            byts = pairs * (2_064_000 if mode == "matrix" else 72_000)
            h = {"pair_distances_per_s": round(pairs * 1e6 / (ms * 1e-3), 1), "ms_per_launch": round(ms, 4), "pairs_per_launch": pairs,
                 "achieved_GBps": round(byts / (ms * 1e-3) / 1e9, 2)}
            # round 4: the distances come off the matrix cores (v_mfma_i32_32x32x32_i8 over 0 / 1 bytes, csrc/hamming.hip): 8 instructions of 32 cycles per
            # 32 x 32 tile of distances -> the launch's matrix-core time at 2.4 GHz; the VALU spreads bits to bytes and (best-2) keeps the two smallest keys
            h["kernel"] = "k_hamming_matrix_mfma" if mode == "matrix" else "k_hamming_best2_mfma"
            h["mfma_time_frac"] = round((pairs * 1e6 / 1024.0) * 8 * 32 / (1024 * 2.4e9) / (ms * 1e-3), 4)
            if mode == "matrix":
                h["frac_hbm"] = round(byts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)     # write-bound: 2 B per distance
            else:
                h["bound"] = "valu + matrix cores (72 KB of traffic per pair: an HBM fraction says nothing here)"
            try:
                sq = json.load(open(_profile("pmc_sq.json")))
                if h["kernel"] in sq.get("kernels_extra", {}) and _fresh(sq, "eao_fusion_amd/csrc/hamming.hip"):
                    ke = sq["kernels_extra"][h["kernel"]]
                    h["valu_frac"] = round(ke["SQ_INSTS_VALU"] / (ms * 1e-3) / VALU_PEAK_WAVE_INSTS, 4)
                    if "SQ_VALU_MFMA_BUSY_CYCLES" in ke and "SQ_BUSY_CYCLES" in ke:
                        h["mfma_busy_cycles_per_launch"] = ke["SQ_VALU_MFMA_BUSY_CYCLES"]
            except Exception:
                pass
            extra["hamming_%s" % mode] = h
    except Exception as ex:  # noqa: BLE001
        extra["hamming_error"] = repr(ex)
    try:
        # measured stream-copy ceiling of this GPU: 1 GiB device-to-device copy, read + write counted
        src = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
        dst = torch.empty_like(src)
        dst.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        extra["stream_copy_GBps"] = round(2 * 5 * (1 << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
        del src, dst
    except Exception as ex:  # noqa: BLE001
        extra["stream_copy_error"] = repr(ex)
    try:
        # the ORB batch end to end through the host API: pageable host frames in, host keypoints + descriptors out (PCIe both ways);
        # never `value`
        ext_h = E.ORBextractor(1000, 1.2, 8, 20, 7)
        fr64 = np.stack([synth.synth_frame(1000 + f, 640, 480) for f in range(64)])
        for _ in range(2):
            kk, _dd = ext_h.extract_batch(fr64)
        t0 = time.perf_counter()
        for _ in range(5):
            kk, _dd = ext_h.extract_batch(fr64)
        dtp = (time.perf_counter() - t0) / 5
        extra["orb_end_to_end_pcie"] = {"ms_per_64_frames": round(dtp * 1e3, 3), "kpts_per_s": round(sum(len(k) for k in kk) / dtp, 1),
                                        "note": "eao_orb_extract_batch: H2D of 19.7 MB of frames + extraction + D2H of 64 x cap x 60 B, pageable host memory"}
        del ext_h
        # ... and through the STREAMING host API (eao_orb_stream_*): three pinned slots, asynchronous submit, the upload of batch
        # k + 1 and the download of batch k - 1 overlap the extraction of batch k.  The frames are in the pinned slots already (a
        # decoder / camera driver writes there directly); the variant that first copies 19.7 MB of pageable frames into the slot is
        # reported beside it.
        ext_s = E.ORBextractor(1000, 1.2, 8, 20, 7)
        sl = ext_s.stream_create(640, 480, 64, 3)
        for s_ in range(3):
            sl[s_]["frames"][:] = fr64
        for rep in range(2):
            for s_ in range(3):
                ext_s.stream_submit(s_)
            for s_ in range(3):
                ext_s.stream_wait(s_)
        nb_ = 30

        def pipeline(refill):
            t0 = time.perf_counter()
            kp = 0
            for k in range(nb_ + 2):
                if k < nb_:
                    if refill:
                        sl[k % 3]["frames"][:] = fr64
                    ext_s.stream_submit(k % 3)
                if k >= 2:
                    ext_s.stream_wait((k - 2) % 3)
                    kp += int(sl[(k - 2) % 3]["n"].sum())
            return (time.perf_counter() - t0) / nb_, kp / nb_
        dts, kps_ = pipeline(False)
        dtr, _ = pipeline(True)
        extra["orb_end_to_end_pcie_streaming"] = {"ms_per_64_frames": round(dts * 1e3, 3), "kpts_per_s": round(kps_ / dts, 1),
                                                  "ms_per_64_frames_with_refill_memcpy": round(dtr * 1e3, 3), "slots": 3, "batches": nb_,
                                                  "note": "eao_orb_stream_submit / _wait: H2D of 19.7 MB + extraction + D2H of 64 x cap x 60 B per batch on three streams, "
                                                          "pinned slots owned by the handle; never `value`"}
        del ext_s
    except Exception as ex:  # noqa: BLE001
        extra["orb_end_to_end_pcie_error"] = repr(ex)
    try:
        # the BA half of the metric as a THROUGHPUT: the 25 independent windows of BASELINE configs[4] in one eao_local_ba_batch
        # call (the window is a grid dimension of every launch), timed at the C-ABI with the arguments packed once
        from eao_fusion_amd import sequence as SQ
        probs = [SQ.window_problem(w) for w in range(SQ.N_WINDOWS)]
        pk = E.Optimizer.pack_batch(probs)
        for _ in range(4):          # (the first three calls of a process spend ~8 ms each inside the HIP runtime: stream / signal pools)
            E._lib.check(L.eao_local_ba_batch(pk["P"], pk["n"], None, pk["R"]))
        tt = []
        for _ in range(7):
            t0 = time.perf_counter()
            E._lib.check(L.eao_local_ba_batch(pk["P"], pk["n"], None, pk["R"]))
            tt.append(time.perf_counter() - t0)
        dtb = float(np.median(tt))
        import ctypes as C2
        dm2, li2 = C2.c_float(), C2.c_int32()
        L.eao_last_lm_timing(C2.byref(dm2), C2.byref(li2))
        Eavg = float(np.mean([len(p["edge_cam"]) for p in probs]))
        extra["ba_batch"] = {"workload": "25 x LocalBundleAdjustment (20 free + 4 fixed KF x 3000 MP, 5+10 LM its), ONE eao_local_ba_batch call",
                             "ms_per_call": round(dtb * 1e3, 3), "ms_per_call_min_max": [round(min(tt) * 1e3, 3), round(max(tt) * 1e3, 3)], "timing": "median of 7 calls", "ms_per_window": round(dtb * 1e3 / len(probs), 4), "device_ms": round(dm2.value, 3),
                             "ba_residual_blocks_per_s": round(Eavg * li2.value / dtb, 1), "ba_scalar_residuals_per_s": round(3 * Eavg * li2.value / dtb, 1),
                             "linearizations": int(li2.value)}
    except Exception as ex:  # noqa: BLE001
        extra["ba_batch_error"] = repr(ex)
    try:
        # PoseOptimization as a throughput: 256 frames (300 correspondences each) in ONE eao_pose_optimization_batch call
        # (the Relocalization candidate loop / offline replays), host arrays in, host results out
        pprobs = [synth.synth_pose(n=300, seed=7000 + k) for k in range(256)]
        # timed AT THE C-ABI like the BA batch: the argument records are packed once (the Python mirror spends ~8 us per frame
        # building them -- test infrastructure, five times the call itself); the mirror's own wall time is reported beside it
        ppk = E.Optimizer.pack_pose_batch(pprobs)
        for _ in range(3):
            E._lib.check(L.eao_pose_optimization_batch(ppk["P"], ppk["n"], ppk["R"]))
        tp = []
        for _ in range(9):
            t0 = time.perf_counter()
            E._lib.check(L.eao_pose_optimization_batch(ppk["P"], ppk["n"], ppk["R"]))
            tp.append(time.perf_counter() - t0)
        dpb = float(np.median(tp))
        t0 = time.perf_counter()
        for _ in range(3):
            E.Optimizer.PoseOptimizationBatch(pprobs)
        dpy = (time.perf_counter() - t0) / 3
        extra["pose_batch"] = {"workload": "256 x PoseOptimization (300 correspondences), ONE eao_pose_optimization_batch call, host arrays in and out",
                               "ms_per_call": round(dpb * 1e3, 3), "ms_per_call_min_max": [round(min(tp) * 1e3, 3), round(max(tp) * 1e3, 3)], "timing": "median of 9 calls at the C-ABI",
                               "us_per_frame": round(dpb * 1e6 / 256, 2), "frames_per_s": round(256 / dpb, 1),
                               "ms_per_call_through_python_mirror": round(dpy * 1e3, 3)}
    except Exception as ex:  # noqa: BLE001
        extra["pose_batch_error"] = repr(ex)
    try:
        p = synth.synth_ba()
        r = E.Optimizer.LocalBundleAdjustment(p)  # warm-up (allocations, code load)
        # timed AT THE C-ABI (the drop-in boundary): arguments prepared once, eao_local_ba called directly -- the Python mirror's
        # own array packing (~50 us per call) is test infrastructure, not part of the product
        import ctypes as C
        Lh = E.load()
        a_cams = np.ascontiguousarray(p["poses"], np.float32); a_fixed = np.ascontiguousarray(p["fixed"], np.uint8)
        a_pts = np.ascontiguousarray(p["points"], np.float32); a_ec = np.ascontiguousarray(p["edge_cam"], np.int32)
        a_ep = np.ascontiguousarray(p["edge_point"], np.int32); a_obs = np.ascontiguousarray(p["obs"], np.float32)
        a_inv = np.ascontiguousarray(p["inv_sigma2"], np.float32)
        Pb = E._lib.BAProblem(len(a_cams), len(a_pts), len(a_ec), E._lib.ptr(a_cams), E._lib.ptr(a_fixed), E._lib.ptr(a_pts), E._lib.ptr(a_ec),
                              E._lib.ptr(a_ep), E._lib.ptr(a_obs), E._lib.ptr(a_inv), p["fx"], p["fy"], p["cx"], p["cy"], p["bf"], 5, 10)
        o_c, o_p, o_e = np.zeros_like(a_cams), np.zeros_like(a_pts), np.zeros(len(a_ec), np.uint8)
        Rb = E._lib.BAResult()
        Rb.cam_Tcw, Rb.points, Rb.edge_outlier = E._lib.ptr(o_c), E._lib.ptr(o_p), E._lib.ptr(o_e)
        dm, li = C.c_float(), C.c_int32()
        reps = 20
        lin = 0
        dev_ms = 0.0
        t0 = time.perf_counter()
        for _ in range(reps):
            E._lib.check(Lh.eao_local_ba(C.byref(Pb), None, C.byref(Rb)))
        wall = (time.perf_counter() - t0) / reps
        for _ in range(3):
            E._lib.check(Lh.eao_local_ba(C.byref(Pb), None, C.byref(Rb)))
            Lh.eao_last_lm_timing(C.byref(dm), C.byref(li))
            lin += li.value
            dev_ms += dm.value
        lin, dev_ms = lin / 3 * reps, dev_ms / 3 * reps
        E_ = len(p["edge_cam"])
        extra["ba"] = {"workload": "LocalBundleAdjustment 20 free + 4 fixed KF x 3000 MP, E=%d stereo edges, 5+10 LM its (BASELINE configs[3])" % E_,
                       "ms_per_lba_wall": round(wall * 1e3, 3), "ms_per_lba_device": round(dev_ms / reps, 3),
                       "timed_at": "C-ABI (eao_local_ba, host buffers in and out)",
                       "linearizations_per_lba": lin / reps,
                       "ba_residual_blocks_per_s": round(E_ * (lin / reps) / wall, 1),
                       "ba_scalar_residuals_per_s": round(3 * E_ * (lin / reps) / wall, 1),
                       "achieved_GBps": round((lin / reps) * (E_ * 520 + 3000 * 360) / wall / 1e9, 3), "iters": [int(x) for x in Rb.iters[:]]}
        # independent windows from several host threads (one HIP stream and arena per thread): the single-window kernels
        # leave most CUs idle, so windows overlap -- the throughput figure for BASELINE configs[4]'s per-window local BA
        import threading
        nth, wreps = 8, 6
        def _work():
            for _ in range(wreps):
                E.Optimizer.LocalBundleAdjustment(p)
        ths = [threading.Thread(target=_work) for _ in range(nth)]
        t0 = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        dtw = time.perf_counter() - t0
        extra["ba"]["concurrent_windows"] = {"host_threads": nth, "windows_per_s": round(nth * wreps / dtw, 1),
                                             "ms_per_window": round(dtw * 1e3 / (nth * wreps), 3),
                                             "ba_residual_blocks_per_s": round(E_ * (lin / reps) * nth * wreps / dtw, 1)}
        # guided matching of one tracked frame (the two SearchByProjection variants of the tracking loop)
        curf, lastf, mpsf = synth.synth_tracking()
        mt = E.ORBmatcher(0.8, True)
        mt.SearchByProjectionPoints(curf, mpsf, 1.0)
        t0 = time.perf_counter()
        for _ in range(20):
            nm1, _m = mt.SearchByProjectionPoints(curf, mpsf, 1.0)
            nm2, _m = mt.SearchByProjectionFrames(curf, lastf, 7.0, False)
        extra["guided_matching"] = {"ms_per_frame_both_searches": round((time.perf_counter() - t0) / 20 * 1e3, 3),
                                    "matches": [int(nm1), int(nm2)], "keypoints": int(len(curf["kp_x"])),
                                    "note": "host buffers in/out, includes H2D/D2H and the host replay of the greedy assignment"}
        pp = synth.synth_pose()
        E.Optimizer.PoseOptimization(pp)
        t0 = time.perf_counter()
        for _ in range(20):
            E.Optimizer.PoseOptimization(pp)
        extra["pose_optimization_ms"] = round((time.perf_counter() - t0) / 20 * 1e3, 3)
        # ... and at the C-ABI, the argument record built once (the figure above includes the Python mirror's array packing)
        import ctypes as C3
        pk1 = E.Optimizer.pack_pose_batch([pp])
        tq = []
        for _ in range(25):
            t0 = time.perf_counter()
            E._lib.check(L.eao_pose_optimization(C3.byref(pk1["P"][0]), C3.byref(pk1["R"][0])))
            tq.append(time.perf_counter() - t0)
        extra["pose_optimization_c_abi_ms"] = round(float(np.median(tq[5:])) * 1e3, 3)
        # the tracking step of independent frames (both guided searches + PoseOptimization) from several host threads: the
        # per-frame kernels are latency-bound single-workgroup work, frames of a batched sequence overlap on the device
        treps = 10
        def _track():
            m2 = E.ORBmatcher(0.8, True)
            for _ in range(treps):
                m2.SearchByProjectionPoints(curf, mpsf, 1.0)
                m2.SearchByProjectionFrames(curf, lastf, 7.0, False)
                E.Optimizer.PoseOptimization(pp)
        _track()
        ths = [threading.Thread(target=_track) for _ in range(nth)]
        t0 = time.perf_counter()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        dtt = time.perf_counter() - t0
        extra["tracking_concurrent"] = {"host_threads": nth, "frames_per_s": round(nth * treps / dtt, 1),
                                        "ms_per_frame": round(dtt * 1e3 / (nth * treps), 3),
                                        "note": "per frame: SearchByProjection(points) + SearchByProjection(frames) + PoseOptimization, host buffers in/out"}
        # the same tracked frame with NOTHING returning to the host in between (row f1): ComputeStereoFromRGBD + grid ->
        # isInFrustum over the local map -> SearchByProjection -> PoseOptimization chained on the device behind the extractor's
        # outputs, one copy back (eao_tracker_track_local_map)
        try:
            from eao_fusion_amd.tracker import Tracker
            curT, lastT, _ = synth.synth_tracking(n=1000, seed=7100, mono_frac=0.0, occupied_frac=0.0)
            NT = len(curT["kp_x"])
            kT = np.zeros(NT, E.orb.KP_DTYPE)
            kT["x"], kT["y"] = np.clip(curT["kp_x"], 1, 638), np.clip(curT["kp_y"], 1, 478)
            kT["angle"], kT["octave"] = curT["kp_angle"], curT["kp_octave"]
            XwT = lastT["Xw"].astype(np.float64)
            dT = np.linalg.norm(XwT, axis=1).astype(np.float32)
            ptsT = dict(active=np.ones(len(XwT), np.uint8), Xw=lastT["Xw"], normal=(XwT / dT[:, None]).astype(np.float32), min_dist_inv=0.6 * dT,
                        max_dist_inv=1.7 * dT, max_dist=(dT * np.float32(1.2) ** (lastT["octave"] - 0.5)).astype(np.float32), descriptors=lastT["descriptors"])
            sfT = curT["scale_factors"]
            trk = Tracker(curT["fx"], curT["fy"], curT["cx"], curT["cy"], curT["mbf"], (0.0, 640.0, 0.0, 480.0), sfT, (np.float32(1) / (sfT * sfT)).astype(np.float32),
                          float(np.log(np.float32(1.2))), 2048, 2048)
            trk.set_local_map(ptsT)
            dk = torch.zeros((2048, 28), dtype=torch.uint8, device=dev); dk[:NT] = torch.from_numpy(kT.view(np.uint8).reshape(NT, 28)).to(dev)
            dd = torch.zeros((2048, 32), dtype=torch.uint8, device=dev); dd[:NT] = torch.from_numpy(np.ascontiguousarray(curT["descriptors"])).to(dev)
            dn = torch.tensor([NT], dtype=torch.int32, device=dev)
            ddep = torch.full((480, 640), 3.0, dtype=torch.float32, device=dev)
            torch.cuda.synchronize()
            stT = torch.cuda.current_stream().cuda_stream
            for _ in range(5):
                rT = trk.track_local_map(dk.data_ptr(), dd.data_ptr(), dn.data_ptr(), ddep.data_ptr(), 640, 640, 480, curT["Tcw"], None, 3.0, 0.8, stT)
            ttT = []
            for _ in range(100):
                t0 = time.perf_counter()
                rT = trk.track_local_map(dk.data_ptr(), dd.data_ptr(), dn.data_ptr(), ddep.data_ptr(), 640, 640, 480, curT["Tcw"], None, 3.0, 0.8, stT)
                ttT.append(time.perf_counter() - t0)
            extra["tracking_frame_device_ms"] = round(float(np.median(ttT)) * 1e3, 4)
            extra["tracking_frame_device"] = {"keypoints": int(rT["n_keypoints"]), "map_points": len(XwT), "matches": int(rT["n_matches"]), "inliers": int(rT["n_inliers"]),
                                              "timing": "median of 100 calls (through the Python mirror)", "ms_min_mean": [round(min(ttT) * 1e3, 4), round(float(np.mean(ttT)) * 1e3, 4)],
                                              "note": "eao_tracker_track_local_map: RGB-D stereo + grid + isInFrustum + SearchByProjection(points) + PoseOptimization on the device, "
                                                      "results in mapped host memory; PoseOptimization's single-workgroup LM (4 rounds) is ~0.135 ms of it"}
            # the stage in front of it (round 4): TrackWithMotionModel's data path on the same handle, and both stages of a tracked frame back to back
            for _ in range(5):
                mT = trk.track_with_motion_model(dk.data_ptr(), dd.data_ptr(), dn.data_ptr(), ddep.data_ptr(), 640, 640, 480, curT["Tcw"], lastT, 15.0, False, True, True, stT)
            tmm, tbb = [], []
            for _ in range(100):
                t0 = time.perf_counter()
                mT = trk.track_with_motion_model(dk.data_ptr(), dd.data_ptr(), dn.data_ptr(), ddep.data_ptr(), 640, 640, 480, curT["Tcw"], lastT, 15.0, False, True, True, stT)
                t1 = time.perf_counter()
                trk.track_local_map(dk.data_ptr(), dd.data_ptr(), dn.data_ptr(), ddep.data_ptr(), 640, 640, 480, mT["Tcw"], None, 3.0, 0.8, stT)
                tmm.append(t1 - t0); tbb.append(time.perf_counter() - t0)
            extra["tracking_motion_model_device"] = {"ms_per_call": round(float(np.median(tmm)) * 1e3, 4), "matches": int(mT["n_matches"]), "kept_after_discard": int(mT["n_inliers"]),
                                                     "ms_motion_model_plus_local_map": round(float(np.median(tbb)) * 1e3, 4),
                                                     "note": "eao_tracker_track_with_motion_model: frame set-up + SearchByProjection(Cur, Last) with the rotation histogram + PoseOptimization + outlier "
                                                             "discard on the device, one copy back; then eao_tracker_track_local_map from its pose (two calls, two copies: the local map of "
                                                             "stage two depends on stage one's matches on the host, src/Tracking.cc:2233-2260)"}
            # ... and the other first stage (round 4): TrackReferenceKeyFrame's data path on a 1000-point keyframe / frame pair filed under 60 vocabulary nodes
            scR = synth.synth_search_scene(n=1000, seed=8100, n_nodes=60)
            K1R, K2R = scR["K1"], scR["K2"]
            NR = len(K2R["kp_x"])
            kR = np.zeros(NR, E.orb.KP_DTYPE)
            kR["x"], kR["y"] = np.clip(K2R["kp_x"], 1, 638), np.clip(K2R["kp_y"], 1, 478)
            kR["angle"], kR["octave"] = K2R["kp_angle"], K2R["kp_octave"]
            mp1R = scR["mp1"]
            kfR = dict(valid=(mp1R >= 0).astype(np.uint8), Xw=np.ascontiguousarray(scR["points"]["Xw"][np.maximum(mp1R, 0)], np.float32), descriptors=K1R["descriptors"],
                       angle=K1R["kp_angle"], fv=scR["fv1"])
            dkR = torch.zeros((2048, 28), dtype=torch.uint8, device=dev); dkR[:NR] = torch.from_numpy(kR.view(np.uint8).reshape(NR, 28)).to(dev)
            ddR = torch.zeros((2048, 32), dtype=torch.uint8, device=dev); ddR[:NR] = torch.from_numpy(np.ascontiguousarray(K2R["descriptors"])).to(dev)
            dnR = torch.tensor([NR], dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            fxR, fyR, cxR, cyR = scR["K"]
            trkR = Tracker(fxR, fyR, cxR, cyR, scR["bf"], (0.0, 640.0, 0.0, 480.0), sfT, (np.float32(1) / (sfT * sfT)).astype(np.float32), float(np.log(np.float32(1.2))), 2048, 2048)
            trr = []
            for i in range(105):
                t0 = time.perf_counter()
                rR = trkR.track_reference_keyframe(dkR.data_ptr(), ddR.data_ptr(), dnR.data_ptr(), ddep.data_ptr(), 640, 640, 480, scR["T2w"], kfR, scR["fv2"], 0.7, True, True, stT)
                if i >= 5: trr.append(time.perf_counter() - t0)
            extra["tracking_reference_keyframe_device"] = {"ms_per_call": round(float(np.median(trr)) * 1e3, 4), "keyframe_keypoints": len(mp1R), "keypoints": NR,
                                                           "matches": int(rR["n_matches"]), "kept_after_discard": int(rR["n_inliers"]),
                                                           "note": "eao_tracker_track_reference_keyframe: frame set-up + SearchByBoW(KF, Frame) node by node + rotation histogram + PoseOptimization + "
                                                                   "outlier discard on the device, one copy back (the frame's feature vector comes from DBoW2 on the host and is an input here)"}
        except Exception as ex:  # noqa: BLE001
            extra["tracking_frame_device_error"] = repr(ex)
        # ---- rows a13-a15: the nine remaining guided searches, each timed at the Python mirror of its C entry point (host buffers in and out: upload, candidate /
        #      pair-distance kernel, download, host replay) on a 1000-point two-keyframe scene; measure_cpu() times the oracle's counterpart beside each
        try:
            from eao_fusion_amd import search as SR
            gS = SR.product()
            scS, pose15S, casesS = search_cases(synth)
            gsr = {}
            for name, fn in casesS:
                ms, r = time_calls(lambda: fn(gS))
                gsr[name] = {"ms_per_call": round(ms, 4), "matches": int(r[0])}
            # the batched LocalMapping-side entry points (round 4): 10 neighbour / target keyframes per call against 10 single calls
            nbS = 10
            k2s = [scS["K2"]] * nbS; fvs = [scS["fv2"]] * nbS
            Fs = [scS["F12"]] * nbS; exs = [scS["ex"]] * nbS; eys = [scS["ey"]] * nbS
            mb, _ = time_calls(lambda: gS.search_for_triangulation_batch(scS["K1"], scS["fv1"], k2s, fvs, Fs, exs, eys, 0, True), reps=8)
            ms1, _ = time_calls(lambda: [gS.search_for_triangulation(scS["K1"], scS["fv1"], scS["K2"], scS["fv2"], scS["F12"], scS["ex"], scS["ey"], 0, True) for _ in range(nbS)], reps=8)
            gsr["search_for_triangulation_batch"] = {"neighbours": nbS, "ms_per_call": round(mb, 4), "ms_ten_single_calls": round(ms1, 4)}
            mb, _ = time_calls(lambda: gS.fuse_search_batch([scS["K2"]] * nbS, 0, [pose15S] * nbS, scS["K"], scS["bf"], scS["points"], 3.0), reps=8)
            ms1, _ = time_calls(lambda: [gS.fuse_search(scS["K2"], 0, pose15S, scS["K"], scS["bf"], scS["points"], 3.0) for _ in range(nbS)], reps=8)
            gsr["fuse_search_batch"] = {"targets": nbS, "ms_per_call": round(mb, 4), "ms_ten_single_calls": round(ms1, 4)}
            # ---- round 5: the same searches through KEYFRAME HANDLES (eao_kf_*): the keyframes live in HBM (uploaded once, outside the timed calls -- a keyframe is
            #      searched many times), the vocabulary-node searches and Fuse select on the device; per call: per-call flags / points up, two launches, one table back
            try:
                gH = SR.product_handles()
                h1, h2 = gH.handle(scS["K1"], scS["fv1"]), gH.handle(scS["K2"], scS["fv2"])
                v1S, v2S = (scS["mp1"] >= 0).astype(np.uint8), (scS["mp2"] >= 0).astype(np.uint8)
                hc = [("search_by_bow_kf_frame", lambda: gH.search_by_bow_h(0, h1, v1S, h2, None, 0.75, True)),
                      ("search_by_bow_kf_kf", lambda: gH.search_by_bow_h(1, h1, v1S, h2, v2S, 0.8, True)),
                      ("search_for_triangulation", lambda: gH.search_for_triangulation_h(h1, [h2], [scS["F12"]], [scS["ex"]], [scS["ey"]], 0, True)),
                      ("fuse_search_pose", lambda: gH.fuse_search_h([h2], 0, [pose15S], scS["K"], scS["bf"], scS["points"], 3.0)),
                      ("fuse_search_sim3", lambda: gH.fuse_search_h([h2], 1, [scS["Scw"].ravel()], scS["K"], scS["bf"], scS["points"], 3.0))]
                for name, fn in hc:
                    ms, r = time_calls(fn)
                    gsr[name]["ms_per_call_handles"] = round(ms, 4)
                    gsr[name]["matches_handles"] = int(np.sum(r[0]))
                for name, fn in casesS:      # the list-based searches: frame resident, host replay as before
                    if "ms_per_call_handles" not in gsr[name]:
                        ms, r = time_calls(lambda: fn(gH))
                        gsr[name]["ms_per_call_handles"] = round(ms, 4)
                mb, _ = time_calls(lambda: gH.search_for_triangulation_h(h1, [h2] * nbS, Fs, exs, eys, 0, True), reps=8)
                gsr["search_for_triangulation_batch"]["ms_per_call_handles"] = round(mb, 4)
                mb, _ = time_calls(lambda: gH.fuse_search_h([h2] * nbS, 0, [pose15S] * nbS, scS["K"], scS["bf"], scS["points"], 3.0), reps=8)
                gsr["fuse_search_batch"]["ms_per_call_handles"] = round(mb, 4)
            except Exception as ex:  # noqa: BLE001
                gsr["handles_error"] = repr(ex)
            gsr["note"] = ("median of 12 calls through the ctypes mirror (array wrapping included, ~0.02 ms); keypoints / map points: %d / %d; where a search loses to "
                           "one CPU thread (extra.cpu_guided_searches) it is the upload + launch + download + host replay of a sub-millisecond problem" % (len(scS["K2"]["kp_x"]), len(scS["points"]["active"])))
            extra["guided_searches"] = gsr
        except Exception as ex:  # noqa: BLE001
            extra["guided_searches_error"] = repr(ex)
        # the Frame glue (isInFrustum over a 20 000-point local map) and a small-map BundleAdjustment (12 KF, 10 its)
        from eao_fusion_amd import frame as FR
        rng = np.random.default_rng(11)
        nmp = 20000
        X = rng.uniform([-8, -5, -2], [8, 5, 14], (nmp, 3)).astype(np.float32)
        nv = (X / np.linalg.norm(X, axis=1, keepdims=True)).astype(np.float32)
        dd = np.linalg.norm(X, axis=1).astype(np.float32)
        mpts = dict(active=np.ones(nmp, np.uint8), Xw=X, normal=nv, min_dist_inv=dd * 0.5, max_dist_inv=dd * 2.0, max_dist=dd * 2.5,
                    descriptors=np.zeros((nmp, 32), np.uint8))
        ffr = dict(Tcw=np.eye(4, dtype=np.float32), Ow=np.zeros(3, np.float32), fx=517.3, fy=516.5, cx=318.6, cy=255.3, mbf=40.0, min_x=0,
                   max_x=640, min_y=0, max_y=480, log_scale_factor=float(np.log(np.float32(1.2))))
        FR.is_in_frustum(ffr, mpts, 0.5)
        t0 = time.perf_counter()
        for _ in range(20):
            rr = FR.is_in_frustum(ffr, mpts, 0.5)
        extra["is_in_frustum"] = {"map_points": nmp, "in_view": int(rr["in_view"].sum()), "ms_per_call": round((time.perf_counter() - t0) / 20 * 1e3, 3),
                                  "note": "host buffers in/out (0.6 MB up, 0.4 MB down)"}
        gp = synth.synth_ba(n_free=11, n_fixed=1, n_points=2000, seed=5100)
        E.Optimizer.BundleAdjustment(gp, 10, bRobust=False)
        t0 = time.perf_counter()
        for _ in range(5):
            rg = E.Optimizer.BundleAdjustment(gp, 10, bRobust=False)
        extra["bundle_adjustment"] = {"workload": "BundleAdjustment 11 free + 1 fixed KF x 2000 MP, E=%d, 10 its, no robust kernel" % len(gp["edge_cam"]),
                                      "ms_per_call": round((time.perf_counter() - t0) / 5 * 1e3, 3), "iters": int(rg["iters"][0])}
        # the same extraction on a 256-frame batch (device-resident in and out, like `value`): the dependent-launch chain of
        # the pyramid and the level-synchronous quad-tree amortise over more frames
        try:
            B2 = 256
            fr2 = np.stack([synth.synth_frame(1000 + f, 640, 480) for f in range(B2)])
            d2 = torch.from_numpy(fr2).to(dev)
            ext2 = E.ORBextractor(1000, 1.2, 8, 20, 7)
            cap2 = ext2.max_keypoints(640, 480)
            k2 = torch.zeros((B2, cap2, 28), dtype=torch.uint8, device=dev)
            e2 = torch.zeros((B2, cap2, 32), dtype=torch.uint8, device=dev)
            n2 = torch.zeros(B2, dtype=torch.int32, device=dev)
            st2 = torch.cuda.current_stream().cuda_stream

            def step2():
                ext2.extract_batch_device(d2.data_ptr(), 640, 480, 640, 640 * 480, B2, k2.data_ptr(), e2.data_ptr(), cap2, n2.data_ptr(), st2)
            for _ in range(20):           # (allocations on the first call, then the clocks: see SETTLE_STEPS)
                step2()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(30):
                step2()
            torch.cuda.synchronize()
            dt2 = (time.perf_counter() - t0) / 30
            extra["orb_batch256"] = {"ms_per_step": round(dt2 * 1e3, 4), "kpts_per_s": round(float(n2.sum().item()) / dt2, 1), "frames_per_step": B2}
            del ext2, d2, k2, e2, n2
        except Exception as ex:  # noqa: BLE001
            extra["orb_batch256_error"] = repr(ex)
        # what Tracking.cc sees through the unchanged call signature: ONE 640x480 frame per ORBextractor::operator() call, host
        # image in, host keypoints / descriptors out (H2D + the latency-bound kernel chain + D2H)
        ext1 = E.ORBextractor(1000, 1.2, 8, 20, 7)
        one = synth.synth_frames(1, seed0=1000)
        for _ in range(3):
            ext1.extract_batch(one)
        ts = []
        for _ in range(30):
            t0 = time.perf_counter()
            k1, _d1 = ext1.extract_batch(one)
            ts.append(time.perf_counter() - t0)
        extra["orb_single_frame_host_api"] = {"ms_per_frame": round(min(ts) * 1e3, 4), "keypoints": int(len(k1[0])),
                                              "note": "eao_orb_extract_batch, batch 1, pageable host buffers in and out (min of 30 calls)"}
        del ext1
        # map scale: more free keyframes than one workgroup factorises -> dense Schur system in HBM, panel / update LDL^T
        gm = synth.synth_ba(n_free=200, n_fixed=1, n_points=20000, seed=5300)
        for _ in range(2):      # (the first call sizes the arena and the pinned mirrors, the second still grows the host crew and the set-up's scratch: steady state from the third)
            E.Optimizer.BundleAdjustment(gm, 10, bRobust=False)
        tsm = []
        for _ in range(5):
            t0 = time.perf_counter()
            rm = E.Optimizer.BundleAdjustment(gm, 10, bRobust=False)
            tsm.append(time.perf_counter() - t0)
        dtm = float(np.median(tsm))
        trials = int(np.sum(rm["trace"]["trials"]))
        Em = len(gm["edge_cam"])
        extra["bundle_adjustment_map_scale"] = {"workload": "BundleAdjustment 200 free + 1 fixed KF x 20000 MP, E=%d, 10 its (1200 x 1200 reduced camera system; every point is seen by 2-8 consecutive keyframes, cyclically: 1 608 of 20 301 keyframe pairs are covisible, 74 of 210 tiles live after fill-in)" % Em,
                                                "ms_per_call": round(dtm * 1e3, 3), "ms_per_call_min_max": [round(min(tsm) * 1e3, 3), round(max(tsm) * 1e3, 3)], "timing": "median of 5 calls behind 2 warm-up calls",
                                                "iters": int(rm["iters"][0]), "lm_trials": trials,
                                                "ba_residual_blocks_per_s": round(Em * int(rm["iters"][0]) / dtm, 1),
                                                # the same per-unit figure as the windows (E x 520 + P x 360 bytes per LM trial) over the call's wall time
                                                "frac_hbm": round((Em * 520 + 20000 * 360) * trials / dtm / 1e9 / HBM_PEAK_GBS, 5)}
        # ... and a MAP as a SLAM system produces it (round 5): 1000 keyframes along a trajectory, each covisible with its +-10 neighbours only -- the reduced camera
        # system is a band; the map-scale path stores and factors only the 64 x 64 tiles of that structure (+ fill-in)
        try:
            gb = synth.synth_ba(n_free=1000, n_fixed=1, n_points=50000, seed=5400, band=11)
            for _ in range(2):
                E.Optimizer.BundleAdjustment(gb, 10, bRobust=False)
            free0 = torch.cuda.mem_get_info()[0]
            tsb = []
            for _ in range(5):
                t0 = time.perf_counter()
                rb_ = E.Optimizer.BundleAdjustment(gb, 10, bRobust=False)
                tsb.append(time.perf_counter() - t0)
            dtb_ = float(np.median(tsb))
            Eb = len(gb["edge_cam"])
            trb = int(np.sum(rb_["trace"]["trials"]))
            extra["bundle_adjustment_map_scale_banded"] = {"workload": "BundleAdjustment 1000 free + 1 fixed KF x 50000 MP, E=%d, 10 its; every keyframe covisible with its +-10 neighbours (6000 x 6000 band system)" % Eb,
                                                           "ms_per_call": round(dtb_ * 1e3, 3), "ms_per_call_min_max": [round(min(tsb) * 1e3, 3), round(max(tsb) * 1e3, 3)], "timing": "median of 5 calls behind 2 warm-up calls",
                                                           "iters": int(rb_["iters"][0]), "lm_trials": trb,
                                                           "ba_residual_blocks_per_s": round(Eb * int(rb_["iters"][0]) / dtb_, 1),
                                                           "frac_hbm": round((Eb * 520 + 50000 * 360) * trb / dtb_ / 1e9 / HBM_PEAK_GBS, 5),
                                                           "device_MB_dense_storage_would_need": round(2 * 6016.0 * 6016 * 8 / 1e6 + 50000 * 1001 * 4 / 1e6, 1),
                                                           "note": "device memory of the library's arena for this problem: see tools/dbg_gba_banded.py (126 MB, torch.cuda.mem_get_info around the first call)"}
            del free0
        except Exception as ex:  # noqa: BLE001
            extra["bundle_adjustment_map_scale_banded_error"] = repr(ex)
    except Exception as ex:  # noqa: BLE001
        extra["ba_error"] = repr(ex)
    return extra


def class_surface_problem(path, synth):
    """The inputs of tests/cpp/adapter_bench.cpp at BASELINE.json's sizes: one 640 x 480 frame (configs[1]'s generator), a 1000-correspondence PoseOptimization,
    the configs[3] LocalBundleAdjustment window (20 + 4 keyframes, 3000 map points), a tracked frame pair for the two SearchByProjection variants and a
    keyframe / frame pair filed under 60 vocabulary nodes for SearchByBoW."""
    import struct
    f32 = lambda a: np.ascontiguousarray(a, np.float32).tobytes()      # noqa: E731
    i32 = lambda a: np.ascontiguousarray(a, np.int32).tobytes()        # noqa: E731
    u8 = lambda a: np.ascontiguousarray(a, np.uint8).tobytes()         # noqa: E731
    with open(path, "wb") as f:
        img = synth.synth_frame(1000)
        f.write(struct.pack("<ii", *img.shape)); f.write(img.tobytes())
        pp = synth.synth_pose()
        f.write(struct.pack("<i", len(pp["points"])))
        f.write(f32(pp["Tcw"])); f.write(f32(pp["points"])); f.write(f32(pp["obs"])); f.write(f32(pp["inv_sigma2"]))
        f.write(f32([pp[k] for k in ("fx", "fy", "cx", "cy", "bf")]))
        bp = synth.synth_ba()
        f.write(struct.pack("<iii", len(bp["poses"]), len(bp["points"]), len(bp["edge_cam"])))
        f.write(f32(bp["poses"])); f.write(u8(bp["fixed"])); f.write(f32(bp["points"])); f.write(i32(bp["edge_cam"])); f.write(i32(bp["edge_point"]))
        f.write(f32(bp["obs"])); f.write(f32(bp["inv_sigma2"])); f.write(f32([bp[k] for k in ("fx", "fy", "cx", "cy", "bf")]))
        cur, last, mps = synth.synth_tracking()
        f.write(struct.pack("<i", len(cur["kp_x"])))
        for k in ("kp_x", "kp_y", "kp_angle", "u_right"):
            f.write(f32(cur[k]))
        f.write(i32(cur["kp_octave"])); f.write(u8(cur["descriptors"])); f.write(u8(cur["occupied"])); f.write(f32(cur["Tcw"]))
        f.write(f32([cur[k] for k in ("fx", "fy", "cx", "cy", "mbf", "mb")])); f.write(f32(cur["scale_factors"]))
        f.write(struct.pack("<i", len(last["valid"])))
        f.write(f32(last["Tcw"])); f.write(u8(last["valid"])); f.write(f32(last["Xw"])); f.write(u8(last["descriptors"])); f.write(i32(last["octave"])); f.write(f32(last["angle"]))
        for k in ("proj_x", "proj_y", "proj_xr", "view_cos"):
            f.write(f32(mps[k]))
        f.write(i32(mps["level"])); f.write(u8(mps["skip"]))
        sc = synth.synth_search_scene(n=1000, seed=8100, n_nodes=60)
        for K, mp, fv in ((sc["K1"], sc["mp1"], sc["fv1"]), (sc["K2"], sc["mp2"], sc["fv2"])):
            f.write(struct.pack("<i", len(K["kp_x"])))
            f.write(f32(K["kp_angle"])); f.write(i32(mp)); f.write(u8(K["descriptors"]))
            f.write(struct.pack("<i", len(fv["node_id"])))
            f.write(np.ascontiguousarray(fv["node_id"], np.uint32).tobytes()); f.write(i32(fv["node_start"])); f.write(np.ascontiguousarray(fv["index"], np.uint32).tobytes())


def measure_class_surface(synth):
    """the hot path timed AT THE C++ CLASS SURFACE DESIGN.md declares as the drop-in boundary -- tests/cpp/adapter_bench.cpp, compiled here with
    g++ against include/eaofusion/*.h and libeaofusion_hip.so, drives ORBextractor::operator(), ORBmatcher::SearchByProjection x 2 / SearchByBoW,
    Optimizer::PoseOptimization(Frame*) and Optimizer::LocalBundleAdjustment(KeyFrame*, bool*, Map*) over stand-ins of the SLAM classes (accessors that lock and clone
    as upstream's do) and reports, per call, the wall time and the part of it spent inside the C-ABI."""
    import subprocess
    import tempfile
    tmp = tempfile.mkdtemp(prefix="eao_class_surface_")
    exe, prob = os.path.join(tmp, "adapter_bench"), os.path.join(tmp, "problem.bin")
    lib = os.path.join(ROOT, "eao_fusion_amd")
    cc = subprocess.run(["g++", "-O2", "-std=c++17", "-DEAOFUSION_FORCE_CV_COMPAT", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "adapter_bench.cpp"),
                         "-o", exe, "-L", lib, "-leaofusion_hip", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-pthread"], capture_output=True, text=True)
    if cc.returncode != 0:
        return {"error": "g++: " + cc.stderr[-400:]}
    class_surface_problem(prob, synth)
    run = subprocess.run([exe, prob], capture_output=True, text=True, timeout=300)
    if run.returncode != 0:
        return {"error": "adapter_bench rc %d: %s" % (run.returncode, (run.stdout + run.stderr)[-400:])}
    out = json.loads(run.stdout)
    # the same LocalBundleAdjustment call over a MapPoint that carries the two allocation-free accessors (ForEachObservation,
    # GetWorldPos(float*)) -- the adapter detects them; everything else in the stand-ins is unchanged
    try:
        exe2 = exe + "_edited"
        cc2 = subprocess.run(["g++", "-O2", "-std=c++17", "-DEAOFUSION_FORCE_CV_COMPAT", "-DEAO_BENCH_EDITED_MAPPOINT", "-I", os.path.join(ROOT, "include"),
                              os.path.join(ROOT, "tests", "cpp", "adapter_bench.cpp"), "-o", exe2, "-L", lib, "-leaofusion_hip", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-pthread"],
                             capture_output=True, text=True)
        if cc2.returncode == 0:
            run2 = subprocess.run([exe2, prob, "lba"], capture_output=True, text=True, timeout=300)
            if run2.returncode == 0:
                out["local_bundle_adjustment_with_accessors"] = json.loads(run2.stdout)["local_bundle_adjustment"]
            else:
                out["local_bundle_adjustment_with_accessors"] = {"error": "rc %d: %s" % (run2.returncode, (run2.stdout + run2.stderr)[-300:])}
        else:
            out["local_bundle_adjustment_with_accessors"] = {"error": "g++: " + cc2.stderr[-300:]}
    except Exception as ex:  # noqa: BLE001
        out["local_bundle_adjustment_with_accessors"] = {"error": repr(ex)}
    # Optimizer::BundleAdjustment over a whole map at the class surface (LoopClosing's call with its abort flag): the 1000-keyframe band, unedited MapPoint and row 2c's
    try:
        gmap = os.path.join(tmp, "map.bin")
        if not os.path.exists(gmap):
            mixed_load_inputs(tmp, synth)
        for key, binary in (("bundle_adjustment_map", exe), ("bundle_adjustment_map_with_accessors", exe + "_edited")):
            if os.path.exists(binary):
                rg = subprocess.run([binary, gmap, "gba"], capture_output=True, text=True, timeout=300)
                out[key] = json.loads(rg.stdout)["bundle_adjustment_map"] if rg.returncode == 0 else {"error": "rc %d: %s" % (rg.returncode, (rg.stdout + rg.stderr)[-300:])}
    except Exception as ex:  # noqa: BLE001
        out["bundle_adjustment_map"] = {"error": repr(ex)}
    out["note"] = ("tests/cpp/adapter_bench.cpp (g++ -O2, a process of its own): median wall time of each call through the reference's class signature, and the share of it inside the "
                   "C-ABI entry point the adapter makes (timed by a wrapper around that very call); adapter_overhead = flattening the object graph + writing the result back")
    return out


def mixed_load_inputs(tmp, synth):
    """Inputs of tests/cpp/mixed_load.cpp: adapter_bench's problem.bin (one frame, a PoseOptimization, the configs[3] window, a tracked frame pair), the 25 windows of
    configs[4] and the 1000-keyframe band map as flat eao_ba_problem arrays."""
    import struct
    prob, wins, gmap = (os.path.join(tmp, n) for n in ("problem.bin", "windows.bin", "map.bin"))
    class_surface_problem(prob, synth)

    def flat(f, p, its1, its2):
        f.write(struct.pack("<iiiii", len(p["poses"]), len(p["points"]), len(p["edge_cam"]), its1, its2))
        f.write(np.ascontiguousarray(p["poses"], np.float32).tobytes()); f.write(np.ascontiguousarray(p["fixed"], np.uint8).tobytes())
        f.write(np.ascontiguousarray(p["points"], np.float32).tobytes()); f.write(np.ascontiguousarray(p["edge_cam"], np.int32).tobytes())
        f.write(np.ascontiguousarray(p["edge_point"], np.int32).tobytes()); f.write(np.ascontiguousarray(p["obs"], np.float32).tobytes())
        f.write(np.ascontiguousarray(p["inv_sigma2"], np.float32).tobytes())
        f.write(np.asarray([p[k] for k in ("fx", "fy", "cx", "cy", "bf")], np.float32).tobytes())
    with open(wins, "wb") as f:
        f.write(struct.pack("<i", 25))
        for w in range(25):
            flat(f, synth.synth_ba(seed=6000 + w), 5, 10)
    with open(gmap, "wb") as f:
        flat(f, synth.synth_ba(n_free=1000, n_fixed=1, n_points=50000, seed=5400, band=11), 10, 0)
    return prob, wins, gmap


def measure_mixed_load(synth, frames=1200, period_us=2000, ab=True):
    """the per-frame calls of the Tracking thread timed while LocalMapping's LocalBundleAdjustment and LoopClosing's map-scale BundleAdjustment
    run on the same GPU from other threads (tests/cpp/mixed_load.cpp, a process of its own, built with hipcc).  Run twice: with the library's stream classes
    (latency / background / bulk priorities: the default) and with EAO_STREAM_PRIORITY=0 (every stream at the default priority, as in rounds 1-5)."""
    import subprocess
    import tempfile
    tmp = tempfile.mkdtemp(prefix="eao_mixed_load_")
    exe = os.path.join(tmp, "mixed_load")
    lib = os.path.join(ROOT, "eao_fusion_amd")
    cc = subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-DEAOFUSION_FORCE_CV_COMPAT", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "mixed_load.cpp"),
                         "-o", exe, "-L", lib, "-leaofusion_hip", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-pthread"], capture_output=True, text=True)
    if cc.returncode != 0:
        return {"error": "hipcc: " + cc.stderr[-400:]}
    files = mixed_load_inputs(tmp, synth)
    out = {}
    for name, env in (("priorities", {}), ("no_priorities", {"EAO_STREAM_PRIORITY": "0"})):
        if name == "no_priorities" and not ab:
            break
        run = subprocess.run([exe, *files, str(frames), str(period_us)], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        try:
            out[name] = json.loads(run.stdout)
        except Exception:  # noqa: BLE001
            out[name] = {"error": "mixed_load rc %d: %s" % (run.returncode, (run.stdout + run.stderr)[-400:])}
    out["note"] = ("tests/cpp/mixed_load.cpp: thread T replays one frame's calls every period_us (ORBextractor::operator() at the class surface, then either the device-resident chain "
                   "DeviceTracker::TrackWithMotionModel + TrackLocalMap or the class-surface SearchByProjection x 2 + PoseOptimization x 2); thread L loops "
                   "Optimizer::LocalBundleAdjustment (configs[3], class surface) or eao_local_ba_batch (25 windows); thread G loops the 1000-keyframe eao_bundle_adjustment.  "
                   "frame_ms = T's per-frame latency; results_identical = every frame's and every background call's result equals the idle run's bit for bit")
    return out


def measure_cpu(frames, synth, extra):
    """CPU baseline = the oracle ("port" of the reference arithmetic, g++ -O3 -march=native), 1 thread, on a bounded
    sample of the SAME workload, timed on this box's host cores."""
    from oracle import oracle as O
    O.build()
    orc = O.OrbOracle(1000, 1.2, 8, 20, 7)
    orc.extract(frames[0])  # warm
    budget, nk, nf = 12.0, 0, 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget and nf < 4 * len(frames):
        k, _ = orc.extract(frames[nf % len(frames)])
        nk += len(k)
        nf += 1
    dt = time.perf_counter() - t0
    cpu = {"value": round(nk / dt, 1), "unit": "kpts/s", "cores": 1, "kind": "port",
           "sample": "%d of the benchmark's synthetic 640x480 frames through oracle/orb_cpu.cpp (1 thread, %.1f s)" % (nf, dt),
           "ms_per_frame": round(dt / nf * 1e3, 2), "host_cpus": os.cpu_count()}
    ex = {}
    try:
        # frame-parallel variant on the host cores (the reported baseline stays the 1-thread figure -- the
        # reference extracts one image per thread; this is the best a CPU box could do on a batch of independent frames)
        import threading
        nth = max(1, min(os.cpu_count() or 1, 32))
        done = [0] * nth
        stop_at = time.perf_counter() + 4.0
        def _w(i):
            o = O.OrbOracle(1000, 1.2, 8, 20, 7)
            j = i
            while time.perf_counter() < stop_at:
                k, _ = o.extract(frames[j % len(frames)])
                done[i] += len(k)
                j += nth
        t0 = time.perf_counter()
        ths = [threading.Thread(target=_w, args=(i,)) for i in range(nth)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        ex["cpu_orb_frame_parallel"] = {"value": round(sum(done) / (time.perf_counter() - t0), 1), "unit": "kpts/s", "cores": nth, "kind": "port"}
    except Exception as e:  # noqa: BLE001
        ex["cpu_orb_frame_parallel_error"] = repr(e)
    try:
        p = synth.synth_ba()
        O.local_ba(p)
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 5.0:
            r = O.local_ba(p)
            reps += 1
        dt = (time.perf_counter() - t0) / reps
        lin = int(r["trace"]["trials"].sum()) if len(r["trace"]["trials"]) else 0  # error passes; linearisations = outer its
        outer = int(r["iters"].sum())
        ex["cpu_ba"] = {"ms_per_lba": round(dt * 1e3, 3), "ba_residual_blocks_per_s": round(len(p["edge_cam"]) * outer / dt, 1),
                        "cores": 1, "kind": "port", "sample": "%d x oracle/lm_cpu.cpp LocalBundleAdjustment on the same window" % reps,
                        "lm_trials": lin}
        a, b = synth.synth_descriptors(1000, 2000)
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 3.0:
            O.hamming_matrix(a, b)
            reps += 1
        dt = (time.perf_counter() - t0) / reps
        ex["cpu_hamming_matrix"] = {"pair_distances_per_s": round(1e6 / dt, 1), "cores": 1, "kind": "port",
                                    "note": "the reference's own SWAR bit count (src/ORBmatcher.cc:1649-1665), ~8x slower than a popcnt loop: faithful to the reference, not the best a CPU can do"}
        curf, lastf, mpsf = synth.synth_tracking()
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 2.0:
            O.search_by_projection_points(curf, mpsf, 1.0, 0.8)
            O.search_by_projection_frames(curf, lastf, 7.0, False, True)
            reps += 1
        ex["cpu_guided_matching"] = {"ms_per_frame_both_searches": round((time.perf_counter() - t0) / reps * 1e3, 3), "cores": 1, "kind": "port"}
        # the nine remaining guided searches on the oracle (oracle/search_cpu.cpp, 1 thread), the same calls measure_extra() times on the product
        try:
            oS = O.search_binding()
            _sc, _p15, casesS = search_cases(synth)
            cs = {}
            for name, fn in casesS:
                ms, r = time_calls(lambda: fn(oS), reps=8)
                g = extra.get("guided_searches", {}).get(name, {})
                cs[name] = {"ms_per_call": round(ms, 4), "matches": int(r[0]),
                            "gpu_over_cpu_time": None if "ms_per_call" not in g else round(g["ms_per_call"] / ms, 2),
                            "gpu_handles_over_cpu_time": None if "ms_per_call_handles" not in g else round(g["ms_per_call_handles"] / ms, 2)}
            cs["note"] = ("oracle/search_cpu.cpp, 1 thread, median of 8 calls through its ctypes binding; gpu_over_cpu_time > 1: the product's call (upload + kernel + download + "
                          "host replay) takes LONGER than one CPU thread on this 1000-point problem")
            ex["cpu_guided_searches"] = cs
        except Exception as e2:  # noqa: BLE001
            ex["cpu_guided_searches_error"] = repr(e2)
        pp = synth.synth_pose()
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 2.0:
            O.pose_optimization(pp)
            reps += 1
        ex["cpu_pose_optimization_ms"] = round((time.perf_counter() - t0) / reps * 1e3, 3)
        gp = synth.synth_ba(n_free=11, n_fixed=1, n_points=2000, seed=5100)
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 2.0:
            O.bundle_adjustment(gp, 10, False)
            reps += 1
        ex["cpu_bundle_adjustment_ms"] = round((time.perf_counter() - t0) / reps * 1e3, 3)
        gm = synth.synth_ba(n_free=200, n_fixed=1, n_points=20000, seed=5300)
        t0 = time.perf_counter()
        O.bundle_adjustment(gm, 10, False)
        ex["cpu_bundle_adjustment_map_scale_ms"] = round((time.perf_counter() - t0) * 1e3, 1)   # (dense LDL^T on the CPU; g2o's sparse solver would be faster)
    except Exception as e:  # noqa: BLE001
        ex["cpu_extra_error"] = repr(e)
    return cpu, ex


if __name__ == "__main__":
    sys.exit(main())
