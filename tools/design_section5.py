"""Renders the numeric tables of DESIGN.md section 5 from the round's committed evidence files and writes them between the
   <!-- MEASURED:BEGIN --> / <!-- MEASURED:END --> markers (the prose around the markers is hand-written).
       python tools/design_section5.py [r06]
   Inputs: profiles/<R>_bench_line.json (python bench.py, default flags), profiles/<R>_kernel_stats_bench.csv (rocprofv3 --kernel-trace --stats of
   the same command), profiles/<R>_pmc_traffic.json, profiles/<R>_ba_pmc_traffic.json, profiles/<R>_ba_{single,batch}_kernel_stats.csv."""
import csv, json, os, sys

R = sys.argv[1] if len(sys.argv) > 1 else "r06"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda name: os.path.join(ROOT, "profiles", "%s_%s" % (R, name))


def stats(path):
    """kernel,calls,avg_us,min_us,max_us,total_ms,pct -- kernel names hold commas (template arguments), so the six numbers are split off from the right"""
    rows = {}
    if not os.path.exists(path):
        return rows
    for ln in open(path):
        if ln.startswith("#") or ln.startswith("kernel,"):
            continue
        parts = ln.rstrip("\n").rsplit(",", 6)
        if len(parts) == 7:
            rows[parts[0]] = dict(zip(("kernel", "calls", "avg_us", "min_us", "max_us", "total_ms", "pct"), parts))
    return rows


def fmt(v, nd=3):
    if v is None:
        return "—"
    if isinstance(v, float):
        return ("%%.%df" % nd) % v
    return str(v)


def main():
    line = json.load(open(P("bench_line.json")))
    ex = line["extra"]
    rf = dict(ex.get("flat", {}), **line["roofline"])      # (round 6: the driver-visible head of the record in `roofline`, the other flat figures in extra.flat)
    out = []
    w = out.append
    w("Source: `profiles/%s_bench_line.json` (`python bench.py`, default flags, HEAD of the round's last kernel edit).  Every row can be recomputed from the"
      " flat scalars of `roofline` in the driver's `BENCH_%s.json`.\n" % (R, R))
    w("| Quantity | Value | Verdict-r5 target (round 5's figure) | Where it is measured |")
    w("|---|---|---|---|")
    rows = [
        ("ORB step, 64 frames (`value`)", "%.4f ms = %.1f M kpts/s (cold inputs %.4f ms)" % (line["ms_per_step"], line["value"] / 1e6, line["ms_per_step_cold"]), "—", "HIP events around K steps, inputs resident"),
        ("dominant kernel `k_fast_cells`", "%.1f µs, %.0f GB/s = %.3f of 8 TB/s" % (rf["avg_launch_ms"] * 1e3, rf["achieved"], rf["frac"]), "—", "stage events on the kernel's own stream"),
        ("LocalBundleAdjustment, one window", "%.3f ms (device %.3f), %.1f×  the scalar port (%.2f ms)" % (rf["ba_single_ms"], rf["ba_single_device_ms"], rf["ba_gpu_over_cpu"], rf["ba_cpu_ms"]), "— (1.013)", "C-ABI `eao_local_ba`"),
        ("LocalBundleAdjustment, 25 windows", "%.3f ms, %.2f × 10⁹ residual blocks/s, frac %.3f" % (rf["ba_batched_ms"], rf["ba_batched_residual_blocks_per_s"] / 1e9, rf["ba_batched_frac"]), "— (2.64)", "C-ABI `eao_local_ba_batch`"),
        ("BundleAdjustment 200 KF × 20 k MP (points seen by 2–8 consecutive keyframes, cyclic)", "%.2f ms, frac %.4f" % (rf["ba_map_scale_ms"], rf["ba_map_scale_frac"]), "≤ 6 (8.96)", "`eao_bundle_adjustment`"),
        ("BundleAdjustment 1000 KF × 50 k MP (±10 band)", "%.2f ms, frac %.4f" % (rf["ba_map_scale_banded_ms"], rf["ba_map_scale_banded_frac"]), "≤ 15 (31.9)", "`eao_bundle_adjustment`"),
        ("PoseOptimization, 1000 correspondences", "%.1f µs at the C-ABI (CPU port %.0f µs)" % (rf["pose_opt_us"], rf["pose_opt_cpu_us"]), "— (199)", "`eao_pose_optimization`"),
        ("tracked frame: motion model + local map", "%.4f ms (%.4f + %.4f)" % (rf["track_frame_ms"], rf["track_motion_model_ms"], rf["track_local_map_ms"]), "— (0.487)", "`eao_tracker_*`, polled done word"),
        ("Hamming 1000 × 1000 matrix (×64 pairs)", "%.1f µs, frac %.3f" % (rf["hamming_matrix_us"], rf["hamming_matrix_frac"]), "—", "`eao_hamming_matrix_device`"),
    ]
    rows.append(("guided searches on keyframe handles: `search_by_bow` kf-frame / kf-kf, `search_for_triangulation`, `fuse_search`", " / ".join("%.4f" % rf["gs_%s_handles_ms" % k] for k in
                 ("search_by_bow_kf_frame", "search_by_bow_kf_kf", "search_for_triangulation", "fuse_search_pose")) + " ms = " + " / ".join("%.2f" % rf["gs_%s_handles_over_cpu" % k] for k in
                 ("search_by_bow_kf_frame", "search_by_bow_kf_kf", "search_for_triangulation", "fuse_search_pose")) + " × one CPU thread", "—", "ctypes mirror, median of 12"))
    rows += [
        ("`search_for_triangulation_batch` (10 neighbours, handles)", "%.4f ms (upload form %.3f)" % (rf["gs_triangulation_batch10_handles_ms"], rf["gs_triangulation_batch10_ms"]), "—", "″"),
        ("`fuse_search_batch` (10 targets, handles)", "%.4f ms (upload form %.3f)" % (rf["gs_fuse_batch10_handles_ms"], rf["gs_fuse_batch10_ms"]), "—", "″"),
        ("class surface `ORBextractor::operator()`", "%.4f ms (C-ABI share %.4f); with the lazy pyramid %.4f" % (rf["cs_orb_call_ms"], rf["cs_orb_call_c_abi_ms"], rf["cs_orb_call_with_pyramid_ms"]), "—", "`tests/cpp/adapter_bench.cpp`"),
        ("class surface `Optimizer::PoseOptimization(Frame*)`", "%.4f ms (C-ABI %.4f)" % (rf["cs_pose_opt_ms"], rf["cs_pose_opt_c_abi_ms"]), "—", "″"),
        ("class surface `Optimizer::LocalBundleAdjustment`", "%.3f ms (C-ABI %.3f; a bare walk of the same accessors %.3f)" % (
            rf["cs_lba_ms"], rf["cs_lba_c_abi_ms"], ex["class_surface"]["local_bundle_adjustment"]["reference_accessor_walk_ms"]), "1.97 without the edit", "″"),
        ("... over a `MapPoint` with the two optional accessors (INTEGRATION.md row 2c)", "%.3f ms (C-ABI %.3f)" % (
            rf["cs_lba_observations_ref_ms"], ex["class_surface"]["local_bundle_adjustment_with_accessors"]["c_abi_ms"]), "≤ 1.3 with the edit", "″, built with `-DEAO_BENCH_EDITED_MAPPOINT`"),
        ("class surface `Optimizer::BundleAdjustment`, 1000 KF × 50 k MP map with LoopClosing's abort flag", "%.1f ms (C-ABI %.2f)" % (
            rf.get("cs_map_ba_ms", float("nan")), rf.get("cs_map_ba_c_abi_ms", float("nan"))), "— (101 ms with round 5's adapter)", "`adapter_bench <map.bin> gba`"),
        ("class surface `SearchByProjection` ×2 / `SearchByBoW`", "%.4f / %.4f / %.4f ms (C-ABI %.4f / %.4f / %.4f)" % (
            rf["cs_sbp_local_map_ms"], rf["cs_sbp_last_frame_ms"], rf["cs_sbow_ms"], rf["cs_sbp_local_map_c_abi_ms"], rf["cs_sbp_last_frame_c_abi_ms"], rf["cs_sbow_c_abi_ms"]), "—", "″"),
    ]
    for r in rows:
        w("| %s | %s | %s | %s |" % r)
    ml = ex.get("mixed_load", {})
    if "priorities" in ml and "error" not in ml["priorities"]:
        w("\nMixed load (`extra.mixed_load`, `tests/cpp/mixed_load.cpp`): thread T replays one frame's calls every %d µs (%d frames per scenario) while LocalMapping's and LoopClosing's"
          " calls loop on other threads of the same process.  Per-frame latency of T in ms; \"stream classes\" = the library's default, \"none\" = `EAO_STREAM_PRIORITY=0` (rounds 1–5)."
          "  Every frame's and every background call's result equals the idle run's bit for bit: **%s**.\n" % (ml["priorities"]["period_us"], ml["priorities"]["frames"], ml["priorities"]["results_identical"]))
        w("| T's calls | beside | stream classes: p50 / p99 / max | p99 ÷ idle p50 | none: p50 / p99 / max | background call, p50 (alone) |")
        w("|---|---|---|---|---|---|")
        names = {"idle": "nothing", "beside_lba": "`Optimizer::LocalBundleAdjustment` loop (configs[3], class surface)", "beside_lba_batch25": "`eao_local_ba_batch` loop (25 windows)",
                 "beside_lba_and_map_ba": "LBA loop + 1000-keyframe `eao_bundle_adjustment` loop"}
        for var, label in (("device_chain", "`ORBextractor::operator()` + `DeviceTracker::TrackWithMotionModel` + `TrackLocalMap`"),
                           ("class_surface", "`operator()` + `SearchByProjection` ×2 + `PoseOptimization` ×2 (class surface)")):
            idle = ml["priorities"][var]["idle"]["frame_ms"]["p50"]
            for sc in ("idle", "beside_lba", "beside_lba_batch25", "beside_lba_and_map_ba"):
                a = ml["priorities"][var][sc]
                b = ml.get("no_priorities", {}).get(var, {}).get(sc)
                fa = a["frame_ms"]
                bg = []
                for k, al in (("lba_class_surface_ms", "lba_class_surface_ms"), ("lba_batch25_ms", "lba_batch25_ms"), ("map_ba_ms", "map_ba_ms")):
                    if k in a:
                        bg.append("%s %.2f (%.2f)" % (k[:-3], a[k]["p50"], ml["priorities"]["alone"][al]["p50"]))
                w("| %s | %s | %.3f / %.3f / %.3f | %.2f | %s | %s |" % (label if sc == "idle" else "″", names[sc], fa["p50"], fa["p99"], fa["max"], fa["p99"] / idle,
                                                                      "%.3f / %.3f / %.3f" % (b["frame_ms"]["p50"], b["frame_ms"]["p99"], b["frame_ms"]["max"]) if b else "—", ", ".join(bg) or "—"))
    # ---- kernel table
    ks = stats(P("kernel_stats_bench.csv"))
    w("\nPer-kernel durations, `profiles/%s_kernel_stats_bench.csv` (`rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 3`; the ORB rows mix the"
      " 64-frame batches with the single-frame and 256-frame calls of the extras, the stage events of the table above are the 64-frame figures):\n" % R)
    w("| Kernel | calls | avg µs | min µs | max µs | % of GPU time |")
    w("|---|---|---|---|---|---|")
    for name, r in list(ks.items())[:8]:
        w("| `%s` | %s | %s | %s | %s | %.1f |" % (name, r["calls"], r["avg_us"], r["min_us"], r["max_us"], float(r["pct"])))
    # ---- traffic
    if os.path.exists(P("pmc_traffic.json")):
        t = json.load(open(P("pmc_traffic.json")))
        alg = {"pyramid": 64 * 1569878, "fast": rf["algorithmic_bytes_per_launch"], "blur": 64 * 1901064, "quadtree": None, "orient_describe": None}
        w("\nCounter traffic of the ORB stages per 64-frame step, `profiles/%s_pmc_traffic.json` (separate `--pmc FETCH_SIZE` / `WRITE_SIZE` passes; HBM bytes = 2 × FETCH_SIZE + WRITE_SIZE,"
          " the gfx950 correction of `MI355X_MICROARCH.md`):\n" % R)
        w("| Stage | kernel | launches | 2·FETCH + WRITE (MB) | algorithmic (MB) | ratio |")
        w("|---|---|---|---|---|---|")
        for st, v in t["kernels"].items():
            a = alg.get(st)
            hb = v["hbm_bytes_per_step_corrected"]
            w("| %s | `%s` | %s | %.1f | %s | %s |" % (st, v["kernel"], v["launches_per_step"], hb / 1e6, "%.1f" % (a / 1e6) if a else "—", "%.2f" % (hb / a) if a else "—"))
    if os.path.exists(P("ba_pmc_traffic.json")):
        t = json.load(open(P("ba_pmc_traffic.json")))
        w("\nCounter traffic of the LM launches, `profiles/%s_ba_pmc_traffic.json`: batched %.1f MB per LM iteration against %.1f MB algorithmic = **%.2f×**; one window %.1f MB against %.2f MB = **%.2f×**."
          % (R, t["batched"]["hbm_bytes_per_iteration"] / 1e6, rf["ba_batched_bytes_per_iteration"] / 1e6, t["batched"]["hbm_bytes_per_iteration"] / rf["ba_batched_bytes_per_iteration"],
             t["single_window"]["hbm_bytes_per_iteration"] / 1e6, rf["ba_single_bytes_per_iteration"] / 1e6, t["single_window"]["hbm_bytes_per_iteration"] / rf["ba_single_bytes_per_iteration"]))
    for tag, fn in (("one window", "ba_single_kernel_stats.csv"), ("25 windows", "ba_batch_kernel_stats.csv")):
        ks = stats(P(fn))
        if ks:
            w("\nLM kernels, %s (`profiles/%s_%s`): " % (tag, R, fn) + "; ".join("`%s` %s µs × %s" % (k, r["avg_us"], r["calls"]) for k, r in list(ks.items())[:5]) + ".")
    text = "\n".join(out) + "\n"
    path = os.path.join(ROOT, "DESIGN.md")
    src = open(path).read()
    b, e = "<!-- MEASURED:BEGIN -->", "<!-- MEASURED:END -->"
    if b not in src:
        raise SystemExit("markers missing in DESIGN.md")
    head, rest = src.split(b, 1)
    _, tail = rest.split(e, 1)
    open(path, "w").write(head + b + "\n" + text + e + tail)
    print("DESIGN.md section 5 tables rewritten from profiles/%s_* (%d lines)" % (R, len(out)))


if __name__ == "__main__":
    main()
