"""Optimizer::BundleAdjustment's adapter over a whole map WITHOUT a GPU (tests/cpp/adapter_bench.cpp `gba-walk`: the library call replaced by an identity result, the
problem handed over dumped): cameras and points in ascending mnId, every observation once, edges point after point in ascending order (what the map-scale set-up's parallel
passes need) whatever order Map::GetAllMapPoints() delivers (the bench shuffles it)."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("edited,n_points,tsan", [(False, 2500, False), (True, 2500, False), (True, 13000, True)],
                         ids=["upstream_accessors", "row_2c_accessors", "three_walk_threads_under_tsan"])
def test_map_adapter_flattens_the_map_in_ascending_order(tmp_path, edited, n_points, tsan):
    """(third case: 13 000 map points = three threads on the read side of the walk, the binary built with -fsanitize=thread: the stand-ins lock as upstream's classes do, the
    adapter must not add a race of its own)"""
    import sys
    sys.path.insert(0, ROOT)
    from eao_fusion_amd import synth
    p = synth.synth_ba(n_free=60, n_fixed=1, n_points=n_points, seed=5770, band=7)
    path = str(tmp_path / "map.bin")
    with open(path, "wb") as f:
        f.write(struct.pack("<iiiii", len(p["poses"]), len(p["points"]), len(p["edge_cam"]), 10, 0))
        for k, dt in (("poses", np.float32), ("fixed", np.uint8), ("points", np.float32), ("edge_cam", np.int32), ("edge_point", np.int32), ("obs", np.float32), ("inv_sigma2", np.float32)):
            f.write(np.ascontiguousarray(p[k], dt).tobytes())
        f.write(np.asarray([p[k] for k in ("fx", "fy", "cx", "cy", "bf")], np.float32).tobytes())
    exe = str(tmp_path / "adapter_bench")
    lib = os.path.join(ROOT, "eao_fusion_amd")
    cc = subprocess.run(["g++", "-O1", "-std=c++17", "-DEAOFUSION_FORCE_CV_COMPAT"] + (["-DEAO_BENCH_EDITED_MAPPOINT"] if edited else []) + (["-g", "-fsanitize=thread"] if tsan else []) +
                        ["-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "adapter_bench.cpp"), "-o", exe, "-L", lib, "-leaofusion_hip",
                         "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-pthread"], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr[-2000:]
    dump = str(tmp_path / "walk.bin")
    env = {k: v for k, v in os.environ.items() if k not in ("LD_PRELOAD", "ASAN_OPTIONS", "UBSAN_OPTIONS")} if tsan else dict(os.environ)
    run = subprocess.run([exe, path, "gba-walk"], capture_output=True, text=True, env=dict(env, EAO_WALK_DUMP=dump, TSAN_OPTIONS="halt_on_error=1"), timeout=600)
    assert run.returncode == 0 and "ThreadSanitizer" not in run.stderr, run.stdout[-500:] + run.stderr[-3000:]
    raw = open(dump, "rb").read()
    nc, npt, ne = np.frombuffer(raw, np.int32, 3)
    off = 12
    def take(dtype, n):
        nonlocal off
        a = np.frombuffer(raw, dtype, n, off); off += a.nbytes
        return a
    cams = take(np.float32, 16 * nc).reshape(nc, 4, 4); fixed = take(np.uint8, nc); pts = take(np.float32, 3 * npt).reshape(npt, 3)
    ecam = take(np.int32, ne); ept = take(np.int32, ne); obs = take(np.float32, 3 * ne).reshape(ne, 3); inv = take(np.float32, ne)
    assert off == len(raw)
    P, F = np.asarray(p["poses"], np.float32), np.asarray(p["fixed"], np.uint8)
    assert (nc, npt, ne) == (len(P), len(p["points"]), len(p["edge_cam"]))
    # adapter_bench's map section: the fixed keyframe is mnId 0, the others c + 1 -- fixed first, then input order
    order = np.argsort(np.where(F != 0, 0, np.arange(len(P)) + 1), kind="stable")
    newcam = np.empty(len(P), np.int64); newcam[order] = np.arange(len(P))
    assert np.array_equal(cams, P[order].reshape(nc, 4, 4)) and np.array_equal(fixed, F[order])
    assert np.array_equal(pts, np.asarray(p["points"], np.float32))
    assert np.all(np.diff(ept) >= 0), "edges are not grouped by ascending point index"
    want = {(int(newcam[p["edge_cam"][e]]), int(p["edge_point"][e])): (np.asarray(p["obs"][e], np.float32).tobytes(), np.float32(p["inv_sigma2"][e]).tobytes()) for e in range(ne)}
    got = {(int(ecam[e]), int(ept[e])): (obs[e].tobytes(), inv[e].tobytes()) for e in range(ne)}
    assert len(got) == ne and got == want
