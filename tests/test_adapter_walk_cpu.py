"""The LocalBundleAdjustment adapter's walk WITHOUT a GPU (tests/cpp/adapter_bench.cpp `lba-walk`: the library call is replaced by an identity result and the problem the
adapter hands over is dumped): over stand-ins with upstream's locking / copying accessors -- and over the optional allocation-free ones of INTEGRATION.md row 2c -- the
flattened problem must be the window the objects were built from: cameras in ascending mnId, points in ascending mnId, every edge once, a point's edges side by side
in the order of its observation map (include/eaofusion/OptimizerImpl.h; reference src/Optimizer.cc:680-905)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("edited", [False, True], ids=["upstream_accessors", "row_2c_accessors"])
def test_lba_adapter_flattens_the_window_it_was_given(tmp_path, edited):
    import sys
    sys.path.insert(0, ROOT)
    import bench
    from eao_fusion_amd import synth
    prob = str(tmp_path / "problem.bin")
    bench.class_surface_problem(prob, synth)
    exe = str(tmp_path / "adapter_bench")
    lib = os.path.join(ROOT, "eao_fusion_amd")
    cc = subprocess.run(["g++", "-O1", "-std=c++17", "-DEAOFUSION_FORCE_CV_COMPAT"] + (["-DEAO_BENCH_EDITED_MAPPOINT"] if edited else []) +
                        ["-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "adapter_bench.cpp"), "-o", exe, "-L", lib, "-leaofusion_hip",
                         "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-pthread"], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr[-2000:]
    dump = str(tmp_path / "walk.bin")
    run = subprocess.run([exe, prob, "lba-walk"], capture_output=True, text=True, env=dict(os.environ, EAO_WALK_DUMP=dump), timeout=300)
    assert run.returncode == 0, run.stdout[-500:] + run.stderr[-1500:]
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
    bp = synth.synth_ba()
    P, F = np.asarray(bp["poses"], np.float32), np.asarray(bp["fixed"], np.uint8)
    # the window as upstream defines it (src/Optimizer.cc:696-738): the map points the LOCAL keyframes see -- a point only fixed cameras observe is not part of it
    ec_in, ep_in = np.asarray(bp["edge_cam"]), np.asarray(bp["edge_point"])
    seen = np.zeros(len(bp["points"]), bool); seen[ep_in[F[ec_in] == 0]] = True
    rank = np.cumsum(seen) - 1
    keep = seen[ep_in]
    assert (nc, npt, ne) == (len(P), int(seen.sum()), int(keep.sum()))
    # adapter_bench numbers the keyframes c + (fixed ? 0 : 100): fixed cameras first, each group in input order
    order = np.argsort(np.arange(len(P)) + np.where(F != 0, 0, 100), kind="stable")
    assert np.array_equal(cams, P[order].reshape(nc, 4, 4)) and np.array_equal(fixed, F[order])
    assert np.array_equal(pts, np.asarray(bp["points"], np.float32)[seen])
    newcam = np.empty(len(P), np.int64); newcam[order] = np.arange(len(P))
    want = {}
    for e in np.flatnonzero(keep):
        want[(int(newcam[ec_in[e]]), int(rank[ep_in[e]]))] = (np.asarray(bp["obs"][e], np.float32).tobytes(), np.float32(bp["inv_sigma2"][e]).tobytes())
    got = {(int(ecam[e]), int(ept[e])): (obs[e].tobytes(), inv[e].tobytes()) for e in range(ne)}
    assert len(got) == ne and got == want
    # a point's edges side by side; every point once
    change = np.flatnonzero(np.diff(ept) != 0)
    assert len(change) + 1 == len(np.unique(ept)) == npt
