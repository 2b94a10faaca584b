"""The header-only C++ adapters (include/eaofusion/) compiled with g++ against small stand-ins of the reference's
Frame / KeyFrame / MapPoint / Map, driven the way Tracking.cc / LocalMapping.cc drive the reference classes.  Results
must equal what the (already parity-tested) C-ABI returns through the Python mirror."""
import os
import struct
import subprocess

import numpy as np
import pytest

from eao_fusion_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_adapters_end_to_end(tmp_path):
    import eao_fusion_amd as E
    exe = str(tmp_path / "adapter_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-DEAOFUSION_FORCE_CV_COMPAT", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "adapter_test.cpp"), "-o", exe,
                           "-L", os.path.join(ROOT, "eao_fusion_amd"), "-leaofusion_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "eao_fusion_amd"), "-Wl,-rpath,/opt/rocm/lib", "-pthread"])
    img = synth.synth_frame(1234)
    pp = synth.synth_pose(n=400, seed=4100)
    bp = synth.synth_ba(n_free=6, n_fixed=3, n_points=400, seed=3100, mono_frac=0.2)
    prob = str(tmp_path / "problem.bin")
    with open(prob, "wb") as f:
        f.write(struct.pack("<ii", *img.shape)); f.write(img.tobytes())
        f.write(struct.pack("<i", len(pp["points"])))
        for k in ("Tcw", "points", "obs", "inv_sigma2"):
            f.write(np.ascontiguousarray(pp[k], np.float32).tobytes())
        f.write(np.array([pp[k] for k in ("fx", "fy", "cx", "cy", "bf")], np.float32).tobytes())
        f.write(struct.pack("<iii", len(bp["poses"]), len(bp["points"]), len(bp["edge_cam"])))
        f.write(bp["poses"].tobytes()); f.write(bp["fixed"].tobytes()); f.write(bp["points"].tobytes())
        f.write(bp["edge_cam"].tobytes()); f.write(bp["edge_point"].tobytes()); f.write(bp["obs"].tobytes()); f.write(bp["inv_sigma2"].tobytes())
        f.write(np.array([bp[k] for k in ("fx", "fy", "cx", "cy", "bf")], np.float32).tobytes())
    res = str(tmp_path / "result.bin")
    out = subprocess.run([exe, prob, res], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    buf = open(res, "rb").read()
    off = 0

    def take(dtype, n):
        nonlocal off
        a = np.frombuffer(buf, dtype=dtype, count=n, offset=off)
        off += a.nbytes
        return a
    # ---- ORBextractor
    nk = int(take(np.int32, 1)[0])
    kps = take(E.KP_DTYPE, nk)
    desc = take(np.uint8, nk * 32).reshape(nk, 32)
    pyr = take(np.int32, 3)
    dd = int(take(np.int32, 1)[0])
    untouched = int(take(np.int32, 1)[0])
    ext = E.ORBextractor(1000, 1.2, 8, 20, 7)
    k2, d2 = ext(img)
    assert np.array_equal(kps, k2) and np.array_equal(desc, d2)
    assert list(pyr) == [480, 640, 179]               # mvImagePyramid[l] is the w x h view (border lives around it)
    assert dd == int(np.unpackbits(desc[0] ^ desc[1]).sum())
    assert untouched == 3
    # ---- ORBmatcher::SearchByProjection over the extractor's own keypoints: (almost) every point re-finds itself
    nm, self_hits = (int(x) for x in take(np.int32, 2))
    assert nm > 0.9 * nk and self_hits > 0.9 * nm
    # ---- PoseOptimization
    inl = int(take(np.int32, 1)[0])
    Tcw = take(np.float32, 16).reshape(4, 4)
    outl = take(np.uint8, len(pp["points"]))
    r = E.Optimizer.PoseOptimization(pp)
    assert inl == r["n_inliers"] and np.array_equal(outl, r["outlier"]) and np.array_equal(Tcw, r["Tcw"])
    # ---- LocalBundleAdjustment (edge insertion order differs from the flat problem => rounding-level differences only)
    nc, npnt = len(bp["poses"]), len(bp["points"])
    poses = take(np.float32, nc * 16).reshape(nc, 4, 4)
    pts = take(np.float32, npnt * 3).reshape(npnt, 3)
    erased, normals, same = (int(x) for x in take(np.int32, 3))
    rb = E.Optimizer.LocalBundleAdjustment(bp)
    assert np.allclose(poses, rb["poses"], rtol=0, atol=2e-6)
    # the reference only gathers map points matched in a LOCAL keyframe (src/Optimizer.cc:693-719): points seen by fixed
    # cameras alone are not part of the window the adapter builds, and keep their value
    free_seen = np.zeros(npnt, bool)
    free_seen[bp["edge_point"][~bp["fixed"].astype(bool)[bp["edge_cam"]]]] = True
    assert free_seen.sum() > 0.9 * npnt
    assert np.allclose(pts[free_seen], rb["points"][free_seen], rtol=0, atol=2e-5)
    assert np.array_equal(pts[~free_seen], bp["points"][~free_seen])
    assert normals == int(free_seen.sum()) and same == 1
    assert erased > 0
    f = bp["fixed"].astype(bool)
    assert np.array_equal(poses[f], bp["poses"][f])    # fixed keyframes are never written back
