"""CPU suite (no GPU): the HOST half of the guided searches -- eao_fusion_amd/csrc/search.hip, the per-query geometry and the order-dependent selection loops
of all eleven searches -- compiled unchanged as plain C++ against a list provider that asks the ORACLE for the candidate lists the product's kernels would
deliver (tests/cpp/host_replay_provider.cpp, oracle/match_cpu.cpp::orc_candidate_lists), and driven through the same C entry points:
  * every search must give the oracle's match table, bit for bit (the replay code is thereby checked apart from the kernels);
  * malformed input -- an octave beyond the pyramid, NaN / Inf in a pose or fundamental matrix, a feature-vector index beyond the frame, prior octaves out of
    range, empty frames -- must come back as EAO_ERR_INVALID, never as a fault.
EAO_HOST_SAN=1 (tools/run_sanitizers.sh) builds the library with AddressSanitizer + UndefinedBehaviorSanitizer: caller-supplied indices then meet ASan."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from eao_fusion_amd import _lib, search, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]


@pytest.fixture(scope="module")
def host(tmp_path_factory, oracle):
    """search.hip as C++ + the oracle-backed list provider -> a scratch shared object bound like the product's searches."""
    so = str(tmp_path_factory.mktemp("hostreplay") / "libeaosearch_hosttest.so")
    olib = os.environ.get("EAO_ORACLE_LIB", os.path.join(ROOT, "oracle", "liboracle.so"))
    flags = SAN if os.environ.get("EAO_HOST_SAN") else ["-O2"]
    subprocess.check_call(["g++", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas", "-D__HIP_PLATFORM_AMD__",
                           "-I/opt/rocm/include"] + flags + ["-x", "c++", os.path.join(ROOT, "eao_fusion_amd", "csrc", "search.hip"), "-x", "none",
                           os.path.join(ROOT, "tests", "cpp", "host_replay_provider.cpp"), "-o", so, olib, "-Wl,-rpath," + os.path.dirname(olib)])
    L = C.CDLL(so)
    L.eao_last_error.restype = C.c_char_p

    class HostError(Exception):
        pass

    def check(status):
        if status != 0:
            raise HostError("status %d: %s" % (status, L.eao_last_error().decode()))
    b = search.ProductBinding(L, check)
    # the two per-frame searches (argument lists of include/eao_fusion.h, mirrored in eao_fusion_amd/_lib.py)
    for name in ("eao_search_by_projection_points", "eao_search_by_projection_frames"):
        fn = getattr(L, name)
        fn.restype, fn.argtypes = _lib.SYMBOLS[name]
    return b, L, check, HostError


def _points(L, check, frame, mps, th, nnratio):
    from eao_fusion_amd.matcher import make_frame_view
    v, keep = make_frame_view(frame, _lib.FrameView)
    a = {k: np.ascontiguousarray(mps[k], np.float32) for k in ("proj_x", "proj_y", "proj_xr", "view_cos")}
    lvl, desc = np.ascontiguousarray(mps["level"], np.int32), np.ascontiguousarray(mps["descriptors"], np.uint8)
    skip = None if mps.get("skip") is None else np.ascontiguousarray(mps["skip"], np.uint8)
    out, nm = np.full(len(lvl), -1, np.int32), C.c_int32()
    check(L.eao_search_by_projection_points(C.byref(v), len(lvl), _lib.ptr(a["proj_x"]), _lib.ptr(a["proj_y"]), _lib.ptr(a["proj_xr"]), _lib.ptr(a["view_cos"]), _lib.ptr(lvl),
                                            _lib.ptr(desc), _lib.ptr(skip), th, nnratio, _lib.ptr(out), C.byref(nm)))
    return nm.value, out


def _frames(L, check, cur, last, th, mono, check_ori=True):
    from eao_fusion_amd.matcher import make_frame_view
    v, keep = make_frame_view(cur, _lib.FrameView)
    Tc, Tl = np.ascontiguousarray(cur["Tcw"], np.float32), np.ascontiguousarray(last["Tcw"], np.float32)
    valid, Xw = np.ascontiguousarray(last["valid"], np.uint8), np.ascontiguousarray(last["Xw"], np.float32)
    desc, octv, ang = np.ascontiguousarray(last["descriptors"], np.uint8), np.ascontiguousarray(last["octave"], np.int32), np.ascontiguousarray(last["angle"], np.float32)
    out, nm = np.full(v.n, -1, np.int32), C.c_int32()
    check(L.eao_search_by_projection_frames(C.byref(v), _lib.ptr(Tc), _lib.ptr(Tl), len(valid), _lib.ptr(valid), _lib.ptr(Xw), _lib.ptr(desc), _lib.ptr(octv), _lib.ptr(ang),
                                            cur["fx"], cur["fy"], cur["cx"], cur["cy"], cur["mbf"], cur["mb"], th, 1 if mono else 0, 1 if check_ori else 0, _lib.ptr(out), C.byref(nm)))
    return nm.value, out


def _same(a, b):
    assert a[0] == b[0], (a[0], b[0])
    for x, y in zip(a[1:], b[1:]):
        assert np.array_equal(x, y)
    return a[0]


@pytest.mark.parametrize("kw", [dict(n=500, seed=8400), dict(n=900, seed=8401, flip=0.09, mono_frac=0.6), dict(n=200, seed=8402, clutter=0.5, n_nodes=12)])
def test_host_replay_of_every_search_equals_the_oracle(host, oracle, kw):
    b, L, check, _ = host
    o = oracle.search_binding()
    sc = synth.synth_search_scene(**kw)
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
    kf = dict(sc["K2"]); kf["occupied"] = (np.arange(len(kf["kp_x"])) % 13 == 0).astype(np.uint8)
    assert _same(b.search_by_projection_sim3(kf, sc["Scw"], sc["K"], P, 10), o.search_by_projection_sim3(kf, sc["Scw"], sc["K"], P, 10)) > 5
    _same(b.search_by_projection_kf(kf, sc["T2w"], sc["K"], P, ang, 15, 100, True), o.search_by_projection_kf(kf, sc["T2w"], sc["K"], P, ang, 15, 100, True))
    for mode, ratio in ((0, 0.75), (1, 0.8)):
        assert _same(b.search_by_bow(mode, s1, s2, ratio, True), o.search_by_bow(mode, s1, s2, ratio, True)) > 5
    k1 = dict(sc["K1"]); k1["occupied"] = ((sc["mp1"] >= 0) & (np.arange(len(sc["mp1"])) % 2 == 0)).astype(np.uint8)
    for only_stereo in (0, 1):
        args = (k1, sc["fv1"], sc["K2"], sc["fv2"], sc["F12"], sc["ex"], sc["ey"], only_stereo, True)
        _same(b.search_for_triangulation(*args), o.search_for_triangulation(*args))
    _same(b.search_for_initialization(sc["K1"], sc["K2"], pm, 100, 0.9, True), o.search_for_initialization(sc["K1"], sc["K2"], pm, 100, 0.9, True))
    assert _same(b.fuse_search(sc["K2"], 0, pose15, sc["K"], sc["bf"], P, 3.0), o.fuse_search(sc["K2"], 0, pose15, sc["K"], sc["bf"], P, 3.0)) > 5
    _same(b.fuse_search(sc["K2"], 1, sc["Scw"], sc["K"], sc["bf"], P, 3.0), o.fuse_search(sc["K2"], 1, sc["Scw"], sc["K"], sc["bf"], P, 3.0))
    P1, P2 = pts_of(sc["mp1"]), pts_of(sc["mp2"])
    args = (sc["K1"], sc["T1w"], P1, sc["K2"], sc["T2w"], P2, sc["K"], 1.0, sc["R12"], sc["t12"], 7.5)
    _same(b.search_by_sim3(*args), o.search_by_sim3(*args))
    # the batched entry points = their single calls
    nm, m = b.search_for_triangulation_batch(k1, sc["fv1"], [sc["K2"]] * 3, [sc["fv2"]] * 3, [sc["F12"]] * 3, [sc["ex"]] * 3, [sc["ey"]] * 3, 0, True)
    want = o.search_for_triangulation(k1, sc["fv1"], sc["K2"], sc["fv2"], sc["F12"], sc["ex"], sc["ey"], 0, True)
    assert all(nm[k] == want[0] and np.array_equal(m[k], want[1]) for k in range(3))
    nf, best = b.fuse_search_batch([sc["K2"]] * 2, 0, [pose15] * 2, sc["K"], sc["bf"], P, 3.0)
    want = o.fuse_search(sc["K2"], 0, pose15, sc["K"], sc["bf"], P, 3.0)
    assert all(nf[k] == want[0] and np.array_equal(best[k], want[1]) for k in range(2))
    # the two per-frame searches
    cur, last, mps = synth.synth_tracking(n=kw["n"], seed=kw["seed"] + 50)
    _same(_points(L, check, cur, mps, 1.0, 0.8), oracle.search_by_projection_points(cur, mps, 1.0, 0.8))
    for th, mono in ((7.0, False), (15.0, True)):
        assert _same(_frames(L, check, cur, last, th, mono), oracle.search_by_projection_frames(cur, last, th, mono, True)) > 5


def test_host_replay_refuses_malformed_input(host):
    """Every case must come back as an error status (an exception here): under EAO_HOST_SAN=1 an out-of-range index that got through would abort the run."""
    b, L, check, HostError = host
    sc = synth.synth_search_scene(n=300, seed=8410)
    P = sc["points"]
    T = sc["T2w"].astype(np.float64)
    pose15 = np.concatenate([T[:3, :3].ravel(), T[:3, 3], -T[:3, :3].T @ T[:3, 3]]).astype(np.float32)
    s1 = dict(descriptors=sc["K1"]["descriptors"], angle=sc["K1"]["kp_angle"], valid=(sc["mp1"] >= 0).astype(np.uint8), fv=sc["fv1"])
    s2 = dict(descriptors=sc["K2"]["descriptors"], angle=sc["K2"]["kp_angle"], valid=(sc["mp2"] >= 0).astype(np.uint8), fv=sc["fv2"])
    bad_oct = dict(sc["K2"]); bad_oct["kp_octave"] = sc["K2"]["kp_octave"].copy(); bad_oct["kp_octave"][7] = 8
    neg_oct = dict(sc["K2"]); neg_oct["kp_octave"] = sc["K2"]["kp_octave"].copy(); neg_oct["kp_octave"][2] = -1
    fv_hi = dict(sc["fv2"]); fv_hi["index"] = fv_hi["index"].copy(); fv_hi["index"][0] = len(sc["K2"]["kp_x"])
    fv_unsorted = dict(sc["fv2"]); fv_unsorted["node_id"] = fv_unsorted["node_id"][::-1].copy()
    nanS = sc["Scw"].copy(); nanS[0, 0] = np.nan
    infT = sc["T2w"].copy(); infT[1, 3] = np.inf
    nanF = sc["F12"].copy(); nanF[2, 2] = np.nan
    nan15 = pose15.copy(); nan15[4] = np.nan
    cases = [
        lambda: b.search_for_triangulation(sc["K1"], sc["fv1"], bad_oct, sc["fv2"], sc["F12"], sc["ex"], sc["ey"], 0, True),
        lambda: b.search_for_triangulation(sc["K1"], sc["fv1"], neg_oct, sc["fv2"], sc["F12"], sc["ex"], sc["ey"], 0, True),
        lambda: b.search_for_triangulation(sc["K1"], sc["fv1"], sc["K2"], fv_hi, sc["F12"], sc["ex"], sc["ey"], 0, True),
        lambda: b.search_for_triangulation(sc["K1"], sc["fv1"], sc["K2"], fv_unsorted, sc["F12"], sc["ex"], sc["ey"], 0, True),
        lambda: b.search_for_triangulation(sc["K1"], sc["fv1"], sc["K2"], sc["fv2"], nanF, sc["ex"], sc["ey"], 0, True),
        lambda: b.search_by_bow(0, s1, dict(s2, fv=fv_hi), 0.75, True),
        lambda: b.fuse_search(bad_oct, 0, pose15, sc["K"], sc["bf"], P, 3.0),
        lambda: b.fuse_search(sc["K2"], 0, nan15, sc["K"], sc["bf"], P, 3.0),
        lambda: b.fuse_search(sc["K2"], 1, nanS, sc["K"], sc["bf"], P, 3.0),
        lambda: b.search_by_projection_sim3(sc["K2"], nanS, sc["K"], P, 10),
        lambda: b.search_by_projection_kf(sc["K2"], infT, sc["K"], P, np.zeros(len(P["active"]), np.float32), 15, 100, True),
        lambda: b.search_by_sim3(sc["K1"], sc["T1w"], P, sc["K2"], infT, P, sc["K"], 1.0, sc["R12"], sc["t12"], 7.5),
    ]
    cur, last, mps = synth.synth_tracking(n=200, seed=8411)
    l_oct = dict(last); l_oct["octave"] = last["octave"].copy(); l_oct["octave"][np.nonzero(last["valid"])[0][0]] = 8
    c_nan = dict(cur); c_nan["Tcw"] = cur["Tcw"].copy(); c_nan["Tcw"][0, 3] = np.nan
    cases += [lambda: _frames(L, check, cur, l_oct, 7.0, False), lambda: _frames(L, check, c_nan, last, 7.0, False)]
    for k, fn in enumerate(cases):
        with pytest.raises(HostError):
            fn()
    # n = 0 everywhere: no error, no match
    empty = {k: (v[:0] if isinstance(v, np.ndarray) and v.ndim >= 1 and len(v) == len(sc["K2"]["kp_x"]) else v) for k, v in sc["K2"].items()}
    P0 = {k: v[:0] for k, v in P.items()}
    assert b.fuse_search(sc["K2"], 0, pose15, sc["K"], sc["bf"], P0, 3.0)[0] == 0
    assert b.search_by_projection_sim3(empty, sc["Scw"], sc["K"], P, 10)[0] == 0
