"""ctypes loader for the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never from the
product package (eao_fusion_amd never imports this module).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# EAO_ORACLE_LIB: an alternative build of the same sources (tools/run_sanitizers.sh points it at an ASan + UBSan build)
LIB_PATH = os.environ.get("EAO_ORACLE_LIB") or os.path.join(HERE, "liboracle.so")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28


def build():
    if os.environ.get("EAO_ORACLE_LIB"):
        return
    subprocess.check_call(["make", "-s", "-C", HERE])


class Trace(C.Structure):
    _fields_ = [("n", C.c_int32), ("lam", C.c_double * 64), ("chi2", C.c_double * 64), ("trials", C.c_int32 * 64)]

    def to_dict(self):
        n = self.n
        return dict(lam=np.array(self.lam[:n]), chi2=np.array(self.chi2[:n]), trials=np.array(self.trials[:n]))


class PoseProblem(C.Structure):
    _fields_ = [("n", C.c_int32), ("Tcw", C.c_void_p), ("Xw", C.c_void_p), ("obs", C.c_void_p),
                ("inv_sigma2", C.c_void_p), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float),
                ("cy", C.c_float), ("bf", C.c_float)]


class BAProblem(C.Structure):
    _fields_ = [("n_cams", C.c_int32), ("n_points", C.c_int32), ("n_edges", C.c_int32),
                ("cam_Tcw", C.c_void_p), ("cam_fixed", C.c_void_p), ("points", C.c_void_p),
                ("edge_cam", C.c_void_p), ("edge_point", C.c_void_p), ("edge_obs", C.c_void_p),
                ("edge_inv_sigma2", C.c_void_p), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float),
                ("cy", C.c_float), ("bf", C.c_float), ("its_first", C.c_int32), ("its_second", C.c_int32)]


class BAPlanes(C.Structure):   # orc_ba_planes
    _fields_ = [("n_planes", C.c_int32), ("plane_world", C.c_void_p), ("n_pedges", C.c_int32), ("pedge_plane", C.c_void_p),
                ("pedge_cam", C.c_void_p), ("pedge_obs", C.c_void_p)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.orc_orb_create.restype = C.c_void_p
        L.orc_orb_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        L.orc_orb_destroy.argtypes = [C.c_void_p]
        L.orc_orb_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_orb_result.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_orb_tables.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        L.orc_orb_level_dims.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_orb_level_image.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_orb_level_candidates.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_orb_level_keypoints.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_resize_linear_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.orc_gaussian_blur7.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_gaussian_taps.argtypes = [C.c_void_p]
        L.orc_fast_atan2.restype = C.c_float
        L.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.orc_fast.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_distribute.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_orb_pattern.restype = C.POINTER(C.c_int8)
        L.orc_descriptor_distance.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_hamming_matrix.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_hamming_best2.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_pose_optimization.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_local_ba.argtypes = [C.c_void_p] * 9
        L.orc_bundle_adjustment.argtypes = [C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 7
        L.orc_bundle_adjustment.restype = C.c_int
        L.orc_bundle_adjustment_planes.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 8
        L.orc_bundle_adjustment_planes.restype = C.c_int
        L.orc_ba_edge_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] + [C.c_double] * 5 + [C.c_void_p] * 3
        L.orc_se3_oplus.argtypes = [C.c_void_p] * 3
        L.orc_huber.argtypes = [C.c_double, C.c_double, C.c_void_p]
        L.orc_Tcw_to_cam7.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_search_by_projection_points.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 7 + [C.c_float, C.c_float, C.c_void_p]
        L.orc_search_by_projection_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_float] * 7 + [C.c_int, C.c_int, C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data if a is not None else None


class OrbOracle:
    """CPU restatement of ORBextractor (reference include/ORBextractor.h:45-111)."""

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7):
        self.L = lib()
        self.nlevels = nlevels
        self.h = self.L.orc_orb_create(nfeatures, scale_factor, nlevels, ini_th, min_th)

    def __del__(self):
        try:
            self.L.orc_orb_destroy(self.h)
        except Exception:
            pass

    def tables(self):
        n = self.nlevels
        sc, inv, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        quota = np.zeros(n, np.int32)
        umax = np.zeros(16, np.int32)
        self.L.orc_orb_tables(self.h, _p(sc), _p(inv), _p(s2), _p(is2), _p(quota), _p(umax))
        return dict(scale=sc, inv_scale=inv, sigma2=s2, inv_sigma2=is2, quota=quota, umax=umax)

    def extract(self, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape
        n = self.L.orc_orb_run(self.h, _p(img), w, h, w)
        if n < 0:
            raise ValueError("unsupported image geometry")
        kps = np.zeros(n, KP_DTYPE)
        desc = np.zeros((n, 32), np.uint8)
        self.L.orc_orb_result(self.h, _p(kps), _p(desc), n)
        return kps, desc

    def level_dims(self, level):
        w, h = C.c_int(), C.c_int()
        self.L.orc_orb_level_dims(self.h, level, C.byref(w), C.byref(h))
        return w.value, h.value

    def level_image(self, level, blurred=False):
        w, h = self.level_dims(level)
        out = np.zeros((h, w), np.uint8)
        n = self.L.orc_orb_level_image(self.h, level, 1 if blurred else 0, _p(out))
        return out if n else None

    def level_candidates(self, level):
        n = self.L.orc_orb_level_candidates(self.h, level, None, 0)
        out = np.zeros((n, 3), np.float32)
        self.L.orc_orb_level_candidates(self.h, level, _p(out), n)
        return out

    def level_keypoints(self, level):
        n = self.L.orc_orb_level_keypoints(self.h, level, None, 0)
        out = np.zeros(n, KP_DTYPE)
        self.L.orc_orb_level_keypoints(self.h, level, _p(out), n)
        return out


def resize_linear(src, dw, dh):
    src = np.ascontiguousarray(src, np.uint8)
    out = np.zeros((dh, dw), np.uint8)
    lib().orc_resize_linear_u8(_p(src), src.shape[1], src.shape[0], _p(out), dw, dh)
    return out


def gaussian_blur7(src):
    src = np.ascontiguousarray(src, np.uint8)
    out = np.zeros_like(src)
    lib().orc_gaussian_blur7(_p(src), src.shape[1], src.shape[0], _p(out))
    return out


def fast(img, th):
    img = np.ascontiguousarray(img, np.uint8)
    cap = img.size
    out = np.zeros((cap, 3), np.float32)
    n = lib().orc_fast(_p(img), img.shape[1], img.shape[0], th, _p(out), cap)
    return out[:n]


def distribute(xyr, min_x, max_x, min_y, max_y, N):
    xyr = np.ascontiguousarray(xyr, np.float32)
    sel = np.zeros(len(xyr) + 8, np.int32)
    n = lib().orc_distribute(_p(xyr), len(xyr), min_x, max_x, min_y, max_y, N, _p(sel), len(sel))
    return sel[:n]


def descriptor_distance(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return lib().orc_descriptor_distance(_p(a), _p(b))


def hamming_matrix(A, B):
    A = np.ascontiguousarray(A, np.uint8)
    B = np.ascontiguousarray(B, np.uint8)
    D = np.zeros((len(A), len(B)), np.uint16)
    lib().orc_hamming_matrix(_p(A), len(A), _p(B), len(B), _p(D))
    return D


def hamming_best2(A, B, mask=None):
    A = np.ascontiguousarray(A, np.uint8)
    B = np.ascontiguousarray(B, np.uint8)
    out = np.zeros((len(A), 4), np.int32)
    if mask is not None:
        mask = np.ascontiguousarray(mask, np.uint8)
    lib().orc_hamming_best2(_p(A), len(A), _p(B), len(B), _p(mask), _p(out))
    return out


def pose_optimization(prob):
    """prob: dict from eao_fusion_amd.synth.synth_pose (Tcw, points, obs, inv_sigma2, fx..bf)."""
    Tcw = np.ascontiguousarray(prob["Tcw"], np.float32)
    Xw = np.ascontiguousarray(prob["points"], np.float32)
    obs = np.ascontiguousarray(prob["obs"], np.float32)
    inv = np.ascontiguousarray(prob["inv_sigma2"], np.float32)
    n = len(Xw)
    P = PoseProblem(n, _p(Tcw), _p(Xw), _p(obs), _p(inv), prob["fx"], prob["fy"], prob["cx"], prob["cy"], prob["bf"])
    T_out = np.zeros((4, 4), np.float32)
    outl = np.zeros(max(n, 1), np.uint8)
    pose_d = np.zeros(7)
    tr = Trace()
    pw = prob.get("plane_world")
    if pw is None:
        ninl = lib().orc_pose_optimization(C.byref(P), _p(T_out), _p(outl), _p(pose_d), C.byref(tr))
        return dict(Tcw=T_out, outlier=outl[:n], n_inliers=ninl, pose_d=pose_d, trace=tr.to_dict())
    pw = np.ascontiguousarray(pw, np.float32)
    po = np.ascontiguousarray(prob["plane_obs"], np.float32)
    ps = np.ascontiguousarray(prob["plane_seen"], np.uint8)
    pout = np.zeros(max(len(pw), 1), np.uint8)
    fn = lib().orc_pose_optimization_planes
    fn.restype = C.c_int
    ninl = fn(C.byref(P), C.c_int(len(pw)), C.c_void_p(_p(pw)), C.c_void_p(_p(po)), C.c_void_p(_p(ps)), C.c_void_p(_p(T_out)),
              C.c_void_p(_p(outl)), C.c_void_p(_p(pout)), C.c_void_p(_p(pose_d)), C.byref(tr))
    return dict(Tcw=T_out, outlier=outl[:n], plane_outlier=pout[:len(pw)], n_inliers=ninl, pose_d=pose_d, trace=tr.to_dict())


def local_ba(prob, its=(5, 10), stop=None):
    cams = np.ascontiguousarray(prob["poses"], np.float32)
    fixed = np.ascontiguousarray(prob["fixed"], np.uint8)
    pts = np.ascontiguousarray(prob["points"], np.float32)
    ec = np.ascontiguousarray(prob["edge_cam"], np.int32)
    ep = np.ascontiguousarray(prob["edge_point"], np.int32)
    obs = np.ascontiguousarray(prob["obs"], np.float32)
    inv = np.ascontiguousarray(prob["inv_sigma2"], np.float32)
    P = BAProblem(len(cams), len(pts), len(ec), _p(cams), _p(fixed), _p(pts), _p(ec), _p(ep), _p(obs), _p(inv),
                  prob["fx"], prob["fy"], prob["cx"], prob["cy"], prob["bf"], its[0], its[1])
    cams_out = np.zeros_like(cams)
    pts_out = np.zeros_like(pts)
    outl = np.zeros(max(len(ec), 1), np.uint8)
    cams_d = np.zeros((len(cams), 7))
    pts_d = np.zeros((len(pts), 3))
    iters = np.zeros(2, np.int32)
    tr = Trace()
    stop_p = None
    if stop is not None:
        stop_arr = np.array([1 if stop else 0], np.uint8)
        stop_p = _p(stop_arr)
    rc = lib().orc_local_ba(C.byref(P), stop_p, _p(cams_out), _p(pts_out), _p(outl), _p(cams_d), _p(pts_d), _p(iters), C.byref(tr))
    return dict(poses=cams_out, points=pts_out, edge_outlier=outl[:len(ec)], cams_d=cams_d, points_d=pts_d,
                iters=iters, trace=tr.to_dict(), aborted=bool(rc))


def bundle_adjustment(prob, iterations=5, robust=True, stop=None):
    """Optimizer::BundleAdjustment over keyframes and map points (oracle/lm_cpu.cpp: orc_bundle_adjustment)"""
    cams = np.ascontiguousarray(prob["poses"], np.float32)
    fixed = np.ascontiguousarray(prob["fixed"], np.uint8)
    pts = np.ascontiguousarray(prob["points"], np.float32)
    ec = np.ascontiguousarray(prob["edge_cam"], np.int32)
    ep = np.ascontiguousarray(prob["edge_point"], np.int32)
    obs = np.ascontiguousarray(prob["obs"], np.float32)
    inv = np.ascontiguousarray(prob["inv_sigma2"], np.float32)
    P = BAProblem(len(cams), len(pts), len(ec), _p(cams), _p(fixed), _p(pts), _p(ec), _p(ep), _p(obs), _p(inv),
                  prob["fx"], prob["fy"], prob["cx"], prob["cy"], prob["bf"], iterations, 0)
    cams_out, pts_out = np.zeros_like(cams), np.zeros_like(pts)
    cams_d, pts_d = np.zeros((len(cams), 7)), np.zeros((len(pts), 3))
    iters = np.zeros(2, np.int32)
    tr = Trace()
    stop_p = None
    if stop is not None:
        stop_arr = np.array([1 if stop else 0], np.uint8)
        stop_p = _p(stop_arr)
    if prob.get("planes") is not None and len(prob["planes"]):
        pw = np.ascontiguousarray(prob["planes"], np.float32)
        pp = np.ascontiguousarray(prob["pedge_plane"], np.int32)
        pc = np.ascontiguousarray(prob["pedge_cam"], np.int32)
        po = np.ascontiguousarray(prob["pedge_obs"], np.float32)
        PL = BAPlanes(len(pw), _p(pw), len(pp), _p(pp), _p(pc), _p(po))
        planes_out = np.zeros_like(pw)
        lib().orc_bundle_adjustment_planes(C.byref(P), C.byref(PL), int(iterations), 1 if robust else 0, stop_p, _p(cams_out), _p(pts_out), _p(planes_out),
           _p(cams_d), _p(pts_d), _p(iters), C.byref(tr))
        return dict(poses=cams_out, points=pts_out, planes=planes_out, cams_d=cams_d, points_d=pts_d, iters=iters, trace=tr.to_dict())
    lib().orc_bundle_adjustment(C.byref(P), int(iterations), 1 if robust else 0, stop_p, _p(cams_out), _p(pts_out), _p(cams_d), _p(pts_d),
                                _p(iters), C.byref(tr))
    return dict(poses=cams_out, points=pts_out, cams_d=cams_d, points_d=pts_d, iters=iters, trace=tr.to_dict())


class FrameView(C.Structure):   # same layout as eao_frame_view
    _fields_ = [("n", C.c_int32), ("kp_x", C.c_void_p), ("kp_y", C.c_void_p), ("kp_octave", C.c_void_p), ("kp_angle", C.c_void_p),
                ("u_right", C.c_void_p), ("descriptors", C.c_void_p), ("occupied", C.c_void_p),
                ("min_x", C.c_float), ("min_y", C.c_float), ("max_x", C.c_float), ("max_y", C.c_float),
                ("grid_inv_w", C.c_float), ("grid_inv_h", C.c_float), ("grid_cols", C.c_int32), ("grid_rows", C.c_int32),
                ("scale_factors", C.c_void_p), ("nlevels", C.c_int32),
                ("log_scale_factor", C.c_float), ("level_sigma2", C.c_void_p), ("inv_level_sigma2", C.c_void_p)]


def _frame_view(frame):
    keep = {}
    for k, dt in (("kp_x", np.float32), ("kp_y", np.float32), ("kp_octave", np.int32), ("kp_angle", np.float32),
                  ("u_right", np.float32), ("descriptors", np.uint8), ("scale_factors", np.float32)):
        keep[k] = np.ascontiguousarray(frame[k], dt)
    occ = frame.get("occupied")
    keep["occupied"] = None if occ is None else np.ascontiguousarray(occ, np.uint8)
    cols, rows = int(frame.get("grid_cols", 64)), int(frame.get("grid_rows", 48))
    inv_w = np.float32(cols) / np.float32(np.float32(frame["max_x"]) - np.float32(frame["min_x"]))
    inv_h = np.float32(rows) / np.float32(np.float32(frame["max_y"]) - np.float32(frame["min_y"]))
    v = FrameView(len(keep["kp_x"]), _p(keep["kp_x"]), _p(keep["kp_y"]), _p(keep["kp_octave"]), _p(keep["kp_angle"]), _p(keep["u_right"]),
                  _p(keep["descriptors"]), _p(keep["occupied"]), frame["min_x"], frame["min_y"], frame["max_x"], frame["max_y"],
                  inv_w, inv_h, cols, rows, _p(keep["scale_factors"]), len(keep["scale_factors"]), 0.0, None, None)
    return v, keep


def search_by_projection_points(frame, mps, th, nnratio):
    v, keep = _frame_view(frame)
    a = {k: np.ascontiguousarray(mps[k], np.float32) for k in ("proj_x", "proj_y", "proj_xr", "view_cos")}
    lvl = np.ascontiguousarray(mps["level"], np.int32)
    desc = np.ascontiguousarray(mps["descriptors"], np.uint8)
    skip = mps.get("skip")
    skip = None if skip is None else np.ascontiguousarray(skip, np.uint8)
    out = np.full(len(lvl), -1, np.int32)
    nm = lib().orc_search_by_projection_points(C.byref(v), len(lvl), _p(a["proj_x"]), _p(a["proj_y"]), _p(a["proj_xr"]), _p(a["view_cos"]),
                                               _p(lvl), _p(desc), _p(skip), th, nnratio, _p(out))
    return nm, out


def search_by_projection_frames(cur, last, th, mono, check_orientation=True):
    v, keep = _frame_view(cur)
    Tc = np.ascontiguousarray(cur["Tcw"], np.float32)
    Tl = np.ascontiguousarray(last["Tcw"], np.float32)
    valid = np.ascontiguousarray(last["valid"], np.uint8)
    Xw = np.ascontiguousarray(last["Xw"], np.float32)
    desc = np.ascontiguousarray(last["descriptors"], np.uint8)
    octv = np.ascontiguousarray(last["octave"], np.int32)
    ang = np.ascontiguousarray(last["angle"], np.float32)
    out = np.full(v.n, -1, np.int32)
    nm = lib().orc_search_by_projection_frames(C.byref(v), _p(Tc), _p(Tl), len(valid), _p(valid), _p(Xw), _p(desc), _p(octv), _p(ang),
                                               cur["fx"], cur["fy"], cur["cx"], cur["cy"], cur["mbf"], cur["mb"], th, 1 if mono else 0,
                                               1 if check_orientation else 0, _p(out))
    return nm, out


def distinctive_descriptors(sets):
    start = np.zeros(len(sets) + 1, np.int32)
    for i, d in enumerate(sets):
        start[i + 1] = start[i] + len(d)
    desc = np.ascontiguousarray(np.concatenate([np.asarray(d, np.uint8).reshape(-1, 32) for d in sets]) if len(sets) else np.zeros((0, 32), np.uint8))
    best = np.full(len(sets), -1, np.int32)
    fn = lib().orc_distinctive_descriptors
    fn.restype = None
    fn(C.c_int(len(sets)), C.c_void_p(_p(start)), C.c_void_p(_p(desc)), C.c_void_p(_p(best)))
    return best


def stereo_matches(orc_left, orc_right, kps_l, desc_l, kps_r, desc_r, mb, mbf):
    """Frame::ComputeStereoMatches over two OrbOracle instances that just extracted the stereo pair."""
    kl = np.ascontiguousarray(kps_l, KP_DTYPE)
    kr = np.ascontiguousarray(kps_r, KP_DTYPE)
    dl = np.ascontiguousarray(desc_l, np.uint8)
    dr = np.ascontiguousarray(desc_r, np.uint8)
    ur = np.full(len(kl), -1, np.float32)
    dp = np.full(len(kl), -1, np.float32)
    fn = lib().orc_stereo_matches
    fn.restype = C.c_int
    fn(C.c_void_p(orc_left.h), C.c_void_p(orc_right.h), C.c_int(len(kl)), C.c_void_p(_p(kl)), C.c_void_p(_p(dl)), C.c_int(len(kr)),
       C.c_void_p(_p(kr)), C.c_void_p(_p(dr)), C.c_float(mb), C.c_float(mbf), C.c_void_p(_p(ur)), C.c_void_p(_p(dp)))
    return ur, dp


# ---------------------------------------------------------------------------------------------------------------------
# Checker bindings of the guided searches and the Frame glue: the product package's marshalling (array packing, struct
# layouts) driven against liboracle.so's `orc_*` entry points.  Test infrastructure -- the product never imports this.
def search_binding():
    """The seven remaining guided searches of oracle/search_cpu.cpp: `orc_<name>` returns the match count and takes no count
    pointer (the product's `eao_<name>` returns a status and writes the count through a last pointer)."""
    from eao_fusion_amd import search as S

    class _SearchOracle(S.Binding):
        prefix = "orc_"

        def __init__(self, L):
            self.lib, self.check = L, None
            for name, args in S.SEARCH_ARGTYPES.items():
                fn = getattr(L, "orc_" + name)
                fn.restype = C.c_int32
                fn.argtypes = args + [C.c_void_p]

        def _call(self, name, args, out):
            return int(getattr(self.lib, "orc_" + name)(*args, _p(out))), out

    return _SearchOracle(lib())


def frame_binding():
    """Frame::isInFrustum / AssignFeaturesToGrid / ComputeStereoFromRGBD of oracle/frame_cpu.cpp (flat argument lists)."""
    from eao_fusion_amd import frame as F
    _P, _I, _F = C.c_void_p, C.c_int32, C.c_float

    class _FrameOracle(F.Binding):
        def __init__(self, L):
            self.lib, self.check = L, None
            L.orc_is_in_frustum.restype = _I
            L.orc_is_in_frustum.argtypes = [_I] + [_P] * 8 + [_F] * 11 + [_P] * 6
            L.orc_assign_features_to_grid.restype = _I
            L.orc_assign_features_to_grid.argtypes = [_I, _P, _P, _F, _F, _F, _F, _I, _I, _P, _P]
            L.orc_stereo_from_rgbd.restype = _I
            L.orc_stereo_from_rgbd.argtypes = [_I, _P, _P, _P, _P, _I, _F, _P, _P]
            L.orc_undistort_keypoints.restype = _I
            L.orc_undistort_keypoints.argtypes = [_I, _P, _P, _F, _F, _F, _F, _P, _I, _P, _P]
            L.orc_compute_image_bounds.restype = _I
            L.orc_compute_image_bounds.argtypes = [_I, _I, _F, _F, _F, _F, _P, _I, _P]

        def _raw_is_in_frustum(self, m, keep, T, Ow, sc, limit, outs):
            R = np.ascontiguousarray(T[:3, :3]); t = np.ascontiguousarray(T[:3, 3])
            self.lib.orc_is_in_frustum(m.n, _p(keep["Xw"]), _p(keep["normal"]), _p(keep["min_dist_inv"]), _p(keep["max_dist_inv"]),
                                       _p(keep["max_dist"]), _p(R), _p(t), _p(Ow), *sc, limit, *outs)

        def _raw_assign(self, n, kx, ky, min_x, min_y, inv_w, inv_h, cols, rows, start, items):
            self.lib.orc_assign_features_to_grid(n, kx, ky, min_x, min_y, inv_w, inv_h, cols, rows, start, items)

        def _raw_rgbd(self, n, kx, ky, ku, d, w, h, mbf, ur, dz):
            self.lib.orc_stereo_from_rgbd(n, kx, ky, ku, d, w, mbf, ur, dz)

        def _raw_undistort(self, n, kx, ky, fx, fy, cx, cy, dist, nc, ox, oy):
            self.lib.orc_undistort_keypoints(n, kx, ky, fx, fy, cx, cy, dist, nc, ox, oy)

        def _raw_bounds(self, cols, rows, fx, fy, cx, cy, dist, nc, out):
            self.lib.orc_compute_image_bounds(cols, rows, fx, fy, cx, cy, dist, nc, out)

    return _FrameOracle(lib())
