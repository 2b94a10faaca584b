"""ctypes mirror of the remaining guided searches (include/eao_fusion.h: eao_search_by_projection_sim3 / _kf, eao_search_by_bow,
eao_search_for_triangulation, eao_search_for_initialization, eao_fuse_search, eao_search_by_sim3), i.e. reference
src/ORBmatcher.cc:159-1326,1474-1601 over plain arrays.

Data conventions (dicts of numpy arrays):
  frame   kp_x, kp_y, kp_angle, u_right (f32), kp_octave (i32), descriptors (n,32) u8, optional occupied (u8), min_x, min_y,
          max_x, max_y, scale_factors (f32), optional log_scale_factor, level_sigma2, inv_level_sigma2
  points  active (u8), Xw (n,3), optional normal (n,3), min_dist_inv, max_dist_inv, max_dist (f32), descriptors (n,32)
  fv      node_id (u32, ascending), node_start (i32, n_nodes+1), index (u32)
`Binding(lib, check)` binds the wrappers to libeaofusion_hip.so (see product() below)."""
import ctypes as C

import numpy as np

_P = C.c_void_p
_I = C.c_int32
_F = C.c_float


class FrameView(C.Structure):
    _fields_ = [("n", _I), ("kp_x", _P), ("kp_y", _P), ("kp_octave", _P), ("kp_angle", _P), ("u_right", _P), ("descriptors", _P),
                ("occupied", _P), ("min_x", _F), ("min_y", _F), ("max_x", _F), ("max_y", _F), ("grid_inv_w", _F), ("grid_inv_h", _F),
                ("grid_cols", _I), ("grid_rows", _I), ("scale_factors", _P), ("nlevels", _I),
                ("log_scale_factor", _F), ("level_sigma2", _P), ("inv_level_sigma2", _P)]


class MapPoints(C.Structure):
    _fields_ = [("n", _I), ("active", _P), ("Xw", _P), ("normal", _P), ("min_dist_inv", _P), ("max_dist_inv", _P), ("max_dist", _P),
                ("desc", _P)]


class FeatureVector(C.Structure):
    _fields_ = [("n_nodes", _I), ("node_id", _P), ("node_start", _P), ("index", _P)]


def _p(a):
    return None if a is None else a.ctypes.data


def frame_view(frame):
    keep = {}
    for k, dt in (("kp_x", np.float32), ("kp_y", np.float32), ("kp_octave", np.int32), ("kp_angle", np.float32),
                  ("u_right", np.float32), ("descriptors", np.uint8), ("scale_factors", np.float32)):
        keep[k] = np.ascontiguousarray(frame[k], dt)
    for k, dt in (("occupied", np.uint8), ("level_sigma2", np.float32), ("inv_level_sigma2", np.float32)):
        v = frame.get(k)
        keep[k] = None if v is None else np.ascontiguousarray(v, dt)
    cols, rows = int(frame.get("grid_cols", 64)), int(frame.get("grid_rows", 48))
    inv_w = np.float32(cols) / np.float32(np.float32(frame["max_x"]) - np.float32(frame["min_x"]))
    inv_h = np.float32(rows) / np.float32(np.float32(frame["max_y"]) - np.float32(frame["min_y"]))
    v = FrameView(len(keep["kp_x"]), _p(keep["kp_x"]), _p(keep["kp_y"]), _p(keep["kp_octave"]), _p(keep["kp_angle"]), _p(keep["u_right"]),
                  _p(keep["descriptors"]), _p(keep["occupied"]), frame["min_x"], frame["min_y"], frame["max_x"], frame["max_y"],
                  inv_w, inv_h, cols, rows, _p(keep["scale_factors"]), len(keep["scale_factors"]),
                  float(frame.get("log_scale_factor", 0.0)), _p(keep["level_sigma2"]), _p(keep["inv_level_sigma2"]))
    return v, keep


def map_points(pts):
    keep = {"active": np.ascontiguousarray(pts["active"], np.uint8), "Xw": np.ascontiguousarray(pts["Xw"], np.float32),
            "descriptors": np.ascontiguousarray(pts["descriptors"], np.uint8)}
    for k in ("min_dist_inv", "max_dist_inv", "max_dist"):
        keep[k] = np.ascontiguousarray(pts[k], np.float32)
    nrm = pts.get("normal")
    keep["normal"] = None if nrm is None else np.ascontiguousarray(nrm, np.float32)
    v = MapPoints(len(keep["active"]), _p(keep["active"]), _p(keep["Xw"]), _p(keep["normal"]), _p(keep["min_dist_inv"]),
                  _p(keep["max_dist_inv"]), _p(keep["max_dist"]), _p(keep["descriptors"]))
    return v, keep


def feature_vector(fv):
    keep = {"node_id": np.ascontiguousarray(fv["node_id"], np.uint32), "node_start": np.ascontiguousarray(fv["node_start"], np.int32),
            "index": np.ascontiguousarray(fv["index"], np.uint32)}
    return FeatureVector(len(keep["node_id"]), _p(keep["node_id"]), _p(keep["node_start"]), _p(keep["index"])), keep


_FV, _MP, _FR = C.POINTER(FeatureVector), C.POINTER(MapPoints), C.POINTER(FrameView)
SEARCH_ARGTYPES = {   # argument types after which every function takes (int32* result_array, int32* count)
    "search_by_projection_sim3": [_FR, _P, _F, _F, _F, _F, _MP, _I],
    "search_by_projection_kf": [_FR, _P, _F, _F, _F, _F, _MP, _P, _F, _I, _I],
    "search_by_bow": [_I, _I, _P, _P, _P, _FV, _I, _P, _P, _P, _FV, _F, _I],
    "search_for_triangulation": [_FR, _FV, _FR, _FV, _P, _F, _F, _I, _I],
    "search_for_initialization": [_I, _P, _P, _P, _FR, _P, _I, _F, _I],
    "fuse_search": [_FR, _I, _P, _F, _F, _F, _F, _F, _MP, _F],
    "search_by_sim3": [_FR, _P, _MP, _FR, _P, _MP, _F, _F, _F, _F, _F, _P, _P, _F],
}


class Binding:
    """The seven searches bound to libeaofusion_hip.so: every function returns a status and writes the match count through its
    last pointer.  (The test suite derives its checker binding from this class; nothing in this package
    knows about it.)"""
    prefix = "eao_"

    def __init__(self, lib, check):
        self.lib, self.check = lib, check
        for name, args in SEARCH_ARGTYPES.items():
            fn = getattr(lib, self.prefix + name)
            fn.restype = _I
            fn.argtypes = args + [_P, C.POINTER(_I)]

    def _call(self, name, args, out):
        n = _I(0)
        self.check(getattr(self.lib, self.prefix + name)(*args, _p(out), C.byref(n)))
        return int(n.value), out

    def search_by_projection_sim3(self, kf, Scw, K, pts, th):
        v, k1 = frame_view(kf)
        m, k2 = map_points(pts)
        S = np.ascontiguousarray(Scw, np.float32)
        out = np.full(v.n, -1, np.int32)
        return self._call("search_by_projection_sim3", [C.byref(v), _p(S), K[0], K[1], K[2], K[3], C.byref(m), int(th)], out)

    def search_by_projection_kf(self, cur, Tcw, K, pts, kf_angle, th, orb_dist, check_orientation=True):
        v, k1 = frame_view(cur)
        m, k2 = map_points(pts)
        T = np.ascontiguousarray(Tcw, np.float32)
        ang = np.ascontiguousarray(kf_angle, np.float32)
        out = np.full(v.n, -1, np.int32)
        return self._call("search_by_projection_kf", [C.byref(v), _p(T), K[0], K[1], K[2], K[3], C.byref(m), _p(ang), float(th), int(orb_dist),
                                                      int(check_orientation)], out)

    def search_by_bow(self, mode, s1, s2, nnratio, check_orientation=True):
        d1, d2 = np.ascontiguousarray(s1["descriptors"], np.uint8), np.ascontiguousarray(s2["descriptors"], np.uint8)
        a1, a2 = np.ascontiguousarray(s1["angle"], np.float32), np.ascontiguousarray(s2["angle"], np.float32)
        v1 = np.ascontiguousarray(s1["valid"], np.uint8)
        v2 = None if s2.get("valid") is None else np.ascontiguousarray(s2["valid"], np.uint8)
        f1, k1 = feature_vector(s1["fv"])
        f2, k2 = feature_vector(s2["fv"])
        out = np.full(len(d1), -1, np.int32)
        return self._call("search_by_bow", [int(mode), len(d1), _p(d1), _p(a1), _p(v1), C.byref(f1), len(d2), _p(d2), _p(a2), _p(v2), C.byref(f2),
                                            float(nnratio), int(check_orientation)], out)

    def search_for_triangulation(self, k1, fv1, k2, fv2, F12, ex, ey, only_stereo, check_orientation=True):
        v1, keep1 = frame_view(k1)
        v2, keep2 = frame_view(k2)
        f1, kf1 = feature_vector(fv1)
        f2, kf2 = feature_vector(fv2)
        F = np.ascontiguousarray(F12, np.float32)
        out = np.full(v1.n, -1, np.int32)
        return self._call("search_for_triangulation", [C.byref(v1), C.byref(f1), C.byref(v2), C.byref(f2), _p(F), float(ex), float(ey),
                                                       int(only_stereo), int(check_orientation)], out)

    def search_for_initialization(self, f1, f2, prev_matched, window, nnratio, check_orientation=True):
        v2, keep2 = frame_view(f2)
        o1 = np.ascontiguousarray(f1["kp_octave"], np.int32)
        a1 = np.ascontiguousarray(f1["kp_angle"], np.float32)
        d1 = np.ascontiguousarray(f1["descriptors"], np.uint8)
        pm = np.array(prev_matched, np.float32, copy=True)
        out = np.full(len(o1), -1, np.int32)
        n, out = self._call("search_for_initialization", [len(o1), _p(o1), _p(a1), _p(d1), C.byref(v2), _p(pm), int(window), float(nnratio),
                                                          int(check_orientation)], out)
        return n, out, pm

    def fuse_search(self, kf, use_sim3, pose, K, bf, pts, th):
        v, k1 = frame_view(kf)
        m, k2 = map_points(pts)
        ps = np.ascontiguousarray(pose, np.float32).ravel()
        out = np.full(m.n, -1, np.int32)
        return self._call("fuse_search", [C.byref(v), int(use_sim3), _p(ps), K[0], K[1], K[2], K[3], float(bf), C.byref(m), float(th)], out)

    def search_by_sim3(self, k1, T1w, pts1, k2, T2w, pts2, K, s12, R12, t12, th):
        v1, keep1 = frame_view(k1)
        v2, keep2 = frame_view(k2)
        m1, km1 = map_points(pts1)
        m2, km2 = map_points(pts2)
        T1, T2 = np.ascontiguousarray(T1w, np.float32), np.ascontiguousarray(T2w, np.float32)
        R, t = np.ascontiguousarray(R12, np.float32), np.ascontiguousarray(t12, np.float32)
        out = np.full(m1.n, -1, np.int32)
        return self._call("search_by_sim3", [C.byref(v1), _p(T1), C.byref(m1), C.byref(v2), _p(T2), C.byref(m2), K[0], K[1], K[2], K[3],
                                             float(s12), _p(R), _p(t), float(th)], out)


class ProductBinding(Binding):
    """... plus the two batched entry points of round 4 (LocalMapping::CreateNewMapPoints / SearchInNeighbors call a search once per neighbour keyframe): the
    product only -- the checker has no batch form, the tests compare a batch with its single calls and those with the oracle."""

    def __init__(self, lib, check):
        super().__init__(lib, check)
        lib.eao_search_for_triangulation_batch.restype = _I
        lib.eao_search_for_triangulation_batch.argtypes = [_FR, _FV, _I, _P, _P, _P, _P, _P, _I, _I, _P, _P]
        lib.eao_fuse_search_batch.restype = _I
        lib.eao_fuse_search_batch.argtypes = [_I, _P, _I, _P, _F, _F, _F, _F, _F, _MP, _F, _P, _P]

    def search_for_triangulation_batch(self, k1, fv1, k2s, fv2s, F12s, exs, eys, only_stereo, check_orientation=True):
        """Returns (nmatches[n_nb], match12[n_nb, n1])."""
        v1, keep1 = frame_view(k1)
        f1, kf1 = feature_vector(fv1)
        views = [frame_view(k) for k in k2s]
        fvs = [feature_vector(f) for f in fv2s]
        nb = len(views)
        vp = (C.POINTER(FrameView) * max(nb, 1))(*[C.pointer(v[0]) for v in views])
        fp = (C.POINTER(FeatureVector) * max(nb, 1))(*[C.pointer(f[0]) for f in fvs])
        F = np.ascontiguousarray(np.asarray(F12s, np.float32).reshape(nb, 9))
        ex, ey = np.ascontiguousarray(exs, np.float32), np.ascontiguousarray(eys, np.float32)
        out = np.full((nb, v1.n), -1, np.int32)
        nm = np.zeros(nb, np.int32)
        self.check(self.lib.eao_search_for_triangulation_batch(C.byref(v1), C.byref(f1), nb, C.cast(vp, _P), C.cast(fp, _P), _p(F), _p(ex), _p(ey),
                                                               int(only_stereo), int(check_orientation), _p(out), _p(nm)))
        return nm, out

    def fuse_search_batch(self, kfs, use_sim3, poses, K, bf, pts, th):
        """Returns (nfused[n_kf], best_kp[n_kf, n_points])."""
        views = [frame_view(k) for k in kfs]
        nk = len(views)
        vp = (C.POINTER(FrameView) * max(nk, 1))(*[C.pointer(v[0]) for v in views])
        m, k2 = map_points(pts)
        ps = np.ascontiguousarray(np.asarray(poses, np.float32).reshape(nk, -1))
        out = np.full((nk, m.n), -1, np.int32)
        nf = np.zeros(nk, np.int32)
        self.check(self.lib.eao_fuse_search_batch(nk, C.cast(vp, _P), int(use_sim3), _p(ps), K[0], K[1], K[2], K[3], float(bf), C.byref(m), float(th), _p(out), _p(nf)))
        return nf, out


class KeyFrameHandle:
    """eao_keyframe: a frame dict (and its feature vector) uploaded once (include/eao_fusion.h, "keyframe handles")."""

    def __init__(self, lib, check, frame, fv=None):
        self.lib, self.check = lib, check
        v, self._keep = frame_view(frame)
        f = None
        if fv is not None:
            f, self._keepfv = feature_vector(fv)
        self.h = _P()
        check(lib.eao_keyframe_create(C.byref(v), None if f is None else C.byref(f), C.byref(self.h)))
        self.n = v.n

    def update_points(self, occupied):
        occ = None if occupied is None else np.ascontiguousarray(occupied, np.uint8)
        self.check(self.lib.eao_keyframe_update_points(self.h, _p(occ)))

    def __del__(self):
        try:
            if self.h:
                self.lib.eao_keyframe_destroy(self.h)
                self.h = None
        except Exception:
            pass


class HandleBinding(ProductBinding):
    """The same searches through keyframe handles (eao_kf_*).  The methods keep the host-array signatures -- a frame dict is turned into a handle on first use
    and cached by the identity of its arrays -- so that the parity tests run every case through both bindings; handle() / the *_h methods are what a caller
    that keeps its keyframes resident uses (and what bench.py times)."""

    def __init__(self, lib, check):
        super().__init__(lib, check)
        K = _P
        sigs = {
            "eao_keyframe_create": [_FR, _FV, C.POINTER(_P)], "eao_keyframe_update_points": [K, _P], "eao_keyframe_size": [K],
            "eao_kf_search_by_bow": [_I, K, _P, K, _P, _F, _I, _P, C.POINTER(_I)],
            "eao_kf_search_for_triangulation": [K, _I, _P, _P, _P, _P, _I, _I, _P, _P],
            "eao_kf_fuse_search": [_I, _P, _I, _P, _F, _F, _F, _F, _F, _MP, _F, _P, _P],
            "eao_kf_search_by_projection_sim3": [K, _P, _P, _F, _F, _F, _F, _MP, _I, _P, C.POINTER(_I)],
            "eao_kf_search_by_projection_kf": [K, _P, _P, _F, _F, _F, _F, _MP, _P, _F, _I, _I, _P, C.POINTER(_I)],
            "eao_kf_search_for_initialization": [_I, _P, _P, _P, K, _P, _I, _F, _I, _P, C.POINTER(_I)],
            "eao_kf_search_by_sim3": [K, _P, _MP, K, _P, _MP, _F, _F, _F, _F, _F, _P, _P, _F, _P, C.POINTER(_I)],
        }
        for name, args in sigs.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = _I, args
        lib.eao_keyframe_destroy.restype, lib.eao_keyframe_destroy.argtypes = None, [K]
        self._cache = {}

    def handle(self, frame, fv=None):
        """A handle of (frame, fv), cached while the dicts' arrays stay the same objects."""
        key = tuple(id(frame.get(k)) for k in ("kp_x", "kp_y", "kp_octave", "kp_angle", "u_right", "descriptors", "occupied", "scale_factors", "level_sigma2",
                                               "inv_level_sigma2")) + (None if fv is None else (id(fv["node_id"]), id(fv["node_start"]), id(fv["index"])),)
        hit = self._cache.get(key)
        if hit is None:
            hit = (KeyFrameHandle(self.lib, self.check, frame, fv), frame, fv)      # (the dicts are kept alive: ids stay unique)
            self._cache[key] = hit
        return hit[0]

    # ---- handle-level calls
    def search_by_bow_h(self, mode, h1, valid1, h2, valid2, nnratio, check_orientation=True):
        v1 = np.ascontiguousarray(valid1, np.uint8)
        v2 = None if valid2 is None else np.ascontiguousarray(valid2, np.uint8)
        out, n = np.full(h1.n, -1, np.int32), _I(0)
        self.check(self.lib.eao_kf_search_by_bow(int(mode), h1.h, _p(v1), h2.h, _p(v2), float(nnratio), int(check_orientation), _p(out), C.byref(n)))
        return int(n.value), out

    def search_for_triangulation_h(self, h1, h2s, F12s, exs, eys, only_stereo, check_orientation=True):
        nb = len(h2s)
        hp = (_P * max(nb, 1))(*[h.h for h in h2s])
        F = np.ascontiguousarray(np.asarray(F12s, np.float32).reshape(nb, 9))
        ex, ey = np.ascontiguousarray(exs, np.float32), np.ascontiguousarray(eys, np.float32)
        out, nm = np.full((nb, h1.n), -1, np.int32), np.zeros(nb, np.int32)
        self.check(self.lib.eao_kf_search_for_triangulation(h1.h, nb, C.cast(hp, _P), _p(F), _p(ex), _p(ey), int(only_stereo), int(check_orientation), _p(out), _p(nm)))
        return nm, out

    def fuse_search_h(self, hs, use_sim3, poses, K, bf, pts, th):
        nk = len(hs)
        hp = (_P * max(nk, 1))(*[h.h for h in hs])
        m, k2 = map_points(pts)
        ps = np.ascontiguousarray(np.asarray(poses, np.float32).reshape(nk, -1))
        out, nf = np.full((nk, m.n), -1, np.int32), np.zeros(nk, np.int32)
        self.check(self.lib.eao_kf_fuse_search(nk, C.cast(hp, _P), int(use_sim3), _p(ps), K[0], K[1], K[2], K[3], float(bf), C.byref(m), float(th), _p(out), _p(nf)))
        return nf, out

    # ---- the host-array signatures, routed through handles
    def search_by_bow(self, mode, s1, s2, nnratio, check_orientation=True):
        def as_frame(s):
            n = len(s["descriptors"])
            z = np.zeros(n, np.float32)
            return dict(kp_x=z, kp_y=z, kp_octave=np.zeros(n, np.int32), kp_angle=s["angle"], u_right=z, descriptors=s["descriptors"], min_x=0.0, min_y=0.0, max_x=640.0,
                        max_y=480.0, scale_factors=np.ones(1, np.float32))
        key = (id(s1["descriptors"]), id(s1["fv"]["node_id"]), id(s2["descriptors"]), id(s2["fv"]["node_id"]))
        hit = self._cache.get(("bow",) + key)
        if hit is None:
            f1, f2 = as_frame(s1), as_frame(s2)
            hit = (self.handle(f1, s1["fv"]), self.handle(f2, s2["fv"]), s1, s2, f1, f2)
            self._cache[("bow",) + key] = hit
        return self.search_by_bow_h(mode, hit[0], s1["valid"], hit[1], s2.get("valid"), nnratio, check_orientation)

    def search_for_triangulation(self, k1, fv1, k2, fv2, F12, ex, ey, only_stereo, check_orientation=True):
        nm, out = self.search_for_triangulation_h(self.handle(k1, fv1), [self.handle(k2, fv2)], [F12], [ex], [ey], only_stereo, check_orientation)
        return int(nm[0]), out[0]

    def search_for_triangulation_batch(self, k1, fv1, k2s, fv2s, F12s, exs, eys, only_stereo, check_orientation=True):
        return self.search_for_triangulation_h(self.handle(k1, fv1), [self.handle(k, f) for k, f in zip(k2s, fv2s)], F12s, exs, eys, only_stereo, check_orientation)

    def fuse_search(self, kf, use_sim3, pose, K, bf, pts, th):
        nf, out = self.fuse_search_h([self.handle(kf)], use_sim3, [np.asarray(pose, np.float32).ravel()], K, bf, pts, th)
        return int(nf[0]), out[0]

    def fuse_search_batch(self, kfs, use_sim3, poses, K, bf, pts, th):
        return self.fuse_search_h([self.handle(k) for k in kfs], use_sim3, poses, K, bf, pts, th)

    def search_by_projection_sim3(self, kf, Scw, K, pts, th):
        h = self.handle(kf)
        m, k2 = map_points(pts)
        S = np.ascontiguousarray(Scw, np.float32)
        occ = None if kf.get("occupied") is None else np.ascontiguousarray(kf["occupied"], np.uint8)
        out, n = np.full(h.n, -1, np.int32), _I(0)
        self.check(self.lib.eao_kf_search_by_projection_sim3(h.h, _p(occ), _p(S), K[0], K[1], K[2], K[3], C.byref(m), int(th), _p(out), C.byref(n)))
        return int(n.value), out

    def search_by_projection_kf(self, cur, Tcw, K, pts, kf_angle, th, orb_dist, check_orientation=True):
        h = self.handle(cur)
        m, k2 = map_points(pts)
        T, ang = np.ascontiguousarray(Tcw, np.float32), np.ascontiguousarray(kf_angle, np.float32)
        occ = None if cur.get("occupied") is None else np.ascontiguousarray(cur["occupied"], np.uint8)
        out, n = np.full(h.n, -1, np.int32), _I(0)
        self.check(self.lib.eao_kf_search_by_projection_kf(h.h, _p(occ), _p(T), K[0], K[1], K[2], K[3], C.byref(m), _p(ang), float(th), int(orb_dist), int(check_orientation),
                                                           _p(out), C.byref(n)))
        return int(n.value), out

    def search_for_initialization(self, f1, f2, prev_matched, window, nnratio, check_orientation=True):
        h2 = self.handle(f2)
        o1, a1, d1 = np.ascontiguousarray(f1["kp_octave"], np.int32), np.ascontiguousarray(f1["kp_angle"], np.float32), np.ascontiguousarray(f1["descriptors"], np.uint8)
        pm = np.array(prev_matched, np.float32, copy=True)
        out, n = np.full(len(o1), -1, np.int32), _I(0)
        self.check(self.lib.eao_kf_search_for_initialization(len(o1), _p(o1), _p(a1), _p(d1), h2.h, _p(pm), int(window), float(nnratio), int(check_orientation), _p(out), C.byref(n)))
        return int(n.value), out, pm

    def search_by_sim3(self, k1, T1w, pts1, k2, T2w, pts2, K, s12, R12, t12, th):
        h1, h2 = self.handle(k1), self.handle(k2)
        m1, km1 = map_points(pts1)
        m2, km2 = map_points(pts2)
        T1, T2 = np.ascontiguousarray(T1w, np.float32), np.ascontiguousarray(T2w, np.float32)
        R, t = np.ascontiguousarray(R12, np.float32), np.ascontiguousarray(t12, np.float32)
        out, n = np.full(m1.n, -1, np.int32), _I(0)
        self.check(self.lib.eao_kf_search_by_sim3(h1.h, _p(T1), C.byref(m1), h2.h, _p(T2), C.byref(m2), K[0], K[1], K[2], K[3], float(s12), _p(R), _p(t), float(th), _p(out),
                                                  C.byref(n)))
        return int(n.value), out


_binding = None
_hbinding = None


def product_handles():
    """The searches through keyframe handles (eao_kf_*), bound to libeaofusion_hip.so."""
    global _hbinding
    if _hbinding is None:
        from . import _lib
        _hbinding = HandleBinding(_lib.load(), _lib.check)
    return _hbinding


def product():
    """The searches bound to libeaofusion_hip.so (fails loudly when it is missing: there is no CPU path)."""
    global _binding
    if _binding is None:
        from . import _lib
        _binding = ProductBinding(_lib.load(), _lib.check)
    return _binding
