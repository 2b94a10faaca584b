"""ctypes mirror of the device-resident tracked frame (include/eao_fusion.h, row f1: eao_tracker_*): Tracking::TrackLocalMap's
data path -- reference src/Tracking.cc:1717-2231, 2587-2641 -- chained on the device behind the extractor's outputs."""
import ctypes as C

import numpy as np

from . import _lib
from .search import MapPoints, map_points

_P, _I, _F = C.c_void_p, C.c_int32, C.c_float


class TrackerCfg(C.Structure):
    _fields_ = [("fx", _F), ("fy", _F), ("cx", _F), ("cy", _F), ("mbf", _F), ("min_x", _F), ("max_x", _F), ("min_y", _F), ("max_y", _F),
                ("grid_cols", _I), ("grid_rows", _I), ("nlevels", _I), ("scale_factors", _P), ("inv_level_sigma2", _P),
                ("log_scale_factor", _F), ("max_keypoints", _I), ("max_map_points", _I)]


class TrackResult(C.Structure):
    _fields_ = [("Tcw", _F * 16), ("n_keypoints", _I), ("n_matches", _I), ("n_edges", _I), ("n_inliers", _I), ("kp_map_point", _P),
                ("kp_outlier", _P), ("kp_u_right", _P), ("kp_depth", _P), ("map_in_view", _P)]


class TrackOptions(C.Structure):   # eao_track_options
    _fields_ = [("min_matches", _I), ("n_planes", _I), ("plane_world", _P), ("plane_obs", _P), ("plane_seen", _P), ("plane_outlier", _P)]


def _bind(L):
    L.eao_tracker_set_options.restype = _I
    L.eao_tracker_set_options.argtypes = [_P, C.POINTER(TrackOptions)]
    L.eao_tracker_set_distortion.restype = _I
    L.eao_tracker_set_distortion.argtypes = [_P, _P, _I]
    L.eao_tracker_create.restype = _I
    L.eao_tracker_create.argtypes = [C.POINTER(TrackerCfg), C.POINTER(_P)]
    L.eao_tracker_destroy.restype = None
    L.eao_tracker_destroy.argtypes = [_P]
    L.eao_tracker_set_local_map.restype = _I
    L.eao_tracker_set_local_map.argtypes = [_P, C.POINTER(MapPoints)]
    L.eao_tracker_track_local_map.restype = _I
    L.eao_tracker_track_local_map.argtypes = [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _F, _F, C.POINTER(TrackResult), _P]
    L.eao_tracker_track_reference_keyframe.restype = _I
    L.eao_tracker_track_reference_keyframe.argtypes = [_P, _P, _P, _P, _P, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _F, _I, _I, C.POINTER(TrackResult), _P]
    L.eao_tracker_track_with_motion_model.restype = _I
    L.eao_tracker_track_with_motion_model.argtypes = [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P, _F, _I, _I, _I, C.POINTER(TrackResult), _P]
    return L


class Tracker:
    def __init__(self, fx, fy, cx, cy, mbf, bounds, scale_factors, inv_level_sigma2, log_scale_factor, max_keypoints, max_map_points,
                 grid=(64, 48)):
        self._L = _bind(_lib.load())
        sf = np.ascontiguousarray(scale_factors, np.float32)
        isg = np.ascontiguousarray(inv_level_sigma2, np.float32)
        cfg = TrackerCfg(fx, fy, cx, cy, mbf, bounds[0], bounds[1], bounds[2], bounds[3], grid[0], grid[1], len(sf), _lib.ptr(sf), _lib.ptr(isg),
                         log_scale_factor, max_keypoints, max_map_points)
        self._h = _P()
        _lib.check(self._L.eao_tracker_create(C.byref(cfg), C.byref(self._h)))
        self.cap = int(max_keypoints)
        self.cap_mp = int(max_map_points)
        self.n_mp = 0

    def __del__(self):
        try:
            if self._h:
                self._L.eao_tracker_destroy(self._h)
        except Exception:
            pass

    def set_local_map(self, pts):
        m, keep = map_points(pts)
        _lib.check(self._L.eao_tracker_set_local_map(self._h, C.byref(m)))
        self.n_mp = int(m.n)

    def set_distortion(self, dist_coef):
        """mDistCoef of the camera (k1, k2, p1, p2[, k3]); persistent.  With a non-zero k1 every track_* call undistorts the frame's keypoints on the device first
        (Frame::UndistortKeyPoints); `bounds` of the constructor are then the undistorted image bounds (frame.compute_image_bounds)."""
        d = np.ascontiguousarray(dist_coef, np.float32)
        self._dist = d
        _lib.check(self._L.eao_tracker_set_distortion(self._h, d.ctypes.data if len(d) else None, len(d)))

    def set_options(self, min_matches=0, planes=None):
        """One-shot options of the next track_* call (eao_tracker_set_options).  planes: dict(plane_world (m, 4), plane_obs (m, 4), plane_seen (m)); the plane outlier
        flags of that call are then in self.plane_outlier afterwards."""
        O = TrackOptions()
        O.min_matches = int(min_matches)
        self.plane_outlier = None
        if planes is not None and len(planes["plane_world"]):
            self._pw = np.ascontiguousarray(planes["plane_world"], np.float32); self._po = np.ascontiguousarray(planes["plane_obs"], np.float32)
            self._ps = np.ascontiguousarray(planes["plane_seen"], np.uint8)
            self.plane_outlier = np.zeros(len(self._ps), np.uint8)
            O.n_planes, O.plane_world, O.plane_obs, O.plane_seen, O.plane_outlier = len(self._ps), _lib.ptr(self._pw), _lib.ptr(self._po), _lib.ptr(self._ps), _lib.ptr(self.plane_outlier)
        _lib.check(self._L.eao_tracker_set_options(self._h, C.byref(O)))

    def track_local_map(self, d_kps, d_desc, d_n, d_depth, depth_pitch, width, height, Tcw_prior, prior=None, th=1.0, nnratio=0.8, stream=0, prior_Xw=None):
        """d_*: integers (HBM addresses).  prior: per keypoint -1 / local-map index / -2 (map point outside the local map, position in
        prior_Xw[k]).  Returns dict(Tcw, n_keypoints, n_matches, n_edges, n_inliers, kp_map_point, kp_outlier, u_right, depth, map_in_view)."""
        T = np.ascontiguousarray(Tcw_prior, np.float32).reshape(4, 4)
        kpmp = np.full(self.cap, -1, np.int32)
        outl = np.zeros(self.cap, np.uint8)
        ur = np.zeros(self.cap, np.float32)
        dz = np.zeros(self.cap, np.float32)
        pr = None
        if prior is not None:
            pr = np.full(self.cap, -1, np.int32)
            pr[:len(prior)] = np.asarray(prior, np.int32)
        px = None
        if prior_Xw is not None:
            px = np.zeros((self.cap, 3), np.float32)
            px[:len(prior_Xw)] = np.asarray(prior_Xw, np.float32).reshape(-1, 3)
        inview = np.zeros(self.cap_mp, np.uint8)
        R = TrackResult()
        R.kp_map_point, R.kp_outlier, R.kp_u_right, R.kp_depth = _lib.ptr(kpmp), _lib.ptr(outl), _lib.ptr(ur), _lib.ptr(dz)
        R.map_in_view = _lib.ptr(inview)
        _lib.check(self._L.eao_tracker_track_local_map(self._h, d_kps, d_desc, d_n, d_depth, depth_pitch, width, height, _lib.ptr(T), _lib.ptr(pr),
                                                      _lib.ptr(px), th, nnratio, C.byref(R), stream))
        n = R.n_keypoints
        return dict(Tcw=np.array(R.Tcw, np.float32).reshape(4, 4), n_keypoints=n, n_matches=R.n_matches, n_edges=R.n_edges, n_inliers=R.n_inliers,
                    kp_map_point=kpmp[:n], kp_outlier=outl[:n], u_right=ur[:n], depth=dz[:n], map_in_view=inview[:self.n_mp])

    def track_reference_keyframe(self, d_kps, d_desc, d_n, d_depth, depth_pitch, width, height, Tcw_last, kf, fv_cur, nnratio=0.7, check_orientation=True,
                                 discard_outliers=True, stream=0):
        """Tracking::TrackReferenceKeyFrame's data path (src/Tracking.cc:1568-1631).  kf: dict(valid, Xw, descriptors, angle, fv) per keyframe keypoint, fv /
        fv_cur: {node id: [keypoint indices]} (DBoW2::FeatureVector).  kp_map_point holds KEYFRAME keypoint indices."""
        from .search import feature_vector
        Tl = np.ascontiguousarray(Tcw_last, np.float32).reshape(4, 4)
        valid = np.ascontiguousarray(kf["valid"], np.uint8)
        Xw = np.ascontiguousarray(kf["Xw"], np.float32)
        desc = np.ascontiguousarray(kf["descriptors"], np.uint8)
        ang = np.ascontiguousarray(kf["angle"], np.float32)
        f1, keep1 = feature_vector(kf["fv"])
        f2, keep2 = feature_vector(fv_cur)
        kpmp = np.full(self.cap, -1, np.int32)
        outl = np.zeros(self.cap, np.uint8)
        ur = np.zeros(self.cap, np.float32)
        dz = np.zeros(self.cap, np.float32)
        R = TrackResult()
        R.kp_map_point, R.kp_outlier, R.kp_u_right, R.kp_depth = _lib.ptr(kpmp), _lib.ptr(outl), _lib.ptr(ur), _lib.ptr(dz)
        _lib.check(self._L.eao_tracker_track_reference_keyframe(self._h, d_kps, d_desc, d_n, d_depth, depth_pitch, width, height, _lib.ptr(Tl), len(valid), _lib.ptr(valid),
                                                               _lib.ptr(Xw), _lib.ptr(desc), _lib.ptr(ang), C.byref(f1), C.byref(f2), nnratio,
                                                               1 if check_orientation else 0, 1 if discard_outliers else 0, C.byref(R), stream))
        n = R.n_keypoints
        return dict(Tcw=np.array(R.Tcw, np.float32).reshape(4, 4), n_keypoints=n, n_matches=R.n_matches, n_edges=R.n_edges, n_inliers=R.n_inliers,
                    kp_map_point=kpmp[:n], kp_outlier=outl[:n], u_right=ur[:n], depth=dz[:n])

    def track_with_motion_model(self, d_kps, d_desc, d_n, d_depth, depth_pitch, width, height, Tcw_cur, last, th, mono=False, check_orientation=True,
                                discard_outliers=True, stream=0):
        """Tracking::TrackWithMotionModel's data path (src/Tracking.cc:1717-2231).  last: dict(Tcw, valid, Xw, descriptors, octave, angle) per
        last-frame keypoint.  kp_map_point holds LAST-FRAME indices."""
        Tc = np.ascontiguousarray(Tcw_cur, np.float32).reshape(4, 4)
        Tl = np.ascontiguousarray(last["Tcw"], np.float32).reshape(4, 4)
        valid = np.ascontiguousarray(last["valid"], np.uint8)
        Xw = np.ascontiguousarray(last["Xw"], np.float32)
        desc = np.ascontiguousarray(last["descriptors"], np.uint8)
        octv = np.ascontiguousarray(last["octave"], np.int32)
        ang = np.ascontiguousarray(last["angle"], np.float32)
        kpmp = np.full(self.cap, -1, np.int32)
        outl = np.zeros(self.cap, np.uint8)
        ur = np.zeros(self.cap, np.float32)
        dz = np.zeros(self.cap, np.float32)
        R = TrackResult()
        R.kp_map_point, R.kp_outlier, R.kp_u_right, R.kp_depth = _lib.ptr(kpmp), _lib.ptr(outl), _lib.ptr(ur), _lib.ptr(dz)
        _lib.check(self._L.eao_tracker_track_with_motion_model(self._h, d_kps, d_desc, d_n, d_depth, depth_pitch, width, height, _lib.ptr(Tc), _lib.ptr(Tl),
                                                              len(valid), _lib.ptr(valid), _lib.ptr(Xw), _lib.ptr(desc), _lib.ptr(octv), _lib.ptr(ang), th,
                                                              1 if mono else 0, 1 if check_orientation else 0, 1 if discard_outliers else 0, C.byref(R), stream))
        n = R.n_keypoints
        return dict(Tcw=np.array(R.Tcw, np.float32).reshape(4, 4), n_keypoints=n, n_matches=R.n_matches, n_edges=R.n_edges, n_inliers=R.n_inliers,
                    kp_map_point=kpmp[:n], kp_outlier=outl[:n], u_right=ur[:n], depth=dz[:n])
