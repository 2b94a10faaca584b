"""ctypes mirror of the Frame glue either side of the matcher (include/eao_fusion.h, row f1): eao_frame_is_in_frustum,
eao_assign_features_to_grid, eao_compute_stereo_from_rgbd -- reference src/Frame.cc:597-614, 638-695, 751-761, 1016-1037.

`Binding(lib, check)` binds the wrappers to libeaofusion_hip.so (see product() below)."""
import ctypes as C

import numpy as np

from .search import MapPoints, map_points

_P = C.c_void_p
_I = C.c_int32
_F = C.c_float


class FrustumFrame(C.Structure):
    _fields_ = [("Tcw", _F * 16), ("Ow", _F * 3), ("fx", _F), ("fy", _F), ("cx", _F), ("cy", _F), ("mbf", _F),
                ("min_x", _F), ("max_x", _F), ("min_y", _F), ("max_y", _F), ("log_scale_factor", _F)]


def _p(a):
    return None if a is None else a.ctypes.data


class Binding:
    """The Frame glue bound to libeaofusion_hip.so.  The three `_raw_*` methods are the only places that touch the library;
    the test suite derives its checker binding from this class by overriding exactly those."""

    def __init__(self, lib, check):
        self.lib, self.check = lib, check
        lib.eao_frame_is_in_frustum.restype = _I
        lib.eao_frame_is_in_frustum.argtypes = [C.POINTER(FrustumFrame), C.POINTER(MapPoints), _F] + [_P] * 6
        lib.eao_assign_features_to_grid.restype = _I
        lib.eao_assign_features_to_grid.argtypes = [_I, _P, _P, _F, _F, _F, _F, _I, _I, _P, _P]
        lib.eao_compute_stereo_from_rgbd.restype = _I
        lib.eao_compute_stereo_from_rgbd.argtypes = [_I, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P, _P]
        lib.eao_undistort_keypoints.restype = _I
        lib.eao_undistort_keypoints.argtypes = [_I, _P, _P, _F, _F, _F, _F, _P, _I, _P, _P]
        lib.eao_compute_image_bounds.restype = _I
        lib.eao_compute_image_bounds.argtypes = [_I, _I, _F, _F, _F, _F, _P, _I, _P]

    def _raw_is_in_frustum(self, m, keep, T, Ow, sc, limit, outs):
        F = FrustumFrame()
        F.Tcw[:] = T.ravel().tolist(); F.Ow[:] = Ow.tolist()
        (F.fx, F.fy, F.cx, F.cy, F.mbf, F.min_x, F.max_x, F.min_y, F.max_y, F.log_scale_factor) = sc
        self.check(self.lib.eao_frame_is_in_frustum(C.byref(F), C.byref(m), limit, *outs))

    def _raw_assign(self, n, kx, ky, min_x, min_y, inv_w, inv_h, cols, rows, start, items):
        self.check(self.lib.eao_assign_features_to_grid(n, kx, ky, min_x, min_y, inv_w, inv_h, cols, rows, start, items))

    def _raw_rgbd(self, n, kx, ky, ku, d, w, h, mbf, ur, dz):
        self.check(self.lib.eao_compute_stereo_from_rgbd(n, kx, ky, ku, d, w, h, w, 0, mbf, ur, dz))

    def _raw_undistort(self, n, kx, ky, fx, fy, cx, cy, dist, nc, ox, oy):
        self.check(self.lib.eao_undistort_keypoints(n, kx, ky, fx, fy, cx, cy, dist, nc, ox, oy))

    def _raw_bounds(self, cols, rows, fx, fy, cx, cy, dist, nc, out):
        self.check(self.lib.eao_compute_image_bounds(cols, rows, fx, fy, cx, cy, dist, nc, out))

    def undistort_keypoints(self, kp_x, kp_y, fx, fy, cx, cy, dist_coef):
        """Frame::UndistortKeyPoints (src/Frame.cc:773-806).  dist_coef: k1, k2, p1, p2[, k3].  Returns (x, y) of mvKeysUn."""
        kx, ky = np.ascontiguousarray(kp_x, np.float32), np.ascontiguousarray(kp_y, np.float32)
        d = np.ascontiguousarray(dist_coef, np.float32)
        ox, oy = np.zeros(len(kx), np.float32), np.zeros(len(kx), np.float32)
        self._raw_undistort(len(kx), _p(kx), _p(ky), float(fx), float(fy), float(cx), float(cy), _p(d) if len(d) else None, len(d), _p(ox), _p(oy))
        return ox, oy

    def compute_image_bounds(self, cols, rows, fx, fy, cx, cy, dist_coef):
        """Frame::ComputeImageBounds (src/Frame.cc:808-842).  Returns float32 [mnMinX, mnMaxX, mnMinY, mnMaxY]."""
        d = np.ascontiguousarray(dist_coef, np.float32)
        out = np.zeros(4, np.float32)
        self._raw_bounds(int(cols), int(rows), float(fx), float(fy), float(cx), float(cy), _p(d) if len(d) else None, len(d), _p(out))
        return out

    def is_in_frustum(self, frame, pts, viewing_cos_limit=0.5):
        """frame: Tcw (4x4 f32), Ow (3), fx, fy, cx, cy, mbf, min_x, max_x, min_y, max_y, log_scale_factor.
        pts: Xw, normal, min_dist_inv, max_dist_inv, max_dist (+ active / descriptors, unread).
        Returns dict(in_view, proj_x, proj_y, proj_xr, view_cos, pred_level); entries of points out of view are -1 / 0."""
        m, keep = map_points(pts)
        n = m.n
        out = dict(in_view=np.zeros(n, np.uint8), proj_x=np.full(n, -1, np.float32), proj_y=np.full(n, -1, np.float32),
                   proj_xr=np.full(n, -1, np.float32), view_cos=np.zeros(n, np.float32), pred_level=np.full(n, -1, np.int32))
        T = np.ascontiguousarray(frame["Tcw"], np.float32).reshape(4, 4)
        Ow = np.ascontiguousarray(frame["Ow"], np.float32)
        sc = [float(frame[k]) for k in ("fx", "fy", "cx", "cy", "mbf", "min_x", "max_x", "min_y", "max_y", "log_scale_factor")]
        outs = [_p(out[k]) for k in ("in_view", "proj_x", "proj_y", "proj_xr", "view_cos", "pred_level")]
        self._raw_is_in_frustum(m, keep, T, Ow, sc, float(viewing_cos_limit), outs)
        return out

    def assign_features_to_grid(self, kp_x, kp_y, min_x, min_y, max_x, max_y, cols=64, rows=48):
        """Returns (cell_start [cols*rows+1], items): mGrid[ix][iy] = items[cell_start[ix*rows+iy] : cell_start[ix*rows+iy+1]]."""
        kx, ky = np.ascontiguousarray(kp_x, np.float32), np.ascontiguousarray(kp_y, np.float32)
        inv_w = np.float32(cols) / np.float32(np.float32(max_x) - np.float32(min_x))     # mfGridElementWidthInv, src/Frame.cc:258-259
        inv_h = np.float32(rows) / np.float32(np.float32(max_y) - np.float32(min_y))
        start = np.zeros(cols * rows + 1, np.int32)
        items = np.full(max(len(kx), 1), -1, np.int32)
        self._raw_assign(len(kx), _p(kx), _p(ky), float(min_x), float(min_y), float(inv_w), float(inv_h), cols, rows, _p(start), _p(items))
        return start, items[:start[-1]]

    def compute_stereo_from_rgbd(self, kp_x, kp_y, kpu_x, depth, mbf):
        """depth: (H, W) float32 image.  Returns (mvuRight, mvDepth)."""
        kx, ky, ku = (np.ascontiguousarray(a, np.float32) for a in (kp_x, kp_y, kpu_x))
        d = np.ascontiguousarray(depth, np.float32)
        ur, dz = np.zeros(len(kx), np.float32), np.zeros(len(kx), np.float32)
        self._raw_rgbd(len(kx), _p(kx), _p(ky), _p(ku), _p(d), d.shape[1], d.shape[0], float(mbf), _p(ur), _p(dz))
        return ur, dz


_product = None


def product():
    global _product
    if _product is None:
        from . import _lib
        _product = Binding(_lib.load(), _lib.check)
    return _product


def is_in_frustum(frame, pts, viewing_cos_limit=0.5):
    return product().is_in_frustum(frame, pts, viewing_cos_limit)


def assign_features_to_grid(kp_x, kp_y, min_x, min_y, max_x, max_y, cols=64, rows=48):
    return product().assign_features_to_grid(kp_x, kp_y, min_x, min_y, max_x, max_y, cols, rows)


def compute_stereo_from_rgbd(kp_x, kp_y, kpu_x, depth, mbf):
    return product().compute_stereo_from_rgbd(kp_x, kp_y, kpu_x, depth, mbf)


def undistort_keypoints(kp_x, kp_y, fx, fy, cx, cy, dist_coef):
    return product().undistort_keypoints(kp_x, kp_y, fx, fy, cx, cy, dist_coef)


def compute_image_bounds(cols, rows, fx, fy, cx, cy, dist_coef):
    return product().compute_image_bounds(cols, rows, fx, fy, cx, cy, dist_coef)
