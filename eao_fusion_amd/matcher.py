"""ORBmatcher -- the distance kernel of the reference matcher (reference include/ORBmatcher.h:37-102,
src/ORBmatcher.cc:37-39,1649-1665).  Distances come from libeaofusion_hip.so; the greedy assignment passes of
the Search* routines replay on the host over these distances."""
import ctypes as C

import numpy as np

from . import _lib


def hamming_matrix(A, B):
    A = np.ascontiguousarray(A, np.uint8)
    B = np.ascontiguousarray(B, np.uint8)
    D = np.zeros((len(A), len(B)), np.uint16)
    _lib.check(_lib.load().eao_hamming_matrix(_lib.ptr(A), len(A), _lib.ptr(B), len(B), _lib.ptr(D)))
    return D


def hamming_best2(A, B, mask=None):
    """Per row of A: (best, second, idx, idx2) over the allowed columns; first column wins ties."""
    A = np.ascontiguousarray(A, np.uint8)
    B = np.ascontiguousarray(B, np.uint8)
    out = np.zeros((len(A), 4), np.int32)
    if mask is not None:
        mask = np.ascontiguousarray(mask, np.uint8)
        assert mask.shape == (len(A), len(B))
    _lib.check(_lib.load().eao_hamming_best2(_lib.ptr(A), len(A), _lib.ptr(B), len(B), _lib.ptr(mask), _lib.ptr(out)))
    return out


def distinctive_descriptors(sets):
    """MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:242-307) for a batch of map points: `sets` is a list of
    (N_s, 32) uint8 arrays (the descriptors of the keyframes observing each point).  Returns the index of the chosen
    descriptor inside each set (-1 for an empty set)."""
    start = np.zeros(len(sets) + 1, np.int32)
    for i, d in enumerate(sets):
        start[i + 1] = start[i] + len(d)
    desc = np.ascontiguousarray(np.concatenate([np.asarray(d, np.uint8).reshape(-1, 32) for d in sets]) if len(sets) else np.zeros((0, 32), np.uint8))
    best = np.full(len(sets), -1, np.int32)
    _lib.check(_lib.load().eao_distinctive_descriptors(len(sets), _lib.ptr(start), _lib.ptr(desc), _lib.ptr(best)))
    return best


class ORBmatcher:
    TH_LOW = 50
    TH_HIGH = 100
    HISTO_LENGTH = 30

    def __init__(self, nnratio=0.6, checkOri=True):
        self.mfNNratio = float(nnratio)
        self.mbCheckOrientation = bool(checkOri)

    @staticmethod
    def DescriptorDistance(a, b):
        return int(hamming_matrix(np.asarray(a).reshape(1, 32), np.asarray(b).reshape(1, 32))[0, 0])


def make_frame_view(frame, view_cls=None):
    """frame: dict with kp_x, kp_y (f32), kp_octave (i32), kp_angle, u_right (f32), descriptors (n,32) u8, optional
    occupied (u8), min_x, min_y, max_x, max_y, scale_factors (f32); the grid is the reference's 64 x 48
    (include/Frame.h:89-90) with mfGridElementWidthInv = 64 / (mnMaxX - mnMinX) (src/Frame.cc:213-214).
    Returns (struct, keepalive list)."""
    cls = view_cls or _lib.FrameView
    keep = {}
    for k, dt in (("kp_x", np.float32), ("kp_y", np.float32), ("kp_octave", np.int32), ("kp_angle", np.float32),
                  ("u_right", np.float32), ("descriptors", np.uint8), ("scale_factors", np.float32)):
        keep[k] = np.ascontiguousarray(frame[k], dt)
    occ = frame.get("occupied")
    keep["occupied"] = None if occ is None else np.ascontiguousarray(occ, np.uint8)
    cols, rows = int(frame.get("grid_cols", 64)), int(frame.get("grid_rows", 48))
    inv_w = np.float32(cols) / np.float32(np.float32(frame["max_x"]) - np.float32(frame["min_x"]))
    inv_h = np.float32(rows) / np.float32(np.float32(frame["max_y"]) - np.float32(frame["min_y"]))
    v = cls(len(keep["kp_x"]), _lib.ptr(keep["kp_x"]), _lib.ptr(keep["kp_y"]), _lib.ptr(keep["kp_octave"]), _lib.ptr(keep["kp_angle"]),
            _lib.ptr(keep["u_right"]), _lib.ptr(keep["descriptors"]), _lib.ptr(keep["occupied"]),
            frame["min_x"], frame["min_y"], frame["max_x"], frame["max_y"], inv_w, inv_h, cols, rows,
            _lib.ptr(keep["scale_factors"]), len(keep["scale_factors"]), 0.0, None, None)
    return v, keep


def _search_points(self, frame, mps, th):
    """SearchByProjection(Frame&, const vector<MapPoint*>&, th) -- reference src/ORBmatcher.cc:45-129.
    mps: dict with proj_x, proj_y, proj_xr, view_cos (f32), level (i32), descriptors (m,32), optional skip (u8).
    Returns (nmatches, match_kp[m])."""
    v, keep = make_frame_view(frame)
    a = {k: np.ascontiguousarray(mps[k], np.float32) for k in ("proj_x", "proj_y", "proj_xr", "view_cos")}
    lvl = np.ascontiguousarray(mps["level"], np.int32)
    desc = np.ascontiguousarray(mps["descriptors"], np.uint8)
    skip = mps.get("skip")
    skip = None if skip is None else np.ascontiguousarray(skip, np.uint8)
    out = np.full(len(lvl), -1, np.int32)
    nm = C.c_int32()
    _lib.check(_lib.load().eao_search_by_projection_points(C.byref(v), len(lvl), _lib.ptr(a["proj_x"]), _lib.ptr(a["proj_y"]),
                                                           _lib.ptr(a["proj_xr"]), _lib.ptr(a["view_cos"]), _lib.ptr(lvl), _lib.ptr(desc),
                                                           _lib.ptr(skip), th, self.mfNNratio, _lib.ptr(out), C.byref(nm)))
    return nm.value, out


def _search_frames(self, cur, last, th, mono):
    """SearchByProjection(Frame& Cur, const Frame& Last, th, bMono) -- reference src/ORBmatcher.cc:1328-1472.
    cur: frame dict + Tcw, fx, fy, cx, cy, mbf, mb; last: dict with Tcw, valid (u8), Xw (n,3), descriptors, octave, angle.
    Returns (nmatches, cur_match[n_cur])."""
    v, keep = make_frame_view(cur)
    Tc = np.ascontiguousarray(cur["Tcw"], np.float32)
    Tl = np.ascontiguousarray(last["Tcw"], np.float32)
    valid = np.ascontiguousarray(last["valid"], np.uint8)
    Xw = np.ascontiguousarray(last["Xw"], np.float32)
    desc = np.ascontiguousarray(last["descriptors"], np.uint8)
    octv = np.ascontiguousarray(last["octave"], np.int32)
    ang = np.ascontiguousarray(last["angle"], np.float32)
    out = np.full(v.n, -1, np.int32)
    nm = C.c_int32()
    _lib.check(_lib.load().eao_search_by_projection_frames(C.byref(v), _lib.ptr(Tc), _lib.ptr(Tl), len(valid), _lib.ptr(valid), _lib.ptr(Xw),
                                                           _lib.ptr(desc), _lib.ptr(octv), _lib.ptr(ang), cur["fx"], cur["fy"], cur["cx"], cur["cy"],
                                                           cur["mbf"], cur["mb"], th, 1 if mono else 0, 1 if self.mbCheckOrientation else 0,
                                                           _lib.ptr(out), C.byref(nm)))
    return nm.value, out


ORBmatcher.SearchByProjectionPoints = _search_points
ORBmatcher.SearchByProjectionFrames = _search_frames
