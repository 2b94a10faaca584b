"""ctypes binding of libeaofusion_hip.so (the C-ABI in include/eao_fusion.h).

The library is the product; this module only loads it.  There is no CPU fallback anywhere in this package:
if the shared object is missing or no MI355X is visible, calls fail loudly.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EAO_LIB_PATH") or os.path.join(HERE, "libeaofusion_hip.so")      # (EAO_LIB_PATH: A/B runs against another build of the library)

EAO_OK, EAO_ERR_INVALID, EAO_ERR_NO_DEVICE, EAO_ERR_CAPACITY, EAO_ERR_INTERNAL = 0, -1, -2, -3, -4


class EaoError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("eao status %d: %s" % (status, msg))
        self.status = status


class OrbCfg(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32)]


class OrbSlot(C.Structure):   # eao_orb_slot
    _fields_ = [("frames", C.c_void_p), ("stride", C.c_int32), ("frame_stride", C.c_int64), ("kps", C.c_void_p), ("desc", C.c_void_p),
                ("n", C.c_void_p), ("cap", C.c_int32)]


class OrbLevelView(C.Structure):   # eao_orb_level_view
    _fields_ = [("data", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32), ("step", C.c_int32)]


class FrameView(C.Structure):
    _fields_ = [("n", C.c_int32), ("kp_x", C.c_void_p), ("kp_y", C.c_void_p), ("kp_octave", C.c_void_p), ("kp_angle", C.c_void_p),
                ("u_right", C.c_void_p), ("descriptors", C.c_void_p), ("occupied", C.c_void_p),
                ("min_x", C.c_float), ("min_y", C.c_float), ("max_x", C.c_float), ("max_y", C.c_float),
                ("grid_inv_w", C.c_float), ("grid_inv_h", C.c_float), ("grid_cols", C.c_int32), ("grid_rows", C.c_int32),
                ("scale_factors", C.c_void_p), ("nlevels", C.c_int32),
                ("log_scale_factor", C.c_float), ("level_sigma2", C.c_void_p), ("inv_level_sigma2", C.c_void_p)]


class PoseProblem(C.Structure):
    _fields_ = [("n", C.c_int32), ("Tcw", C.c_void_p), ("Xw", C.c_void_p), ("obs", C.c_void_p),
                ("inv_sigma2", C.c_void_p), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float),
                ("cy", C.c_float), ("bf", C.c_float),
                ("n_planes", C.c_int32), ("plane_world", C.c_void_p), ("plane_obs", C.c_void_p), ("plane_seen", C.c_void_p)]


class PoseResult(C.Structure):
    _fields_ = [("Tcw", C.c_float * 16), ("outlier", C.c_void_p), ("n_inliers", C.c_int32),
                ("lm_iterations", C.c_int32), ("plane_outlier", C.c_void_p)]


class BAProblem(C.Structure):
    _fields_ = [("n_cams", C.c_int32), ("n_points", C.c_int32), ("n_edges", C.c_int32),
                ("cam_Tcw", C.c_void_p), ("cam_fixed", C.c_void_p), ("points", C.c_void_p),
                ("edge_cam", C.c_void_p), ("edge_point", C.c_void_p), ("edge_obs", C.c_void_p),
                ("edge_inv_sigma2", C.c_void_p), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float),
                ("cy", C.c_float), ("bf", C.c_float), ("its_first", C.c_int32), ("its_second", C.c_int32)]


class BAPlanes(C.Structure):   # eao_ba_planes
    _fields_ = [("n_planes", C.c_int32), ("plane_world", C.c_void_p), ("n_pedges", C.c_int32), ("pedge_plane", C.c_void_p),
                ("pedge_cam", C.c_void_p), ("pedge_obs", C.c_void_p)]


class BAResult(C.Structure):
    _fields_ = [("cam_Tcw", C.c_void_p), ("points", C.c_void_p), ("edge_outlier", C.c_void_p),
                ("iters", C.c_int32 * 2), ("aborted", C.c_int32), ("chi2", C.c_double * 2)]


# every symbol include/eao_fusion.h declares: (restype, argtypes)
_P = C.c_void_p
_I = C.c_int32
SYMBOLS = {
    "eao_last_error": (C.c_char_p, []),
    "eao_device_check": (_I, []),
    "eao_version": (C.c_char_p, []),
    "eao_orb_create": (_I, [C.POINTER(OrbCfg), C.POINTER(_P)]),
    "eao_orb_destroy": (None, [_P]),
    "eao_orb_tables": (_I, [_P, _P, _P, _P, _P, _P]),
    "eao_orb_max_keypoints": (_I, [_P, _I, _I, C.POINTER(_I)]),
    "eao_orb_extract": (_I, [_P, _P, _I, _I, _I, _P, _P, _I, C.POINTER(_I)]),
    "eao_orb_extract_batch": (_I, [_P, _P, _I, _I, _I, C.c_int64, _I, _P, _P, _I, _P]),
    "eao_orb_extract_batch_device": (_I, [_P, _P, _I, _I, _I, C.c_int64, _I, _P, _P, _I, _P, _P]),
    "eao_orb_stream_create": (_I, [_P, _I, _I, _I, _I]),
    "eao_orb_stream_slot": (_I, [_P, _I, C.POINTER(OrbSlot)]),
    "eao_orb_stream_submit": (_I, [_P, _I, _I]),
    "eao_orb_stream_wait": (_I, [_P, _I]),
    "eao_orb_level": (_I, [_P, _I, _I, _I, C.POINTER(_I), C.POINTER(_I), _P]),
    "eao_orb_pyramid": (_I, [_P, _I, _I, C.POINTER(OrbLevelView)]),
    "eao_orb_set_keep_pyramid": (_I, [_P, _I]),
    "eao_orb_extract_ref": (_I, [_P, _P, _I, _I, _I, C.POINTER(_P), C.POINTER(_P), C.POINTER(_I)]),
    "eao_orb_level_candidates": (_I, [_P, _I, _I, _P, _I, C.POINTER(_I)]),
    "eao_orb_set_profiling": (_I, [_P, _I]),
    "eao_orb_last_timing": (_I, [_P, _P]),
    "eao_orb_lanes": (_I, [_I]),
    "eao_hamming_matrix": (_I, [_P, _I, _P, _I, _P]),
    "eao_hamming_best2": (_I, [_P, _I, _P, _I, _P, _P]),
    "eao_hamming_matrix_device": (_I, [_P, _I, _P, _I, _I, _P, _P]),
    "eao_hamming_best2_device": (_I, [_P, _I, _P, _I, _I, _P, _P, _P]),
    "eao_hamming_best2_sequence_device": (_I, [_P, _I, _P, _I, _P, _I, _P, _P]),
    "eao_search_by_projection_points": (_I, [C.POINTER(FrameView), _I, _P, _P, _P, _P, _P, _P, _P, C.c_float, C.c_float, _P, C.POINTER(_I)]),
    "eao_search_by_projection_frames": (_I, [C.POINTER(FrameView), _P, _P, _I, _P, _P, _P, _P, _P] + [C.c_float] * 7 + [_I, _I, _P, C.POINTER(_I)]),
    # the remaining guided searches: argument lists live in search.py (SEARCH_ARGTYPES), bound there
    "eao_search_by_projection_sim3": None, "eao_search_by_projection_kf": None, "eao_search_by_bow": None,
    "eao_search_for_triangulation": None, "eao_search_for_initialization": None, "eao_fuse_search": None, "eao_search_by_sim3": None,
    "eao_search_for_triangulation_batch": None, "eao_fuse_search_batch": None,
    # keyframe handles (round 5): argument lists live in search.py (HandleBinding)
    "eao_keyframe_create": None, "eao_keyframe_update_points": None, "eao_keyframe_destroy": None, "eao_keyframe_size": None,
    "eao_kf_search_by_bow": None, "eao_kf_search_for_triangulation": None, "eao_kf_fuse_search": None, "eao_kf_search_by_projection_sim3": None,
    "eao_kf_search_by_projection_kf": None, "eao_kf_search_for_initialization": None, "eao_kf_search_by_sim3": None,
    # f1, second half (the device-resident tracked frame): argument lists live in tracker.py
    "eao_tracker_create": None, "eao_tracker_destroy": None, "eao_tracker_set_local_map": None, "eao_tracker_track_local_map": None, "eao_tracker_set_options": None, "eao_tracker_set_distortion": None,
    "eao_tracker_track_with_motion_model": None, "eao_tracker_track_reference_keyframe": None, "eao_abi_version": (_I, []),
    # f1 (Frame glue): argument lists live in frame.py
    "eao_frame_is_in_frustum": None, "eao_assign_features_to_grid": None, "eao_compute_stereo_from_rgbd": None, "eao_undistort_keypoints": None,
    "eao_compute_image_bounds": None,
    "eao_distinctive_descriptors": (_I, [_I, _P, _P, _P]),
    "eao_compute_stereo_matches": (_I, [_P, _P, _I, _I, _P, _P, _I, _P, _P, C.c_float, C.c_float, _P, _P]),
    "eao_pose_optimization": (_I, [C.POINTER(PoseProblem), C.POINTER(PoseResult)]),
    "eao_pose_optimization_batch": (_I, [C.POINTER(PoseProblem), C.c_int32, C.POINTER(PoseResult)]),
    "eao_local_ba": (_I, [C.POINTER(BAProblem), _P, C.POINTER(BAResult)]),
    "eao_local_ba_batch": (_I, [C.POINTER(BAProblem), _I, _P, C.POINTER(BAResult)]),
    "eao_bundle_adjustment": (_I, [C.POINTER(BAProblem), _I, _P, C.POINTER(BAResult)]),
    "eao_bundle_adjustment_planes": (_I, [C.POINTER(BAProblem), C.POINTER(BAPlanes), _I, _P, C.POINTER(BAResult), _P]),
    "eao_last_lm_trace": (_I, [_P, _P, _P, _I, C.POINTER(_I)]),
    "eao_last_lm_timing": (_I, [C.POINTER(C.c_float), C.POINTER(_I)]),
    "eao_bundle_adjustment_plan": (_I, [_I, _I, _P, _P, _I, _P, _P, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _I]),
}

_lib = None


def load():
    """Load the shared object (after torch, if torch is in the process, so that both resolve the same
    libamdhip64.so.7 -- see DESIGN.md 'process model')."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, sig in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError = ABI drift, fail loudly
            if sig is not None:
                fn.restype, fn.argtypes = sig
        _lib = L
    return _lib


def check(status):
    if status != EAO_OK:
        raise EaoError(status, load().eao_last_error().decode("utf-8", "replace"))


def ptr(a):
    return None if a is None else a.ctypes.data
