"""Optimizer -- PoseOptimization / LocalBundleAdjustment over flattened problems (reference
include/Optimizer.h:55-56, src/Optimizer.cc:325-673,675-1138); compute is eao_pose_optimization /
eao_local_ba in libeaofusion_hip.so."""
import ctypes as C

import numpy as np

from . import _lib


def _trace():
    L = _lib.load()
    lam = np.zeros(128)
    chi = np.zeros(128)
    tr = np.zeros(128, np.int32)
    n = C.c_int32()
    _lib.check(L.eao_last_lm_trace(_lib.ptr(lam), _lib.ptr(chi), _lib.ptr(tr), 128, C.byref(n)))
    return dict(lam=lam[:n.value], chi2=chi[:n.value], trials=tr[:n.value])


def _timing():
    ms, lin = C.c_float(), C.c_int32()
    if _lib.load().eao_last_lm_timing(C.byref(ms), C.byref(lin)) != 0:
        return None
    return dict(device_ms=ms.value, linearizations=lin.value)


class Optimizer:
    @staticmethod
    def PoseOptimization(prob):
        """prob: Tcw (4,4) f32, points (n,3) f32, obs (n,3) f32 [u, v, ur (<0 = mono)], inv_sigma2 (n,), fx..bf, optional
        plane_world / plane_obs (m,4) f32 + plane_seen (m,) u8 (the plane edges of src/Optimizer.cc:456-535).
        Returns dict(Tcw, outlier, [plane_outlier,] n_inliers) -- n_inliers is the reference's return value."""
        Tcw = np.ascontiguousarray(prob["Tcw"], np.float32)
        Xw = np.ascontiguousarray(prob["points"], np.float32)
        obs = np.ascontiguousarray(prob["obs"], np.float32)
        inv = np.ascontiguousarray(prob["inv_sigma2"], np.float32)
        n = len(Xw)
        outl = np.zeros(max(n, 1), np.uint8)
        pw = prob.get("plane_world")
        m = 0 if pw is None else len(pw)
        pw = None if pw is None else np.ascontiguousarray(pw, np.float32)
        po = None if pw is None else np.ascontiguousarray(prob["plane_obs"], np.float32)
        ps = None if pw is None else np.ascontiguousarray(prob["plane_seen"], np.uint8)
        pout = np.zeros(max(m, 1), np.uint8)
        P = _lib.PoseProblem(n, _lib.ptr(Tcw), _lib.ptr(Xw), _lib.ptr(obs), _lib.ptr(inv), prob["fx"], prob["fy"],
                             prob["cx"], prob["cy"], prob["bf"], m, _lib.ptr(pw), _lib.ptr(po), _lib.ptr(ps))
        R = _lib.PoseResult()
        R.outlier = _lib.ptr(outl)
        R.plane_outlier = _lib.ptr(pout)
        _lib.check(_lib.load().eao_pose_optimization(C.byref(P), C.byref(R)))
        out = dict(Tcw=np.array(R.Tcw, np.float32).reshape(4, 4), outlier=outl[:n], n_inliers=R.n_inliers,
                   lm_iterations=R.lm_iterations, trace=_trace(), timing=_timing())
        if m:
            out["plane_outlier"] = pout[:m]
        return out

    @staticmethod
    def pack_pose_batch(probs):
        """The argument arrays of eao_pose_optimization_batch for a list of frames (kept alive by the returned object)."""
        nb = len(probs)
        Ps = (_lib.PoseProblem * max(nb, 1))()
        Rs = (_lib.PoseResult * max(nb, 1))()
        keep = []
        for b, prob in enumerate(probs):
            Tcw = np.ascontiguousarray(prob["Tcw"], np.float32)
            Xw = np.ascontiguousarray(prob["points"], np.float32)
            obs = np.ascontiguousarray(prob["obs"], np.float32)
            inv = np.ascontiguousarray(prob["inv_sigma2"], np.float32)
            n = len(Xw)
            pw = prob.get("plane_world")
            m = 0 if pw is None else len(pw)
            pw = None if pw is None else np.ascontiguousarray(pw, np.float32)
            po = None if pw is None else np.ascontiguousarray(prob["plane_obs"], np.float32)
            ps = None if pw is None else np.ascontiguousarray(prob["plane_seen"], np.uint8)
            outl = np.zeros(max(n, 1), np.uint8)
            pout = np.zeros(max(m, 1), np.uint8)
            Ps[b] = _lib.PoseProblem(n, _lib.ptr(Tcw), _lib.ptr(Xw), _lib.ptr(obs), _lib.ptr(inv), prob["fx"], prob["fy"],
                                     prob["cx"], prob["cy"], prob["bf"], m, _lib.ptr(pw), _lib.ptr(po), _lib.ptr(ps))
            Rs[b].outlier = _lib.ptr(outl)
            Rs[b].plane_outlier = _lib.ptr(pout)
            keep.append((Tcw, Xw, obs, inv, pw, po, ps, outl, pout, n, m))
        return dict(P=Ps, R=Rs, n=nb, keep=keep)

    @staticmethod
    def PoseOptimizationBatch(probs, packed=None):
        """One PoseOptimization per element of `probs` (the candidate loop of Tracking::Relocalization, src/Tracking.cc:2786-2940)
        in a single eao_pose_optimization_batch call.  Returns a list of the dicts PoseOptimization returns."""
        pk = packed or Optimizer.pack_pose_batch(probs)
        nb, Rs, keep = pk["n"], pk["R"], pk["keep"]
        _lib.check(_lib.load().eao_pose_optimization_batch(pk["P"], nb, Rs))
        outs = []
        for b in range(nb):
            outl, pout, n, m = keep[b][7:]
            out = dict(Tcw=np.array(Rs[b].Tcw, np.float32).reshape(4, 4), outlier=outl[:n], n_inliers=Rs[b].n_inliers,
                       lm_iterations=Rs[b].lm_iterations)
            if m:
                out["plane_outlier"] = pout[:m]
            outs.append(out)
        return outs

    @staticmethod
    def LocalBundleAdjustment(prob, stop=None, its=(5, 10), gba=None):
        """prob: poses (n_cams,4,4) f32, fixed (n_cams,) u8, points (n_points,3) f32, edge_cam, edge_point (E,) i32,
        obs (E,3) f32, inv_sigma2 (E,) f32, fx..bf.  stop: optional np.uint8[1] polled between LM iterations."""
        cams = np.ascontiguousarray(prob["poses"], np.float32)
        fixed = np.ascontiguousarray(prob["fixed"], np.uint8)
        pts = np.ascontiguousarray(prob["points"], np.float32)
        ec = np.ascontiguousarray(prob["edge_cam"], np.int32)
        ep = np.ascontiguousarray(prob["edge_point"], np.int32)
        obs = np.ascontiguousarray(prob["obs"], np.float32)
        inv = np.ascontiguousarray(prob["inv_sigma2"], np.float32)
        P = _lib.BAProblem(len(cams), len(pts), len(ec), _lib.ptr(cams), _lib.ptr(fixed), _lib.ptr(pts), _lib.ptr(ec),
                           _lib.ptr(ep), _lib.ptr(obs), _lib.ptr(inv), prob["fx"], prob["fy"], prob["cx"], prob["cy"],
                           prob["bf"], its[0], its[1])
        cams_out = np.zeros_like(cams)
        pts_out = np.zeros_like(pts)
        outl = np.zeros(max(len(ec), 1), np.uint8)
        R = _lib.BAResult()
        R.cam_Tcw, R.points, R.edge_outlier = _lib.ptr(cams_out), _lib.ptr(pts_out), _lib.ptr(outl)
        stop_p = None
        if stop is not None:
            stop = np.ascontiguousarray(stop, np.uint8)
            stop_p = _lib.ptr(stop)
        planes_out = None
        if gba is None:
            _lib.check(_lib.load().eao_local_ba(C.byref(P), stop_p, C.byref(R)))
        elif prob.get("planes") is not None and len(prob["planes"]):
            pw = np.ascontiguousarray(prob["planes"], np.float32)
            pp = np.ascontiguousarray(prob["pedge_plane"], np.int32)
            pc = np.ascontiguousarray(prob["pedge_cam"], np.int32)
            po = np.ascontiguousarray(prob["pedge_obs"], np.float32)
            PL = _lib.BAPlanes(len(pw), _lib.ptr(pw), len(pp), _lib.ptr(pp), _lib.ptr(pc), _lib.ptr(po))
            planes_out = np.zeros_like(pw)
            _lib.check(_lib.load().eao_bundle_adjustment_planes(C.byref(P), C.byref(PL), 1 if gba else 0, stop_p, C.byref(R), _lib.ptr(planes_out)))
        else:
            _lib.check(_lib.load().eao_bundle_adjustment(C.byref(P), 1 if gba else 0, stop_p, C.byref(R)))
        out = dict(poses=cams_out, points=pts_out, edge_outlier=outl[:len(ec)], iters=np.array(R.iters[:]),
                   aborted=bool(R.aborted), chi2=np.array(R.chi2[:]), trace=_trace(), timing=_timing())
        if planes_out is not None:
            out["planes"] = planes_out
        return out

    @staticmethod
    def pack_batch(probs, its=(5, 10)):
        """The argument arrays of eao_local_ba_batch for a list of windows (kept alive by the returned object)."""
        n = len(probs)
        P, R = (_lib.BAProblem * n)(), (_lib.BAResult * n)()
        keep, outs = [], []
        for w, prob in enumerate(probs):
            cams = np.ascontiguousarray(prob["poses"], np.float32); fixed = np.ascontiguousarray(prob["fixed"], np.uint8)
            pts = np.ascontiguousarray(prob["points"], np.float32); ec = np.ascontiguousarray(prob["edge_cam"], np.int32)
            ep = np.ascontiguousarray(prob["edge_point"], np.int32); obs = np.ascontiguousarray(prob["obs"], np.float32)
            inv = np.ascontiguousarray(prob["inv_sigma2"], np.float32)
            P[w] = _lib.BAProblem(len(cams), len(pts), len(ec), _lib.ptr(cams), _lib.ptr(fixed), _lib.ptr(pts), _lib.ptr(ec), _lib.ptr(ep),
                                  _lib.ptr(obs), _lib.ptr(inv), prob["fx"], prob["fy"], prob["cx"], prob["cy"], prob["bf"], its[0], its[1])
            co, po, ol = np.zeros_like(cams), np.zeros_like(pts), np.zeros(max(len(ec), 1), np.uint8)
            R[w].cam_Tcw, R[w].points, R[w].edge_outlier = _lib.ptr(co), _lib.ptr(po), _lib.ptr(ol)
            keep.append((cams, fixed, pts, ec, ep, obs, inv))
            outs.append((co, po, ol, len(ec)))
        return dict(P=P, R=R, n=n, keep=keep, outs=outs)

    @staticmethod
    def LocalBundleAdjustmentBatch(probs, stop=None, its=(5, 10), packed=None):
        """n independent windows through ONE eao_local_ba_batch call (the window is a grid dimension of every launch).
        Returns a list of result dicts shaped like LocalBundleAdjustment's (without the per-window LM trace)."""
        pk = packed or Optimizer.pack_batch(probs, its)
        stop_p = None
        if stop is not None:
            stop = np.ascontiguousarray(stop, np.uint8)
            stop_p = _lib.ptr(stop)
        _lib.check(_lib.load().eao_local_ba_batch(pk["P"], pk["n"], stop_p, pk["R"]))
        tm = _timing()
        return [dict(poses=co, points=po, edge_outlier=ol[:ne], iters=np.array(pk["R"][w].iters[:]), aborted=bool(pk["R"][w].aborted),
                     chi2=np.array(pk["R"][w].chi2[:]), timing=tm) for w, (co, po, ol, ne) in enumerate(pk["outs"])]

    @staticmethod
    def BundleAdjustment(prob, nIterations=5, stop=None, bRobust=True):
        """Optimizer::BundleAdjustment over keyframes and map points (reference src/Optimizer.cc:55-323): one
        optimize(nIterations) call, Huber kernels only when bRobust, nothing is erased.  prob as for LocalBundleAdjustment
        (fixed[i] = 1 for the keyframe with mnId == 0); optional planes (m,4) f32 + pedge_plane / pedge_cam (Ep,) i32 + pedge_obs
        (Ep,4) f32: the MapPlane vertices and EdgePlane edges of :203-252 (eao_bundle_adjustment_planes)."""
        return Optimizer.LocalBundleAdjustment(prob, stop, (int(nIterations), 0), gba=bool(bRobust))
