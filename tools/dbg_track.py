"""Latency of eao_tracker_track_local_map behind device-resident extractor outputs (the configuration of bench.py's
extra.tracking_frame_device_ms): min / median of 100 calls."""
import sys, time; sys.path.insert(0, ".")
import numpy as np, torch
import eao_fusion_amd as E
from eao_fusion_amd import synth
from eao_fusion_amd.tracker import Tracker
dev = torch.device("cuda", 0)
curT, lastT, _ = synth.synth_tracking(n=1000, seed=7100, mono_frac=0.0, occupied_frac=0.0)
NT = len(curT["kp_x"])
kT = np.zeros(NT, E.orb.KP_DTYPE)
kT["x"], kT["y"] = np.clip(curT["kp_x"], 1, 638), np.clip(curT["kp_y"], 1, 478)
kT["angle"], kT["octave"] = curT["kp_angle"], curT["kp_octave"]
XwT = lastT["Xw"].astype(np.float64)
dT = np.linalg.norm(XwT, axis=1).astype(np.float32)
ptsT = dict(active=np.ones(len(XwT), np.uint8), Xw=lastT["Xw"], normal=(XwT / dT[:, None]).astype(np.float32), min_dist_inv=0.6 * dT,
            max_dist_inv=1.7 * dT, max_dist=(dT * np.float32(1.2) ** (lastT["octave"] - 0.5)).astype(np.float32), descriptors=lastT["descriptors"])
sfT = curT["scale_factors"]
trk = Tracker(curT["fx"], curT["fy"], curT["cx"], curT["cy"], curT["mbf"], (0.0, 640.0, 0.0, 480.0), sfT, (np.float32(1) / (sfT * sfT)).astype(np.float32),
              float(np.log(np.float32(1.2))), 2048, 2048)
trk.set_local_map(ptsT)
dk = torch.zeros((2048, 28), dtype=torch.uint8, device=dev); dk[:NT] = torch.from_numpy(kT.view(np.uint8).reshape(NT, 28)).to(dev)
dd = torch.zeros((2048, 32), dtype=torch.uint8, device=dev); dd[:NT] = torch.from_numpy(np.ascontiguousarray(curT["descriptors"])).to(dev)
dn = torch.tensor([NT], dtype=torch.int32, device=dev)
ddep = torch.full((480, 640), 3.0, dtype=torch.float32, device=dev)
torch.cuda.synchronize()
st = torch.cuda.current_stream().cuda_stream
ts = []
for i in range(105):
    t0 = time.perf_counter()
    r = trk.track_local_map(dk.data_ptr(), dd.data_ptr(), dn.data_ptr(), ddep.data_ptr(), 640, 640, 480, curT["Tcw"], None, 3.0, 0.8, st)
    if i >= 5: ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e3
print("eao_tracker_track_local_map: min %.4f median %.4f ms (%d keypoints, %d matches, %d inliers)" % (ts.min(), np.median(ts), r["n_keypoints"], r["n_matches"], r["n_inliers"]))
# the stage in front of it (round 4): TrackWithMotionModel's data path, and both stages back to back (what a tracked frame costs behind the extractor)
ts, tb = [], []
for i in range(105):
    t0 = time.perf_counter()
    m = trk.track_with_motion_model(dk.data_ptr(), dd.data_ptr(), dn.data_ptr(), ddep.data_ptr(), 640, 640, 480, curT["Tcw"], lastT, 15.0, False, True, True, st)
    t1 = time.perf_counter()
    r = trk.track_local_map(dk.data_ptr(), dd.data_ptr(), dn.data_ptr(), ddep.data_ptr(), 640, 640, 480, m["Tcw"], None, 3.0, 0.8, st)
    if i >= 5: ts.append(t1 - t0); tb.append(time.perf_counter() - t0)
ts, tb = np.array(ts) * 1e3, np.array(tb) * 1e3
print("eao_tracker_track_with_motion_model: min %.4f median %.4f ms (%d matches, %d kept); motion model + local map back to back: min %.4f median %.4f ms per frame"
      % (ts.min(), np.median(ts), m["n_matches"], m["n_inliers"], tb.min(), np.median(tb)))
