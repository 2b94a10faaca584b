"""A/B of the pose kernel's launch geometry: four waves (default) vs eight (EAO_POSE_WAVES=8), per-wave cycle stamps.
   python tools/dbg_pose_waves.py            (spawns itself once per geometry and size)"""
import os, sys, subprocess, time
sys.path.insert(0, '.')
if len(sys.argv) > 1:
    n = int(sys.argv[1])
    import eao_fusion_amd as E
    from eao_fusion_amd import synth
    p = synth.synth_pose(n=n)
    os.environ.pop("EAO_DEBUG_STAMPS", None)
    for i in range(3): r = E.Optimizer.PoseOptimization(p)
    ts = []
    for i in range(20):
        r = E.Optimizer.PoseOptimization(p); ts.append(r['timing']['device_ms'] if 'timing' in r and 'device_ms' in r['timing'] else 0)
    print("n=%d waves=%s device ms min %.4f med %.4f  inliers %d" % (n, os.environ.get("EAO_POSE_WAVES", "4"), min(ts), sorted(ts)[10], r['n_inliers']), flush=True)
    sys.exit(0)
for n in (300, 700, 1000):
    for w in ("4", "8"):
        env = dict(os.environ, EAO_POSE_WAVES=w)
        subprocess.run([sys.executable, __file__, str(n)], env=env)
        env["EAO_DEBUG_STAMPS"] = "1"
        subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0,'.'); import eao_fusion_amd as E; from eao_fusion_amd import synth; E.Optimizer.PoseOptimization(synth.synth_pose(n=%d))" % n], env=env)
