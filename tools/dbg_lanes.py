# timing of the host-API batch path (includes H2D/D2H) and of pure device replays, without torch (ROCm runtime of /opt/rocm)
import sys, time, os; sys.path.insert(0,'.')
import numpy as np, ctypes as C
import eao_fusion_amd as E
from eao_fusion_amd import synth, _lib
L=_lib.load()
hip=C.CDLL("libamdhip64.so.7")
hip.hipMalloc.argtypes=[C.c_void_p,C.c_size_t]; hip.hipMemcpy.argtypes=[C.c_void_p,C.c_void_p,C.c_size_t,C.c_int]
imgs = synth.synth_frames(int(os.environ.get("EAO_DBG_BATCH", "64")))
ext = E.ORBextractor(1000,1.2,8,20,7)
B,H,W=imgs.shape; cap=ext.max_keypoints(W,H)
def dmalloc(n):
    p=C.c_void_p(); assert hip.hipMalloc(C.byref(p), n)==0; return p
d_img=dmalloc(imgs.nbytes); hip.hipMemcpy(d_img, imgs.ctypes.data, imgs.nbytes, 1)
d_k=dmalloc(B*cap*28); d_d=dmalloc(B*cap*32); d_n=dmalloc(B*4)
stream = None
if os.environ.get("EAO_DBG_STREAM") == "own":      # a non-blocking stream of the caller's instead of the null stream
    stream = C.c_void_p(); assert hip.hipStreamCreateWithFlags(C.byref(stream), 1) == 0
def step(): _lib.check(L.eao_orb_extract_batch_device(ext._h, d_img, W,H,W,W*H,B,d_k,d_d,cap,d_n,stream))
if os.environ.get('EAO_DBG_PROF'): ext.set_profiling(True)   # stages run one after the other
for _ in range(3): step()
hip.hipDeviceSynchronize()
t=time.perf_counter()
K=int(os.environ.get("EAO_DBG_STEPS", "30"))
for _ in range(K): step()
t_enq=time.perf_counter()-t
hip.hipDeviceSynchronize()
print('host enqueue ms/step %.4f' % (t_enq/K*1e3))
print("lanes", os.environ.get("EAO_ORB_LANES"), "graph", os.environ.get("EAO_ORB_GRAPH"), "ms/step %.4f" % ((time.perf_counter()-t)/K*1e3))
