# per-phase cycles of k_quadtree (EAO_DEBUG_STAMPS=1): one 64-frame batch through the host API, stamps of frame 0 on stderr
import sys, os; sys.path.insert(0, '.')
os.environ["EAO_DEBUG_STAMPS"] = "1"
import eao_fusion_amd as E
from eao_fusion_amd import synth
imgs = synth.synth_frames(64)
ext = E.ORBextractor(1000, 1.2, 8, 20, 7)
ext.set_profiling(True)
for _ in range(2): ext.extract_batch(imgs)
print(ext.last_timing())
