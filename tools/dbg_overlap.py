"""Does a pinned H2D copy on one stream overlap a kernel on another on this runtime?  (torch only)"""
import time, torch
h = torch.empty(20 << 20, dtype=torch.uint8).pin_memory(); d = torch.empty(20 << 20, dtype=torch.uint8, device="cuda")
a = torch.randn(4096, 4096, device="cuda"); b = torch.randn(4096, 4096, device="cuda")
x = torch.zeros(64 << 20, dtype=torch.float32, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def k_mm():
    with torch.cuda.stream(s1): torch.mm(a, b)
def k_el():
    with torch.cuda.stream(s1):
        for _ in range(2): x.add_(1.0)
def cp():
    with torch.cuda.stream(s2): d.copy_(h, non_blocking=True)
print("copy alone %.3f  mm alone %.3f  elementwise alone %.3f" % (t(cp), t(k_mm), t(k_el)))
print("copy + mm on two streams %.3f   copy + elementwise %.3f" % (t(lambda: (k_mm(), cp())), t(lambda: (k_el(), cp()))))
