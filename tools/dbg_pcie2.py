"""PCIe copies on this box: direction, size, one or two streams (pinned host memory, torch)."""
import time, torch
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for mb in (0.25, 1, 4, 20):
    nby = int(mb * (1 << 20))
    h = torch.empty(nby, dtype=torch.uint8).pin_memory(); d = torch.empty(nby, dtype=torch.uint8, device="cuda")
    a = t(lambda: d.copy_(h, non_blocking=True)); b = t(lambda: h.copy_(d, non_blocking=True))
    print("%5.2f MiB: H2D %.3f ms (%.1f GB/s)   D2H %.3f ms (%.1f GB/s)" % (mb, a * 1e3, nby / a / 1e9, b * 1e3, nby / b / 1e9))
h1 = torch.empty(20 << 20, dtype=torch.uint8).pin_memory(); d1 = torch.empty(20 << 20, dtype=torch.uint8, device="cuda")
h2 = torch.empty(4 << 20, dtype=torch.uint8).pin_memory(); d2 = torch.empty(4 << 20, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def both():
    with torch.cuda.stream(s1): d1.copy_(h1, non_blocking=True)
    with torch.cuda.stream(s2): h2.copy_(d2, non_blocking=True)
def serial():
    d1.copy_(h1, non_blocking=True); h2.copy_(d2, non_blocking=True)
print("20 MiB H2D + 4 MiB D2H: two streams %.3f ms, one stream %.3f ms" % (t(both) * 1e3, t(serial) * 1e3))
# a device kernel writing into mapped pinned memory (zero-copy store): what a blit of our own would get
hm = torch.empty(4 << 20, dtype=torch.uint8).pin_memory()
