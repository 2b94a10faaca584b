"""Cost of a cross-stream event hand-over under PyTorch's HIP runtime: A on s1 -> event -> B on s2 -> event -> A ... against the
same kernels back to back on one stream."""
import time, torch
x = torch.ones(8 << 20, device="cuda"); y = torch.ones(8 << 20, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def one_stream(n):
    with torch.cuda.stream(s1):
        for _ in range(n):
            x.mul_(1.0001); y.mul_(1.0001)
def two_streams(n):
    e1 = torch.cuda.Event(); e2 = torch.cuda.Event()
    for _ in range(n):
        with torch.cuda.stream(s1):
            x.mul_(1.0001); e1.record(s1)
        s2.wait_event(e1)
        with torch.cuda.stream(s2):
            y.mul_(1.0001); e2.record(s2)
        s1.wait_event(e2)
def records_only(n):
    e1 = torch.cuda.Event()
    with torch.cuda.stream(s1):
        for _ in range(n):
            x.mul_(1.0001); e1.record(s1); y.mul_(1.0001); e1.record(s1)
for fn in (one_stream, two_streams, records_only):
    fn(50); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(500); torch.cuda.synchronize()
    print("%-14s %.2f us per kernel pair" % (fn.__name__, (time.perf_counter() - t0) / 500 * 1e6))
