"""Pinned host -> HBM copy rate: by copy size on one stream, and 2 MB copies (one scan) dealt over 1 / 2 / 4 streams."""
import torch, time
x = torch.empty(64*1024*1024//4, dtype=torch.float32).pin_memory()
d = torch.empty_like(x, device="cuda")
for n in (2*1024*1024//4, 16*1024*1024//4, 64*1024*1024//4):
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(20): d[:n].copy_(x[:n], non_blocking=True)
    torch.cuda.synchronize(); dt=time.perf_counter()-t
    print("H2D pinned %5.1f MB: %.1f GB/s" % (n*4/1e6, 20*n*4/dt/1e9))
n = 2*1024*1024//4
for ns in (1, 2, 4):
    ss = [torch.cuda.Stream() for _ in range(ns)]
    torch.cuda.synchronize(); t=time.perf_counter()
    for i in range(160):
        with torch.cuda.stream(ss[i % ns]):
            o = (i % 32) * n
            d[o:o+n].copy_(x[o:o+n], non_blocking=True)
    torch.cuda.synchronize(); dt=time.perf_counter()-t
    print("160 copies of 2.1 MB over %d stream(s): %.1f GB/s" % (ns, 160*n*4/dt/1e9))
