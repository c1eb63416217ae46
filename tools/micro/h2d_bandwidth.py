"""Pinned host -> HBM copy rate at three sizes (the include-h2d leg of bench.py moves 1.05 MB per scan)."""
import torch, time
x = torch.empty(64*1024*1024//4, dtype=torch.float32).pin_memory()
d = torch.empty_like(x, device="cuda")
for n in (2*1024*1024//4, 16*1024*1024//4, 64*1024*1024//4):
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(20): d[:n].copy_(x[:n], non_blocking=True)
    torch.cuda.synchronize(); dt=time.perf_counter()-t
    print("H2D pinned %5.1f MB: %.1f GB/s" % (n*4/1e6, 20*n*4/dt/1e9))
