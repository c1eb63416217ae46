"""H2D bandwidth of a 16 MB pinned -> device copy, per stream and per pinned block, in THIS process (is the upload mode's occasional
4x slow run a property of the copy stream's DMA queue, of where the pinned pages sit, or of the process?)."""
import os, sys, time
import torch
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
blocks = [torch.empty((1 << 22,), dtype=torch.float32).pin_memory() for _ in range(3)]          # 16 MB each
dst = torch.empty((1 << 22,), dtype=torch.float32, device=dev)
streams = [torch.cuda.Stream(dev) for _ in range(6)]
def bw(s, src, reps=5):
    with torch.cuda.stream(s):
        dst.copy_(src, non_blocking=True); s.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            dst.copy_(src, non_blocking=True)
        s.synchronize()
    return reps * src.numel() * 4 / (time.perf_counter() - t) / 1e9
try:
    cpu = int(open("/proc/self/stat").read().rsplit(")", 1)[1].split()[36])
except Exception:
    cpu = -1
print("cpu of this thread:", cpu)
for bi, b in enumerate(blocks):
    print("pinned block %d: " % bi + "  ".join("%5.1f" % bw(s, b) for s in streams) + "  GB/s on streams 0..5")
