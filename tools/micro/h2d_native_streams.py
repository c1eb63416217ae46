"""Pinned host -> HBM through caelo_upload_many: 2.1 MB copies (one scan each) on 1 / 2 / 4 streams, and 8 at a time behind one call."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "cae-lo_amd"))
import numpy as np, torch
import caelo; caelo.configure_runtime()
from caelo import _ffi
lib = _ffi.load()
N = 2 * 1024 * 1024
host = [torch.empty(N // 4, dtype=torch.float32).pin_memory() for _ in range(32)]
dev = [torch.empty(N // 4, dtype=torch.float32, device="cuda") for _ in range(32)]
dst = np.array([t.data_ptr() for t in dev], dtype=np.uint64)
src = np.array([t.data_ptr() for t in host], dtype=np.uint64)
nb = np.full(32, N, dtype=np.uint64)
for ns in (1, 2, 4):
    ss = [torch.cuda.Stream() for _ in range(ns)]
    for rep in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        for it in range(20):
            for b in range(4):          # four "batches" of 8 scans
                for k in range(ns):     # the batch's scans dealt over the streams
                    idx = np.arange(b * 8 + k, b * 8 + 8, ns)
                    d_, s_, n_ = dst[idx].copy(), src[idx].copy(), nb[idx].copy()   # (kept alive across the call)
                    _ffi.check(lib.caelo_upload_many(d_.ctypes.data, s_.ctypes.data, n_.ctypes.data, len(idx), C.c_void_p(ss[k].cuda_stream)))
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("640 copies of 2.1 MB over %d stream(s): %.1f GB/s" % (ns, 640 * N / dt / 1e9))

# eight scans per call from consecutive slots of ONE pinned ring into consecutive slots of one device allocation (an experiment build of
# caelo_upload_many turned such a batch into one pitched / one span copy: 55 / 46 GB/s here, slower inside the pipeline -- DESIGN 4.18)
cap = 160000 * 16
hring = torch.empty(32 * cap // 4, dtype=torch.float32).pin_memory()
dring = torch.empty(8 * cap // 4, dtype=torch.float32, device="cuda")
src2 = np.array([hring.data_ptr() + i * cap for i in range(32)], dtype=np.uint64)
dst2 = np.array([dring.data_ptr() + i * cap for i in range(8)], dtype=np.uint64)
nb2 = np.full(8, N, dtype=np.uint64)
st = torch.cuda.Stream()
for rep in range(2):
    torch.cuda.synchronize(); t = time.perf_counter()
    for it in range(20):
        for b in range(4):
            s_ = src2[b * 8:b * 8 + 8].copy()
            _ffi.check(lib.caelo_upload_many(dst2.ctypes.data, s_.ctypes.data, nb2.ctypes.data, 8, C.c_void_p(st.cuda_stream)))
    torch.cuda.synchronize(); dt = time.perf_counter() - t
print("80 calls of 8 x 2.1 MB from a contiguous ring: %.1f GB/s" % (640 * N / dt / 1e9))
