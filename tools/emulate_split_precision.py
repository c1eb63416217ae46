"""How much precision do split low-precision products cost the descriptors?  A float64 CPU emulation of the encoder (PyTorch conv3d /
matmul in f64) in which chosen layers take their operands as sums of f16 (or bf16) terms and only some of the partial products are
kept -- the arithmetic of k_enc_stage1x / k_enc_conv3 / k_enc_dense1p -- compared with an exact f64 evaluation and with the f32 CPU
oracle, on every 8th key point of a synthetic frame.  Test infrastructure (it drives the oracle); nothing in the product imports it.
Findings it was written for (round 3): the f32 oracle is itself 1.3e-6 from the f64 network; f16 x 2 with three partial products
on all four layers adds 6e-7; flushing f16 subnormals would cost 1e-4 (the hardware honours them: tools/micro/f16_mfma_subnormal.hip).
    python tools/emulate_split_precision.py        (CPU, ~1 min)"""
import os, sys, numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO+'/cae-lo_amd'); sys.path.insert(0, REPO+'/oracle')
import oracle as orc
from caelo import synth
import torch.nn.functional as F
resp_m, enc_m = orc.load_models(REPO+'/weights/SphericalRingPCRespondLayer.h5', REPO+'/weights/EncoderModel4VoxelPatch.h5')
pc = synth.make_scan(0, quantum=1e-3)
ring, cnt = orc.ProjectPC2SphericalRing(pc)
resp = resp_m.predict(ring[None, 0:64, 0:1792, 0:3])[0]
kp, kpix, _ = orc.GetKeyPtsByAE(ring, cnt, resp)
vox = orc.Voxelization(pc[:, 0:3])
sel = np.arange(0, len(kp), 8)
bits = [orc.patches_bits(kp[sel], vox[6+s], s)[0] for s in range(3)]
bits = np.concatenate(bits, 0)
ref = enc_m.predict_bits(bits)            # f32 oracle [N,20]
x = torch.from_numpy(orc.unpack_patches(bits)).double() if hasattr(orc,'unpack_patches') else None
print(bits.shape, ref.shape, [w.shape for w in enc_m.w])
W = [torch.from_numpy(np.asarray(w)).double() for w in enc_m.w]
if x is None:
    b = np.ascontiguousarray(bits, dtype='<u8').view(np.uint8).reshape(bits.shape[0], 512)
    # bit ((iy&3)*16+iz) of word (ix*4+iy/4)
    u = np.unpackbits(b, axis=1, bitorder='little').reshape(bits.shape[0], 16, 4, 4, 16)  # ix, iy/4, iy&3, iz
    x = torch.from_numpy(u.reshape(bits.shape[0],16,16,16).astype(np.float64))
x = x.reshape(-1,1,16,16,16)
def split16(t, n):
    parts=[]; r=t.clone()
    for i in range(n):
        h = r.float().half().double()   # f16 RNE (via f32: r is exactly f32-representable? approx)
        parts.append(h); r = r-h
    return parts
def splitbf(t, n):
    parts=[]; r=t.clone()
    for i in range(n):
        h = r.float().bfloat16().double(); parts.append(h); r=r-h
    return parts
def conv(xp, wp, b, terms):
    # wp: list of split weights [27*Cin*Cout] keras layout [3,3,3,Cin,Cout]
    acc=None
    for (i,j) in terms:
        w = wp[j].permute(4,3,0,1,2)
        y = F.conv3d(xp[i], w, None, padding=1)
        acc = y if acc is None else acc+y
    return acc + b.view(1,-1,1,1,1)
def run(mode):
    # mode: dict layer-> (split fn, nx, nw, terms) or None for exact
    def lay(xin, w, b, key):
        m = mode.get(key)
        wk = w
        if m is None:
            return conv([xin],[wk],b,[(0,0)])
        f,nx,nw,terms = m
        xin32 = xin.float().double()  # activations are f32 in the kernel
        return conv(f(xin32,nx), f(wk.float().double(),nw), b, terms)
    h = torch.tanh(lay(x, W[0], W[1], 'c1'))
    h = F.max_pool3d(h, 2)
    bg = torch.tanh(W[1]).view(1,-1,1,1,1)
    if mode.get('c2delta'):
        pass
    h = torch.tanh(lay(h, W[2], W[3], 'c2')); h = F.max_pool3d(h,2)
    h = torch.tanh(lay(h, W[4], W[5], 'c3'))
    flat = h.permute(0,2,3,4,1).reshape(h.shape[0], -1)
    m = mode.get('d1')
    if m is None:
        z = flat @ W[6]
    else:
        f,nx,nw,terms = m
        xs = f(flat.float().double(), nx); ws = f(W[6], nw)
        z = sum(xs[i] @ ws[j] for i,j in terms)
    z = torch.tanh(z + W[7])
    return torch.tanh(z @ W[8] + W[9]).numpy()
exact = run({})
def rep(name, out):
    e64 = np.abs(out-exact).max(); eo = (np.abs(out-ref)/np.maximum(np.abs(ref),0.1)).max()
    print('%-40s max abs vs f64-exact %.3e   rel vs f32 oracle (floor .1) %.3e' % (name, e64, eo))
rep('f64 exact', exact)
print('f32 oracle vs f64 exact: max abs %.3e' % np.abs(ref-exact).max())
T3=[(0,0),(0,1),(1,0)]; T4=T3+[(1,1)]
T6=[(0,0),(0,1),(1,0),(1,1),(0,2),(2,0)]
rep('conv2 f16x2 (3 terms)', run({'c2':(split16,2,2,T3)}))
rep('conv2 f16x2 (4 terms)', run({'c2':(split16,2,2,T4)}))
rep('conv2 bf16x3 (6 terms)', run({'c2':(splitbf,3,3,T6)}))
rep('conv1 w f16x2', run({'c1':(split16,1,2,[(0,0),(0,1)])}))
rep('c1,c2,c3,d1 all f16x2 4 terms', run({'c1':(split16,1,2,[(0,0),(0,1)]),'c2':(split16,2,2,T4),'c3':(split16,2,2,T4),'d1':(split16,2,2,T4)}))
rep('c3,d1 bf16x3 (current)', run({'c3':(splitbf,3,3,T6),'d1':(splitbf,3,3,T6)}))
rep('c1 c2 f16x2, c3 d1 bf16x3', run({'c1':(split16,1,2,[(0,0),(0,1)]),'c2':(split16,2,2,T4),'c3':(splitbf,3,3,T6),'d1':(splitbf,3,3,T6)}))
rep('all f16x2 3 terms', run({'c1':(split16,1,2,[(0,0),(0,1)]),'c2':(split16,2,2,T3),'c3':(split16,2,2,T3),'d1':(split16,2,2,T3)}))
def split16f(t, n):
    parts=[]; r=t.clone()
    for i in range(n):
        h = r.float().half().double()
        h = torch.where(h.abs() < 2.0**-14, torch.zeros_like(h), h)   # flush f16 subnormals
        parts.append(h); r = r-h
    return parts
rep('FLUSH conv2 f16x2 3 terms', run({'c2':(split16f,2,2,T3)}))
rep('FLUSH all f16x2 3 terms', run({'c1':(split16f,1,2,[(0,0),(0,1)]),'c2':(split16f,2,2,T3),'c3':(split16f,2,2,T3),'d1':(split16f,2,2,T3)}))
# scaled variant: lo parts scaled by 2^11 to stay normal? emulate "lo computed then flushed if < 2^-14" is above.
for w in W[::2]: print(tuple(w.shape), float(w.abs().max()), float(w.abs().min()), float((w.abs()<2.0**-14).double().mean()))
