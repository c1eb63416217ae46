"""Brute-force search of LDS layouts for the conv2 A operand of k_enc_stage1x (round 4): position pitch, unit swizzle and A row order
such that every ds_read_b128 of the three fragments touches each of the 64 banks once (MI355X_MICROARCH.md: a b128 read is served in four
groups of 16 lanes).  Prints the best (worst-case conflict degree, total, y pitch, row map, swizzle); the kernel uses (10, cur, y0)."""
import itertools
HW = [list(range(0,4))+list(range(12,16))+list(range(20,28)),
      list(range(4,12))+list(range(16,20))+list(range(28,32)),
      list(range(32,36))+list(range(44,48))+list(range(52,60)),
      list(range(36,44))+list(range(48,52))+list(range(60,64))]
# fragments: (tap per g, term per g)
FR = [([0,1,2,0],[0,0,0,0]), ([0,1,2,0],[1,1,1,1]), ([1,2,1,2],[0,0,1,1])]
def rowmaps():
    # n -> (yl, z): candidates
    out = {}
    out['cur'] = lambda n: ((1 if (n>>2) in (1,2) else 0), 4*((n>>2)>>1) + (n&3))
    out['lin'] = lambda n: (n>>3, n&7)
    out['alt'] = lambda n: (n&1, n>>1)
    out['q'] = lambda n: ((n>>2)&1, 4*(n>>3) + (n&3))
    return out
def swz():
    return {'none': lambda x,y,z: 0, 'z0': lambda x,y,z: z&1, 'z1': lambda x,y,z: (z>>1)&1, 'z2': lambda x,y,z: (z>>2)&1,
            'y0': lambda x,y,z: y&1, 'z0y0': lambda x,y,z: (z^y)&1, 'z1y0': lambda x,y,z: ((z>>1)^y)&1, 'z2y0': lambda x,y,z: ((z>>2)^y)&1,
            'z0z1': lambda x,y,z: (z^(z>>1))&1, 'z0z2': lambda x,y,z:(z^(z>>2))&1, 'z1z2': lambda x,y,z:((z>>1)^(z>>2))&1}
res = []
for PY in range(10, 18):
  for rmn, rm in rowmaps().items():
    for sn, sw in swz().items():
      worst = 0; tot = 0
      for ybase in range(0, 9):       # padded y of the slab's first row (yp = 2yi + kb, 0..8)
        for fr_t, fr_u in FR:
          for grp in HW:
            banks = {}
            for lane in grp:
              g, n = lane >> 4, lane & 15
              yl, z = rm(n)
              yp = ybase + yl; zp = z + fr_t[g]      # padded z = z + tap (tap 0..2 <-> z-1..z+1 with +1 pad)
              q = yp * PY + zp
              unit = fr_u[g] ^ sw(0, yp, zp)
              addr = 32 * q + 16 * unit
              b = (addr // 4) % 64
              for k in range(4):
                banks[(b + k) % 64] = banks.get((b + k) % 64, 0) + 1
            c = max(banks.values())
            worst = max(worst, c); tot += c
      res.append((worst, tot, PY, rmn, sn))
res.sort()
for r in res[:15]: print(r)
