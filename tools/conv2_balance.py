"""How evenly does stage 1's static deal of tile pairs (wave w owns xp = (w - 2 yi) mod 4 for every row pair yi) spread the
conv2 MFMA rows of a patch over the four wavefronts?  CPU estimate on the golden frame's 3072 patches (tests/golden/frame_q0.npz):
executed (m-tile, tap-row) blocks per pair from the same occupancy rule the kernel uses, then max-over-wavefronts / average.

    python tools/conv2_balance.py

Printed for DESIGN.md 4.1: round 1's static deal reaches 78 % (58 / 74 / 86 % at scales 0 / 1 / 2), the XOR Latin square the kernel
uses now 83 % (the best of all 24^3 one-pair-per-row-pair deals on the two golden frames), greedy longest-first taking 92 %,
in-order taking 86 % -- and the measured kernel with in-patch taking (LDS claim per pair, runtime-addressed pair body) was SLOWER
(358 -> 405 us per 8-frame launch): the claim round trip and the lost compile-time addressing cost more than the balance gains."""
import os
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = np.load(os.path.join(REPO, "tests", "golden", "frame_q0.npz"))
pb = g["patch_bits"]
vox = np.unpackbits(pb.reshape(-1, 64).view(np.uint8), axis=1, bitorder="little").reshape(-1, 16, 16, 16).astype(bool)
n = vox.shape[0]
pad = np.zeros((n, 18, 18, 18), bool)
pad[:, 1:17, 1:17, 1:17] = vox
cell = np.zeros((n, 8, 8, 8), bool)          # pooled cells whose 4^3 receptive field holds a set voxel
for a in range(4):
    for b in range(4):
        for c in range(4):
            cell |= pad[:, a:a + 16:2, b:b + 16:2, c:c + 16:2]
nz = np.zeros((n, 10, 10), bool)             # padded (x, y) rows with a non-background cell
nz[:, 1:9, 1:9] = cell.any(axis=3)
rows = np.zeros((n, 4, 4), int)              # executed tap rows (6 MFMAs each) of pair (xp, yi)
for xp in range(4):
    for yi in range(4):
        for xt in range(2):
            for ka in range(3):
                for kb in range(3):
                    rows[:, xp, yi] += nz[:, 2 * xp + xt + ka, 2 * yi + kb] | nz[:, 2 * xp + xt + ka, 2 * yi + kb + 1]
tot = rows.sum(axis=(1, 2))
print("cells / patch by scale:", [round(float(cell[s::3].sum(axis=(1, 2, 3)).mean()), 1) for s in range(3)])
print("executed share of the dense conv2: %.3f (%.0f MFMAs per patch)" % (tot.mean() / (16 * 18), tot.mean() * 6))
for name, deal in (("round 1's deal xp = (w - 2 yi) mod 4", lambda w, yi: (w - 2 * yi) & 3),
                   ("XOR Latin square xp = w ^ (0, 3, 1, 2)[yi] (the kernel's)", lambda w, yi: w ^ (0, 3, 1, 2)[yi])):
    per_wave = np.zeros((n, 4), int)
    for w in range(4):
        for yi in range(4):
            per_wave[:, w] += rows[:, deal(w, yi), yi]
    print("%s: average / busiest wavefront = %.3f" % (name, per_wave.sum() / 4 / per_wave.max(axis=1).sum()),
          [round(float(per_wave[s::3].sum() / 4 / per_wave[s::3].max(axis=1).sum()), 3) for s in range(3)])


def taking(r, order):
    loads = [0, 0, 0, 0]
    for v in order(r.flatten()):
        if v:
            loads[loads.index(min(loads))] += v
    return max(loads)


print("greedy longest-first taking: %.3f" % (tot.sum() / 4 / sum(taking(rows[i], lambda v: sorted(v, reverse=True)) for i in range(n))))
print("in-order taking:             %.3f" % (tot.sum() / 4 / sum(taking(rows[i], lambda v: v) for i in range(n))))
