"""CPU model of the brick partition (sph_bricks.h: greedy z cut under tmax = 256 targets and smax = 1,792 shell records) on the
headline box's REST lattice, for the two grid origins of the SPH_OPT_BRICK_ORIGIN experiment (commit 359bfce, reverted):
how many bricks, how full.  No GPU.  Usage: python tools/brick_origin_model.py"""
import numpy as np

h = np.float32(0.04); d = np.float32(0.02)


def cells(n, start=0.04):
    x = (np.float32(start) + d * np.arange(n, dtype=np.float32)).astype(np.float32)
    return np.floor(x / h).astype(int)      # IEEE divide + trunc, as the hash does


nx, ny, nz = 125, 75, 50
ox_ = np.bincount(cells(246), minlength=nx); oy_ = np.bincount(cells(74), minlength=ny); oz_ = np.bincount(cells(96), minlength=nz)
occ = ox_[:, None, None] * oy_[None, :, None] * oz_[None, None, :]
print("lattice planes per cell, first cells of an axis:", ox_[:12], "(the f32 quotient puts a node that sits on a cell face on either side)")


def bricks(OX, OY, tmax=256, smax=1792, BX=4, BY=2, BZ=4):
    out = []
    for bx in range((nx + OX + BX - 1) // BX):
        for by in range((ny + OY + BY - 1) // BY):
            x0, y0 = bx * BX - OX, by * BY - OY
            T = occ[max(x0, 0):min(x0 + BX, nx), max(y0, 0):min(y0 + BY, ny), :].sum((0, 1))
            S = occ[max(x0 - 1, 0):min(x0 + BX + 1, nx), max(y0 - 1, 0):min(y0 + BY + 1, ny), :].sum((0, 1))
            Tp = np.concatenate([[0], np.cumsum(T)]); Sp = np.concatenate([[0], np.cumsum(S)])
            z = 0
            while z < nz:
                if T[z] == 0:
                    z += 1; continue
                e = 1; t0 = Tp[z]; s0 = Sp[max(z - 1, 0)]
                while z + e < nz and e < BZ and T[z + e] != 0 and Tp[z + e + 1] - t0 <= tmax and Sp[min(z + e + 2, nz)] - s0 <= smax:
                    e += 1
                out.append((Tp[z + e] - Tp[z], Sp[min(z + e + 1, nz)] - Sp[max(z - 1, 0)], e))
                z += e
    return np.array(out)


for OX, OY in ((0, 0), (3, 1)):
    b = bricks(OX, OY)
    print(f"origin offset ({OX}, {OY}): {len(b)} bricks, {(b[:, 0] >= 160).sum()} heavy, targets per brick mean {b[:, 0].mean():.1f}, "
          f"shell records mean {b[:, 1].mean():.1f}, heights 1..4: {np.bincount(b[:, 2], minlength=5)[1:]}, "
          f"targets < 128 / < 160 / < 192 / < 224 / <= 256: {np.histogram(b[:, 0], bins=[0, 128, 160, 192, 224, 257])[0]}")
