"""CPU model (no GPU) of VERDICT r03 next #5: "order each cell's staged records by z so that a lane's hits in a run are one
interval and its filter can stop at the interval's end".

A run = the three z-cells (cz-1, cz, cz+1) of one of the nine (x, y) columns around the target's cell.  With the records of
every cell ordered by z, the candidates a target can reach in z form ONE interval of the run: z_j in (z_i - h, z_i + h).
Its HITS do not: inside that interval a candidate still misses when it is too far in x / y (the interval is a slab of the
column, the neighbourhood a ball), so the hit bits stay scattered and the emission loop -- whose trip count is the wave's
largest per-run hit count -- is untouched.  What the order can save is FILTER slots.  This script settles a box with the CPU
oracle (test infrastructure) and counts, per wave of 64 consecutive targets and in lock-step (a trip = 8 candidates, the
wave runs as many trips per run as its busiest lane needs):
  today      : every lane tests its whole run
  z-exact    : every lane tests exactly the records with |z_j - z_i| < h (cells fully sorted by z: the bound)
  z-bins(B)  : cells ordered into B z-bins each; a lane tests the bins its window touches
and the hits per lane, the emission trips (unchanged by construction) for reference.
Usage: python tools/zwindow_model.py [--n 28] [--steps 1500]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from emission_model import settle  # noqa: E402


def analyse(x, h, label, bins=(2, 3, 4, 8), BX=4, BY=2, BZ=4):
    cell = np.floor(x / h).astype(np.int64)
    cell -= cell.min(0)
    nx, ny, nz = cell.max(0) + 1
    key = (cell[:, 0] * ny + cell[:, 1]) * nz + cell[:, 2]
    # order by cell, then by z inside the cell (the proposed LDS order; the HBM order would stay the reference's)
    order = np.lexsort((x[:, 2], key))
    x, cell, key = x[order], cell[order], key[order]
    G = nx * ny * nz
    cnt = np.bincount(key, minlength=G)
    end = np.cumsum(cnt)
    beg = end - cnt
    N = len(x)
    z = x[:, 2]
    full = np.zeros((N, 9), np.int64)
    zex = np.zeros((N, 9), np.int64)
    zb = {B: np.zeros((N, 9), np.int64) for B in bins}
    hits = np.zeros((N, 9), np.int64)
    for r, (dx, dy) in enumerate((a, b) for a in (-1, 0, 1) for b in (-1, 0, 1)):
        cx, cy = cell[:, 0] + dx, cell[:, 1] + dy
        ok = (cx >= 0) & (cx < nx) & (cy >= 0) & (cy < ny)
        for dz in (-1, 0, 1):
            cz = cell[:, 2] + dz
            okz = ok & (cz >= 0) & (cz < nz)
            k = (np.where(okz, cx, 0) * ny + np.where(okz, cy, 0)) * nz + np.where(okz, cz, 0)
            b, e = beg[k], end[k]
            L = np.where(okz, e - b, 0)
            full[:, r] += L
            for t in range(int(L.max())):
                m = t < L
                j = np.where(m, b + t, 0)
                dzz = np.abs(z - z[j])
                inwin = m & (dzz < h)
                zex[:, r] += inwin
                dd = x - x[j]
                hits[:, r] += m & ((dd * dd).sum(1) < h * h)
                for B in bins:
                    # candidate's bin inside its cell and the bins the target's window touches in that cell
                    zc = np.where(okz, cz, 0) * h
                    bj = np.minimum(((z[j] - zc) / (h / B)).astype(np.int64), B - 1)
                    lo = np.clip(np.floor((z - h - zc) / (h / B)), 0, B - 1).astype(np.int64)
                    hi = np.clip(np.floor((z + h - zc) / (h / B)), 0, B - 1).astype(np.int64)
                    zb[B][:, r] += m & (bj >= lo) & (bj <= hi) & (z + h > zc) & (z - h < zc + h)
    # waves of 64 consecutive targets of interior bricks (reference order inside the brick is immaterial for the counts)
    res = {"waves": 0, "full": 0, "zex": 0, "hits": 0.0, "emit": 0, "own_full": 0.0, "own_zex": 0.0}
    res.update({f"zb{B}": 0 for B in bins})
    for bx in range(1, (nx - 2) // BX):
        for by in range(1, (ny - 2) // BY):
            for bz in range(1, (nz - 2) // BZ):
                t = []
                for ix in range(bx * BX, bx * BX + BX):
                    for iy in range(by * BY, by * BY + BY):
                        k0 = (ix * ny + iy) * nz + bz * BZ
                        t.extend(range(beg[k0], end[k0 + BZ - 1]))
                t = np.array(t, np.int64)
                for w in range(0, len(t) - 63, 64):
                    ids = t[w:w + 64]
                    r8 = lambda a: ((a + 7) // 8) * 8
                    res["full"] += int(r8(full[ids]).max(0).sum())
                    res["zex"] += int(r8(zex[ids]).max(0).sum())
                    for B in bins:
                        res[f"zb{B}"] += int(r8(zb[B][ids]).max(0).sum())
                    res["own_full"] += float(full[ids].sum(1).mean())
                    res["own_zex"] += float(zex[ids].sum(1).mean())
                    res["hits"] += float(hits[ids].sum(1).mean())
                    res["emit"] += int(hits[ids].max(0).sum())
                    res["waves"] += 1
    w = max(res["waves"], 1)
    print(f"{label}: {N} particles, {cnt[cnt > 0].mean():.2f} per cell, {w} interior waves")
    print(f"   a lane's own candidates: whole runs {res['own_full'] / w:.0f}, inside its z window {res['own_zex'] / w:.0f}; hits {res['hits'] / w:.1f}; emission trips per wave (unchanged by the order) {res['emit'] / w:.1f}")
    line = f"   lock-step filter slots per wave (8 per trip): today {res['full'] / w:.0f}, z-exact {res['zex'] / w:.0f} ({100 * res['zex'] / res['full']:.0f} %)"
    for B in bins:
        line += f", {B} bins per cell {res[f'zb{B}'] / w:.0f} ({100 * res[f'zb{B}'] / res['full']:.0f} %)"
    print(line)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=28)
    ap.add_argument("--steps", type=int, default=1500)
    a = ap.parse_args()
    x0, x1, geom = settle(a.n, a.steps, os.cpu_count())
    h = 4 * geom.particle_radius
    analyse(x0.astype(np.float64), h, "rest lattice")
    if a.steps:
        analyse(x1.astype(np.float64), h, f"after {a.steps} steps (settled)")


if __name__ == "__main__":
    main()
