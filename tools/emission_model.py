"""Lock-step trip counts of the density sweep's emission loop under different orderings (CPU model, no GPU).

A wave = 64 consecutive targets of a brick; every target has 9 column runs.  The emission loop runs, per phase, as
many trips as the wave's busiest lane has hits in the run it handles in that phase.  This script settles a small
box with the CPU oracle (test infrastructure), then counts trips for
  natural   : every lane walks its runs in (dx, dy) order                                (r01 kernel)
  mirrored  : every lane walks its runs near side first (dx order reversed in the upper half of its cell, same for dy)
  sorted    : every lane walks its runs by descending own hit count (bound for any per-lane permutation)
  merged    : one loop over the lane's total (bound for any scheme)
Usage: python tools/emission_model.py [--n 28] [--steps 1500] [--fill 1.0]
"""
from __future__ import annotations

import argparse
import copy
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def settle(n, steps, threads):
    import scenes
    d = 0.02
    start = (0.04, 0.04, 0.04)
    sd = scenes.fluid_only(counts=(n, n, n), start=start, velocity=(0.6, -1.0, 0.3),
                           domain_end=(start[0] + n * d + 0.12, start[1] + n * d * 1.6, start[2] + n * d + 0.12))
    cfg, sc = scenes.build(copy.deepcopy(sd))
    o = scenes.make_oracle(cfg, sc, omp_threads=threads)
    o.initialize()
    x0 = o.by_pid("x").copy()
    if steps:
        o.step(steps)
    return x0, o.by_pid("x").copy(), sc.geom


def analyse(x, h, label, BX=4, BY=2, BZ=4, margin=1.0002, sort_cells=False):
    cell = np.floor(x / h).astype(np.int64)
    lo = cell.min(0)
    cell -= lo
    nx, ny, nz = cell.max(0) + 1
    key = (cell[:, 0] * ny + cell[:, 1]) * nz + cell[:, 2]
    order = np.argsort(key, kind="stable")
    x = x[order]; cell = cell[order]; key = key[order]
    G = nx * ny * nz
    cnt = np.bincount(key, minlength=G)
    end = np.cumsum(cnt)
    beg = end - cnt
    frac = x / h - np.floor(x / h)
    h2 = h * h * margin
    N = len(x)
    # per target: hits per run [9] in natural (dx, dy) order, run lengths [9]
    hits = np.zeros((N, 9), np.int32)
    hits_c = np.zeros((N, 9, 4), np.int32)     # hits per 32-candidate chunk of the run (the kernel's mask registers)
    rlen = np.zeros((N, 9), np.int32)
    for r, (dx, dy) in enumerate((a, b) for a in (-1, 0, 1) for b in (-1, 0, 1)):
        cx = cell[:, 0] + dx; cy = cell[:, 1] + dy
        ok = (cx >= 0) & (cx < nx) & (cy >= 0) & (cy < ny)
        zlo = np.maximum(cell[:, 2] - 1, 0); zhi = np.minimum(cell[:, 2] + 1, nz - 1)
        klo = (np.where(ok, cx, 0) * ny + np.where(ok, cy, 0)) * nz + zlo
        khi = (np.where(ok, cx, 0) * ny + np.where(ok, cy, 0)) * nz + zhi
        b = beg[klo]; e = end[khi]
        L = np.where(ok, e - b, 0)
        rlen[:, r] = L
        for k in range(int(L.max())):
            m = k < L
            j = np.where(m, b + k, 0)
            dd = x - x[j]
            hit = (m & ((dd * dd).sum(1) < h2)).astype(np.int32)
            hits[:, r] += hit
            hits_c[:, r, min(k // 32, 3)] += hit
    # interior targets only (full neighbourhoods), grouped into bricks then waves of 64 consecutive targets
    sx = frac[:, 0] >= 0.5; sy = frac[:, 1] >= 0.5
    perm_nat = np.arange(9)
    res = dict(natural=0, mirrored=0, sorted=0, merged=0, ideal=0.0, filt_nat=0, waves=0, lanes=0)
    bxs = range(1, (nx - 2) // BX); bys = range(1, (ny - 2) // BY); bzs = range(1, (nz - 2) // BZ)
    idx_of = {}
    for bx in bxs:
        for by in bys:
            for bz in bzs:
                t = []
                cells = []
                for ix in range(bx * BX, bx * BX + BX):
                    for iy in range(by * BY, by * BY + BY):
                        k0 = (ix * ny + iy) * nz + bz * BZ
                        t.extend(range(beg[k0], end[k0 + BZ - 1]))
                        cells.extend(range(k0, k0 + BZ))
                t = np.array(t, np.int64)
                if len(t) < 64:
                    continue
                if sort_cells == "octant":   # lanes assigned by the octant of the target inside its cell (then by cell)
                    oc = (frac[t, 0] >= 0.5) * 4 + (frac[t, 1] >= 0.5) * 2 + (frac[t, 2] >= 0.5) * 1
                    t = t[np.argsort(oc, kind="stable")]
                elif sort_cells == "octant27":   # finer: thirds of the cell in every axis
                    oc = np.minimum((frac[t, 0] * 3).astype(int), 2) * 9 + np.minimum((frac[t, 1] * 3).astype(int), 2) * 3 + np.minimum((frac[t, 2] * 3).astype(int), 2)
                    t = t[np.argsort(oc, kind="stable")]
                elif sort_cells:      # lanes assigned cell by cell in order of the cell's candidate count (longest runs first)
                    work = [int(rlen[beg[c]].sum()) if cnt[c] else 0 for c in cells]
                    order_c = sorted(range(len(cells)), key=lambda q: -work[q])
                    t = np.array([p_ for q in order_c for p_ in range(beg[cells[q]], end[cells[q]])], np.int64)
                for w in range(0, len(t) - 63, 64):
                    ids = t[w:w + 64]
                    H = hits[ids]
                    res["natural"] += int(H.max(0).sum())
                    res["chunked"] = res.get("chunked", 0) + int(hits_c[ids].max(0).sum())
                    # mirrored: phase (px, py) -> run (sx ? 2-px : px, sy ? 2-py : py)
                    M = np.empty_like(H)
                    for p in range(9):
                        px, py = divmod(p, 3)
                        rx = np.where(sx[ids], 2 - px, px); ry = np.where(sy[ids], 2 - py, py)
                        M[:, p] = H[np.arange(64), rx * 3 + ry]
                    res["mirrored"] += int(M.max(0).sum())
                    res["sorted"] += int((-np.sort(-H, axis=1)).max(0).sum())
                    # centre run first, then the four edge runs and the four corner runs each by descending count
                    gs = np.concatenate([H[:, [4]], -np.sort(-H[:, [1, 3, 5, 7]], axis=1), -np.sort(-H[:, [0, 2, 6, 8]], axis=1)], axis=1)
                    res["group_sorted"] = res.get("group_sorted", 0) + int(gs.max(0).sum())
                    Hc = hits_c[ids][:, :, 0]      # first 32-candidate chunk only (what the kernel keeps in registers)
                    over = int(hits_c[ids][:, :, 1:].max(0).sum())
                    res["k_mirror"] = res.get("k_mirror", 0) + over + int(np.stack([Hc[np.arange(64), np.where(sx[ids], 2 - p // 3, p // 3) * 3 + np.where(sy[ids], 2 - p % 3, p % 3)] for p in range(9)], 1).max(0).sum())
                    res["k_sorted"] = res.get("k_sorted", 0) + over + int((-np.sort(-Hc, axis=1)).max(0).sum())
                    gk = np.concatenate([Hc[:, [4]], -np.sort(-Hc[:, [1, 3, 5, 7]], axis=1), -np.sort(-Hc[:, [0, 2, 6, 8]], axis=1)], axis=1)
                    res["k_group"] = res.get("k_group", 0) + over + int(gk.max(0).sum())
                    res["merged"] += int(H.sum(1).max())
                    # antipodal pairs emitted jointly: (dx,dy) with (-dx,-dy), centre alone
                    res["paired"] = res.get("paired", 0) + int(sum((H[:, a] + H[:, 8 - a]).max() for a in range(4)) + H[:, 4].max())
                    S = -np.sort(-H, axis=1)
                    # balanced pairs after a full sort: largest alone, then (2nd + 9th), (3rd + 8th), (4th + 7th), (5th + 6th)
                    res["bal_pairs"] = res.get("bal_pairs", 0) + int(S[:, 0].max() + sum((S[:, 1 + a] + S[:, 8 - a]).max() for a in range(4)))
                    # three joint loops: (1st + 6th + 7th), (2nd + 5th + 8th), (3rd + 4th + 9th)
                    res["bal_triples"] = res.get("bal_triples", 0) + int((S[:, 0] + S[:, 5] + S[:, 6]).max() + (S[:, 1] + S[:, 4] + S[:, 7]).max() + (S[:, 2] + S[:, 3] + S[:, 8]).max())
                    res["ideal"] += float(H.sum(1).mean())
                    res["filt_nat"] += int((((rlen[ids] + 7) // 8) * 8).max(0).sum())
                    res["filt_sorted"] = res.get("filt_sorted", 0) + int((((-np.sort(-rlen[ids], axis=1) + 7) // 8) * 8).max(0).sum())
                    res["filt_own"] = res.get("filt_own", 0) + float((((rlen[ids] + 7) // 8) * 8).sum(1).mean())
                    res["waves"] += 1
    w = max(res["waves"], 1)
    print(f"{label}: {N} particles, mean cell occupancy {cnt[cnt > 0].mean():.2f} (max {cnt.max()}), {w} interior waves")
    print(f"   mean hits per lane {res['ideal'] / w:.1f};  emission trips per wave: per 32-candidate chunk (the kernel) {res.get('chunked', 0) / w:.1f}, per run {res['natural'] / w:.1f}, "
          f"mirrored {res['mirrored'] / w:.1f}, centre + sorted edges + sorted corners {res.get('group_sorted', 0) / w:.1f}, antipodal pairs {res.get('paired', 0) / w:.1f}, sorted {res['sorted'] / w:.1f}, merged {res['merged'] / w:.1f};  "
          f"filter candidates per wave (lock-step, 8 per trip) {res['filt_nat'] / w:.0f}, runs walked longest first {res.get('filt_sorted', 0) / w:.0f}, a lane's own {res.get('filt_own', 0) / w:.0f}")
    print(f"   joint loops: largest + four balanced pairs {res.get('bal_pairs', 0) / w:.1f}, three balanced triples {res.get('bal_triples', 0) / w:.1f}")
    print(f"   as built (first chunk in registers, further chunks at once): mirrored {res.get('k_mirror', 0) / w:.1f}, sorted {res.get('k_sorted', 0) / w:.1f}, centre + sorted edges + sorted corners {res.get('k_group', 0) / w:.1f}")
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=28)
    ap.add_argument("--steps", type=int, default=1500)
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    a = ap.parse_args()
    x0, x1, geom = settle(a.n, a.steps, a.threads)
    h = geom.support_radius if hasattr(geom, "support_radius") else 4 * geom.particle_radius
    analyse(x0.astype(np.float64), h, "rest lattice")
    analyse(x0.astype(np.float64), h, "rest lattice, lanes by cell workload", sort_cells=True)
    if a.steps:
        analyse(x1.astype(np.float64), h, f"after {a.steps} steps")
        analyse(x1.astype(np.float64), h, f"after {a.steps} steps, lanes by cell workload", sort_cells=True)
        analyse(x1.astype(np.float64), h, f"after {a.steps} steps, lanes by octant of the target in its cell", sort_cells="octant")
        analyse(x1.astype(np.float64), h, f"after {a.steps} steps, lanes by thirds of the cell (27 classes)", sort_cells="octant27")


if __name__ == "__main__":
    main()
