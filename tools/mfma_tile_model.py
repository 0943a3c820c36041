"""Pair-slot accounting of the density sweep's candidate filter on the matrix pipe (VERDICT r04 "next" #3), CPU model, no GPU.

Today (VALU): one target per lane walks its own nine column runs (the 3 z-cells around its cell), 8 candidates per trip; a
wave runs as many trips per run as its busiest lane needs (lock-step).  Cost per wave-level candidate slot: 1 ds_read_b128 +
sub + 3 FMA + v_alignbit = 6.9 issue units (profiles/r04m_valu_issue_rates_and_packed_filter.txt).

Matrix pipe (v_mfma_f32_16x16x4_f32, exact f32, masks bit-identical: profiles/archive/r03g_ubench_mfma_filter.txt): one
instruction tests 16 candidates (rows) against 16 targets (columns) = 256 pair slots = 4 wave-level candidate slots.  All 16
targets of an instruction share its candidate rows, so a tile is 16 CONSECUTIVE targets of the wave (lane = target, as the
emission loop needs: four tiles per wave, the 4-bit row pieces of a lane quarter reach their target's lane through a 4 x 4
v_permlane16/32_swap transpose of four mask registers).  What a tile has to test per run:
  * its column's whole staged segment (all shell cells of the column, BZ + 2 of them) -- "segment", or the union of its
    targets' z windows -- "window"; both padded to multiples of 16 rows;
  * a tile whose 16 targets straddle two (x, y) columns needs both columns' runs (a second set of instructions for it).
Per instruction the VALU pays the accumulator -> hit-mask conversion: 4 accumulator registers x (v_cmp_lt_f32 + v_addc_co_u32)
= 8 units (C = 0; with four live C = -thr tuples, 16 VGPRs, it is 4 v_alignbit = 6.4), + 1 ds_read_b32 for the A operand.
The matrix pipe pays 8 passes = 32 cycles per instruction.

The script settles a box with the CPU oracle (test infrastructure), cuts it into the kernel's bricks (4 x 2 columns, as many
z layers <= 4 as keep <= 256 targets), takes waves of 64 consecutive targets and prints, per wave: today's lock-step slots
and issue units, the tiles' instructions, conversion units and matrix-pipe cycles.
Usage: python tools/mfma_tile_model.py [--n 28] [--steps 1500]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from emission_model import settle  # noqa: E402

VALU_PER_SLOT = 6.9         # issue units per wave-level candidate slot of today's filter (sub 1.0 + 3 x fma 1.44 + alignbit 1.6)
CONV_PER_MFMA = 8.0         # 4 x (v_cmp 1.0 + v_addc 1.0)
CONV_PER_MFMA_C = 6.4       # 4 x v_alignbit 1.6 (needs the four -thr accumulator tuples live: 16 VGPRs)
MFMA_CYCLES = 32            # v_mfma_f32_16x16x4_f32: 8 passes


def analyse(x, h, label, BX=4, BY=2, BZ=4, tmax=256):
    cell = np.floor(x / h).astype(np.int64)
    cell -= cell.min(0)
    nx, ny, nz = (int(v) for v in cell.max(0) + 1)
    key = (cell[:, 0] * ny + cell[:, 1]) * nz + cell[:, 2]
    order = np.argsort(key, kind="stable")
    cell, key = cell[order], key[order]
    cnt = np.bincount(key, minlength=nx * ny * nz).reshape(nx, ny, nz)
    res = dict(waves=0, slots=0, mf_seg=0, mf_win=0, mf_seg_nostr=0, tiles=0, straddle=0, targets=0)
    for bx in range(1, (nx - 2) // BX):
        for by in range(1, (ny - 2) // BY):
            z = 1
            while z < nz - 1:
                # the builder's greedy cut: as many layers (<= BZ) as keep the brick's targets within one round
                e = 1
                tot = int(cnt[bx * BX:bx * BX + BX, by * BY:by * BY + BY, z].sum())
                while e < BZ and z + e < nz - 1:
                    add = int(cnt[bx * BX:bx * BX + BX, by * BY:by * BY + BY, z + e].sum())
                    if tot + add > tmax:
                        break
                    tot += add
                    e += 1
                z0, z1 = z, z + e
                z = z1
                if tot == 0:
                    continue
                # targets in brick order: column (x, y), z cell, member
                tcol, tz = [], []
                for ix in range(bx * BX, bx * BX + BX):
                    for iy in range(by * BY, by * BY + BY):
                        for iz in range(z0, z1):
                            n = int(cnt[ix, iy, iz])
                            tcol += [(ix, iy)] * n
                            tz += [iz] * n
                T = len(tcol)
                for w in range(0, T - 63, 64):            # full waves only
                    res["waves"] += 1
                    res["targets"] += 64
                    lanes = range(w, w + 64)
                    # today: per run, the busiest lane's 3-cell window rounded up to trips of 8
                    for dx in (-1, 0, 1):
                        for dy in (-1, 0, 1):
                            m = 0
                            for t in lanes:
                                ix, iy = tcol[t]
                                L = int(cnt[ix + dx, iy + dy, tz[t] - 1:tz[t] + 2].sum())
                                m = max(m, (L + 7) // 8 * 8)
                            res["slots"] += m
                    # tiles of 16 consecutive targets
                    for g in range(4):
                        tl = list(range(w + 16 * g, w + 16 * g + 16))
                        cols = []
                        for t in tl:
                            if tcol[t] not in cols:
                                cols.append(tcol[t])
                        res["tiles"] += 1
                        res["straddle"] += len(cols) > 1
                        for k, (ix, iy) in enumerate(cols):
                            zs = [tz[t] for t in tl if tcol[t] == (ix, iy)]
                            zlo, zhi = max(min(zs) - 1, 0), min(max(zs) + 1, nz - 1)
                            for dx in (-1, 0, 1):
                                for dy in (-1, 0, 1):
                                    seg = int(cnt[ix + dx, iy + dy, max(z0 - 1, 0):min(z1 + 1, nz)].sum())
                                    win = int(cnt[ix + dx, iy + dy, zlo:zhi + 1].sum())
                                    res["mf_seg"] += (seg + 15) // 16
                                    res["mf_win"] += (win + 15) // 16
                                    if k == 0:
                                        res["mf_seg_nostr"] += (seg + 15) // 16
    w = max(res["waves"], 1)
    occ = cnt[cnt > 0].mean()
    print(f"{label}: {len(x)} particles, {occ:.2f} per non-empty cell, {w} full interior waves, "
          f"{100.0 * res['straddle'] / max(res['tiles'], 1):.0f} % of the 16-target tiles straddle two columns")
    today = res["slots"] / w
    print(f"   today      : {today:7.0f} lock-step candidate slots per wave  -> {today * VALU_PER_SLOT:7.0f} VALU issue units")
    for name, key in (("segment", "mf_seg"), ("window ", "mf_win"), ("segment, no straddle cost (bound)", "mf_seg_nostr")):
        n = res[key] / w
        print(f"   MFMA {name}: {n:6.1f} instructions per wave = {n * 4:6.0f} slot-equivalents ({n * 4 / today:4.2f} x today's) -> "
              f"conversion {n * CONV_PER_MFMA:6.0f} units (cmp+addc) / {n * CONV_PER_MFMA_C:6.0f} (alignbit, 16 VGPRs more) + {n:4.0f} ds_read_b32; "
              f"matrix pipe {n * MFMA_CYCLES:6.0f} cycles")
    return res


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
