#!/usr/bin/env python
"""Per-workgroup time line of the two brick sweeps at FOUR (density) / FIVE (force) resident workgroups per CU -- not a
sweep timed alone with sections ablated ("measure first what the fixed part is made of").

Profiling build only (SPH_HIP_LIB_VARIANT=profile): every workgroup of k_gather_brick stamps the 100 MHz wall clock at
entry / after the step-A barrier / after the step-B barrier / before its finish / at exit and records the CU it ran on
(sph_gather.hip: SPH_TS).  From the rows of one launch this prints, per state (rest lattice, settled) and sweep:
  * the phases of a workgroup's residence (A: brick entry + column table, B: staging loads + tile in LDS, C: filter +
    emission / pair loop of lane 0's wave, F: finish + the wait for the slowest wave), mean and percentiles;
  * per CU: how many workgroups are resident on average, and which share of the CU's busy time has NO resident
    workgroup in its compute phase (all of them staging, finishing or draining) -- the part of the fixed cost that is
    exposed rather than hidden behind the neighbours' compute;
  * the launch's ramp: how long until the last CU has its first workgroup, and how long the tail is in which fewer than
    half of the CUs still hold work.
Usage: python tools/brick_timeline.py [--out gpurun_out/r04/brick_timeline.txt] [--settle 2000]"""
import argparse
import copy
import ctypes as C
import os
import sys

os.environ["SPH_HIP_LIB_VARIANT"] = "profile"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import bench  # noqa: E402
from sph_taichi_amd import ParticleSystem, SimConfig, _lib  # noqa: E402


def read_rows(ps, n_blocks):
    buf = np.zeros((n_blocks, 8), dtype=np.uint64)
    fn = ps._lib.sph_profile_read
    fn.restype = C.c_int32
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    rc = fn(ps._ctx, buf.ctypes.data_as(C.c_void_p), buf.nbytes)
    assert rc == 0, rc
    return buf


def union_length(iv):
    """total length of the union of intervals [(a, b)]"""
    if len(iv) == 0:
        return 0.0
    iv = iv[np.argsort(iv[:, 0])]
    tot, ca, cb = 0.0, iv[0, 0], iv[0, 1]
    for a, b in iv[1:]:
        if a > cb:
            tot += cb - ca
            ca, cb = a, b
        else:
            cb = max(cb, b)
    return tot + (cb - ca)


def analyse(rows, label, out):
    t = rows[:, :5].astype(np.float64) / 100.0          # microseconds
    done = (rows[:, 4] > 0) & (rows[:, 2] > 0)
    t = t[done]
    meta = rows[done]
    if len(t) == 0:
        out.append(f"{label}: no rows")
        return
    t0 = t[:, 0].min()
    t = t - t0
    A, B, Cc, F = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3]
    span = t[:, 4].max()
    T = (meta[:, 6] >> np.uint64(32)).astype(np.int64)
    shell = (meta[:, 6] & np.uint64(0xffffffff)).astype(np.int64)
    out.append(f"== {label}: {len(t)} workgroups with targets, launch span {span:.1f} us; targets/brick mean {T.mean():.0f}, shell records mean {shell.mean():.0f}")
    pct = lambda v: f"mean {v.mean():6.2f}  p10 {np.percentile(v, 10):6.2f}  p50 {np.percentile(v, 50):6.2f}  p90 {np.percentile(v, 90):6.2f}"
    out.append(f"   A  entry -> column table (brick entry, cell_end loads, scan, barrier) : {pct(A)} us")
    out.append(f"   B  staging (target + shell loads in flight, tile writes, barrier)     : {pct(B)} us")
    out.append(f"   C  compute of lane 0's wave (filter + emission | pair loop)           : {pct(Cc)} us")
    out.append(f"   F  finish (stores) + wait for the workgroup's slowest wave            : {pct(F)} us")
    res = t[:, 4] - t[:, 0]
    out.append(f"   residence                                                             : {pct(res)} us   (A+B = {100 * (A + B).sum() / res.sum():.1f} % of it, C = {100 * Cc.sum() / res.sum():.1f} %, F = {100 * F.sum() / res.sum():.1f} %)")
    heavy = T >= 160
    if heavy.any() and (~heavy).any():
        out.append(f"   heavy bricks (>= 160 targets, {heavy.sum()}): A+B {np.mean((A + B)[heavy]):.2f}, C {Cc[heavy].mean():.2f}, F {F[heavy].mean():.2f} us;  light ({(~heavy).sum()}): A+B {np.mean((A + B)[~heavy]):.2f}, C {Cc[~heavy].mean():.2f}, F {F[~heavy].mean():.2f} us")
    # per CU
    hw = meta[:, 5]
    cu_key = ((hw >> np.uint64(32)) & np.uint64(0xf)) * np.uint64(256) + ((hw >> np.uint64(8)) & np.uint64(0xff))
    keys = np.unique(cu_key)
    busy_tot = comp_tot = res_tot = 0.0
    first = []
    for k in keys:
        m = cu_key == k
        tt = t[m]
        busy = union_length(np.stack([tt[:, 0], tt[:, 4]], 1))
        comp = union_length(np.stack([tt[:, 2], tt[:, 3]], 1))
        busy_tot += busy
        comp_tot += comp
        res_tot += (tt[:, 4] - tt[:, 0]).sum()
        first.append(tt[:, 0].min())
    out.append(f"   {len(keys)} CUs seen; resident workgroups per busy CU: {res_tot / busy_tot:.2f}; share of CU busy time with NO workgroup computing: {100 * (1 - comp_tot / busy_tot):.1f} %  ({(busy_tot - comp_tot) / len(keys):.1f} us per CU of {busy_tot / len(keys):.1f} us busy, launch span {span:.1f} us)")
    first = np.array(first)
    ends = np.sort(t[:, 4])
    # tail: from the moment fewer than half of the CUs hold a workgroup until the end
    ev = np.concatenate([np.stack([t[:, 0], np.ones(len(t))], 1), np.stack([t[:, 4], -np.ones(len(t))], 1)])
    ev = ev[np.argsort(ev[:, 0])]
    live = np.cumsum(ev[:, 1])
    half = len(keys) * 2            # fewer than 2 workgroups per CU on average
    idx = np.where(live >= half)[0]
    t_tail = ev[idx[-1], 0] if len(idx) else 0.0
    out.append(f"   ramp: last CU gets its first workgroup at {first.max():.1f} us; drain: from {t_tail:.1f} us on fewer than 2 workgroups per CU are resident ({span - t_tail:.1f} us tail = {100 * (span - t_tail) / span:.1f} % of the span)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/brick_timeline.txt")
    ap.add_argument("--settle", type=int, default=2000)
    ap.add_argument("--workload", default="c3p_uniform_1.75M")
    a = ap.parse_args()
    assert _lib.profiling_variant()
    sd = bench.scene_dict(a.workload)
    ps = ParticleSystem(SimConfig(config=copy.deepcopy(sd)))
    solver = ps.build_solver()
    solver.initialize()
    gx, gy, gz = (int(v) for v in ps.grid_num)
    n_blocks = ((gx + 3) // 4) * ((gy + 1) // 2) * gz
    n_blocks = (n_blocks + 7) // 8 * 8 + 8
    out = [f"tools/brick_timeline.py: {a.workload}, {ps.particle_max_num} particles; per-workgroup wall-clock stamps (10 ns) of one launch each"]
    for state, pre in (("rest lattice (steps 25..)", 25), (f"settled ({a.settle} steps on)", a.settle)):
        solver.step(pre)
        for bit, sweep in ((1 << 28, "density + EOS sweep (4 workgroups / CU)"), (1 << 29, "force sweep (5 workgroups / CU)")):
            ps.set_option(_lib.OPT_DEBUG_ABLATE, bit)
            solver.step(1)
            ps.sync()
            rows = read_rows(ps, n_blocks)
            ps.set_option(_lib.OPT_DEBUG_ABLATE, 0)
            analyse(rows, f"{state}: {sweep}", out)
    ps.close()
    txt = "\n".join(out)
    print(txt)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    open(a.out, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
