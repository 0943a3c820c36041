#!/usr/bin/env python
"""VERDICT r05 "next" #1 (neighbour lists that survive several steps), costed by MEASUREMENT on the GPU before anything is
built.  On the headline box (C3', 1,747,584 particles) in its settled state -- and on the developed C1 dam break, the fast
flow -- this measures the three quantities a skin-list scheme stands on:

  displacement : max over particles of |x(t0 + k) - x(t0)| for k = 1..K steps, in units of h -- a list built at t0 with
                 skin s*h stays a superset while 2 * max displacement <= s*h, so this gives the rebuild interval per skin;
                 next to it the largest RELATIVE displacement of any neighbour pair would be the true criterion, but it is
                 not computable without the pair list: the per-particle bound is what a device-side guard can test;
  cell changes : how many particles change their cell per step (a list reused WITHOUT a skin is exact only while no
                 particle has changed cell: the 27-cell shell of the rebuild-time cell must contain every r < h partner);
  list sweeps  : the time of a LIST-READING sweep over the same state (DFSPH's factor sweep: same lists, same staging, a
                 pair term of the density sweep's size) against the filtering density sweep it would replace, and the
                 sort phase the reuse steps would skip -- the best case of a reuse step.

Usage: python tools/list_reuse_probe.py [--workload c3p_uniform_1.75M] [--settle 2000] [--horizon 40] > out.json
"""
import argparse
import copy
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3p_uniform_1.75M")
    ap.add_argument("--settle", type=int, default=2000)
    ap.add_argument("--horizon", type=int, default=40)
    args = ap.parse_args()
    import bench
    from sph_taichi_amd import ParticleSystem, SimConfig, _lib
    sd = bench.scene_dict(args.workload)
    ps = ParticleSystem(SimConfig(config=copy.deepcopy(sd)))
    solver = ps.build_solver()
    N = ps.particle_max_num
    h = float(ps.support_radius)
    dt = float(sd["Configuration"]["timeStepSize"])
    solver.initialize()
    solver.step(args.settle)
    ps.sync()

    def by_pid(name):
        a = getattr(ps, name).to_numpy()
        out = np.empty_like(a)
        out[ps.pid.to_numpy()] = a
        return out

    x0 = by_pid("x")
    g_prev = None
    v = by_pid("v")
    speed = np.sqrt((v.astype(np.float64) ** 2).sum(axis=1))
    rows = []
    xp = x0
    pid_prev = ps.pid.to_numpy().copy()
    gid_prev = np.empty(N, dtype=np.int64)
    gid_prev[pid_prev] = ps.grid_ids.to_numpy()
    for k in range(1, args.horizon + 1):
        solver.step(1)
        x = by_pid("x")
        pid = ps.pid.to_numpy()
        gid = np.empty(N, dtype=np.int64)
        gid[pid] = ps.grid_ids.to_numpy()
        # grid_ids are those of the sort at the START of the step, i.e. of the positions the previous step left
        d0 = np.sqrt(((x.astype(np.float64) - x0) ** 2).sum(axis=1))
        d1 = np.sqrt(((x.astype(np.float64) - xp) ** 2).sum(axis=1))
        rows.append({"k": k, "max_disp_over_h": round(float(d0.max()) / h, 5), "p999_disp_over_h": round(float(np.quantile(d0, 0.999)) / h, 5),
                     "mean_disp_over_h": round(float(d0.mean()) / h, 6), "max_step_disp_over_h": round(float(d1.max()) / h, 5),
                     "cell_changes_in_step": int((gid != gid_prev).sum())})
        xp, gid_prev = x, gid
    # rebuild interval per skin: the largest k with 2 * max displacement(k) <= s * h
    intervals = {}
    for s in (0.02, 0.05, 0.1, 0.2, 0.4):
        ok = [r["k"] for r in rows if 2.0 * r["max_disp_over_h"] <= s]
        intervals[str(s)] = {"steps_a_list_survives": (max(ok) if ok else 0),
                             "list_length_factor": round((1.0 + s) ** 3, 3)}
    st = _lib.SphStats()
    ps._call("sph_get_stats", st)
    mean_list = st.list_entries / max(st.targets - st.list_overflow_targets - st.lds_overflow_targets, 1)

    # ---- sweep times on this state (wall clock over many launches between two synchronisations) ----
    def timed(fn, reps):
        fn(); ps.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ps.sync()
        return (time.perf_counter() - t0) / reps * 1e3

    ps.set_option(_lib.OPT_TIMING, 1)
    ps._call("sph_reset_timings")
    solver.step(64)
    ps.sync()
    tm = _lib.SphTimings()
    ps._call("sph_get_timings", tm)
    ps.set_option(_lib.OPT_TIMING, 0)
    kk = max(int(tm.steps), 1)
    phases = {"sort": tm.sort_ms / kk, "density_filtering": tm.neighbour_ms / kk, "force": tm.force_ms / kk, "integrate": tm.integrate_ms / kk}
    t_step = timed(lambda: solver.step(1), 64)
    # list-reading sweep of the density sweep's size over the SAME positions: DFSPH's density sweep writes lists + stg records of
    # its own kind, the factor sweep (DFSPH.py:116-154: one W-gradient per pair, no neighbour gather) reads them
    ps._call("sph_initialize_particle_system")
    ps._call("sph_dfsph_compute_densities")
    t_df_density = timed(lambda: ps._call("sph_dfsph_compute_densities"), 32)
    t_list_read = timed(lambda: ps._call("sph_dfsph_compute_DFSPH_factor"), 64)
    t_sort = timed(lambda: ps._call("sph_sort"), 32)
    out = {
        "workload": args.workload, "particles": N, "h": h, "dt": dt, "settled_steps": args.settle,
        "speed": {"max": round(float(speed.max()), 4), "p999": round(float(np.quantile(speed, 0.999)), 4), "mean": round(float(speed.mean()), 4),
                  "max_speed_dt_over_h": round(float(speed.max()) * dt / h, 5)},
        "mean_list_entries": round(mean_list, 2), "max_list_entries": int(st.max_list),
        "displacement": rows, "rebuild_interval_by_skin": intervals,
        "ms": {"step": round(t_step, 4), **{k_: round(v_, 4) for k_, v_ in phases.items()},
               "list_writing_density_sweep_dfsph_kind": round(t_df_density, 4), "list_reading_sweep": round(t_list_read, 4),
               "stand_alone_sort": round(t_sort, 4)},
    }
    # the best case of a reuse step: no sort, the density sweep as a list reader, the force sweep over lists (1 + s)^3 longer
    best = {}
    for s, iv in intervals.items():
        f = iv["list_length_factor"]
        reuse = t_list_read * (0.35 + 0.65 * f) + phases["force"] * (0.3 + 0.7 * f) + phases["integrate"]
        k = iv["steps_a_list_survives"]
        best[s] = {"reuse_step_ms_best_case": round(reuse, 4), "survives": k}
    out["model"] = {"note": "reuse step = list-reading density sweep + force sweep, both with their pair loops (65 % / 70 % of the sweep: "
                            "profiles/r04m ablation rows) scaled by the list-length factor (1 + s)^3; no sort, no empty launches, no guard -- "
                            "a lower bound.  A rebuild step costs at least today's step (its filter must look two cells out, see DESIGN).",
                    "by_skin": best}
    print(json.dumps(out))
    ps.close()


if __name__ == "__main__":
    main()
