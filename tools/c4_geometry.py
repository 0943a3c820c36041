"""BASELINE.json config 5 in its own geometry on ONE MI355X: the (16, 4, 3.4) tank with the 512 x 165 x 165 =
13,939,200-particle dam-break column at its -x end.

  (1) one context: scene + upload + initialize, ms/step over the first collapse, invariants;
  (2) the same scene as 8 logical slabs in one process (distributed.run_local_slabs: the exchange is a device
      pointer hand-over, everything else is what a rank does), cut by particle count, with re-cut every K steps:
      the column collapses into the empty 5.8 m of the tank, so the balancing cuts must travel -- the unbalanced
      case the re-cut exists for.  Checks conservation and compares positions with (1) after the same steps.

Usage: python tools/c4_geometry.py [--steps 200] [--recut-every 10] [--scale 1.0] [--out gpurun_out/c4.json]
  --scale s < 1 shrinks the column's particle counts (CPU-side smoke / quicker GPU runs).
"""
from __future__ import annotations

import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


from sph_taichi_amd.distributed import c4_dambreak_scene as c4_scene  # noqa: E402  (the scene lives with the slab bench now)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2500,
                    help="total steps; positions are compared with the one-context run after 200 steps (asserted <= 1e-4) and at the end (reported)")
    ap.add_argument("--recut-every", type=int, default=10)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--out", default="gpurun_out/c4_geometry.json")
    a = ap.parse_args()
    from sph_taichi_amd import ParticleSystem, SimConfig, _lib
    from sph_taichi_amd.distributed import SlabSolver, run_local_slabs, gather_by_pid
    sd, n = c4_scene(a.scale)
    res = {"particles": n, "domain": sd["Configuration"]["domainEnd"], "steps": a.steps}

    t0 = time.perf_counter()
    ps = ParticleSystem(SimConfig(config=copy.deepcopy(sd)))
    solver = ps.build_solver()
    solver.initialize()
    ps.sync()
    res["cells"] = int(np.prod(ps.grid_num))
    res["setup_s"] = round(time.perf_counter() - t0, 2)
    def snapshot():
        out = np.empty((n, 3), np.float32)
        pid_ = ps.pid.to_numpy()
        out[pid_] = ps.x.to_numpy()
        assert np.isfinite(out).all() and np.array_equal(np.sort(pid_), np.arange(n))
        return out

    first = min(200, a.steps)
    solver.step(10); ps.sync()
    ps.set_option(_lib.OPT_TIMING, 1); ps._call("sph_reset_timings")
    t0 = time.perf_counter(); solver.step(first - 10); ps.sync(); dt = time.perf_counter() - t0
    tm = _lib.SphTimings(); ps._call("sph_get_timings", tm); k = max(int(tm.steps), 1)
    res["one_context"] = {"first_collapse": {"steps": first, "ms_per_step": round(dt / (first - 10) * 1e3, 4),
                          "steps_per_s_at_1.74M": round((first - 10) / dt * n / 1747584, 1),
                          "sort": round(tm.sort_ms / k, 4), "neighbour": round(tm.neighbour_ms / k, 4),
                          "force": round(tm.force_ms / k, 4), "integrate": round(tm.integrate_ms / k, 4)}}
    x_first = snapshot()
    ps.set_option(_lib.OPT_TIMING, 0)
    if a.steps > first:
        solver.step(a.steps - first - 50); ps.sync()
        ps.set_option(_lib.OPT_TIMING, 1); ps._call("sph_reset_timings")
        t0 = time.perf_counter(); solver.step(50); ps.sync(); dt = time.perf_counter() - t0
        tm = _lib.SphTimings(); ps._call("sph_get_timings", tm); k = max(int(tm.steps), 1)
        res["one_context"]["developed"] = {"after_steps": a.steps, "ms_per_step": round(dt / 50 * 1e3, 4),
                                           "steps_per_s_at_1.74M": round(50 / dt * n / 1747584, 1),
                                           "sort": round(tm.sort_ms / k, 4), "neighbour": round(tm.neighbour_ms / k, 4),
                                           "force": round(tm.force_ms / k, 4), "integrate": round(tm.integrate_ms / k, 4)}
    st = _lib.SphStats(); ps._call("sph_get_stats", st)
    res["one_context"].update(max_list_entries=st.max_list, list_overflow_targets=st.list_overflow_targets,
                              lds_overflow_targets=st.lds_overflow_targets, max_cell_occupancy=st.max_cell_occupancy)
    x_ref = snapshot()
    res["one_context"]["front_x_max"] = float(x_ref[:, 0].max())
    ps.close()
    print(json.dumps(res["one_context"]), flush=True)

    world = a.world
    t0 = time.perf_counter()
    solvers = [SlabSolver(sd, r, world, device=0, recut_every=a.recut_every) for r in range(world)]
    run_local_slabs(solvers, 0, initialize=True)
    cuts0 = list(solvers[0].cuts)
    owned0 = [int(s.owned_range[1]) for s in solvers]
    for s in solvers:
        s.ps.sync()
    setup = time.perf_counter() - t0
    rel = lambda x, r: float(np.linalg.norm(x.astype(np.float64) - r) / np.linalg.norm(r.astype(np.float64)))
    t0 = time.perf_counter()
    run_local_slabs(solvers, first)
    for s in solvers:
        s.ps.sync()
    dt_first = time.perf_counter() - t0
    err_first = rel(gather_by_pid(solvers, "x", n), x_first)
    t0 = time.perf_counter()
    if a.steps > first:
        run_local_slabs(solvers, a.steps - first)
        for s in solvers:
            s.ps.sync()
    dt = time.perf_counter() - t0
    owned1 = [int(s.owned_range[1]) for s in solvers]
    assert sum(owned1) == n, (sum(owned1), n)
    x = gather_by_pid(solvers, "x", n)
    err = rel(x, x_ref)
    res["logical_slabs"] = {"world": world, "recut_every": a.recut_every, "setup_s": round(setup, 2),
                            "ms_per_step_all_slabs_in_lock_step_first": round(dt_first / first * 1e3, 4),
                            "ms_per_step_all_slabs_in_lock_step_rest": round(dt / max(a.steps - first, 1) * 1e3, 4),
                            "cuts_start": cuts0, "cuts_end": list(solvers[0].cuts),
                            "recut_events_per_slab": [int(s.stats.get("recuts", 0)) for s in solvers],
                            "owned_start": owned0, "owned_end": owned1,
                            "imbalance_start": round(max(owned0) / (n / world), 4),
                            "imbalance_end": round(max(owned1) / (n / world), 4),
                            "imbalance_end_with_the_starting_cuts": None,
                            f"rel_l2_x_vs_one_context_after_{first}": err_first,
                            f"rel_l2_x_vs_one_context_after_{a.steps}": err}
    # what the imbalance would be now had the cuts stayed where they started
    lay = np.clip((x[:, 0].astype(np.float32) / np.float32(0.04)).astype(np.int64), 0, cuts0[-1] - 1)
    frozen = [int(((lay >= cuts0[r]) & (lay < cuts0[r + 1])).sum()) for r in range(world)]
    res["logical_slabs"]["imbalance_end_with_the_starting_cuts"] = round(max(frozen) / (n / world), 4)
    for s in solvers:
        s.close()
    print(json.dumps(res["logical_slabs"]), flush=True)
    assert err_first <= 1e-4, err_first
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
