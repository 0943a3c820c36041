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


def c4_scene(scale=1.0):
    cfg = {
        "domainStart": [0.0, 0.0, 0.0], "domainEnd": [16.0, 4.0, 3.4], "particleRadius": 0.01,
        "numberOfStepsPerRenderUpdate": 1, "density0": 1000, "simulationMethod": 0,
        "gravitation": [0.0, -9.81, 0.0], "timeStepSize": 0.0004, "stiffness": 50000, "exponent": 7,
        "boundaryHandlingMethod": 0, "exportFrame": False, "exportPly": False, "exportObj": False,
    }
    d = 0.02
    counts = tuple(max(int(round(c * scale)), 8) for c in (512, 165, 165))
    corner = (0.04, 0.04, 0.04)
    end = [c + (n - 0.5) * d for c, n in zip(corner, counts)]
    return {"Configuration": cfg,
            "FluidBlocks": [{"objectId": 0, "start": list(corner), "end": end, "translation": [0.0, 0.0, 0.0],
                             "scale": [1, 1, 1], "velocity": [0.0, 0.0, 0.0], "density": 1000.0,
                             "color": [50, 100, 200]}]}, counts[0] * counts[1] * counts[2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
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
    solver.step(10); ps.sync()
    ps.set_option(_lib.OPT_TIMING, 1); ps._call("sph_reset_timings")
    t0 = time.perf_counter(); solver.step(a.steps - 10); ps.sync(); dt = time.perf_counter() - t0
    tm = _lib.SphTimings(); ps._call("sph_get_timings", tm); k = max(int(tm.steps), 1)
    st = _lib.SphStats(); ps._call("sph_get_stats", st)
    res["one_context"] = {"ms_per_step": round(dt / (a.steps - 10) * 1e3, 4),
                          "steps_per_s_at_1.74M": round((a.steps - 10) / dt * n / 1747584, 1),
                          "sort": round(tm.sort_ms / k, 4), "neighbour": round(tm.neighbour_ms / k, 4),
                          "force": round(tm.force_ms / k, 4), "integrate": round(tm.integrate_ms / k, 4),
                          "max_list_entries": st.max_list, "list_overflow_targets": st.list_overflow_targets,
                          "lds_overflow_targets": st.lds_overflow_targets}
    x_ref = np.empty((n, 3), np.float32)
    pid = ps.pid.to_numpy()
    x_ref[pid] = ps.x.to_numpy()
    assert np.isfinite(x_ref).all() and np.array_equal(np.sort(pid), np.arange(n))
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
    t0 = time.perf_counter()
    run_local_slabs(solvers, a.steps)
    for s in solvers:
        s.ps.sync()
    dt = time.perf_counter() - t0
    owned1 = [int(s.owned_range[1]) for s in solvers]
    assert sum(owned1) == n, (sum(owned1), n)
    x = gather_by_pid(solvers, "x", n)
    err = float(np.linalg.norm(x.astype(np.float64) - x_ref) / np.linalg.norm(x_ref.astype(np.float64)))
    res["logical_slabs"] = {"world": world, "recut_every": a.recut_every, "setup_s": round(setup, 2),
                            "ms_per_step_all_slabs_in_lock_step": round(dt / a.steps * 1e3, 4),
                            "cuts_start": cuts0, "cuts_end": list(solvers[0].cuts),
                            "recuts": int(solvers[0].stats.get("recuts", 0)),
                            "owned_start": owned0, "owned_end": owned1,
                            "imbalance_start": round(max(owned0) / (n / world), 4),
                            "imbalance_end": round(max(owned1) / (n / world), 4),
                            "rel_l2_x_vs_one_context": err}
    for s in solvers:
        s.close()
    print(json.dumps(res["logical_slabs"]), flush=True)
    assert err <= 1e-4, err
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
