"""A/B table of the SPH_OPT_KERNEL_VARIANT instances of the two brick sweeps (GPU).

For every variant mask: a fresh context on the rest lattice (5 warm-up + `--steps` timed steps, like bench.py's
default line), then -- on ONE context settled for `--settle` steps -- the same masks back to back on the developed
flow (mask 0 repeated at the end to show the drift of the state itself).  Per-phase times are the HIP-event
buckets of sph_step.  Writes one JSON document.

  python tools/variant_sweep.py --out gpurun_out/variants.json [--variants 0,1,4,5,...] [--workload c3p_uniform_1.75M]
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

import bench  # noqa: E402


def timed(ps, solver, lib, steps, warmup):
    ps.set_option(lib.OPT_TIMING, 0)
    solver.step(warmup)
    ps.set_option(lib.OPT_TIMING, 1)
    ps._call("sph_reset_timings")
    ps.sync()
    t0 = time.perf_counter()
    solver.step(steps)
    ps.sync()
    dt = time.perf_counter() - t0
    tm = lib.SphTimings()
    ps._call("sph_get_timings", tm)
    k = max(int(tm.steps), 1)
    return {"ms_per_step": round(dt / steps * 1e3, 4), "sort": round(tm.sort_ms / k, 4),
            "neighbour": round(tm.neighbour_ms / k, 4), "force": round(tm.force_ms / k, 4),
            "integrate": round(tm.integrate_ms / k, 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3p_uniform_1.75M")
    ap.add_argument("--variants", default="25,0,1,24")
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--settle", type=int, default=2000)
    ap.add_argument("--settled-steps", type=int, default=60)
    ap.add_argument("--shapes", default="0", help="SPH_OPT_BRICK_SHAPE values to cross the variants with (0 = adaptive height, 1 = fixed 4x2x4)")
    ap.add_argument("--dump-settled", default="", help="write the settled positions (float32 [N,3], by pid) to this .npy")
    ap.add_argument("--out", default="gpurun_out/variants.json")
    a = ap.parse_args()
    from sph_taichi_amd import ParticleSystem, SimConfig, _lib
    variants = [int(v) for v in a.variants.split(",")]
    shapes = [int(v) for v in a.shapes.split(",")]
    sd = bench.scene_dict(a.workload)
    out = {"workload": a.workload, "steps": a.steps, "settle": a.settle, "rest": {}, "settled": {}}

    def fresh():
        ps = ParticleSystem(SimConfig(config=copy.deepcopy(sd)))
        solver = ps.build_solver()
        solver.initialize()
        return ps, solver

    combos = [(sh, v) for sh in shapes for v in variants]
    name = lambda sh, v: str(v) if shapes == [0] else f"shape{sh}_var{v}"
    for sh, v in combos:
        ps, solver = fresh()
        ps.set_option(_lib.OPT_BRICK_SHAPE, sh)
        ps.set_option(_lib.OPT_KERNEL_VARIANT, v)
        out["rest"][name(sh, v)] = timed(ps, solver, _lib, a.steps, 5)
        print(f"[rest] shape {sh} variant {v:3d}: {out['rest'][name(sh, v)]}", flush=True)
        ps.close()
    if a.settle > 0:
        ps, solver = fresh()
        solver.step(a.settle)
        ps.sync()
        if a.dump_settled:
            import numpy as np
            x = ps.x.to_numpy()
            np.save(a.dump_settled, x.astype(np.float32))
        for sh, v in combos + [combos[0]]:
            ps.set_option(_lib.OPT_BRICK_SHAPE, sh)
            ps.set_option(_lib.OPT_KERNEL_VARIANT, v)
            key = name(sh, v) if name(sh, v) not in out["settled"] else f"{name(sh, v)}_again"
            out["settled"][key] = timed(ps, solver, _lib, a.settled_steps, 3)
            print(f"[settled {a.settle}] shape {sh} variant {v:3d}: {out['settled'][key]}", flush=True)
        st = _lib.SphStats()
        ps._call("sph_get_stats", st)
        out["settled_neighbourhood"] = {"mean_list_entries": round(st.list_entries / max(st.targets, 1), 2),
                                        "max_list_entries": st.max_list, "max_cell_occupancy": st.max_cell_occupancy}
        ps.close()
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
