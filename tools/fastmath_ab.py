#!/usr/bin/env python
"""A/B of the fast-math choice (VERDICT r03 "weak" #1, next-round #2): the two brick sweeps of the fused WCSPH step with
v_rsq_f32 / v_rcp_f32 and the one-expression spline (the product) against their SPH_OPT_EXACT_MATH instances (IEEE sqrt and
divide, the two-branch spline, no FMA contraction: the reference's f32 expressions as the oracle evaluates them), on
  c2   dragon_bath.json equivalent to step 300 (floor impact at ~60 ... the pressure wave ... lateral spreading),
  c1   the 64^3 dam-break, 100 steps from the developed state (2,500 steps in),
  big  ref_big_fluid_wall: 10,240 particles thrown into a corner, 50 steps of the reference's own source (fixture),
each against the CPU oracle (c2, c1) or the reference-executed fixture (big).  If the post-impact errors of the fast build
are rounding, the exact build brings them down to the oracle-vs-reference level; if not, there is a defect to find.
Writes gpurun_out/<tag>/parity_fastmath_ab.json (copied to profiles/r04_parity_fastmath_ab.json)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import scenes  # noqa: E402
import test_gpu_fullsize as fs  # noqa: E402
import test_golden as tg  # noqa: E402
from sph_taichi_amd import _lib  # noqa: E402


def errs(ps, o):
    return {f: scenes.rel_l2(scenes.ps_by_pid(ps, f), o.by_pid(f)) for f in ("x", "v", "density")}


def march(sd, checkpoints, exact, arrays=None, oracle=None):
    cfg, sc = scenes.build(sd)
    if arrays is not None:
        sc.arrays["x"], sc.arrays["v"] = arrays["x"].copy(), arrays["v"].copy()
    o = scenes.make_oracle(cfg, sc, omp_threads=fs._threads())
    ps, solver = scenes.make_ps(sd, arrays=arrays)
    ps.set_option(_lib.OPT_EXACT_MATH, exact)
    o.initialize(); solver.initialize()
    out, done = {}, 0
    for n in checkpoints:
        o.step(n - done); solver.step(n - done)
        done = n
        out[str(n)] = errs(ps, o)
    ps.close()
    return out


def big_fixture(exact):
    path = [p for p in tg.BIG if "fluid_wall" in p][0]
    z, sd, steps = tg._load(path)
    ps, solver = scenes.make_ps(sd)
    ps.set_option(_lib.OPT_EXACT_MATH, exact)
    solver.initialize()
    out, done = {}, 0
    for n in (1, 10, 25, steps):
        solver.step(n - done)
        done = n
        gi = ps.grid_ids.to_numpy()
        same = bool(np.array_equal(gi, z[f"step{n}/grid_ids"]))
        e = {"cell_ids_identical": same, "cell_ids_differing": int((gi != z[f"step{n}/grid_ids"]).sum())}
        if same:
            for f in ("x", "v", "density"):
                e[f] = scenes.rel_l2(getattr(ps, f).to_numpy(), z[f"step{n}/{f}"])
        out[str(n)] = e
    ps.close()
    return out


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
    res = {"what": __doc__.split("\n\n")[0].replace("\n", " "),
           "errors": "relative L2 against the CPU oracle (c2, c1) / the reference-executed fixture (big), by persistent id",
           "fast": {}, "exact": {}}
    t0 = time.time()
    sd1 = scenes.fluid_only(velocity=(0.0, 0.0, 0.0), **fs.C1)
    x, v = fs._developed_state(sd1, 2500)
    for name, exact in (("fast", 0), ("exact", 1)):
        res[name]["c2_dragon_bath"] = march(fs.dragon_bath_scene(), (50, 100, 200, 300), exact)
        res[name]["c1_dambreak_developed"] = march(sd1, (25, 50, 100), exact, arrays={"x": x, "v": v})
        res[name]["ref_big_fluid_wall"] = big_fixture(exact)
        print(name, json.dumps(res[name]), flush=True)
    res["oracle_vs_reference_on_ref_big_fluid_wall_step50"] = {"x": 2.3e-5, "v": 4e-4, "density": 1.1e-4,
                                                               "source": "tests/test_golden.py (measured r03, CPU)"}
    res["seconds"] = round(time.time() - t0, 1)
    out = os.path.join(ROOT, "gpurun_out", tag)
    os.makedirs(out, exist_ok=True)
    json.dump(res, open(os.path.join(out, "parity_fastmath_ab.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
