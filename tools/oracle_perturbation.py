#!/usr/bin/env python
"""CPU only: how far do two CORRECT f32 evaluations of the reference's formulas drift apart on the scenes of
tools/fastmath_ab.py?  The oracle against itself with (1) the neighbours visited in the reverse order (every sum added up
in another order), (2) the integer Tait exponent by repeated multiplication instead of powf, (3) both.  These differences
are ulp-level by construction; what they have grown into after a floor impact is the noise floor any HIP-vs-oracle
comparison sits on.  Writes gpurun_out/<tag>/oracle_perturbation.json (-> profiles/r04_oracle_perturbation.json)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import scenes  # noqa: E402
import test_gpu_fullsize as fs  # noqa: E402
from oracle.oracle import Oracle, usable_cpus  # noqa: E402


def run(sd, checkpoints, perturbs, arrays=None):
    cfg, sc = scenes.build(sd)
    if arrays is not None:
        sc.arrays["x"], sc.arrays["v"] = arrays["x"].copy(), arrays["v"].copy()
    os_ = {p: Oracle(scenes.solver_params(cfg, sc), sc.arrays, n_objects=max(sc.n_objects, 1),
                     rigid_body_ids=sorted(sc.object_id_rigid_body), dynamic_ids=sorted(sc.dynamic_rigid_ids),
                     omp_threads=usable_cpus(), perturb=p) for p in [0] + list(perturbs)}
    for o in os_.values():
        o.initialize()
    out, done = {p: {} for p in perturbs}, 0
    for n in checkpoints:
        for o in os_.values():
            o.step(n - done)
        done = n
        for p in perturbs:
            out[p][str(n)] = {f: scenes.rel_l2(os_[p].by_pid(f), os_[0].by_pid(f)) for f in ("x", "v", "density")}
        print(n, json.dumps({p: out[p][str(n)] for p in perturbs}), flush=True)
    return out


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
    t0 = time.time()
    names = {1: "reversed_neighbour_order", 2: "tait_exponent_by_multiplication", 3: "both"}
    res = {"what": __doc__.split("\n\n")[0].replace("\n", " "), "threads": usable_cpus(),
           "errors": "relative L2 of the perturbed oracle against the unperturbed one, by persistent id"}
    c2 = run(fs.dragon_bath_scene(), (50, 100, 200, 300), (1, 2, 3))
    res["c2_dragon_bath"] = {names[p]: c2[p] for p in c2}
    res["seconds"] = round(time.time() - t0, 1)
    out = os.path.join(ROOT, "gpurun_out", tag)
    os.makedirs(out, exist_ok=True)
    json.dump(res, open(os.path.join(out, "oracle_perturbation.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
