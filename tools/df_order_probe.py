"""Why does bench.py's DFSPH run-ahead A/B report other iteration counts after ~40 steps at 1.75 M particles?  The second block
restarts from the same positions and velocities by persistent id -- in the particle ORDER the first block ended in.  This probe
separates the two suspects: (1) the default loop twice, the second time from the restored state in the other order; (2) the
default loop and the run-ahead loop from two FRESH contexts (the same order).  If (1) differs and (2) does not, the order is the
cause and the run-ahead loop is exact.  Usage: python tools/df_order_probe.py [steps=63]"""
import copy, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from sph_taichi_amd import ParticleSystem, SimConfig, _lib

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 63
sd = bench.scene_dict("c3p_uniform_1.75M", "dfsph")


def counts(solver, n):
    out = []
    for _ in range(n):
        solver.step(1)
        st = solver.stats()
        out.append((st["iterations_v"], st["iterations"]))
    return out


def by_pid(ps, name):
    pid = ps.pid.to_numpy()
    a = getattr(ps, name).to_numpy()
    o = np.empty_like(a); o[pid] = a
    return o

res = {}
ps = ParticleSystem(SimConfig(config=copy.deepcopy(sd)), device=0); s = ps.build_solver(); s.initialize()
x0, v0 = by_pid(ps, "x"), by_pid(ps, "v")
a = counts(s, steps)
xa = by_pid(ps, "x")
pid = ps.pid.to_numpy()
ps.x.from_numpy(x0[pid]); ps.v.from_numpy(v0[pid]); s.initialize()
b = counts(s, steps)
xb = by_pid(ps, "x")
ps.close()
res["default_then_default_from_restored_state_in_the_other_order"] = {
    "same_counts": a == b, "first_step_that_differs": next((i for i, (p, q) in enumerate(zip(a, b)) if p != q), None),
    "rel_l2_x": float(np.linalg.norm(xa - xb) / np.linalg.norm(xa))}
ps = ParticleSystem(SimConfig(config=copy.deepcopy(sd)), device=0); s = ps.build_solver()
ps.set_option(_lib.OPT_DF_RUNAHEAD, 1); s.initialize()
c = counts(s, steps)
xc = by_pid(ps, "x")
ps.close()
res["default_vs_runahead_from_fresh_contexts"] = {
    "same_counts": a == c, "first_step_that_differs": next((i for i, (p, q) in enumerate(zip(a, c)) if p != q), None),
    "bit_identical_x": bool(np.array_equal(xa, xc))}
res["steps"] = steps
res["iterations_total_default"] = [int(sum(p for p, _ in a)), int(sum(q for _, q in a))]
print(json.dumps(res))
