"""Long runs of the bench workloads: no NaN/inf, every particle inside the walls, pid set intact, densities sane.
Usage: python tools/soak.py [steps]"""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from sph_taichi_amd import ParticleSystem, SimConfig

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
for workload, solver_name, n in [("c1_dambreak_262k", "wcsph", steps), ("c2_dragon_bath", "wcsph", steps),
                                 ("c3_armadillo_equiv", "wcsph", steps // 2), ("c1_dambreak_262k", "dfsph", steps // 10),
                                 ("c2_dragon_bath", "dfsph", steps // 10)]:
    sd = bench.scene_dict(workload, solver_name)
    ps = ParticleSystem(SimConfig(config=copy.deepcopy(sd)))
    solver = ps.build_solver()
    solver.initialize()
    t0 = time.perf_counter()
    done = 0
    while done < n:
        k = min(500, n - done)
        solver.step(k)
        done += k
    ps.sync()
    dt = time.perf_counter() - t0
    x, v, rho = ps.x.to_numpy(), ps.v.to_numpy(), ps.density.to_numpy()
    mat, dyn = ps.material.to_numpy(), ps.is_dynamic.to_numpy()
    pid = ps.pid.to_numpy()
    g = ps._scene.geom
    pad = np.float32(g.padding)
    hi = (np.asarray(g.domain_size) - g.padding).astype(np.float32)
    fl = mat == 1
    ok = {
        "finite": bool(np.isfinite(x).all() and np.isfinite(v).all() and np.isfinite(rho).all()),
        "pid_set": bool(np.array_equal(np.sort(pid), np.arange(pid.size))),
        "fluid_inside_walls": bool((x[fl] >= pad - 1e-6).all() and (x[fl] <= hi + 1e-6).all()),
        "sorted": bool(np.all(np.diff(ps.grid_ids.to_numpy()) >= 0)),
    }
    extra = f"rho[{rho[fl].min():.0f},{rho[fl].max():.0f}] |v|max {np.abs(v).max():.2f} y_mean {x[fl,1].mean():.3f}"
    if solver_name == "dfsph":
        st = solver.stats()
        extra += f" iters/step {st['total_iterations_v']/max(st['steps'],1):.1f}+{st['total_iterations']/max(st['steps'],1):.1f}"
    print(f"{workload:22s} {solver_name} {n:5d} steps {dt/n*1e3:.3f} ms/step  {ok}  {extra}", flush=True)
    assert all(ok.values()), ok
    ps.close()
print("soak ok")
