"""One context, 14 M particles (the 8-GPU workload of BASELINE.md on a single MI355X): indexing at scale + timing."""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sph_taichi_amd import ParticleSystem, SimConfig, _lib
from sph_taichi_amd.distributed import slab_bench_scene

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sd, n = slab_bench_scene(world)
t0 = time.perf_counter()
ps = ParticleSystem(SimConfig(config=copy.deepcopy(sd)))
solver = ps.build_solver()
solver.initialize()
ps.sync()
print(f"{n} particles, {int(np.prod(ps.grid_num))} cells: scene + upload + initialize {time.perf_counter() - t0:.1f} s", flush=True)
solver.step(10); ps.sync()
ps.set_option(_lib.OPT_TIMING, 1); ps._call("sph_reset_timings")
t0 = time.perf_counter(); solver.step(50); ps.sync(); dt = time.perf_counter() - t0
tm = _lib.SphTimings(); ps._call("sph_get_timings", tm); k = tm.steps
print(f"{dt / 50 * 1e3:.3f} ms/step  ({50 / dt * n / 1747584:.1f} steps/s at 1.74 M)  sort {tm.sort_ms/k:.3f} density {tm.neighbour_ms/k:.3f} "
      f"force {tm.force_ms/k:.3f} integrate {tm.integrate_ms/k:.3f}", flush=True)
x = ps.x.to_numpy(); pid = ps.pid.to_numpy(); gi = ps.grid_ids.to_numpy()
assert np.isfinite(x).all() and np.array_equal(np.sort(pid), np.arange(n)) and np.all(np.diff(gi) >= 0)
rho = ps.density.to_numpy()
print(f"ok: rho in [{rho.min():.1f}, {rho.max():.1f}], uniform-fluid sweep state {ps.get_option(_lib.OPT_UNIFORM_FLUID_STATE)}")
ps.close()
