"""GPU cost of slab mode for ranks WITH neighbours, without a second GPU: P logical slabs of the bench's tiled scene
run in lock-step in one process (device-to-device hand-over, no transport), against the same particles in one
context.  The slabs run one after the other on the one GPU, so (slab time) / P is a rank's GPU work per step --
ghost layers, boundary launches, packers, inserts included; the exchange itself is not."""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sph_taichi_amd import ParticleSystem, SimConfig
from sph_taichi_amd.distributed import SlabSolver, slab_bench_scene, run_local_slabs

P = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
sd, n = slab_bench_scene(P)
ps = ParticleSystem(SimConfig(config=copy.deepcopy(sd)))
solver = ps.build_solver(); solver.initialize(); solver.step(10); ps.sync()
t0 = time.perf_counter(); solver.step(steps); ps.sync(); t1 = time.perf_counter()
plain = (t1 - t0) / steps * 1e3
print(f"one context, {n} particles : {plain:.3f} ms/step = {plain / P:.3f} ms per 1.75 M")
ps.close()
solvers = [SlabSolver(sd, r, P, device=0) for r in range(P)]
run_local_slabs(solvers, 1, initialize=True)
run_local_slabs(solvers, 10)
for s in solvers: s.ps.sync()
t0 = time.perf_counter(); run_local_slabs(solvers, steps)
for s in solvers: s.ps.sync()
t1 = time.perf_counter()
slab = (t1 - t0) / steps * 1e3
print(f"{P} logical slabs          : {slab:.3f} ms/step = {slab / P:.3f} ms per slab-step "
      f"(+{(slab / plain - 1) * 100:.1f} % over one context); owned {[s.owned_range[1] for s in solvers]}")
for s in solvers: s.close()
