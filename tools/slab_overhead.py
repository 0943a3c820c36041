"""How much does the slab driver's per-step host logic cost?  world = 1 (no neighbours, no messages) against the
plain device loop on the same scene."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sph_taichi_amd import ParticleSystem, SimConfig, _lib
from sph_taichi_amd.distributed import SlabSolver, slab_bench_scene, LocalTransport
import copy

sd, n = slab_bench_scene(1)
ps = ParticleSystem(SimConfig(config=copy.deepcopy(sd)))
solver = ps.build_solver(); solver.initialize(); solver.step(10); ps.sync()
t0 = time.perf_counter(); solver.step(100); ps.sync(); t1 = time.perf_counter()
print(f"plain sph_step            : {(t1 - t0) * 10:.3f} ms/step  ({n} particles)")
ps.close()

class NoTransport:
    def start_counts(self, a, b): self._pending = None
    def exchange(self, sL, nL, sR, nR, alloc): return None, 0, None, 0
    def all_reduce_sum(self, t): return t
s = SlabSolver(sd, 0, 1, device=0)
s.attach(NoTransport()); s.initialize(); s.step(10); s.ps.sync()
t0 = time.perf_counter(); s.step(100); s.ps.sync(); t1 = time.perf_counter()
print(f"SlabSolver world=1 (no msg): {(t1 - t0) * 10:.3f} ms/step")
s.ps.set_option(_lib.OPT_TIMING, 1); s.ps._call("sph_reset_timings")
t0 = time.perf_counter(); s.step(100); s.ps.sync(); t1 = time.perf_counter()
tm = _lib.SphTimings(); s.ps._call("sph_get_timings", tm)
print(f"  with SPH_OPT_TIMING      : {(t1 - t0) * 10:.3f} ms/step; events: sort {tm.sort_ms/tm.steps:.3f} density {tm.neighbour_ms/tm.steps:.3f} "
      f"force {tm.force_ms/tm.steps:.3f} advect {tm.integrate_ms/tm.steps:.3f}; host {s.host_ms}")
s.close()
# the same with the native RCCL transport (csrc/sph_comm.hip; world = 1: communicator of one rank, no neighbours -- what
# is measured is the driver's per-step calls with the exchange entry points in the path)
from sph_taichi_amd.distributed import NativeTransport
s = SlabSolver(sd, 0, 1, device=0)
tr = NativeTransport(s.ps, torch.device("cuda", 0), rank=0, world=1)
s.attach(tr); s.initialize(); s.step(10); s.ps.sync()
t0 = time.perf_counter(); s.step(100); s.ps.sync(); t1 = time.perf_counter()
print(f"SlabSolver world=1 (NativeTransport): {(t1 - t0) * 10:.3f} ms/step; host {s.host_ms}")
tr.close(); s.close()
