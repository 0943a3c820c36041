"""What the GPU box's host gives this container: visible CPUs, cgroup quota, and how the CPU oracle scales with OpenMP threads."""
import os, sys, time, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/proc/loadavg"):
    try:
        print(f, open(f).read().strip())
    except OSError as e:
        print(f, "n/a")
import bench
from oracle.oracle import Oracle, lib
from sph_taichi_amd.config_builder import SimConfig
from sph_taichi_amd import scene as scene_mod
sd = bench.scene_dict("c1_dambreak_262k")
cfg = SimConfig(config=copy.deepcopy(sd)); sc = scene_mod.build_scene(cfg); g = sc.geom
params = dict(particle_radius=g.particle_radius, domain_size=list(g.domain_size), density_0=cfg.get_cfg("density0"),
              stiffness=cfg.get_cfg("stiffness"), exponent=cfg.get_cfg("exponent"), dt=cfg.get_cfg("timeStepSize"),
              g=cfg.get_cfg("gravitation"), simulation_method=0, fluid_particle_num=sc.fluid_particle_num)
for timing in (False, True):
    for th in (1, 4, 8, 16, 32, 64, 128, 256):
        if th > (os.cpu_count() or 1):
            continue
        o = Oracle(params, sc.arrays, n_objects=1, omp_threads=th, timing_build=timing)
        o.initialize(); o.step(1)
        t0 = time.perf_counter(); o.step(3); dt = (time.perf_counter() - t0) / 3
        print(f"{'timing' if timing else 'parity'} build, {th:3d} threads: {dt*1e3:8.1f} ms/step (262k particles)", flush=True)
