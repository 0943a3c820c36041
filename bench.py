#!/usr/bin/env python
"""bench.py -- WCSPH steps/s + ms/step breakdown (sort / neighbour / force) on MI355X.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one WCSPHSolver.step() (SPHBase.step(), reference sph_base.py:263-271)
over the synthetic uniform box of BASELINE.md (C3': 246x74x96 = 1,747,584 fluid
particles in the armadillo scene's (5,3,2) domain; BASELINE.json's metric is quoted
at 1.74 M particles).  Inputs are resident in HBM before the timed region.

`value` = particle-steps/s / 1,747,584, i.e. "steps/s at 1.74 M particles": at
N = 1 it is exactly the job's steps/s; at N > 1 (x-slab sharding, every rank owns
one 1.75 M slab: weak scaling) it is the whole-job aggregate.

The JSON line also carries
  roofline      : the dominant kernel (the density + EOS sweep) -- algorithmic bytes per
                  launch (32*N + 4*G) / its mean launch time from HIP events
                  on the kernel's own stream, against the 8 TB/s HBM peak; `traffic`
                  (rocprofv3 FETCH_SIZE*2 + WRITE_SIZE) and the VALU figures come from
                  profiles/pmc_traffic.json and are quoted only if that file was
                  measured on the same kernel sources (fingerprint), else null;
  roofline_valu : wave-level VALU instructions per launch / launch time against
                  1024 SIMD-32 x 2.4 GHz / 2 cycles, and the useful FLOP rate of
                  SURVEY 8d against 157.3 TFLOP/s;
  cpu_baseline  : the CPU oracle (oracle/sph_oracle.c, "port" of the reference
                  algorithm with OpenMP; timing build -O3 -march=native) on the CPUs
                  this container may use (cgroup quota), median of three bounded
                  samples of the same workload;
  neighbourhood : list lengths / cell occupancy of the last density sweep.
--settle K times a developed flow, --variant M an A/B kernel instance, --ablate the
section table of DESIGN.md section 4.4 (stderr).
"""
from __future__ import annotations

import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

REF_PARTICLES = 1_747_584          # C3' 
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

CFG = {
    "domainStart": [0.0, 0.0, 0.0], "domainEnd": [5.0, 3.0, 2.0], "particleRadius": 0.01,
    "numberOfStepsPerRenderUpdate": 1, "density0": 1000, "simulationMethod": 0,
    "gravitation": [0.0, -9.81, 0.0], "timeStepSize": 0.0004, "stiffness": 50000, "exponent": 7,
    "boundaryHandlingMethod": 0, "exportFrame": False, "exportPly": False, "exportObj": False,
}

WORKLOADS = {
    # name: (domainEnd, fluid lattice counts, lower corner)
    "c3p_uniform_1.75M": ([5.0, 3.0, 2.0], (246, 74, 96), (0.04, 0.04, 0.04)),
    "c1_dambreak_262k": ([3.2, 2.0, 1.4], (64, 64, 64), (0.04, 0.04, 0.04)),
    "c0_dragon_fluid_423k": ([5.0, 3.0, 2.0], (55, 140, 55), (0.3, 0.1, 0.7)),
    # the headline box thrown against its +x / +z walls: with --settle K the timed steps run on a developed,
    # compressed / sloshing state instead of the rest lattice (VERDICT r01 "weak" #3)
    "c3p_slosh_1.75M": ([5.0, 3.0, 2.0], (246, 74, 96), (0.04, 0.04, 0.04)),
}
INITIAL_VELOCITY = {"c0_dragon_fluid_423k": [0.0, -1.0, 0.0], "c3p_slosh_1.75M": [1.0, -0.5, 0.5]}


DFSPH_DT = 0.004    # timeStepSize of every *_dfsph.json scene of the reference (data/scenes/)


def scene_dict(workload: str, solver: str = "wcsph"):
    sd = _scene_dict(workload)
    if solver == "dfsph":
        sd["Configuration"]["simulationMethod"] = 4
        sd["Configuration"]["timeStepSize"] = DFSPH_DT
    return sd


def _scene_dict(workload: str):
    if workload in ("c2_dragon_bath", "c3_armadillo_equiv"):
        # the reference's two demo scenes with their bodies taken from the committed voxel fixtures
        # (tests/golden/*.npy; /root/reference does not exist on the GPU box)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import test_gpu_fullsize as fs
        return fs.dragon_bath_scene() if workload == "c2_dragon_bath" else fs.armadillo_equiv_scene(body_y=2.0)
    dom, counts, corner = WORKLOADS[workload]
    cfg = copy.deepcopy(CFG)
    cfg["domainEnd"] = dom
    d = 2 * cfg["particleRadius"]
    end = [c + (n - 0.5) * d for c, n in zip(corner, counts)]
    vel = INITIAL_VELOCITY.get(workload, [0.0, 0.0, 0.0])
    return {"Configuration": cfg,
            "FluidBlocks": [{"objectId": 0, "start": list(corner), "end": end, "translation": [0.0, 0.0, 0.0],
                             "scale": [1, 1, 1], "velocity": vel, "density": 1000.0, "color": [50, 100, 200]}]}


def cpu_baseline(sd, sample_steps: int, repeats: int = 3):
    """Time the CPU oracle on a bounded sample of the same workload (rank 0, N=1 only): the timing build of
    oracle/sph_oracle.c (-O3 -march=native, compiled here on this host), OpenMP over every
    core the process may use, median of `repeats` samples of `sample_steps` steps each."""
    from oracle.oracle import Oracle, lib
    from sph_taichi_amd.config_builder import SimConfig
    from sph_taichi_amd import scene as scene_mod
    cfg = SimConfig(config=copy.deepcopy(sd))
    sc = scene_mod.build_scene(cfg)
    g = sc.geom
    params = dict(particle_radius=g.particle_radius, domain_size=list(g.domain_size),
                  density_0=cfg.get_cfg("density0"), stiffness=cfg.get_cfg("stiffness"),
                  exponent=cfg.get_cfg("exponent"), dt=cfg.get_cfg("timeStepSize"), g=cfg.get_cfg("gravitation"),
                  simulation_method=cfg.get_cfg("simulationMethod") or 0, fluid_particle_num=sc.fluid_particle_num)
    build = "-O3 -march=native (oracle/Makefile: libsph_oracle_timing.so)"
    try:
        L = lib(timing=True)
        timing = True
    except Exception as e:      # no compiler on this host: time the parity build and say so
        L = lib()
        timing = False
        build = f"-O2 -ffp-contract=off parity build (timing build failed: {type(e).__name__})"
    from oracle.oracle import usable_cpus
    threads = usable_cpus()          # scheduler affinity capped by the cgroup CPU quota (16 on the GPU boxes, whose nproc is 256)
    o = Oracle(params, sc.arrays, n_objects=max(sc.n_objects, 1), rigid_body_ids=sorted(sc.object_id_rigid_body),
               dynamic_ids=sorted(sc.dynamic_rigid_ids), omp_threads=threads, timing_build=timing)
    o.initialize()
    o.step(1)                                   # warm-up (page faults, first sort)
    samples, phases = [], None
    for _ in range(max(repeats, 1)):
        t0 = time.perf_counter()
        ms = o.step(sample_steps)
        samples.append(time.perf_counter() - t0)
        phases = ms
    samples.sort()
    dt = samples[len(samples) // 2]
    n = sc.particle_max_num
    return {"value": round(sample_steps / dt * n / REF_PARTICLES, 4), "unit": "steps/s at 1.74M particles",
            "cores": threads, "threads": threads, "nproc": os.cpu_count(), "kind": "port",
            "cores_note": "cores = CPUs this container may use (affinity capped by the cgroup quota); nproc = what the host reports",
            "sample": f"median of {len(samples)} samples of {sample_steps} steps of the same {n}-particle workload "
                      f"after 1 warm-up step; oracle/sph_oracle.c, {build}, {threads} OpenMP threads",
            "ms_per_step": round(dt / sample_steps * 1e3, 2),
            "samples_ms_per_step": [round(x / sample_steps * 1e3, 2) for x in samples],
            "phase_ms": {k: round(v / sample_steps, 2) for k, v in zip(("sort", "neighbour", "force", "integrate"), phases)}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c3p_uniform_1.75M",
                    choices=sorted(WORKLOADS) + ["c2_dragon_bath", "c3_armadillo_equiv"])
    ap.add_argument("--solver", default="wcsph", choices=["wcsph", "dfsph"],
                    help="dfsph: the same workload under DFSPHSolver (simulationMethod 4, dt = 4e-3) -- a supplementary "
                         "line, not BASELINE.json's metric")
    ap.add_argument("--settle", type=int, default=0,
                    help="run this many untimed steps first, so that the timed ones see a developed flow (dam-break front, "
                         "sloshing, compression at the walls) instead of the initial lattice")
    ap.add_argument("--recut-every", type=int, default=0,
                    help="--gpus N: re-cut the slabs every K steps (0 = never: the tiled workload is balanced by construction)")
    ap.add_argument("--gather-impl", type=int, default=1)
    ap.add_argument("--brick-shape", type=int, default=0)
    ap.add_argument("--fused", type=int, default=1)
    ap.add_argument("--variant", type=int, default=-1, help="SPH_OPT_KERNEL_VARIANT mask (-1 = the library's default)")
    ap.add_argument("--cpu-steps", type=int, default=10, help="CPU-oracle sample size (0 = skip the baseline leg)")
    ap.add_argument("--sweep", action="store_true", help="also time every gather variant (stderr table)")
    ap.add_argument("--ablate", action="store_true", help="profiling: time the sweeps with sections skipped (stderr)")
    ap.add_argument("--ablate-mask", type=int, default=0, help="profiling: run the whole bench with this ablation mask (results invalid)")
    args = ap.parse_args()

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # SPH_DIST_BACKEND=gloo lets several ranks share one GPU (how the N>1 path is exercised on a 1-GPU box)
    backend = os.environ.get("SPH_DIST_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        from sph_taichi_amd.distributed import run_slab_bench
        line = run_slab_bench(args, rank, world, local_rank)
        if rank == 0:
            print(json.dumps(line), flush=True)
        dist.destroy_process_group()
        return

    from sph_taichi_amd import ParticleSystem, SimConfig, _lib
    sd = scene_dict(args.workload, args.solver)
    ps = ParticleSystem(SimConfig(config=copy.deepcopy(sd)), device=local_rank)
    solver = ps.build_solver()
    N = ps.particle_max_num
    G = int(ps.grid_num[0] * ps.grid_num[1] * ps.grid_num[2])

    def run(impl, shape, fused, steps, warmup):
        ps.set_option(_lib.OPT_GATHER_IMPL, impl)
        ps.set_option(_lib.OPT_BRICK_SHAPE, shape)
        ps.set_option(_lib.OPT_FUSED_STEP, fused)
        ps.set_option(_lib.OPT_TIMING, 0)
        solver.step(warmup)
        ps.set_option(_lib.OPT_TIMING, 1)
        ps._call("sph_reset_timings")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        solver.step(steps)
        ps.sync()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        tm = _lib.SphTimings()
        ps._call("sph_get_timings", tm)
        return dt, tm

    solver.initialize()
    ps.set_option(_lib.OPT_KERNEL_VARIANT, args.variant)
    if args.settle > 0:
        solver.step(args.settle)
        ps.sync()
    if args.ablate_mask:
        solver.step(args.warmup)
        solver.dt[None] = 0.0
        ps.set_option(_lib.OPT_DEBUG_ABLATE, args.ablate_mask)
    if args.sweep:
        for impl, shape, fused in [(0, 0, 1), (1, 0, 1), (1, 1, 1), (1, 0, 0)]:
            dt, tm = run(impl, shape, fused, max(args.steps // 4, 10), 5)
            k = max(tm.steps, 1)
            print(f"[sweep] impl={impl} shape={shape} fused={fused}: {dt / max(args.steps // 4, 10) * 1e3:.3f} ms/step "
                  f"sort={tm.sort_ms / k:.3f} neigh={tm.neighbour_ms / k:.3f} force={tm.force_ms / k:.3f} "
                  f"integ={tm.integrate_ms / k:.3f}", file=sys.stderr, flush=True)

    if args.ablate:
        # same particle state for every variant: sweeps only (no advect), positions frozen by dt = 0
        solver.step(args.warmup)
        solver.dt[None] = 0.0
        for mask, what in [(0, "full"), (1, "no phase 2"), (2, "no list write-out (force sweep reads stale lists)"),
                           (64, "force: one list-entry load per 4 pairs"), (128, "force: no neighbour gather"),
                           (192, "force: neither"), (16, "filter only, no hit emitted"), (32, "no pair term in the emission loop"), (34, "emission: bit loop only"),
                           (4, "no phase 1"), (7, "staging + target setup only")]:
            ps.set_option(_lib.OPT_DEBUG_ABLATE, mask)
            dt, tm = run(args.gather_impl, args.brick_shape, 1, 20, 2)
            kk = max(tm.steps, 1)
            print(f"[ablate {mask}] {what:32s} neigh={tm.neighbour_ms / kk:.3f} force={tm.force_ms / kk:.3f}",
                  file=sys.stderr, flush=True)
        ps.set_option(_lib.OPT_DEBUG_ABLATE, 0)
        solver.dt[None] = CFG["timeStepSize"]
    if args.solver == "dfsph":
        it0 = solver.stats()
        dt, tm = run(args.gather_impl, args.brick_shape, 1, args.steps, args.warmup)
        it1 = solver.stats()
        k = max(int(tm.steps), 1)
        iv = (it1["total_iterations_v"] - it0["total_iterations_v"]) / (args.warmup + args.steps)
        ip = (it1["total_iterations"] - it0["total_iterations"]) / (args.warmup + args.steps)
        sweeps = 2 + (1 + 2 * iv) + 1 + (1 + 2 * ip)     # density, factor | divergence solve | forces | pressure solve
        line = {
            "metric": "DFSPH steps/sec at 1.74 M particles (supplementary; BASELINE.json's metric is the WCSPH step)",
            "value": round(args.steps / dt * N / REF_PARTICLES, 3), "unit": "steps/s", "n_gpus": 1,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "solver": "dfsph", "particles": N, "cells": G, "dt": DFSPH_DT,
                       "gather_impl": args.gather_impl, "parallelism": "1 GPU"},
            "breakdown_ms": {"sort": round(tm.sort_ms / k, 4), "neighbour": round(tm.neighbour_ms / k, 4),
                             "force": round(tm.force_ms / k, 4), "integrate": round(tm.integrate_ms / k, 4),
                             "sum_of_phases": round(tm.total_ms / k, 4)},
            "dfsph": {"divergence_iterations_per_step": round(iv, 2), "pressure_iterations_per_step": round(ip, 2),
                      "neighbour_sweeps_per_step": round(sweeps, 2),
                      "ms_per_sweep": round((tm.neighbour_ms + tm.force_ms) / k / sweeps, 4),
                      "simulated_time_per_wall_second": round(args.steps / dt * DFSPH_DT, 3)},
            "roofline": None,
        }
        ps.close()
        line["cpu_baseline"] = cpu_baseline(sd, args.cpu_steps) if args.cpu_steps > 0 else None
        print(json.dumps(line), flush=True)
        return
    dt, tm = run(args.gather_impl, args.brick_shape, args.fused, args.steps, args.warmup)
    k = max(int(tm.steps), 1)
    ms_per_step = dt / args.steps * 1e3
    steps_per_s = args.steps / dt
    value = steps_per_s * N / REF_PARTICLES
    force_ms = tm.force_ms / k
    neigh_ms = tm.neighbour_ms / k
    # Per-launch algorithmic bytes: density+EOS sweep 32 N + 4 G, fused force sweep 60 N + 4 G.  With no
    # dynamic rigid body each phase is exactly one launch, so the HIP-event phase time is that kernel's duration.
    kernels = {
        "k_gather_brick<GM_DENSITY_EOS>": (32.0 * N + 4.0 * G, neigh_ms),
        "k_gather_brick<GM_FORCE_FUSED>": (60.0 * N + 4.0 * G, force_ms),
    }
    one_gather = ps.get_option(_lib.OPT_UNIFORM_FLUID_STATE) == 1 and args.fused == 1
    if one_gather:      # the force sweep that actually ran (SPH_OPT_UNIFORM_FLUID: all fluid masses equal)
        kernels["k_gather_brick<GM_FORCE_FUSED_U>"] = kernels.pop("k_gather_brick<GM_FORCE_FUSED>")
    if not args.gather_impl:
        kernels = {k_.replace("brick", "simple"): v for k_, v in kernels.items()}
    dominant = max(kernels, key=lambda k_: kernels[k_][1])
    alg_bytes, dom_ms = kernels[dominant]
    achieved = alg_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    # Counter-derived figures (HBM bytes, VALU instruction counts) come from the committed rocprofv3 PMC passes
    # (profiles/pmc_traffic.json, tools/gpu_pmc.sh + tools/refresh_pmc.py).  They are quoted ONLY when that file was
    # measured on the kernel sources this library was built from (same fingerprint), on this workload and variant;
    # otherwise they are null -- never a number from another revision next to a live launch time.
    traffic = valu_busy = None
    pmc_k = {}
    pmc_note = "no PMC file for this revision"
    try:
        from sph_taichi_amd import build as _build
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if pm.get("kernel_fingerprint") != _build._fingerprint():
            pmc_note = "profiles/pmc_traffic.json was measured on other kernel sources (fingerprint differs): not quoted"
        elif not (pm["workload"] == args.workload and args.gather_impl == 1 and args.brick_shape == 0 and args.fused == 1
                  and args.variant == -1 and args.settle == 0):
            pmc_note = "profiles/pmc_traffic.json covers the default line only (workload / variant / state differ): not quoted"
        else:
            pmc_k = pm["kernels"]
            kk = pmc_k[dominant]
            traffic = kk["fetch_kb"] * 1024 * 2 + kk["write_kb"] * 1024
            valu_busy = kk.get("valu_busy_frac")
            pmc_note = f"profiles/pmc_traffic.json ({pm.get('source')}), same kernel fingerprint"
    except Exception as e:
        pmc_note = f"PMC file unusable ({type(e).__name__})"
    # VALU roofline of the same kernel.  Peak issue rate: 1024 SIMD-32 x 2.4 GHz / 2 cycles per wave64 instruction
    # (MI355X_MICROARCH.md; measured here 0.85-0.9 G wave-instructions/s per SIMD at the clock the chip sustains,
    # profiles/r02a_ubench_valu_table1.txt -- and half / a quarter of that for the 4- and 8-cycle opcode classes).
    # Useful work: SURVEY 8d's ~2.3 kFLOP (density) / ~4.6 kFLOP (force) per particle against 157.3 TFLOP/s.
    VALU_PEAK_GINST = 1024 * 2.4 / 2.0
    flop_per_particle = 2300.0 if "DENSITY" in dominant else 4600.0
    roofline_valu = {"kernel": dominant, "unit": "G wave-instructions/s", "peak": VALU_PEAK_GINST,
                     "peak_measured_full_rate_ops": round(1024 * 0.875, 1),
                     "useful_tflops": round(flop_per_particle * N / (dom_ms * 1e-3) / 1e12, 2) if dom_ms > 0 else None,
                     "peak_tflops": 157.3, "achieved": None, "frac": None, "valu_wave_insts_per_launch": None,
                     "insts_per_particle": None, "source": pmc_note}
    if dom_ms > 0:
        roofline_valu["useful_frac_of_fp32_peak"] = round(roofline_valu["useful_tflops"] / 157.3, 4)
    if pmc_k.get(dominant, {}).get("valu_wave_insts") and dom_ms > 0:
        wi = pmc_k[dominant]["valu_wave_insts"]
        roofline_valu.update(valu_wave_insts_per_launch=wi, achieved=round(wi / (dom_ms * 1e-3) / 1e9, 1),
                             frac=round(wi / (dom_ms * 1e-3) / 1e9 / VALU_PEAK_GINST, 4),
                             insts_per_particle=round(wi * 64 / N, 1))
    line = {
        "metric": "WCSPH steps/sec at 1.74 M particles (+ ms/step breakdown sort/neighbour/force)",
        "value": round(value, 3), "unit": "steps/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": args.workload, "particles": N, "cells": G, "dt": CFG["timeStepSize"],
                   "gather_impl": args.gather_impl, "brick_shape": args.brick_shape, "fused": args.fused,
                   "kernel_variant": ps.get_option(_lib.OPT_KERNEL_VARIANT), "parallelism": "1 GPU"},
        "breakdown_ms": {"sort": round(tm.sort_ms / k, 4), "neighbour": round(neigh_ms, 4),
                         "force": round(force_ms, 4), "integrate": round(tm.integrate_ms / k, 4),
                         "sum_of_phases": round(tm.total_ms / k, 4)},
        "steps_per_s_job": round(steps_per_s, 3),
        "roofline": {"kernel": dominant, "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                     "alg_bytes_per_launch": alg_bytes, "avg_launch_ms": round(dom_ms, 4),
                     "valu_busy_frac": valu_busy, "counters": pmc_note,
                     "note": "fraction of the HBM roof on ALGORITHMIC bytes, as the contract asks; the sweep itself is "
                             "bound by VALU issue and the LDS / vector-memory pipes (roofline_valu, DESIGN.md section 4), "
                             "so this fraction measures how far the sweep is from a pure streaming pass, not HBM "
                             "saturation; traffic = rocprofv3 FETCH_SIZE*2 + WRITE_SIZE per launch"},
        "roofline_valu": roofline_valu,
        "roofline_kernels": {k_: {"alg_bytes": v[0], "avg_launch_ms": round(v[1], 4),
                                  "achieved_GBs": round(v[0] / (v[1] * 1e-3) / 1e9, 2) if v[1] > 0 else 0.0}
                             for k_, v in kernels.items()},
    }
    # secondary: whole-step algorithmic bytes (360 N + 20 G) against the same peak
    step_bytes = 360.0 * N + 20.0 * G
    line["roofline_step"] = {"alg_bytes": step_bytes, "achieved_GBs": round(step_bytes / (ms_per_step * 1e-3) / 1e9, 2),
                             "frac": round(step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
    st = _lib.SphStats()
    ps._call("sph_get_stats", st)
    line["config"]["settle_steps"] = args.settle
    line["neighbourhood"] = {
        "mean_list_entries": round(st.list_entries / max(st.targets - st.list_overflow_targets - st.lds_overflow_targets, 1), 2),
        "max_list_entries": st.max_list, "list_overflow_targets": st.list_overflow_targets,
        "lds_overflow_targets": st.lds_overflow_targets, "max_cell_occupancy": st.max_cell_occupancy,
        "mean_cell_occupancy": round(N / max(st.nonempty_cells, 1), 2),
        "note": "last density sweep of the timed region (sph_get_stats); list entries = superset filter incl. self"}
    ps.close()
    if args.cpu_steps > 0:
        line["cpu_baseline"] = cpu_baseline(sd, args.cpu_steps)
    else:
        line["cpu_baseline"] = None
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
