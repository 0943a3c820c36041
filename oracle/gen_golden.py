#!/usr/bin/env python
"""Generate tests/golden/ref_*.npz by EXECUTING THE REFERENCE'S OWN SOURCE
(/root/reference/{particle_system,sph_base,WCSPH,config_builder}.py, unmodified)
under the serial pure-Python `taichi` stand-in of oracle/taichi_shim/.

Runs only in the build container (needs /root/reference); the .npz files it
writes are committed and are what tests/test_golden.py checks the C oracle (CPU)
and the HIP path (GPU) against.

Caveats recorded with each fixture (see oracle/taichi_shim/taichi/__init__.py):
loops run serially in index order (=> the stable counting-sort order); Taichi's
compiler-level rounding (fast-math, pow lowering, constant folding) and
ti.polar_decompose are not reproduced; Python-scope np.float64 attributes that
Taichi would bake as f32 constants (domain_size) are cast to f32 up front.
"""
from __future__ import annotations

import copy
import json
import os
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("SPH_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(HERE, "taichi_shim"))
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

FIELDS = ["object_id", "x", "x_0", "v", "acceleration", "m_V", "m", "density", "pressure", "material", "color",
          "is_dynamic", "grid_ids", "grid_particles_num"]


def snapshot(ps):
    return {f: getattr(ps, f).to_numpy().copy() for f in FIELDS}


def run_reference(scene_dict, n_steps, per_kernel_first_step=True):
    import taichi as ti                      # the shim
    from config_builder import SimConfig     # the reference's
    from particle_system import ParticleSystem
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as fh:
        json.dump(scene_dict, fh)
        path = fh.name
    sys.stdout = open(os.devnull, "w")       # the reference prints the whole config
    try:
        cfg = SimConfig(scene_file_path=path)
        ps = ParticleSystem(cfg, GGUI=False)
    finally:
        sys.stdout = sys.__stdout__
    os.unlink(path)
    # Python-scope f64 values that Taichi bakes into kernels as f32 constants
    ps.domain_size = ps.domain_size.astype(np.float32)
    solver = ps.build_solver()
    out = {"initial": snapshot(ps)}
    solver.initialize()
    out["initialized"] = snapshot(ps)
    for s in range(n_steps):
        if s == 0 and per_kernel_first_step:
            # SPHBase.step() unrolled (sph_base.py:263-271) to capture every kernel's output
            ps.initialize_particle_system(); out["k_sort"] = snapshot(ps)
            solver.compute_moving_boundary_volume(); out["k_bvol"] = snapshot(ps)
            solver.compute_densities(); out["k_density"] = snapshot(ps)
            solver.compute_non_pressure_forces(); out["k_nonpressure"] = snapshot(ps)
            solver.compute_pressure_forces(); out["k_pressure"] = snapshot(ps)
            solver.advect(); out["k_advect"] = snapshot(ps)
            solver.solve_rigid_body()
            solver.enforce_boundary_3D(ps.material_fluid)
        else:
            solver.step()
        out[f"step{s + 1}"] = snapshot(ps)
    assert ti.oob_reads == 0, f"{ti.oob_reads} out-of-range field reads: the scene hits undefined behaviour"
    return out


def flatten(d):
    return {f"{stage}/{field}": arr for stage, fields in d.items() for field, arr in fields.items()}


def main():
    import scenes
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    jobs = {
        # fluid block thrown at the floor/wall corner: walls, surface tension, viscosity, EOS
        "ref_fluid_wall": (scenes.fluid_only(counts=(6, 7, 5), start=(0.05, 0.05, 0.05), velocity=(-6.0, -8.0, -4.0),
                                             domain_end=(0.6, 0.6, 0.5)), 8),
        # fluid on a static slab with a dynamic block dropping in: K4 (both), Akinci pressure, coupling scatter
        "ref_fluid_rigid": (scenes.fluid_with_rigid_blocks(fluid_counts=(8, 8, 6), static_counts=(12, 2, 10),
                                                           dyn_counts=(4, 4, 4)), 8),
    }
    for name, (sd, steps) in jobs.items():
        t0 = time.time()
        res = run_reference(copy.deepcopy(sd), steps)
        path = os.path.join(ROOT, "tests", "golden", name + ".npz")
        np.savez_compressed(path, scene=json.dumps(sd), steps=steps, **flatten(res))
        n = res["initial"]["x"].shape[0]
        print(f"{name}: {n} particles, {steps} steps, {time.time() - t0:.1f} s -> {path}")


if __name__ == "__main__":
    main()
