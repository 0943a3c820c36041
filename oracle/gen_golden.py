#!/usr/bin/env python
"""Generate tests/golden/ref_*.npz by EXECUTING THE REFERENCE'S OWN SOURCE
(/root/reference/{particle_system,sph_base,WCSPH,DFSPH,config_builder}.py, unmodified)
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
import io
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
    out = {f: getattr(ps, f).to_numpy().copy() for f in FIELDS}
    for f in ("dfsph_factor", "density_adv"):       # simulationMethod 4 only (particle_system.py:115-117)
        if hasattr(ps, f):
            out[f] = getattr(ps, f).to_numpy().copy()
    return out


def run_reference(scene_dict, n_steps, per_kernel_first_step=True, keep_steps=None, light_fields=None, state_fn=None):
    """`state_fn(initial arrays) -> {"x", "v"}`: the start state written into the reference's own fields after its
    constructor and BEFORE solver.initialize() (stage "state" of the fixture; the tests give the oracle and the HIP path the
    same arrays) -- for starts the scene file cannot express (a body turned against its rest shape)."""
    import taichi as ti                      # the shim
    from config_builder import SimConfig     # the reference's
    from particle_system import ParticleSystem
    os.chdir(ROOT)      # geometryFile paths in the fixtures are relative to the repo root
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
    if state_fn is not None:
        st = state_fn(out["initial"])
        ps.x.from_numpy(np.ascontiguousarray(st["x"], dtype=np.float32))
        ps.v.from_numpy(np.ascontiguousarray(st["v"], dtype=np.float32))
        out["state"] = {"x": ps.x.to_numpy().copy(), "v": ps.v.to_numpy().copy()}
    solver.initialize()
    out["initialized"] = snapshot(ps)
    dfsph = cfg.get_cfg("simulationMethod") == 4
    iters = []
    for s in range(n_steps):
        if dfsph:
            sys.stdout = buf = io.StringIO()     # "DFSPH - iteration V: k Avg density err: e" (DFSPH.py:258, 353)
            try:
                if s == 0 and per_kernel_first_step:
                    # SPHBase.step() / DFSPHSolver.substep() unrolled (sph_base.py:263-271, DFSPH.py:400-408)
                    ps.initialize_particle_system(); out["k_sort"] = snapshot(ps)
                    solver.compute_moving_boundary_volume(); out["k_bvol"] = snapshot(ps)
                    solver.compute_densities(); out["k_density"] = snapshot(ps)
                    solver.compute_DFSPH_factor(); out["k_factor"] = snapshot(ps)
                    solver.compute_density_change(); out["k_density_change"] = snapshot(ps)
                    solver.divergence_solve(); out["k_divergence"] = snapshot(ps)
                    solver.compute_non_pressure_forces(); out["k_nonpressure"] = snapshot(ps)
                    solver.predict_velocity(); out["k_predict"] = snapshot(ps)
                    solver.compute_density_adv(); out["k_density_adv"] = snapshot(ps)
                    solver.pressure_solve(); out["k_pressure_solve"] = snapshot(ps)
                    solver.advect(); out["k_advect"] = snapshot(ps)
                    solver.solve_rigid_body()
                    solver.enforce_boundary_3D(ps.material_fluid)
                else:
                    solver.step()
            finally:
                sys.stdout = sys.__stdout__
            nums = [int(l.split(":")[1].split()[0]) for l in buf.getvalue().splitlines() if l.startswith("DFSPH")]
            iters.append(nums)
        elif s == 0 and per_kernel_first_step:
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
        if keep_steps is None or (s + 1) in keep_steps:
            snap = snapshot(ps)
            if light_fields is not None and (s + 1) != n_steps:
                snap = {f: snap[f] for f in light_fields}      # intermediate stages of the big fixtures: the trajectory only
            out[f"step{s + 1}"] = snap
    assert ti.oob_reads == 0, f"{ti.oob_reads} out-of-range field reads: the scene hits undefined behaviour"
    if dfsph:
        out["solver"] = {"iterations": np.array(iters, dtype=np.int32)}   # [step][divergence, pressure]
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
    # shape-matched RigidBodies (sph_base.py:87-89, 182-260) through the reference's own load_rigid_body; the mesh
    # is a cube OBJ written next to the fixture (the trimesh stand-in voxelises it, see taichi_shim/trimesh)
    obj = os.path.join(ROOT, "tests", "golden", "cube_0p1.obj")
    scenes.write_cube_obj(obj, (0.0, 0.0, 0.0), 0.1)
    # (the voxel points sit on the pitch lattice, i.e. half of them exactly on cell boundaries: the bodies get
    # sideways velocity so that later hashes are not decided by the last bit of the shape-matching sums)
    d3 = scenes.fluid_with_rigid_bodies("tests/golden/cube_0p1.obj",
                                        body_velocities=((0.3, -2.0, 0.2), (-0.25, -2.0, 0.35)))
    d3["FluidBlocks"][0]["end"] = scenes.lattice_end((0.1, 0.1, 0.1), (10, 8, 8))
    jobs["ref_rigid_bodies"] = (d3, 8)
    d4 = copy.deepcopy(d3)
    d4["Configuration"]["simulationMethod"] = 4
    d4["Configuration"]["timeStepSize"] = 0.002
    jobs["ref_dfsph_rigid_bodies"] = (d4, 6)
    # DFSPH (simulationMethod 4, DFSPH.py): fluid hitting a wall corner; fluid on a static slab with a dynamic block
    d1 = scenes.fluid_only(counts=(8, 9, 6), start=(0.05, 0.05, 0.05), velocity=(-4.0, -6.0, -3.0),
                           domain_end=(0.6, 0.6, 0.5))      # both solvers iterate (up to 3 / 2 extra iterations)
    d2 = scenes.fluid_with_rigid_blocks(fluid_counts=(8, 8, 6), static_counts=(12, 2, 10), dyn_counts=(4, 4, 4))
    d2["FluidBlocks"][0]["velocity"] = [0.0, 0.0, 0.0]
    d2["RigidBlocks"][1]["velocity"] = [0.0, -8.0, 0.0]     # the cube hits the fluid: coupling reaction of DFSPH.py:394
    for d in (d1, d2):
        d["Configuration"]["simulationMethod"] = 4
        d["Configuration"]["timeStepSize"] = 0.002
    jobs["ref_dfsph_wall"] = (d1, 8)
    jobs["ref_dfsph_rigid"] = (d2, 8)
    # two fluids of different density (particle_system.py:231 m = m_V0 * density: different particle masses) running
    # into each other above a static slab: the general (non-uniform-mass) force sweep, WCSPH.py:47-112
    d5 = scenes.fluid_with_rigid_blocks(fluid_counts=(6, 6, 5), static_counts=(16, 2, 9), dyn_counts=(3, 3, 3))
    f0 = d5["FluidBlocks"][0]
    f0["velocity"] = [1.5, -0.5, 0.0]
    s1 = (f0["start"][0] + 6 * 0.02 + 0.03, f0["start"][1], f0["start"][2])
    f1 = copy.deepcopy(f0)
    f1.update(objectId=3, start=list(s1), end=scenes.lattice_end(s1, (5, 6, 5)), density=600.0, velocity=[-1.5, -0.5, 0.0],
              color=[200, 100, 50])
    d5["FluidBlocks"].append(f1)
    jobs["ref_two_fluids"] = (d5, 8)
    # The HIGH faces (VERDICT r03 "missing" #5).  The reference flattens neighbour cells without a bounds check
    # (particle_system.py:381-383): for a particle in the LAST y layer, cy + 1 = n_y aliases into cell (cx + 1, 0, cz), in
    # the last z layer cz + 1 = n_z into (cx, cy + 1, 0) -- far-away real cells whose particles then fail the distance
    # test -- where the oracle and the HIP path skip the cell.  A wall-clamped particle sits exactly in the last layer
    # ((size - pad) / h = n - 1), so a block thrown at the (+y, +z) edge keeps particles there for many steps.  The +x
    # face is left alone: there the reference reads out of bounds (`ti.oob_reads` asserts that no run does).
    d7 = scenes.fluid_only(counts=(6, 7, 5), start=(0.2, 0.436, 0.476), velocity=(1.0, 8.0, 6.0), domain_end=(0.6, 0.6, 0.6))
    jobs["ref_high_faces_fluid"] = (d7, 14)
    # ... and a shape-matched dynamic body (its particles are clamped by enforce_boundary_3D(solid), sph_base.py:260) next
    # to the fluid: solids in the last layers too, boundary volumes and the coupling scatter across the aliased lookups
    d8 = scenes.fluid_only(counts=(8, 6, 6), start=(0.15, 0.455, 0.455), velocity=(0.0, 7.0, 5.0), domain_end=(0.6, 0.6, 0.6))
    d8["RigidBodies"] = [{"objectId": 1, "geometryFile": "tests/golden/cube_0p1.obj", "translation": [0.315, 0.45, 0.45],
                          "rotationAxis": [0, 0, 1], "rotationAngle": 0, "scale": [1, 1, 1], "velocity": [0.2, 7.0, 5.0],
                          "density": 600.0, "color": [255, 255, 255], "isDynamic": True}]
    jobs["ref_high_faces_rigid"] = (d8, 14)
    d9 = copy.deepcopy(d7)                                   # ... the high faces under DFSPH (every DFSPH sweep walks the same lookups)
    d9["Configuration"]["simulationMethod"] = 4
    d9["Configuration"]["timeStepSize"] = 0.001
    jobs["ref_dfsph_high_faces"] = (d9, 10)
    # (The cell-0 quirk -- particle_system.py:384 starts the range of cell c at prefix[max(0, c - 1)], so flat cell 0's own
    # particles are in nobody's neighbourhood -- cannot be pinned by execution: a particle in a cell with coordinate 0 looks up
    # cell coordinate -1, and for the corner cell the flat index goes negative: the shim counted 992 out-of-range reads on a
    # scene with a static slab around the origin.  It is undefined behaviour in the reference, like the +x face.)
    d6 = copy.deepcopy(d5)                                   # ... and under DFSPH (the general *_ITER sweeps)
    d6["Configuration"]["simulationMethod"] = 4
    d6["Configuration"]["timeStepSize"] = 0.002
    jobs["ref_dfsph_two_fluids"] = (d6, 6)
    # Degenerate polar decompositions (VERDICT r04 "missing" #6; sph_base.py:200-222): a one-layer body (rank-2 A), a body
    # started turned by 179 degrees and one started mirrored (det A < 0), each touching fluid.  Their start state is not
    # expressible in a scene file: `state_fn` (tests/scenes.py::degenerate_bodies) moves the bodies' particles about their
    # centres of mass after the reference's constructor, x_0 keeps the rest shape.
    state_fns = {}
    for which in ("flat", "turned"):
        sdd, fn = scenes.degenerate_bodies("tests/golden/cube_0p1.obj", which)
        jobs[f"ref_rigid_{which}_body"] = (sdd, 10)
        state_fns[f"ref_rigid_{which}_body"] = fn
    # The BIG family (VERDICT r02 "weak" #1: what pins the oracle was 200-900 particles over 6-8 steps): >= 10 k
    # particles, 50 steps, through wall impact / the block's plunge.  Hours of serial Python each, so they run only
    # when named on the command line; stages kept: initial, initialized, steps 1 / 10 / 25 (x, v, density, pressure,
    # grid_ids) and the complete state after step 50.
    big = {
        "ref_big_fluid_wall": (scenes.fluid_only(counts=(32, 20, 16), start=(0.08, 0.08, 0.08), velocity=(-6.0, -8.0, -4.0),
                                                 domain_end=(1.0, 0.8, 0.6)), 50),
        "ref_big_fluid_rigid": (scenes.fluid_with_rigid_blocks(fluid_counts=(24, 22, 16), static_counts=(30, 2, 22),
                                                               dyn_counts=(8, 8, 8)), 50),
    }
    only = sys.argv[1:]
    for name in only:
        if name in big:
            jobs[name] = big[name]
    for name, (sd, steps) in jobs.items():
        if only and name not in only:
            continue
        t0 = time.time()
        if name in big:
            res = run_reference(copy.deepcopy(sd), steps, per_kernel_first_step=False, keep_steps={1, 10, 25, steps},
                                light_fields=("x", "v", "density", "pressure", "grid_ids"))
        else:
            res = run_reference(copy.deepcopy(sd), steps, state_fn=state_fns.get(name))
        path = os.path.join(ROOT, "tests", "golden", name + ".npz")
        np.savez_compressed(path, scene=json.dumps(sd), steps=steps, **flatten(res))
        n = res["initial"]["x"].shape[0]
        print(f"{name}: {n} particles, {steps} steps, {time.time() - t0:.1f} s -> {path}")


if __name__ == "__main__":
    main()
