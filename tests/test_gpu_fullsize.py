"""Full-size cases of BASELINE.md section 3 on the GPU.

C2  dragon_bath.json equivalent (423,500 fluid + 18,496 static dragon voxels): HIP vs oracle.
C3  armadillo_bath_dynamic.json equivalent (1,723,968 fluid + 3 dynamic bodies, stand-in mesh):
    HIP vs oracle through first contact (two-way coupling + shape matching at scale).
C3' uniform 1,747,584-particle box: size-independent properties of the sort and the sweeps.
C1  64^3 dam-break (262,144 particles): HIP vs oracle from the first collapse and from a developed flow.
The oracle runs multi-threaded here (OpenMP), so only tolerance comparisons are made against it.

The deep cases (VERDICT r01 "Next round" #1) write their error-vs-N curves to gpurun_out/parity_curves.json
(copied to profiles/ by the round's GPU script).
"""
import copy
import json
import os

import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

from sph_taichi_amd.workloads import dragon_bath_scene, armadillo_equiv_scene, DEMO_CFG      # noqa: E402

CFG = DEMO_CFG


def _threads():
    from oracle.oracle import max_threads
    return max_threads()


def test_c2_dragon_bath_equivalent():
    sd = dragon_bath_scene()
    cfg, sc = scenes.build(sd)
    assert sc.fluid_particle_num == 55 * 140 * 55 and sc.solid_particle_num == 18496
    o = scenes.make_oracle(cfg, sc, omp_threads=_threads())
    ps, solver = scenes.make_ps(sd)
    o.initialize(); solver.initialize()
    assert np.array_equal(ps.grid_ids.to_numpy(), o["grid_ids"])
    assert np.array_equal(ps.pid.to_numpy(), o["pid"])
    assert np.array_equal(ps.grid_particles_num.to_numpy(), o["grid_particles_num"])
    solid = o["material"] == 0
    assert np.allclose(ps.m_V.to_numpy()[solid], o["m_V"][solid], rtol=2e-5)       # static boundary volumes
    n = 10
    o.step(n); solver.step(n)
    err = scenes.rel_l2(scenes.ps_by_pid(ps, "x"), o.by_pid("x"))
    assert err <= 1e-4, f"C2 position rel-L2 after {n} steps: {err:.3e}"
    assert np.allclose(scenes.ps_by_pid(ps, "density"), o.by_pid("density"), rtol=5e-5)
    ps.close()


def test_c3_armadillo_equivalent_dynamic_bodies():
    sd = armadillo_equiv_scene()
    cfg, sc = scenes.build(sd)
    assert sc.fluid_particle_num == 246 * 73 * 96 and sorted(sc.dynamic_rigid_ids) == [1, 2, 3]
    # The reference accumulates the shape-matching sums (5,917 terms per body) in f32; a serial run -- the
    # oracle's default -- carries ~1e-5 relative summation error per step that drifts the bodies.  The HIP path
    # reduces in f64.  (1) against the oracle with f64 accumulators the bodies agree tightly; (2) against the
    # f32-serial oracle the whole system is still inside the 1e-4 budget.
    o = scenes.make_oracle(cfg, sc, omp_threads=_threads(), rigid_sums_f64=True)
    o32 = scenes.make_oracle(cfg, sc, omp_threads=_threads())
    ps, solver = scenes.make_ps(sd)
    o.initialize(); o32.initialize(); solver.initialize()
    assert np.array_equal(ps.pid.to_numpy(), o["pid"])
    assert np.allclose(ps.rigid_rest_cm.to_numpy()[1:4], o["rigid_rest_cm"][1:4], rtol=2e-6)
    n = 30
    o.step(n); o32.step(n); solver.step(n)
    x, x_ref = scenes.ps_by_pid(ps, "x"), o.by_pid("x")
    err = scenes.rel_l2(x, x_ref)
    assert err <= 1e-4, f"C3 position rel-L2 after {n} steps: {err:.3e}"
    rigid = sc.arrays["material"] == 0
    assert scenes.rel_l2(x[rigid], x_ref[rigid]) <= 2e-5            # diagnostic: the oracle with f64 accumulators
    assert scenes.rel_l2(x, o32.by_pid("x")) <= 1e-4
    # The bodies alone against the reference-order f32 oracle (the reference's own arithmetic on a serial backend): its
    # f32 running sums over 5,917 terms per body carry ~1e-5 of summation error per step, the HIP path's exact sums
    # none, so THIS difference is the reference's rounding, not ours -- measured 1.4e-4 after 30 steps (r03, value in
    # gpurun_out/parity_curves.json) where the f64-accumulating oracle above agrees to 2e-5.  Stated bound: 2e-4.
    e32 = scenes.rel_l2(x[rigid], o32.by_pid("x")[rigid])
    # The triangle that attributes it (r04): the two ORACLES -- same formulas, same order, f64 against f32 accumulators --
    # are as far from each other on the bodies as the HIP path is from the f32 one, while the HIP path sits on the f64 one.
    # No evaluation with accurate sums can be within 1e-4 of the serial-f32 result; only one that reproduces its rounding.
    e_oracles = scenes.rel_l2(x_ref[rigid], o32.by_pid("x")[rigid])
    e64 = scenes.rel_l2(x[rigid], x_ref[rigid])
    _record_curve("c3_bodies_vs_f32_reference_order_oracle", [(n, e32)], bound=2e-4,
                  hip_vs_f64_sums_oracle=float(e64), f64_sums_oracle_vs_f32_sums_oracle=float(e_oracles))
    assert e32 <= 2e-4, f"C3 bodies vs the f32 reference-order oracle after {n} steps: {e32:.3e}"
    # (the triangle's sides are recorded above, not asserted against each other: ADVICE r04 -- a relation between measured
    # rounding errors fails on any legitimate accuracy improvement; the absolute bounds are what is tested)
    v = scenes.ps_by_pid(ps, "v")
    free_fall = -5.0 - 9.81 * n * 4e-4
    light = sc.arrays["object_id"] == 3                     # density 300: decelerated hard by the fluid
    assert v[light, 1].mean() > free_fall + 0.05, "the bodies never touched the fluid: coupling not exercised"
    assert scenes.rel_l2(v[rigid], o.by_pid("v")[rigid]) <= 2e-3
    ps.close()


def test_c3p_properties_at_full_size():
    sd = scenes.fluid_only(counts=(246, 74, 96), start=(0.04, 0.04, 0.04), velocity=(0.3, -0.5, 0.2),
                           domain_end=(5.0, 3.0, 2.0))
    ps, solver = scenes.make_ps(sd)
    N = ps.particle_max_num
    assert N == 1_747_584
    solver.initialize()
    solver.step(3)
    ps.initialize_particle_system()
    gi = ps.grid_ids.to_numpy()
    prefix = ps.grid_particles_num.to_numpy()
    pid = ps.pid.to_numpy()
    assert np.all(np.diff(gi) >= 0), "sortedness"
    assert prefix[-1] == N and np.all(np.diff(prefix) >= 0)
    assert np.array_equal(np.bincount(gi, minlength=prefix.size).cumsum(), prefix), "prefix == cumsum(histogram)"
    assert np.array_equal(np.sort(pid), np.arange(N)), "the permutation is a bijection"
    x = ps.x.to_numpy()
    cell = (x / np.float32(0.04)).astype(np.int64)
    assert np.array_equal((cell[:, 0] * 75 + cell[:, 1]) * 50 + cell[:, 2], gi), "key == hash(x) after the scatter"
    ps.initialize_particle_system()                         # idempotence: sorting sorted data is the identity
    assert np.array_equal(ps.pid.to_numpy(), pid)
    # pure-fluid pressure forces conserve momentum (symmetric formula, equal m_V)
    solver.compute_densities()
    ps.density.from_numpy(ps.density.to_numpy() * np.float32(1.3))
    ps.acceleration.from_numpy(np.zeros((N, 3), np.float32))
    solver.compute_pressure_forces()
    a = ps.acceleration.to_numpy().astype(np.float64)
    assert np.abs(a.sum(axis=0)).max() < 2e-4 * np.abs(a).sum(axis=0).max()
    # every gather implementation produces the same densities at full size
    solver.compute_densities()
    rho1 = ps.density.to_numpy()
    from sph_taichi_amd import _lib
    ps.set_option(_lib.OPT_GATHER_IMPL, 0)
    solver.compute_densities()
    assert np.allclose(ps.density.to_numpy(), rho1, rtol=3e-6)
    ps.close()


def test_dfsph_dragon_bath_equivalent():
    """dragon_bath_dfsph.json equivalent (dt = 4e-3, 442 k particles): HIP DFSPH vs the oracle (OpenMP) over the first
    steps -- positions to the parity tolerance, solver iteration counts within one of each other."""
    sd = dragon_bath_scene()
    sd["Configuration"]["simulationMethod"] = 4
    sd["Configuration"]["timeStepSize"] = 0.004
    cfg, sc = scenes.build(sd)
    o = scenes.make_oracle(cfg, sc, omp_threads=_threads())
    ps, solver = scenes.make_ps(sd)
    o.initialize(); solver.initialize()
    steps = 3
    its = []
    for _ in range(steps):
        o.step(1)
        its.append((o.s.last_iterations_v, o.s.last_iterations))
    solver.step(steps)
    st = solver.stats()
    assert abs(st["total_iterations_v"] - sum(a + 1 for a, _ in its)) <= 1
    assert abs(st["total_iterations"] - sum(b + 1 for _, b in its)) <= 1
    assert scenes.rel_l2(scenes.ps_by_pid(ps, "x"), o.by_pid("x")) <= 1e-4
    assert np.array_equal(np.sort(ps.pid.to_numpy()), np.arange(sc.particle_max_num))
    ps.close()


def test_high_fluid_wcsph_scene():
    """data/scenes/high_fluid_wcsph.json as it is (VERDICT r02 "missing" #4): a 0.6 x 5.4 x 0.6 column of 30 x 270 x 30 =
    243,000 particles in a (2, 6, 2) tank -- 50 x 150 x 50 cells, tall and narrow: y is the middle axis of the flatten
    order, so the brick list's column groups are few and their z extent short.  100 steps against the oracle."""
    sd = {"Configuration": dict(scenes.BASE_CFG, domainEnd=[2.0, 6.0, 2.0]),
          "FluidBlocks": [{"objectId": 0, "start": [0.0, 0.0, 0.0], "end": [0.6, 5.4, 0.6], "translation": [0.1, 0.1, 0.1],
                           "scale": [1, 1, 1], "velocity": [0.0, 0.0, 0.0], "density": 1000.0, "color": [50, 100, 200]}]}
    cfg, sc = scenes.build(sd)
    assert sc.particle_max_num == 30 * 270 * 30 and tuple(sc.geom.grid_num) == (50, 150, 50)
    o = scenes.make_oracle(cfg, sc, omp_threads=_threads())
    ps, solver = scenes.make_ps(sd)
    o.initialize(); solver.initialize()
    assert np.array_equal(ps.grid_ids.to_numpy(), o["grid_ids"]) and np.array_equal(ps.pid.to_numpy(), o["pid"])
    curve = []
    for n in (25, 50, 100):
        k = n - (curve[-1][0] if curve else 0)
        o.step(k); solver.step(k)
        curve.append((n, scenes.rel_l2(scenes.ps_by_pid(ps, "x"), o.by_pid("x"))))
    _record_curve("high_fluid_wcsph_243k", curve)
    assert curve[-1][1] <= 1e-4, curve
    from sph_taichi_amd import _lib
    st = _lib.SphStats()
    ps._call("sph_get_stats", st)
    assert st.lds_overflow_targets == 0 and st.list_overflow_targets == 0
    assert np.array_equal(np.sort(ps.pid.to_numpy()), np.arange(sc.particle_max_num))
    ps.close()


def test_dragon_bath_dynamic_dfsph_scene():
    """data/scenes/dragon_bath_dynamic_dfsph.json equivalent (VERDICT r02 "missing" #4): the dragon (18,496 voxels from the
    fixture) as a DYNAMIC shape-matched body under DFSPHSolver, dt = 4e-3, next to the falling 423,500-particle block:
    boundary volumes of a moving body, shape matching of 18 k particles, the solid wall pass (the dragon starts 1 cm
    above the floor padding and lands within the first steps), DFSPH's coupling terms -- against the oracle."""
    sd = dragon_bath_scene()
    sd["RigidBodies"][0]["isDynamic"] = True
    sd["Configuration"]["simulationMethod"] = 4
    sd["Configuration"]["timeStepSize"] = 0.004
    cfg, sc = scenes.build(sd)
    o = scenes.make_oracle(cfg, sc, omp_threads=_threads())
    ps, solver = scenes.make_ps(sd)
    o.initialize(); solver.initialize()
    steps = 4
    its = []
    for _ in range(steps):
        o.step(1)
        its.append((o.s.last_iterations_v, o.s.last_iterations))
    solver.step(steps)
    st = solver.stats()
    assert abs(st["total_iterations_v"] - sum(a + 1 for a, _ in its)) <= 1
    assert abs(st["total_iterations"] - sum(b + 1 for _, b in its)) <= 1
    x, xo = scenes.ps_by_pid(ps, "x"), o.by_pid("x")
    assert scenes.rel_l2(x, xo) <= 1e-4
    body = sc.arrays["material"] == 0
    assert np.abs(x[body] - sc.arrays["x"][body]).max() > 1e-4, "the dragon did not move"
    assert scenes.rel_l2(x[body], xo[body]) <= 1e-5
    ps.close()


# ---------------------------------------------------------------------------------------------------------------
# deep cases: north_star's own statement ("dragon_bath.json inputs, <= 1e-4 rel-L2 on positions after N steps")
# at an N where the scene's physics is actually exercised, and BASELINE.json's other configs against the oracle
# ---------------------------------------------------------------------------------------------------------------
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _record_curve(name, curve, **extra):
    """Append one case's error-vs-N curve to $SPH_TEST_EVIDENCE_DIR/parity_curves.json (best effort: the file is evidence,
    not part of the assertion)."""
    try:
        path = scenes.evidence_path("parity_curves.json")
        if path is None:
            return
        data = json.load(open(path)) if os.path.exists(path) else {}
        data[name] = dict(extra, rel_l2_x=[[int(n), float(e)] for n, e in curve])
        for k in ("rel_l2_density", "rel_l2_v"):
            if k in extra:
                data[name][k] = [[int(n), float(e)] for n, e in extra[k]]
        json.dump(data, open(path, "w"), indent=1)
    except OSError:
        pass


def _march(ps, solver, o, checkpoints, tol, name, **extra):
    """Advance both sides through `checkpoints` (cumulative step counts), asserting rel-L2(x) <= tol at each."""
    import time
    curve, done, t_cpu, t_gpu = [], 0, 0.0, 0.0
    c_rho, c_v = [], []
    for n in checkpoints:
        t0 = time.perf_counter()
        o.step(n - done)
        t1 = time.perf_counter()
        solver.step(n - done); ps.sync()
        t_cpu += t1 - t0; t_gpu += time.perf_counter() - t1
        done = n
        curve.append((n, scenes.rel_l2(scenes.ps_by_pid(ps, "x"), o.by_pid("x"))))
        c_rho.append((n, scenes.rel_l2(scenes.ps_by_pid(ps, "density"), o.by_pid("density"))))
        c_v.append((n, scenes.rel_l2(scenes.ps_by_pid(ps, "v"), o.by_pid("v"))))
    _record_curve(name, curve, particles=int(ps.particle_max_num), tolerance=tol, rel_l2_density=c_rho, rel_l2_v=c_v,
                  oracle_ms_per_step=round(t_cpu / done * 1e3, 2), hip_ms_per_step=round(t_gpu / done * 1e3, 4), **extra)
    for n, e in curve:
        assert e <= tol, f"{name}: position rel-L2 after {n} steps = {e:.3e} (curve {curve})"
    return curve


def _developed_state(sd, k):
    """Run the HIP solver k steps from the scene's initial state and return (x, v) by persistent id."""
    ps, solver = scenes.make_ps(sd)
    solver.initialize()
    solver.step(k)
    x, v = scenes.ps_by_pid(ps, "x"), scenes.ps_by_pid(ps, "v")
    ps.close()
    return x, v


def _restart_pair(sd, x, v, threads):
    """A fresh HIP system and a fresh oracle, both started from the same (x, v) in persistent-id order."""
    cfg, sc = scenes.build(sd)
    sc.arrays["x"] = x.copy(); sc.arrays["v"] = v.copy()
    o = scenes.make_oracle(cfg, sc, omp_threads=threads)
    ps, solver = scenes.make_ps(sd, arrays={"x": x, "v": v})
    o.initialize(); solver.initialize()
    # cell ids, prefix array and the permutation are bit-exact on the developed (irregular) state too
    assert np.array_equal(ps.grid_ids.to_numpy(), o["grid_ids"])
    assert np.array_equal(ps.grid_particles_num.to_numpy(), o["grid_particles_num"])
    assert np.array_equal(ps.pid.to_numpy(), o["pid"])
    return ps, solver, o, sc


def test_c2_dragon_bath_to_floor_impact():
    """dragon_bath.json (fixture voxel set): the block (y >= 0.1, v = -1) reaches the floor after ~60 steps;
    300 steps cover the impact, the wall pass (sph_base.py:149-179), the pressure wave travelling up the
    column (WCSPH.py:46-85) and the first lateral spreading."""
    sd = dragon_bath_scene()
    cfg, sc = scenes.build(sd)
    o = scenes.make_oracle(cfg, sc, omp_threads=_threads())
    ps, solver = scenes.make_ps(sd)
    o.initialize(); solver.initialize()
    _march(ps, solver, o, (50, 100, 200, 300), 1e-4, "c2_dragon_bath", scene="data/scenes/dragon_bath.json")
    fluid = sc.arrays["material"] == 1
    x, v = scenes.ps_by_pid(ps, "x")[fluid], scenes.ps_by_pid(ps, "v")[fluid]
    p = scenes.ps_by_pid(ps, "pressure")[fluid]
    pad = np.float32(0.04)
    assert (x[:, 1] <= pad * 1.001).sum() > 1000, "the block never reached the floor"
    assert p.max() > 1e3, "no pressure built up: the impact was not exercised"
    assert v[:, 1].max() > -0.5, "the bottom layers were not decelerated"
    # Derived fields.  Up to the impact they agree like the positions do (step 200: density 2e-7, v 4e-6, recorded in
    # gpurun_out/parity_curves.json).  The impact amplifies ulp-level differences between ANY two f32 evaluations of the
    # same formulas: with stiffness 5e4 and exponent 7 a relative density difference of 1e-7 is a pressure difference of
    # ~4e-2 Pa on particles that are being stopped from 3 m/s within a few steps.  Round 4 measured what that floor is
    # (profiles/r04_parity_fastmath_ab.json): at step 300 the CPU restatement against ITSELF with nothing changed but the
    # order in which neighbours are visited differs by x 5.1e-6 / v 3.7e-3 / density 7.6e-5; the HIP path from it by
    # 6.7e-6 / 4.2e-3 / 9.5e-5 (6.0e-6 / 3.8e-3 / 7.8e-5 with SPH_OPT_EXACT_MATH: IEEE sqrt and divide instead of
    # v_rsq / v_rcp are worth ~10 % of it, not the bulk round 3 assumed).  Bounds = 2 x those measurements.
    assert scenes.rel_l2(scenes.ps_by_pid(ps, "density"), o.by_pid("density")) <= 2e-4
    assert scenes.rel_l2(scenes.ps_by_pid(ps, "v"), o.by_pid("v")) <= 1e-2
    ps.close()


def test_c2_fluid_onto_dragon():
    """The dragon_bath body with a fluid block dropped ONTO it (the scene's own block sits 2 m away and would need
    ~3,000 steps to arrive): Akinci boundary pressure (WCSPH.py:58-68) against 18,496 static body particles at scale."""
    sd = dragon_bath_scene()
    sd["FluidBlocks"][0].update(start=[2.7, 0.98, 0.62], end=[4.3, 1.9, 1.38], translation=[0.0, 0.0, 0.0],
                                velocity=[0.0, -2.0, 0.0])
    cfg, sc = scenes.build(sd)
    assert sc.solid_particle_num == 18496 and sc.fluid_particle_num > 100_000
    o = scenes.make_oracle(cfg, sc, omp_threads=_threads())
    ps, solver = scenes.make_ps(sd)
    o.initialize(); solver.initialize()
    _march(ps, solver, o, (50, 100, 150, 200), 1e-4, "c2_fluid_onto_dragon")
    fluid = sc.arrays["material"] == 1
    v = scenes.ps_by_pid(ps, "v")[fluid]
    free_fall = -2.0 - 9.81 * 200 * 4e-4
    assert v[:, 1].max() > free_fall + 1.0, "no fluid particle was stopped by the body"
    assert np.abs(v[:, [0, 2]]).max() > 0.5, "no lateral deflection: the body contact was not exercised"
    ps.close()


C1 = dict(counts=(64, 64, 64), start=(0.04, 0.04, 0.04), domain_end=(3.2, 2.0, 1.4))


def test_c1_dambreak_first_collapse():
    """BASELINE.json config 2 (64^3 cube in a 3.2 x 2.0 x 1.4 tank) against the oracle over the first 150 steps of
    the collapse, started with a sideways velocity so the free faces move from step 1."""
    sd = scenes.fluid_only(velocity=(1.0, -0.5, 0.3), **C1)
    cfg, sc = scenes.build(sd)
    assert sc.particle_max_num == 262_144
    o = scenes.make_oracle(cfg, sc, omp_threads=_threads())
    ps, solver = scenes.make_ps(sd)
    o.initialize(); solver.initialize()
    assert np.array_equal(ps.pid.to_numpy(), o["pid"]) and np.array_equal(ps.grid_ids.to_numpy(), o["grid_ids"])
    _march(ps, solver, o, (50, 100, 150), 1e-4, "c1_dambreak_first_collapse")
    assert scenes.rel_l2(scenes.ps_by_pid(ps, "density"), o.by_pid("density")) <= 1e-5
    ps.close()


def test_c1_dambreak_developed_flow():
    """The same dam-break after 2,500 steps (1 s: the front has crossed the tank and hit the far wall; cells hold
    0-20 particles, neighbour counts 5-60), then 100 steps of HIP against the oracle from that state."""
    sd = scenes.fluid_only(velocity=(0.0, 0.0, 0.0), **C1)
    x, v = _developed_state(sd, 2500)
    assert x[:, 0].max() > 2.5, "the front has not crossed the tank"
    ps, solver, o, sc = _restart_pair(sd, x, v, _threads())
    occ = np.diff(np.concatenate([[0], o["grid_particles_num"]]))
    curve = _march(ps, solver, o, (25, 50, 100), 1e-4, "c1_dambreak_developed", warm_steps=2500,
                   max_cell_occupancy=int(occ.max()))
    assert occ.max() >= 10, "not a developed state"
    assert scenes.rel_l2(scenes.ps_by_pid(ps, "v"), o.by_pid("v")) <= 5e-3
    ps.close()


def test_c3p_headline_workload_vs_oracle():
    """The benchmarked workload (1,747,584 particles) against the oracle, not only against itself: 30 steps with a
    non-zero initial velocity (every pair term active), then 20 more from a state 1,500 steps into the sloshing."""
    sd = scenes.fluid_only(counts=(246, 74, 96), start=(0.04, 0.04, 0.04), velocity=(0.3, -0.5, 0.2),
                           domain_end=(5.0, 3.0, 2.0))
    cfg, sc = scenes.build(sd)
    assert sc.particle_max_num == 1_747_584
    o = scenes.make_oracle(cfg, sc, omp_threads=_threads())
    ps, solver = scenes.make_ps(sd)
    o.initialize(); solver.initialize()
    assert np.array_equal(ps.pid.to_numpy(), o["pid"]) and np.array_equal(ps.grid_ids.to_numpy(), o["grid_ids"])
    _march(ps, solver, o, (10, 30), 1e-4, "c3p_from_lattice")
    assert scenes.rel_l2(scenes.ps_by_pid(ps, "density"), o.by_pid("density")) <= 1e-5
    del o
    solver.step(1470)
    x, v = scenes.ps_by_pid(ps, "x"), scenes.ps_by_pid(ps, "v")
    ps.close()
    ps, solver, o, sc = _restart_pair(sd, x, v, _threads())
    _march(ps, solver, o, (10, 20), 1e-4, "c3p_developed", warm_steps=1500)
    ps.close()
