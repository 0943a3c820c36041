"""SPH_OPT_KERNEL_VARIANT: every instance of the fused step's two brick sweeps (baseline run-by-run emission, group-sorted
emission; plain / branch-free / early-entry force loops) computes the reference's sums -- each against
the CPU oracle on the coupled scene, on crowded cells (list overflow -> exact walk), and against variant 0."""
import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu

# one step on the coupled scene, max |err| / max |ref| against the oracle (bounds: <= 3 x the recording run's worst, see scenes.bound)
# (recording run r06a, profiles/r06a_variant_errors_recording_run.json: density 2.7e-7, pressure 2.1e-6, acceleration 2.0e-6; before: 2e-5 / 2e-4 / 5e-4)
ONE_STEP_TOL = (("density", 1e-6), ("pressure", 7e-6), ("acceleration", 7e-6))
VARIANTS = [0, 1, 8, 16, 24, 25, 57, 27, 29]  # baseline; GROUPS; the force bits; default = GROUPS | BF | DEEP; 57 = default | MFMA (the filter on the
# matrix pipe); 27 / 29 = default | GAT_LDS / GAT_LDS4 (round 6: the force sweep's second neighbour record staged in LDS, smaller tiles)


def _system(sd, arrays, variant):
    from sph_taichi_amd import _lib
    ps, solver = scenes.make_ps(sd, arrays)
    ps.set_option(_lib.OPT_KERNEL_VARIANT, variant)
    assert ps.get_option(_lib.OPT_KERNEL_VARIANT) == variant
    ps.set_option(_lib.OPT_KERNEL_VARIANT, -1)            # -1 = the library's default mask
    assert ps.get_option(_lib.OPT_KERNEL_VARIANT) == _lib.VAR_DEFAULT
    ps.set_option(_lib.OPT_KERNEL_VARIANT, variant)
    return ps, solver


@pytest.mark.parametrize("variant", VARIANTS)
def test_variant_follows_the_oracle(variant):
    sd = scenes.fluid_with_rigid_blocks()
    cfg, sc = scenes.build(sd)
    scenes.jitter(sc, 0.1, seed=4)
    o = scenes.make_oracle(cfg, sc)
    ps, solver = _system(sd, sc.arrays, variant)
    o.initialize(); solver.initialize()
    o.step(1); solver.step(1)
    for name, tol in ONE_STEP_TOL:
        ref, got = o.by_pid(name), scenes.ps_by_pid(ps, name)
        err = float(np.abs(got.astype(np.float64) - ref).max()) / max(float(np.abs(ref).max()), 1e-30)
        scenes.bound("variant_errors", f"oracle one step: {name} (variant {variant})", err, tol)
    o.step(19); solver.step(19)
    err = scenes.rel_l2(scenes.ps_by_pid(ps, "x"), o.by_pid("x"))
    scenes.bound("variant_errors", f"oracle 20 steps: rel-L2(x) (variant {variant})", err, 2e-7)   # north_star: 1e-4; measured 5.1e-8
    ps.close()


@pytest.mark.parametrize("variant", [0, 1, 24, 25, 57, 27, 29])
def test_variant_on_crowded_cells(variant):
    """12^3 particles in a (1.5 h)^3 box: > 95 neighbours each, so every list overflows (the force sweep must
    fall back to the exact walk; the range-checked list stores must drop rows >= LISTCAP and nothing else)."""
    sd = scenes.fluid_only(counts=(12, 12, 12), start=(0.3, 0.3, 0.3))
    cfg, sc = scenes.build(sd)
    rng = np.random.default_rng(0)
    sc.arrays["x"] = (0.3 + rng.uniform(0, 0.06, size=sc.arrays["x"].shape)).astype(np.float32)
    sc.arrays["x_0"] = sc.arrays["x"].copy()
    o = scenes.make_oracle(cfg, sc)
    ps, solver = _system(sd, sc.arrays, variant)
    o.initialize(); solver.initialize()
    o.step(1); solver.step(1)
    for name, tol in (("density", 4e-6), ("acceleration", 8e-6)):      # measured 1.2e-6 / 2.6e-6 (before: 5e-5 / 5e-3)
        ref, got = o.by_pid(name), scenes.ps_by_pid(ps, name)
        err = float(np.abs(got.astype(np.float64) - ref).max()) / max(float(np.abs(ref).max()), 1e-30)
        scenes.bound("variant_errors", f"crowded: {name} (variant {variant})", err, tol)
    ps.close()


def test_variants_agree_with_each_other_on_a_ragged_lattice():
    """A lattice whose nodes sit on cell faces (run lengths 1..27 per cell, the benchmark's initial state): the padded
    filter reads past every run's end; all variants must produce variant 0's densities and accelerations."""
    sd = scenes.fluid_only(counts=(18, 14, 16), start=(0.04, 0.04, 0.04), velocity=(0.3, -0.2, 0.1))
    cfg, sc = scenes.build(sd)
    base = None
    for variant in VARIANTS:
        ps, solver = _system(sd, sc.arrays, variant)
        solver.initialize()
        solver.step(3)
        got = {n: scenes.ps_by_pid(ps, n).astype(np.float64) for n in ("density", "acceleration", "x")}
        ps.close()
        if base is None:
            base = got
            continue
        for n, tol in (("density", 1e-7), ("acceleration", 2e-6), ("x", 1e-7)):   # measured 0 / 5.5e-7 (the matrix-pipe variant's emission order) / 0
            err = float(np.abs(got[n] - base[n]).max()) / max(float(np.abs(base[n]).max()), 1e-30)
            scenes.bound("variant_errors", f"ragged lattice vs variant 0: {n} (variant {variant})", err, tol)


@pytest.mark.parametrize("variant", [0, 24, 25, 57, 27, 29])
def test_long_lists_between_64_and_95_entries(variant):
    """A slab compressed to ~2.5 x rest density (spacing 0.74 d): 65..95 list entries per interior particle -- beyond
    the 63 of round 1, inside LISTCAP = 95 -- so the list rows >= 64 are written by the density sweep and read back
    by the force sweep; no target may fall back to the exact walk."""
    from sph_taichi_amd import _lib
    import ctypes as C
    sd = scenes.fluid_only(counts=(14, 5, 14), start=(0.3, 0.3, 0.3), velocity=(0.2, 0.0, -0.1))
    cfg, sc = scenes.build(sd)
    i, j, k = np.meshgrid(np.arange(14), np.arange(5), np.arange(14), indexing="ij")
    x = (0.3 + 0.0148 * np.stack([i, j, k], -1).reshape(-1, 3)).astype(np.float32)
    sc.arrays["x"] = x
    sc.arrays["x_0"] = x.copy()
    o = scenes.make_oracle(cfg, sc)
    ps, solver = _system(sd, sc.arrays, variant)
    o.initialize(); solver.initialize()
    o.step(1); solver.step(1)
    st = _lib.SphStats()
    ps._call("sph_get_stats", st)
    assert 64 < st.max_list <= 95 and st.list_overflow_targets == 0 and st.lds_overflow_targets == 0, \
        (st.max_list, st.list_overflow_targets, st.lds_overflow_targets)
    for name, tol in (("density", 1.5e-6), ("acceleration", 4e-6)):     # measured 5.0e-7 / 1.2e-6 (before: 5e-5 / 2e-3)
        ref, got = o.by_pid(name), scenes.ps_by_pid(ps, name)
        err = float(np.abs(got.astype(np.float64) - ref).max()) / max(float(np.abs(ref).max()), 1e-30)
        scenes.bound("variant_errors", f"long lists: {name} (variant {variant})", err, tol)
    ps.close()


def test_exact_math_instance_follows_the_oracle_at_least_as_closely():
    """SPH_OPT_EXACT_MATH: the two brick sweeps with IEEE sqrt / divide, the two-branch spline and no
    FMA contraction -- the reference's f32 expressions as the oracle evaluates them.  It must follow the oracle, and after
    one step its densities and accelerations must be no further from it than the fast-math build's (the A/B at scale is
    tools/fastmath_ab.py -> profiles/r04_parity_fastmath_ab.json).  The production option refuses section ablation."""
    from sph_taichi_amd import _lib
    sd = scenes.fluid_with_rigid_blocks()
    cfg, sc = scenes.build(sd)
    scenes.jitter(sc, 0.1, seed=4)
    errs = {}
    for exact in (0, 1):
        o = scenes.make_oracle(cfg, sc)
        ps, solver = scenes.make_ps(sd, sc.arrays)
        ps.set_option(_lib.OPT_EXACT_MATH, exact)
        assert ps.get_option(_lib.OPT_EXACT_MATH) == exact
        o.initialize(); solver.initialize()
        o.step(1); solver.step(1)
        e = {}
        for name in ("density", "pressure", "acceleration"):
            ref, got = o.by_pid(name), scenes.ps_by_pid(ps, name)
            e[name] = float(np.abs(got.astype(np.float64) - ref).max()) / max(float(np.abs(ref).max()), 1e-30)
        o.step(19); solver.step(19)
        e["x20"] = scenes.rel_l2(scenes.ps_by_pid(ps, "x"), o.by_pid("x"))
        errs[exact] = e
        if not _lib.profiling_variant():
            with pytest.raises(_lib.SphError, match="profiling build"):
                ps.set_option(_lib.OPT_DEBUG_ABLATE, 1)
        ps.close()
    for name, tol in ONE_STEP_TOL + (("x20", 2e-7),):
        for exact in (0, 1):
            scenes.bound("variant_errors", f"exact math {exact}: {name}", errs[exact][name], tol)
    assert errs[1]["density"] <= 2.0 * errs[0]["density"] + 1e-6 and errs[1]["acceleration"] <= 2.0 * errs[0]["acceleration"] + 1e-6, errs


def test_phase_events_in_every_kth_step():
    """SPH_OPT_TIMING k (include/sph_hip.h): the five per-phase events are recorded in every k-th step of a call sequence, the
    sums cover the timed steps only, and the trajectory does not know about them."""
    from sph_taichi_amd import _lib
    sd = scenes.fluid_only(counts=(12, 10, 8))
    out = {}
    for k in (0, 1, 4):
        ps, solver = scenes.make_ps(sd)
        solver.initialize()
        ps.set_option(_lib.OPT_TIMING, k)
        assert ps.get_option(_lib.OPT_TIMING) == k
        ps._call("sph_reset_timings")
        solver.step(10)
        solver.step(3)          # the phase runs on across calls: steps 0, 4, 8 and 12 of the 13
        ps.sync()
        tm = _lib.SphTimings()
        ps._call("sph_get_timings", tm)
        assert int(tm.steps) == {0: 0, 1: 13, 4: 4}[k]
        if k:
            assert tm.total_ms > 0.0 and abs(tm.total_ms - (tm.sort_ms + tm.neighbour_ms + tm.force_ms + tm.integrate_ms)) <= 1e-6 * tm.total_ms
        out[k] = scenes.ps_by_pid(ps, "x")
        ps.close()
    assert np.array_equal(out[0], out[1]) and np.array_equal(out[0], out[4])


def test_pure_fluid_instance_of_the_density_sweep_is_bit_identical():
    """A context without any solid particle runs an instance of the density sweep that takes m_V_j = m_V0 from a register
    instead of the tile (one LDS read per hit less; chosen by the launcher after a device-side check).  Same fma, same
    operand values: densities, pressures and the trajectory must be BIT-identical to the general instance -- a form that
    scaled the sum of W once was 2 % faster too and moved the violent reference-executed fixture from 1.5e-6 to 7.8e-4
    (DESIGN_HISTORY.md, round 5).  Scenes with solids never select it."""
    from sph_taichi_amd import _lib
    sd = scenes.fluid_only(counts=(18, 14, 16), start=(0.04, 0.04, 0.04), velocity=(0.6, -1.0, 0.3))
    cfg, sc = scenes.build(sd)
    scenes.jitter(sc, 0.1, seed=11)
    out = []
    for off in (False, True):
        ps, solver = scenes.make_ps(sd, sc.arrays)
        assert ps.get_option(_lib.OPT_PURE_FLUID_INSTANCE) == 1      # the default
        if off:
            ps.set_option(_lib.OPT_PURE_FLUID_INSTANCE, 0)           # the A/B switch (ABI 5; an environment variable before)
        solver.initialize()
        solver.step(25)
        assert ps.get_option(_lib.OPT_UNIFORM_FLUID_STATE) == 1
        out.append({n: scenes.ps_by_pid(ps, n) for n in ("x", "v", "density", "pressure")})
        ps.close()
    for n in out[0]:
        assert np.array_equal(out[0][n], out[1][n]), f"{n} differs between the pure-fluid and the general instance"


@pytest.mark.parametrize("dfsph", [False, True])
def test_brick_column_records_change_nothing(dfsph, tmp_path):
    """SPH_OPT_BRICK_RECORDS (round 6): the list-writing density sweep leaves every brick's column tables in HBM and the list
    readers (the fused force sweep; every later sweep of a DFSPH step) load them instead of recomputing them from the cell
    array.  The tables are the same integers, so positions, velocities, densities -- and under DFSPH the solver's iteration
    counts -- must be BIT-identical with the option off, on a scene with solids, bodies and partly filled bricks."""
    from sph_taichi_amd import _lib
    sd = scenes.fluid_with_rigid_bodies(str(tmp_path / "cube.obj"), fluid_velocity=(0.8, -1.0, 0.0))
    if dfsph:
        sd = scenes.as_dfsph(sd)
    cfg, sc = scenes.build(sd)
    out = []
    for rec in (1, 0):
        ps, solver = scenes.make_ps(sd, sc.arrays)
        assert ps.get_option(_lib.OPT_BRICK_RECORDS) == 1          # the default
        ps.set_option(_lib.OPT_BRICK_RECORDS, rec)
        solver.initialize()
        solver.step(6 if dfsph else 20)
        got = {n: scenes.ps_by_pid(ps, n) for n in ("x", "v", "density")}
        if dfsph:
            st = solver.stats()
            got["iterations"] = np.array([st["total_iterations_v"], st["total_iterations"]])
        out.append(got)
        ps.close()
    for n in out[0]:
        assert np.array_equal(out[0][n], out[1][n]), f"{n} differs with the brick records on / off"
