"""The oracle's grid/sort/27-cell traversal against an O(N^2) NumPy brute force
that uses no grid at all (oracle/brute.py)."""
import numpy as np
import pytest

import scenes
from oracle.brute import Brute

FIELDS = ["m_V", "density", "pressure", "acceleration", "v", "x"]


def _by_pid(o, name):
    return o.by_pid(name)


@pytest.mark.parametrize("scene_fn,steps", [(scenes.fluid_only, 3), (scenes.fluid_with_rigid_blocks, 3)])
def test_step_matches_bruteforce(scene_fn, steps):
    cfg, sc = scenes.build(scene_fn())
    scenes.jitter(sc, 0.15, seed=11)
    o = scenes.make_oracle(cfg, sc)
    b = Brute(scenes.solver_params(cfg, sc), sc.arrays)
    o.initialize()
    b.boundary_volume(dynamic=False)
    b.boundary_volume(dynamic=True)
    assert np.allclose(_by_pid(o, "m_V"), b.a["m_V"], rtol=2e-5)
    for _ in range(steps):
        o.step(1)
        b.step()
        for f in FIELDS:
            ref, got = b.a[f], _by_pid(o, f)
            scale = max(np.abs(ref).max(), 1e-30)
            assert np.abs(got - ref).max() <= 3e-4 * scale, f


def test_kernel_by_kernel_with_coupling():
    cfg, sc = scenes.build(scenes.fluid_with_rigid_blocks())
    scenes.jitter(sc, 0.2, seed=2)
    o = scenes.make_oracle(cfg, sc)
    b = Brute(scenes.solver_params(cfg, sc), sc.arrays)
    o.initialize()
    b.boundary_volume(False); b.boundary_volume(True)
    for name in ("compute_densities", "compute_non_pressure_forces", "compute_pressure_forces", "advect"):
        getattr(o, name)(); getattr(b, name)()
        for f in FIELDS:
            ref, got = b.a[f], _by_pid(o, f)
            assert np.abs(got - ref).max() <= 2e-4 * max(np.abs(ref).max(), 1e-30), (name, f)
    dyn = (sc.arrays["material"] == 0) & (sc.arrays["is_dynamic"] == 1)
    assert np.abs(b.a["acceleration"][dyn] - np.asarray(cfg.get_cfg("gravitation"), np.float32)).max() > 1e-3, \
        "scene must exercise the two-way coupling scatter"


def test_dfsph_neighbour_sums_match_bruteforce():
    """DFSPH.py:116-221 (factor, density change with its 20-neighbour switch, advected density): oracle vs O(N^2)."""
    sd = scenes.as_dfsph(scenes.fluid_with_rigid_blocks())
    sd["FluidBlocks"][0]["velocity"] = [0.5, -1.0, 0.3]
    cfg, sc = scenes.build(sd)
    scenes.jitter(sc, 0.2, seed=5)
    rng = np.random.default_rng(1)
    sc.arrays["v"] = (sc.arrays["v"] + rng.normal(0, 0.3, size=sc.arrays["v"].shape)).astype(np.float32)
    o = scenes.make_oracle(cfg, sc)
    b = Brute(scenes.solver_params(cfg, sc), sc.arrays)
    o.initialize()
    b.boundary_volume(False); b.boundary_volume(True)
    o.compute_densities(); b.compute_densities()
    fluid = sc.arrays["material"] == 1
    o.compute_DFSPH_factor()
    ref = b.dfsph_factor()
    assert np.abs(_by_pid(o, "dfsph_factor") - ref)[fluid].max() <= 2e-4 * np.abs(ref).max()
    o.compute_density_change()
    ref = b.dfsph_density_change()
    got = _by_pid(o, "density_adv")
    assert (ref[fluid] > 0).any() and (ref[fluid] == 0).any(), "both sides of the neighbour-count switch must occur"
    assert np.abs(got - ref)[fluid].max() <= 3e-4 * np.abs(ref).max()
    o.compute_density_adv()
    ref = b.dfsph_density_adv()
    assert np.abs(_by_pid(o, "density_adv") - ref)[fluid].max() <= 2e-5


def test_dfsph_invariants():
    """A rigidly translating fluid has zero velocity divergence: density change 0, advected density = max(rho/rho0, 1),
    and one divergence-solver iteration leaves the velocities untouched."""
    sd = scenes.as_dfsph(scenes.fluid_only(counts=(8, 8, 8), velocity=(1.0, 2.0, -0.5)))
    cfg, sc = scenes.build(sd)
    scenes.jitter(sc, 0.2, seed=9)
    o = scenes.make_oracle(cfg, sc)
    o.initialize()
    o.compute_densities(); o.compute_DFSPH_factor()
    assert (o["dfsph_factor"] <= 0).all()
    v0 = o["v"].copy()
    o.compute_density_change()
    assert np.abs(o["density_adv"]).max() == 0.0
    assert o.divergence_solve() == 0 and np.array_equal(o["v"], v0)
    o.compute_density_adv()
    assert np.allclose(o["density_adv"], np.maximum(o["density"] / 1000.0, 1.0), rtol=1e-6)
