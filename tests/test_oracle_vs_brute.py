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
