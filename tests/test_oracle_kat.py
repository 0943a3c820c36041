"""Known-answer tests that pin the CPU oracle to the formulas of the reference
(sources cited per test)."""
import numpy as np
import pytest

import scenes

H, D = 0.04, 0.02
K = 8 / np.pi / H ** 3


@pytest.fixture(scope="module")
def orc():
    cfg, sc = scenes.build(scenes.fluid_only(counts=(5, 5, 5), start=(0.3, 0.1, 0.7),
                                             domain_end=(5.0, 3.0, 2.0)))
    return scenes.make_oracle(cfg, sc)


def test_geometry_constants(orc):
    # particle_system.py:36-38, 43-44
    s = orc.s
    assert s.support_radius == pytest.approx(0.04, rel=1e-7)
    assert s.particle_diameter == pytest.approx(0.02, rel=1e-7)
    assert s.m_V0 == pytest.approx(6.4e-6, rel=1e-6)
    assert list(orc.grid_num) == [125, 75, 50] and orc.G == 468750
    assert orc["m"][0] == pytest.approx(6.4e-3, rel=1e-6)   # particle_system.py:230-231


def test_cubic_kernel_values(orc):
    # sph_base.py:23-44
    assert orc.cubic_kernel(0.0) == pytest.approx(39788.7358, rel=1e-6)
    assert orc.cubic_kernel(D) == pytest.approx(K / 4, rel=1e-6)           # 9947.1839
    assert orc.cubic_kernel(np.sqrt(2) * D) == pytest.approx(1999.4847, rel=2e-6)
    assert orc.cubic_kernel(np.sqrt(3) * D) == pytest.approx(191.3628, rel=2e-5)
    assert orc.cubic_kernel(H * 1.0001) == 0.0
    assert orc.cubic_kernel(0.5 * H) == pytest.approx(K * 0.25, rel=1e-6)  # both branches meet at q = 1/2


def test_cubic_kernel_derivative_values(orc):
    # sph_base.py:46-68
    g = orc.cubic_kernel_derivative([D, 0, 0])
    assert g[0] == pytest.approx(-1492077.59, rel=2e-6) and g[1] == 0 and g[2] == 0
    g = orc.cubic_kernel_derivative([0.03, 0, 0])
    assert g[0] == pytest.approx(-373019.40, rel=2e-5)
    assert np.all(orc.cubic_kernel_derivative([0.5e-5, 0, 0]) == 0)       # r_norm <= 1e-5
    assert np.all(orc.cubic_kernel_derivative([0.0401, 0, 0]) == 0)       # q > 1
    # gradient is the derivative of W along r
    r = 0.013
    fd = (orc.cubic_kernel(r + 1e-5) - orc.cubic_kernel(r - 1e-5)) / 2e-5
    assert orc.cubic_kernel_derivative([r, 0, 0])[0] == pytest.approx(fd, rel=2e-3)


def test_tait_eos():
    # WCSPH.py:75-76 with stiffness 50000, exponent 7: p(1010) = 3606.7676
    cfg, sc = scenes.build(scenes.fluid_only(counts=(2, 2, 2)))
    o = scenes.make_oracle(cfg, sc)
    o.initialize_particle_system()
    o["density"][:] = 1010.0
    o.compute_pressure_forces()
    assert o["pressure"][0] == pytest.approx(3606.7676, rel=2e-5)
    o["density"][:] = 900.0
    o.compute_pressure_forces()
    assert np.all(o["density"] == 1000.0) and np.all(o["pressure"] == 0.0)   # clamp, WCSPH.py:75


def test_5x5x5_cube_one_step(orc):
    # 5x5x5 cube at (0.3,0.1,0.7), v0 = (0,-1,0), one step
    o = orc
    o.initialize()
    pid = o["pid"].copy()
    centre = int(np.where(pid == (2 * 25 + 2 * 5 + 2))[0][0])
    corner = int(np.where(pid == 0)[0][0])
    o.initialize_particle_system()
    o.compute_densities()
    assert o["density"][centre] == pytest.approx(799.9777, rel=2e-6)
    assert o["density"][corner] == pytest.approx(485.2487, rel=2e-6)
    o.compute_non_pressure_forces()
    a = o["acceleration"][corner].copy()
    assert a == pytest.approx([2.8275, -6.9825, 2.8275], rel=2e-4)
    ac = o["acceleration"][centre]
    assert abs(ac[0]) < 1e-3 and ac[1] == pytest.approx(-9.81, abs=1e-3) and abs(ac[2]) < 1e-3
    o.compute_pressure_forces()
    assert o["pressure"][corner] == 0.0 and o["density"][centre] == 1000.0
    o.advect()
    assert o["v"][centre][1] == pytest.approx(-1.003924, rel=1e-6)
    assert o["v"][corner] == pytest.approx([0.001131, -1.002793, 0.001131], rel=2e-3)


def test_interior_lattice_density():
    # 26 lattice neighbours: rho = 0.8 * rho0 * 0.99997247 = 799.97797
    cfg, sc = scenes.build(scenes.fluid_only(counts=(7, 7, 7), start=(0.2, 0.2, 0.2)))
    o = scenes.make_oracle(cfg, sc)
    o.initialize_particle_system()
    o.compute_densities()
    mid = int(np.where(o["pid"] == (3 * 49 + 3 * 7 + 3))[0][0])
    assert o["density"][mid] == pytest.approx(799.97797, rel=3e-6)


def test_sort_invariants():
    # particle_system.py:311-369: keys sorted, prefix[G-1] == N, permutation is a bijection, stable
    cfg, sc = scenes.build(scenes.fluid_with_rigid_blocks())
    scenes.jitter(sc, 0.3, seed=3)
    rng = np.random.default_rng(1)
    perm = rng.permutation(sc.particle_max_num)
    for k in sc.arrays:
        sc.arrays[k] = sc.arrays[k][perm]
    sc.arrays["pid"] = np.arange(sc.particle_max_num, dtype=np.int32)   # persistent id = index at upload time
    o = scenes.make_oracle(cfg, sc)
    x_before = o["x"].copy()
    o.initialize_particle_system()
    gi, pid = o["grid_ids"], o["pid"]
    assert np.all(np.diff(gi) >= 0)
    assert o["grid_particles_num"][-1] == o.N
    assert np.array_equal(np.sort(pid), np.arange(o.N))
    same = gi[1:] == gi[:-1]
    assert np.all(pid[1:][same] > pid[:-1][same])                       # stable inside a cell
    assert np.array_equal(o["x"], x_before[pid])                         # payload follows the key
    # independent restatement: stable argsort of the keys
    keys = (np.floor(x_before / np.float32(0.04)).astype(np.int64) * [30 * 20, 20, 1]).sum(axis=1)
    assert np.array_equal(pid, np.argsort(keys, kind="stable"))


def test_pure_fluid_momentum():
    # symmetric pressure formula + equal m_V => sum_i a_pressure_i = 0 (WCSPH.py:56-57)
    cfg, sc = scenes.build(scenes.fluid_only(counts=(8, 8, 8), start=(0.2, 0.2, 0.2)))
    scenes.jitter(sc, 0.25, seed=5)
    o = scenes.make_oracle(cfg, sc)
    o.initialize_particle_system()
    o.compute_densities()
    o["density"][:] *= 1.3                         # force p > 0
    o["acceleration"][:] = 0
    o.compute_pressure_forces()
    a = o["acceleration"].astype(np.float64)
    assert np.abs(a.sum(axis=0)).max() < 1e-3 * np.abs(a).sum(axis=0).max()


def test_polar_rotation_matches_svd():
    from oracle.oracle import polar_rotation
    rng = np.random.default_rng(0)
    for _ in range(50):
        A = rng.normal(size=(3, 3))
        U, s, Vt = np.linalg.svd(A)
        if np.linalg.det(U @ Vt) < 0:
            U[:, -1] *= -1
        R = polar_rotation(A)
        assert np.allclose(R, U @ Vt, atol=2e-6)
        assert np.linalg.det(R.astype(np.float64)) == pytest.approx(1.0, abs=1e-5)
    assert np.all(polar_rotation(np.zeros((3, 3))) == 0)


def _closest_proper_rotation(A):
    """What ti.polar_decompose's rotation factor is for ANY A (taichi's svd keeps det U = det V = +1 and lets the smallest
    singular value go negative): U V^T of an SVD whose U and V are both proper rotations."""
    U, s, Vt = np.linalg.svd(np.asarray(A, np.float64))
    if np.linalg.det(U) < 0:
        U[:, -1] *= -1
    if np.linalg.det(Vt) < 0:
        Vt[-1, :] *= -1
    return U @ Vt


def test_polar_rotation_of_degenerate_matrices():
    """VERDICT r04 "missing" #6: the matrices shape matching produces for degenerate bodies -- rank 2 (a one-layer body), a
    180-degree turn (trace R = -1), a reflection (det A < 0) -- where Newton, Jacobi and Taichi's SVD are most likely to part.
    A = R0 S with a symmetric positive (semi)definite S of DISTINCT eigenvalues, so that the closest proper rotation is
    unique; the oracle's Jacobi form must return it."""
    from oracle.oracle import polar_rotation
    rng = np.random.default_rng(3)

    def rot(axis, deg):
        k = np.asarray(axis, np.float64) / np.linalg.norm(axis)
        K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        t = np.deg2rad(deg)
        return np.eye(3) + np.sin(t) * K + (1 - np.cos(t)) * (K @ K)
    for trial in range(20):
        Q = rot(rng.normal(size=3), rng.uniform(0, 360))
        R0 = rot(rng.normal(size=3), rng.choice([0.0, 25.0, 179.0, 180.0]))
        for eig, mirror in (((3.0, 1.5, 0.0), False), ((3.0, 1.5, 0.4), True), ((2.0, 1.0, 0.5), False)):
            S = Q @ np.diag(eig) @ Q.T
            A = R0 @ S
            if mirror:                                  # reflect through the plane normal to S's smallest eigenvector
                n = Q[:, 2]
                A = R0 @ (np.eye(3) - 2.0 * np.outer(n, n)) @ S
            want = _closest_proper_rotation(A)
            got = polar_rotation(A.astype(np.float32)).astype(np.float64)
            assert np.linalg.det(got) == pytest.approx(1.0, abs=1e-5)
            assert np.abs(got - want).max() <= 5e-6, (trial, eig, mirror)
            if not mirror and eig[2] > 0:
                assert np.abs(got - R0).max() <= 5e-6      # a rigid turn of a full-rank body is recovered, 180 degrees included
