"""Independent O(N^2) NumPy restatement of one WCSPH step (no grid, no sort) and of the DFSPH neighbour sums.

TEST INFRASTRUCTURE ONLY.  Cross-checks oracle/sph_oracle.c's neighbour search:
the neighbour set here is { j != i : |x_i - x_j| < h } computed from a dense
distance matrix, so any mistake in the oracle's cell hashing / prefix sum /
27-cell traversal shows up as a difference.  Summation order differs from the
reference (row-wise pairwise sums), so comparisons use a ~1e-5 tolerance.

Formulas follow /root/reference: sph_base.py:23-68 (kernels), :91-113 (boundary
volume), :149-179 (walls), WCSPH.py:19-149 (density, forces, EOS, advect).
Valid when no particle sits in grid cell 0 (the reference never sees those as
neighbours, particle_system.py:384) and every particle is inside the domain.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


def _W(r, h, k):
    q = r / f32(h)
    w1 = f32(k) * (f32(6.0) * q * q * q - f32(6.0) * q * q + f32(1.0))
    w2 = f32(k) * f32(2.0) * (f32(1.0) - q) ** 3
    return np.where(q <= f32(0.5), w1, np.where(q <= f32(1.0), w2, f32(0.0))).astype(f32)


def _gradW(rvec, rn, h, kd):
    q = rn / f32(h)
    with np.errstate(divide="ignore", invalid="ignore"):
        gq = rvec / (rn * f32(h))[..., None]
    c1 = f32(kd) * q * (f32(3.0) * q - f32(2.0))
    c2 = f32(kd) * (-(f32(1.0) - q) * (f32(1.0) - q))
    c = np.where(q <= f32(0.5), c1, c2)
    ok = (rn > f32(1e-5)) & (q <= f32(1.0))
    return np.where(ok[..., None], c[..., None] * gq, f32(0.0)).astype(f32)


class Brute:
    def __init__(self, params, arrays):
        r = float(params["particle_radius"])
        self.d = 2 * r
        self.h = 4.0 * r
        self.m_V0 = 0.8 * self.d ** 3
        self.rho0 = f32(params["density_0"])
        self.stiffness = f32(params["stiffness"])
        self.exponent = f32(params["exponent"])
        self.dt = f32(params["dt"])
        self.g = np.asarray(params["g"], dtype=f32)
        self.dom = np.asarray(params["domain_size"], dtype=f32)
        self.k = 8 / np.pi / self.h ** 3
        self.kd = 6.0 * 8 / np.pi / self.h ** 3
        self.nu = 0.01
        self.sigma = 0.01
        self.a = {k: np.array(v, copy=True) for k, v in arrays.items()}
        for name in ("x", "v", "acceleration"):
            self.a[name] = np.asarray(self.a.get(name, np.zeros((len(arrays["x"]), 3))), dtype=f32)
        for name in ("m_V", "m", "density", "pressure"):
            self.a[name] = np.asarray(self.a[name], dtype=f32)

    def _pairs(self):
        x = self.a["x"]
        rvec = (x[:, None, :] - x[None, :, :]).astype(f32)
        rn = np.sqrt((rvec[..., 0] * rvec[..., 0] + rvec[..., 1] * rvec[..., 1]).astype(f32)
                     + rvec[..., 2] * rvec[..., 2]).astype(f32)
        nb = rn < f32(self.h)
        np.fill_diagonal(nb, False)
        return rvec, rn, nb

    def boundary_volume(self, dynamic):
        a = self.a
        rvec, rn, nb = self._pairs()
        solid = a["material"] == 0
        tgt = solid & ((a["is_dynamic"] != 0) if dynamic else (a["is_dynamic"] == 0))
        W = _W(rn, self.h, self.k)
        delta = _W(np.zeros(1, f32), self.h, self.k)[0] + np.where(nb & solid[None, :], W, f32(0)).sum(axis=1, dtype=f32)
        a["m_V"] = np.where(tgt, f32(1.0) / delta * f32(3.0), a["m_V"]).astype(f32)

    def compute_densities(self):
        a = self.a
        rvec, rn, nb = self._pairs()
        W = _W(rn, self.h, self.k)
        den = np.where(nb, a["m_V"][None, :] * W, f32(0)).sum(axis=1, dtype=f32)
        rho = (a["m_V"] * _W(np.zeros(1, f32), self.h, self.k)[0] + den) * self.rho0
        fluid = a["material"] == 1
        a["density"] = np.where(fluid, rho, a["density"]).astype(f32)

    def compute_non_pressure_forces(self):
        a = self.a
        rvec, rn, nb = self._pairs()
        fluid = a["material"] == 1
        pair = nb & fluid[None, :]
        r2 = (rn * rn).astype(f32)
        Wst = np.where(r2 > f32(self.d * self.d), _W(rn, self.h, self.k), _W(np.full(1, self.d, f32), self.h, self.k)[0])
        c_st = (f32(self.sigma) / a["m"])[:, None] * a["m"][None, :]
        st = -(np.where(pair, c_st * Wst, f32(0))[..., None] * rvec).sum(axis=1, dtype=f32)
        vij = (a["v"][:, None, :] - a["v"][None, :, :]).astype(f32)
        v_xy = (vij * rvec).sum(axis=2, dtype=f32)
        gw = _gradW(rvec, rn, self.h, self.kd)
        c_v = f32(10 * self.nu) * (a["m"] / a["density"])[None, :] * v_xy / (r2 + f32(0.01 * self.h ** 2))
        visc = (np.where(pair, c_v, f32(0))[..., None] * gw).sum(axis=1, dtype=f32)
        acc = self.g[None, :] + np.where(fluid[:, None], st + visc, f32(0))
        static = (a["material"] == 0) & (a["is_dynamic"] == 0)
        a["acceleration"] = np.where(static[:, None], f32(0), acc).astype(f32)

    def compute_pressure_forces(self):
        a = self.a
        fluid = a["material"] == 1
        rho = np.where(fluid, np.maximum(a["density"], self.rho0), a["density"]).astype(f32)
        a["density"] = rho
        p = self.stiffness * (np.power(rho / self.rho0, self.exponent, dtype=f32) - f32(1.0))
        a["pressure"] = np.where(fluid, p, a["pressure"]).astype(f32)
        rvec, rn, nb = self._pairs()
        gw = _gradW(rvec, rn, self.h, self.kd)
        dpi = (a["pressure"] / (rho * rho)).astype(f32)
        dpj_f = dpi[None, :]
        dpj_s = (a["pressure"] / (self.rho0 * self.rho0)).astype(f32)[:, None]
        dpj = np.where(fluid[None, :], dpj_f, dpj_s)
        c = -self.rho0 * a["m_V"][None, :] * (dpi[:, None] + dpj)
        f = np.where((nb & fluid[:, None])[..., None], c[..., None] * gw, f32(0)).astype(f32)  # f[i,j]
        dv = f.sum(axis=1, dtype=f32)
        dyn_solid = (a["material"] == 0) & (a["is_dynamic"] != 0)
        back = -(f * (self.rho0 / a["density"])[None, :, None])
        back = np.where(dyn_solid[None, :, None], back, f32(0)).sum(axis=0, dtype=f32)
        static = (a["material"] == 0) & (a["is_dynamic"] == 0)
        acc = a["acceleration"] + np.where(fluid[:, None], dv, f32(0)) + back
        a["acceleration"] = np.where(static[:, None], f32(0), acc).astype(f32)

    def advect(self):
        a = self.a
        dyn = (a["is_dynamic"] != 0)[:, None]
        v = a["v"] + self.dt * a["acceleration"]
        a["v"] = np.where(dyn, v, a["v"]).astype(f32)
        a["x"] = np.where(dyn, a["x"] + self.dt * a["v"], a["x"]).astype(f32)

    def enforce_boundary_3D(self, particle_type):
        a = self.a
        sel = (a["material"] == particle_type) & (a["is_dynamic"] != 0)
        pad = f32(self.h)
        x, v = a["x"].copy(), a["v"].copy()
        hi = x > (self.dom - pad)[None, :]
        lo = x <= pad
        n = hi.astype(f32) - lo.astype(f32)
        xn = np.where(hi, (self.dom - pad)[None, :], x)
        xn = np.where(lo, pad, xn)
        ln = np.sqrt((n * n).sum(axis=1, dtype=f32))
        hit = sel & (ln > f32(1e-6))
        with np.errstate(divide="ignore", invalid="ignore"):
            vec = n / ln[:, None]
        vd = (v * vec).sum(axis=1, dtype=f32)
        vn = v - f32(1.5) * vd[:, None] * vec
        a["x"] = np.where(sel[:, None], xn, x).astype(f32)
        a["v"] = np.where(hit[:, None], vn, v).astype(f32)

    # ---- DFSPH (DFSPH.py:116-221): the three neighbour sums, no solver loops ----
    def dfsph_factor(self):
        """DFSPH.py:116-154: -1 / (sum_{fluid j} |m_V_j gradW_ij|^2 + |sum_j m_V_j gradW_ij|^2), 0 if the sum <= 1e-6."""
        a = self.a
        rvec, rn, nb = self._pairs()
        g = a["m_V"][None, :, None] * _gradW(rvec, rn, self.h, self.kd)
        g = np.where(nb[..., None], g, f32(0))
        fluid = a["material"] == 1
        sq = (g * g).sum(axis=2, dtype=f32)
        s = np.where(fluid[None, :], sq, f32(0)).sum(axis=1, dtype=f32)
        gi = g.sum(axis=1, dtype=f32)
        s = s + (gi * gi).sum(axis=1, dtype=f32)
        with np.errstate(divide="ignore"):
            fac = np.where(s > f32(1e-6), f32(-1.0) / s, f32(0))
        return np.where(fluid, fac, f32(0)).astype(f32)

    def dfsph_velocity_divergence(self):
        """sum_j m_V_j (v_i - v_j) . gradW_ij and the neighbour count (DFSPH.py:183-197, 212-221)."""
        a = self.a
        rvec, rn, nb = self._pairs()
        gw = _gradW(rvec, rn, self.h, self.kd)
        vij = (a["v"][:, None, :] - a["v"][None, :, :]).astype(f32)
        term = a["m_V"][None, :] * (vij * gw).sum(axis=2, dtype=f32)
        return np.where(nb, term, f32(0)).sum(axis=1, dtype=f32), nb.sum(axis=1)

    def dfsph_density_change(self):
        div, cnt = self.dfsph_velocity_divergence()
        adv = np.where(cnt < 20, f32(0), np.maximum(div, f32(0)))
        return np.where(self.a["material"] == 1, adv, f32(0)).astype(f32)

    def dfsph_density_adv(self):
        div, _ = self.dfsph_velocity_divergence()
        adv = np.maximum(self.a["density"] / self.rho0 + self.dt * div, f32(1.0))
        return np.where(self.a["material"] == 1, adv, f32(0)).astype(f32)

    def step(self):
        self.boundary_volume(dynamic=True)
        self.compute_densities()
        self.compute_non_pressure_forces()
        self.compute_pressure_forces()
        self.advect()
        self.enforce_boundary_3D(1)
