"""DFSPHSolver with the reference's surface (/root/reference/DFSPH.py:5-408).
step() runs SPHBase.step() with DFSPHSolver.substep() inside libsph_hip (one C
call per batch of steps: sph_dfsph_step); the per-kernel methods and the two
solver loops stay individually callable, as thin wrappers over one C-ABI call each.
"""
from __future__ import annotations

import ctypes as C

from . import _lib
from .sph_base import SPHBase


class DFSPHSolver(SPHBase):
    def __init__(self, particle_system):
        super().__init__(particle_system)
        self.surface_tension = 0.01                              # DFSPH.py:9
        self.dt[None] = self.ps.cfg.get_cfg("timeStepSize")      # DFSPH.py:10
        self.enable_divergence_solver = True                     # DFSPH.py:12
        self.m_max_iterations_v = 100                            # DFSPH.py:14
        self.m_max_iterations = 100                              # DFSPH.py:15
        self.m_eps = 1e-5                                        # DFSPH.py:17
        self.max_error_V = 0.1                                   # DFSPH.py:19
        self.max_error = 0.05                                    # DFSPH.py:20
        self.verbose = False                                     # the reference prints one line per solve
        self._push()

    # ---- parameters ---------------------------------------------------------
    def _push(self):
        super()._push()
        p = _lib.SphDfsphParams()
        p.enable_divergence_solver = int(bool(self.enable_divergence_solver))
        p.m_max_iterations_v = int(self.m_max_iterations_v)
        p.m_max_iterations = int(self.m_max_iterations)
        p.fluid_particle_num = int(self.ps.fluid_particle_num)
        p.m_eps = float(self.m_eps)
        p.max_error_V = float(self.max_error_V)
        p.max_error = float(self.max_error)
        self.ps._call("sph_dfsph_set_params", C.byref(p))

    def stats(self):
        """Iteration counts / average density errors of the last solves (what DFSPH.py:258, :353 print)."""
        s = _lib.SphDfsphStats()
        self.ps._call("sph_dfsph_get_stats", C.byref(s))
        return {f: getattr(s, f) for f, _ in s._fields_}

    # ---- kernels ------------------------------------------------------------
    def compute_densities(self):
        self.ps._call("sph_dfsph_compute_densities")             # DFSPH.py:37-47

    def compute_non_pressure_forces(self):
        self.ps._call("sph_dfsph_compute_non_pressure_forces")   # DFSPH.py:100-112

    def advect(self):
        self.ps._call("sph_dfsph_advect")                        # DFSPH.py:100-107

    def compute_DFSPH_factor(self):
        self.ps._call("sph_dfsph_compute_DFSPH_factor")          # DFSPH.py:116-139

    def compute_density_change(self):
        self.ps._call("sph_dfsph_compute_density_change")        # DFSPH.py:157-180

    def compute_density_adv(self):
        self.ps._call("sph_dfsph_compute_density_adv")           # DFSPH.py:200-209

    def compute_density_error(self, offset: float) -> float:
        out = C.c_float()
        self.ps._call("sph_dfsph_compute_density_error", C.c_float(offset), C.byref(out))   # DFSPH.py:224-230
        return float(out.value)

    def multiply_time_step(self, field, time_step: float):
        """DFSPH.py:233-237; the reference only ever passes ps.dfsph_factor."""
        if field is not self.ps.dfsph_factor:
            raise NotImplementedError("multiply_time_step is only defined on ps.dfsph_factor (all the reference uses)")
        self.ps._call("sph_dfsph_multiply_time_step", C.c_float(time_step))

    def divergence_solver_iteration_kernel(self):
        self.ps._call("sph_dfsph_divergence_solver_iteration_kernel")   # DFSPH.py:285-296

    def pressure_solve_iteration_kernel(self):
        self.ps._call("sph_dfsph_pressure_solve_iteration_kernel")      # DFSPH.py:356-367

    def predict_velocity(self):
        self.ps._call("sph_dfsph_predict_velocity")              # DFSPH.py:388-394

    # ---- solver loops ---------------------------------------------------------
    def divergence_solver_iteration(self):
        """DFSPH.py:278-283."""
        self.divergence_solver_iteration_kernel()
        self.compute_density_change()
        density_err = self.compute_density_error(0.0)
        return density_err / self.ps.fluid_particle_num

    def pressure_solve_iteration(self):
        """DFSPH.py:350-354."""
        self.pressure_solve_iteration_kernel()
        self.compute_density_adv()
        density_err = self.compute_density_error(self.density_0)
        return density_err / self.ps.fluid_particle_num

    def divergence_solve(self):
        """DFSPH.py:240-275 (the loop runs inside the library)."""
        self.ps._call("sph_dfsph_divergence_solve")
        if self.verbose:
            s = self.stats()
            print(f"DFSPH - iteration V: {s['iterations_v']} Avg density err: {s['avg_density_err_v']}")

    def pressure_solve(self):
        """DFSPH.py:324-348 (the loop runs inside the library)."""
        self.ps._call("sph_dfsph_pressure_solve")
        if self.verbose:
            s = self.stats()
            print(f"DFSPH - iterations: {s['iterations']} Avg density Err: {s['avg_density_err']:.4f}")

    def substep(self):
        """DFSPH.py:400-408."""
        self.compute_densities()
        self.compute_DFSPH_factor()
        if self.enable_divergence_solver:
            self.divergence_solve()
        self.compute_non_pressure_forces()
        self.predict_velocity()
        self.pressure_solve()
        self.advect()

    def step(self, n_steps: int = 1):
        """SPHBase.step() (sph_base.py:263-271) with the device-side loop, unless a subclass overrides substep()
        or the scene asks for per-step OBJ export (sph_base.py:251-257)."""
        if type(self).substep is not DFSPHSolver.substep or self.ps.cfg.get_cfg("exportObj") or self.verbose:
            for _ in range(n_steps):
                self._reference_step()
            return
        self._push()    # the solver knobs are plain attributes, as in the reference; pick up any change
        ids, n = self._dynamic_ids()
        self.ps._call("sph_dfsph_step", int(n_steps), ids, n)
