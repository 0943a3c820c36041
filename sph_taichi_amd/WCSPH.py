"""WCSPHSolver with the reference's surface (/root/reference/WCSPH.py:5-156).
step() runs the whole SPHBase.step() sequence on the device in one C call
(sph_step); the per-kernel methods stay individually callable."""
from __future__ import annotations

from .sph_base import SPHBase


class WCSPHSolver(SPHBase):
    def __init__(self, particle_system):
        super().__init__(particle_system)
        self.exponent = self.ps.cfg.get_cfg("exponent")          # WCSPH.py:9-10
        self.stiffness = self.ps.cfg.get_cfg("stiffness")        # WCSPH.py:12-13
        self.surface_tension = 0.01                              # WCSPH.py:15
        self.dt[None] = self.ps.cfg.get_cfg("timeStepSize")      # WCSPH.py:16
        self._push()

    def compute_densities(self):
        self.ps._call("sph_compute_densities")                   # WCSPH.py:33-43

    def compute_non_pressure_forces(self):
        self.ps._call("sph_compute_non_pressure_forces")         # WCSPH.py:128-140

    def compute_pressure_forces(self):
        self.ps._call("sph_compute_pressure_forces")             # WCSPH.py:70-85

    def advect(self):
        self.ps._call("sph_advect")                              # WCSPH.py:143-149

    def substep(self):
        """WCSPH.py:152-156."""
        self.compute_densities()
        self.compute_non_pressure_forces()
        self.compute_pressure_forces()
        self.advect()

    def step(self, n_steps: int = 1):
        """SPHBase.step() (sph_base.py:263-271).  Uses the device-side loop unless a
        subclass overrides substep() or the scene asks for per-step OBJ export
        (which needs R and cm on the host every step, sph_base.py:251-257)."""
        if type(self).substep is not WCSPHSolver.substep or self.ps.cfg.get_cfg("exportObj"):
            for _ in range(n_steps):
                self._reference_step()
            return
        self._push()    # stiffness / exponent / viscosity / surface_tension / g are plain attributes, as in the reference
        ids, n = self._dynamic_ids()
        self.ps._call("sph_step", int(n_steps), ids, n)
