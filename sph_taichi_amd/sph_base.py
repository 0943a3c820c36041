"""SPHBase with the reference's method surface (/root/reference/sph_base.py),
every method a thin wrapper over one C-ABI call of libsph_hip.so.  The smoothing
kernels (sph_base.py:23-68) live in csrc/sph_internal.h (sph_W / sph_gradW)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .field import HostScalar


class SPHBase:
    def __init__(self, particle_system):
        self.ps = particle_system
        self.g = np.array(self.ps.cfg.get_cfg("gravitation"))     # sph_base.py:13
        self.viscosity = 0.01                                       # sph_base.py:15
        self.density_0 = self.ps.cfg.get_cfg("density0")            # sph_base.py:18
        self.dt = HostScalar(1e-4, float, on_set=self._on_dt)       # sph_base.py:20-21
        self._params_pushed = False

    def _on_dt(self, value):
        if getattr(self.ps, "_ctx", None):
            self.ps._call("sph_set_dt", C.c_float(value))

    def _push(self):
        self.ps._push_solver_params(self)
        self._params_pushed = True

    def _dynamic_ids(self):
        ids = [i for i in self.ps.object_id_rigid_body if self.ps.object_collection[i]["isDynamic"]]
        arr = (C.c_int32 * max(len(ids), 1))(*ids)
        return arr, len(ids)

    # ---- sph_base.py:80-113 ------------------------------------------------
    def initialize(self):
        self._push()
        self.ps.initialize_particle_system()
        # A restarted ParticleSystem (state=...) holds displaced bodies: their REST centre is that of x_0, which is what the
        # reference's compute_com(x) gives at this point of an uninterrupted run (x_0 == x, sph_base.py:80-90, 182-192).
        from . import _lib
        restarted = getattr(self.ps, "_restarted", False)
        if restarted:
            self.ps.set_option(_lib.OPT_RIGID_SUMS_FROM_X0, 1)
        try:
            for r_obj_id in self.ps.object_id_rigid_body:
                self.compute_rigid_rest_cm(r_obj_id)
        finally:
            if restarted:
                self.ps.set_option(_lib.OPT_RIGID_SUMS_FROM_X0, 0)
        self.compute_static_boundary_volume()
        self.compute_moving_boundary_volume()

    def compute_rigid_rest_cm(self, object_id: int):
        self.ps._call("sph_compute_rigid_rest_cm", int(object_id))

    def compute_static_boundary_volume(self):
        self.ps._call("sph_compute_boundary_volume", 0)

    def compute_moving_boundary_volume(self):
        self.ps._call("sph_compute_boundary_volume", 1)

    def substep(self):
        pass

    # ---- sph_base.py:149-179 -------------------------------------------------
    def enforce_boundary_3D(self, particle_type: int):
        self.ps._call("sph_enforce_boundary_3D", int(particle_type))

    def enforce_boundary_2D(self, particle_type: int):
        raise NotImplementedError("2-D scenes are not supported (no reference scene is 2-D)")

    # ---- sph_base.py:195-260 -------------------------------------------------
    def compute_com_kernel(self, object_id: int):
        cm = (C.c_float * 3)()
        self.ps._call("sph_compute_com", int(object_id), cm)
        return np.array(list(cm), dtype=np.float32)

    def solve_constraints(self, object_id: int, want_R: bool = True):
        if not want_R:
            self.ps._call("sph_solve_constraints", int(object_id), None)
            return None
        R = (C.c_float * 9)()
        self.ps._call("sph_solve_constraints", int(object_id), R)
        return np.array(list(R), dtype=np.float32).reshape(3, 3)

    def solve_rigid_body(self):
        export_obj = self.ps.cfg.get_cfg("exportObj")
        for r_obj_id in self.ps.object_id_rigid_body:
            obj = self.ps.object_collection[r_obj_id]
            if obj["isDynamic"]:
                R = self.solve_constraints(r_obj_id, want_R=bool(export_obj))
                if export_obj:
                    # sph_base.py:253-257: move the export mesh with the body
                    cm = self.compute_com_kernel(r_obj_id)
                    ret = R.astype(np.float64) @ (obj["restPosition"] - obj["restCenterOfMass"]).T
                    obj["mesh"].vertices = cm + ret.T
                self.enforce_boundary_3D(self.ps.material_solid)

    def _reference_step(self):
        """sph_base.py:263-271 through the individual kernels."""
        self._push()    # the solver's knobs are plain attributes in the reference, read by every kernel: pick up any change
        self.ps.initialize_particle_system()
        self.compute_moving_boundary_volume()
        self.substep()
        self.solve_rigid_body()
        if self.ps.dim == 2:
            self.enforce_boundary_2D(self.ps.material_fluid)
        elif self.ps.dim == 3:
            self.enforce_boundary_3D(self.ps.material_fluid)

    def step(self):
        self._reference_step()
