"""ParticleSystem with the reference's attribute / method surface
(/root/reference/particle_system.py:10-495), backed by a HIP context.

What stays on the host (NumPy): scene ingestion (sph_taichi_amd/scene.py).
What moved to the GPU: every per-particle array (owned by libsph_hip's context,
exposed here as DeviceField objects with the reference's names) and the
neighbour-structure kernels update_grid_id / prefix sum / counting_sort.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib, scene as _scene
from .config_builder import SimConfig
from .field import DeviceField, HostScalar


class ParticleSystem:
    def __init__(self, config: SimConfig, GGUI=False, device: int = 0, stream=None, scene_dir: str | None = None,
                 verbose: bool = False, slab: dict | None = None, populate: bool = True, state: dict | None = None):
        """`slab` (multi-GPU, no reference counterpart) = dict(x_lo, x_hi, halo, capacity[, nx_slack]): this
        context owns the global cell layers [x_lo, x_hi) plus `halo` ghost layers on each side; `nx_slack` more
        layers are allocated so that a re-cut (`set_slab_window`) can widen the slab.

        `state` = {"x", "v"}: f32[N, 3] by persistent id -- a restart: the scene file says WHAT the particles are, `state`
        where they are and how fast (a slab rank keeps those whose restart position lies in its layers).

        `populate=False` stops where the reference's constructor stands after its allocations
        (particle_system.py:91-145): every field exists with `particle_max_num` zeroed rows and
        `particle_num[None] == 0`; the caller fills it with `add_cube` / `add_particles`, as the
        reference's own constructor does next (:148-211)."""
        self.cfg = config
        self.GGUI = GGUI
        self.slab = slab
        self._restarted = state is not None     # SPHBase.initialize / SlabSolver.initialize: rest cm from x_0, not from x
        x_filter = None
        if slab is not None:
            g0 = _scene.Geometry(config)
            nx = int(g0.grid_num[0])
            x_filter = lambda xs: ((_scene.x_layer_of(xs, g0.grid_size, nx) >= slab["x_lo"])
                                   & (_scene.x_layer_of(xs, g0.grid_size, nx) < slab["x_hi"]))
        sc = _scene.build_scene(config, base_dir=scene_dir, verbose=verbose, x_filter=x_filter, state=state)
        g = sc.geom
        self._scene = sc
        # ---- scalars, same names as the reference (particle_system.py:17-46) ----
        self.domain_start = g.domain_start
        self.domain_end = g.domain_end
        self.domain_size = g.domain_size
        self.dim = g.dim
        self.simulation_method = config.get_cfg("simulationMethod")
        self.material_solid = _scene.MATERIAL_SOLID
        self.material_fluid = _scene.MATERIAL_FLUID
        self.particle_radius = g.particle_radius
        self.particle_diameter = g.particle_diameter
        self.support_radius = g.support_radius
        self.m_V0 = g.m_V0
        self.grid_size = g.grid_size
        self.grid_num = g.grid_num
        self.padding = g.padding
        self.object_collection = sc.object_collection
        self.object_id_rigid_body = sc.object_id_rigid_body
        self.fluid_particle_num = sc.fluid_particle_num
        self.solid_particle_num = sc.solid_particle_num
        self.particle_max_num = sc.particle_max_num
        self.num_rigid_bodies = sc.num_rigid_bodies
        self.particle_num = HostScalar(0, int)
        if verbose:
            print("grid size: ", self.grid_num)
            print(f"Current particle num: {self.particle_num[None]}, Particle max num: {self.particle_max_num}")

        # ---- device context (replaces the ti.field allocations of :91-145) ----
        self._lib = _lib.load()
        self._params = self._make_params(sc)
        slack = int(slab.get("nx_slack", 0)) if slab is not None else 0
        self._params.grid_num[0] += slack            # the allocation; the window is set right below
        ctx = C.c_void_p()
        stream_ptr = C.c_void_p(int(stream)) if stream else None
        rc = self._lib.sph_create(C.byref(self._params), int(device), stream_ptr, C.byref(ctx))
        if rc != 0:
            msg = self._lib.sph_last_error(None)
            raise _lib.SphError(f"sph_create failed (rc={rc}): {msg.decode() if msg else ''}")
        self._ctx = ctx
        if slack:
            self._params.grid_num[0] -= slack
            self._call("sph_slab_set_window", int(self._params.cell_origin[0]), int(self._params.grid_num[0]))
        n = self.count if slab is not None else (lambda: self.particle_max_num)
        F = _lib
        mk = lambda fid, dt, vec=0, w=True, name="": DeviceField(self, fid, dt, n, vec, w, name)
        self.object_id = mk(F.F_OBJECT_ID, np.int32, name="object_id")
        self.x = mk(F.F_X, np.float32, 3, name="x")
        self.x_0 = mk(F.F_X_0, np.float32, 3, name="x_0")
        self.v = mk(F.F_V, np.float32, 3, name="v")
        self.acceleration = mk(F.F_ACCELERATION, np.float32, 3, name="acceleration")
        self.m_V = mk(F.F_M_V, np.float32, name="m_V")
        self.m = mk(F.F_M, np.float32, name="m")
        self.density = mk(F.F_DENSITY, np.float32, name="density")
        self.pressure = mk(F.F_PRESSURE, np.float32, name="pressure")
        self.material = mk(F.F_MATERIAL, np.int32, name="material")
        self.color = mk(F.F_COLOR, np.int32, 3, name="color")
        self.is_dynamic = mk(F.F_IS_DYNAMIC, np.int32, name="is_dynamic")
        self.grid_ids = mk(F.F_GRID_IDS, np.int32, w=False, name="grid_ids")
        self.pid = mk(F.F_PID, np.int32, name="pid")
        if self.simulation_method == 4:                       # particle_system.py:115-117
            self.dfsph_factor = mk(F.F_DFSPH_FACTOR, np.float32, name="dfsph_factor")
            self.density_adv = mk(F.F_DENSITY_ADV, np.float32, name="density_adv")
        self.grid_particles_num = DeviceField(self, F.F_GRID_PARTICLES_NUM, np.int32,
                                              lambda: int(np.prod(self._local_grid_num)), 0, False,
                                              "grid_particles_num")
        if self.num_rigid_bodies > 0:
            self.rigid_rest_cm = DeviceField(self, F.F_RIGID_REST_CM, np.float32, lambda: sc.n_objects, 3, True,
                                             "rigid_rest_cm")
        self.x_vis_buffer = None
        if self.GGUI:
            self.x_vis_buffer = np.zeros((self.particle_max_num, 3), dtype=np.float32)
            self.color_vis_buffer = np.zeros((self.particle_max_num, 3), dtype=np.float32)

        # ---- upload the initial particles (the reference's _add_particles, :260-284) ----
        if not populate:
            if slab is not None:
                raise ValueError("populate=False is a single-domain feature (a slab rank subsets the scene file)")
            return              # sph_create zero-fills every array (x = 0, material 0 = solid, static), like ti.field
        for name, arr in sc.arrays.items():
            if name == "pid" and slab is None:
                continue            # default pid = index at creation
            getattr(self, name).from_numpy(arr)
        self.particle_num[None] = self.particle_max_num

    # ------------------------------------------------------------------
    def _make_params(self, sc, solver=None):
        g = sc.geom
        cfg = self.cfg
        p = _lib.SphParams()
        if self.slab is None:
            p.n_particles = sc.particle_max_num
            p.capacity = max(sc.particle_max_num, 1)
            local_grid = [int(v) for v in g.grid_num]
            p.cell_origin = (C.c_int32 * 3)(0, 0, 0)
            p.cold_capacity = 0
        else:
            sl = self.slab
            p.n_particles = int(sc.arrays["x"].shape[0])
            p.capacity = max(int(sl["capacity"]), p.n_particles, 1)
            local_grid = [sl["x_hi"] - sl["x_lo"] + 2 * sl["halo"], int(g.grid_num[1]), int(g.grid_num[2])]
            p.cell_origin = (C.c_int32 * 3)(sl["x_lo"] - sl["halo"], 0, 0)
            p.cold_capacity = max(sc.particle_max_num, p.capacity)
        self._local_grid_num = local_grid
        p.grid_num = (C.c_int32 * 3)(*local_grid)
        p.n_objects = sc.n_objects
        p.grid_size = g.grid_size
        p.support_radius = g.support_radius
        p.particle_diameter = g.particle_diameter
        p.m_V0 = g.m_V0
        viscosity = getattr(solver, "viscosity", 0.01)                      # sph_base.py:15
        p.density_0 = getattr(solver, "density_0", cfg.get_cfg("density0") or 1000.0)
        p.stiffness = getattr(solver, "stiffness", cfg.get_cfg("stiffness") or 50000.0)
        p.exponent = getattr(solver, "exponent", cfg.get_cfg("exponent") or 7.0)
        p.viscosity = viscosity
        p.surface_tension = getattr(solver, "surface_tension", 0.01)       # WCSPH.py:15
        p.dt = solver.dt[None] if solver is not None else (cfg.get_cfg("timeStepSize") or 1e-4)
        grav = getattr(solver, "g", None)
        if grav is None:
            grav = cfg.get_cfg("gravitation") or [0.0, -9.81, 0.0]
        p.g = (C.c_float * 3)(*[float(v) for v in grav])
        p.domain_size = (C.c_float * 3)(*[float(v) for v in g.domain_size])
        p.padding = g.padding
        p.wall_hi = (C.c_float * 3)(*[float(v) - g.padding for v in g.domain_size])   # f64 fold, then f32
        kc = _scene.kernel_constants(g.support_radius, viscosity, g.dim)
        p.k_w, p.k_dw, p.visc_d_nu, p.visc_eps = kc["k_w"], kc["k_dw"], kc["visc_d_nu"], kc["visc_eps"]
        return p

    def set_slab_window(self, x_lo, x_hi):
        """Re-cut: this context now owns the global cell layers [x_lo, x_hi) (effective at the next sort)."""
        sl = self.slab
        nx = int(x_hi) - int(x_lo) + 2 * sl["halo"]
        self._call("sph_slab_set_window", int(x_lo) - sl["halo"], nx)   # fails if wider than the allocation (nx_slack)
        sl["x_lo"], sl["x_hi"] = int(x_lo), int(x_hi)
        self._local_grid_num[0] = nx
        self._params.grid_num[0] = nx
        self._params.cell_origin[0] = sl["x_lo"] - sl["halo"]

    def _push_solver_params(self, solver):
        self._params = self._make_params(self._scene, solver)
        self._call("sph_set_params", C.byref(self._params))

    def _call(self, name, *args):
        rc = getattr(self._lib, name)(self._ctx, *args)
        _lib.check(self._lib, self._ctx, rc, name)

    def count(self) -> int:
        """Live particle count of this context (== particle_max_num on a single GPU)."""
        n = C.c_int32()
        self._call("sph_get_particle_count", C.byref(n))
        return int(n.value)

    def set_option(self, option: int, value: int):
        self._call("sph_set_option", int(option), int(value))

    def get_option(self, option: int) -> int:
        v = C.c_int32()
        rc = self._lib.sph_get_option(self._ctx, int(option), C.byref(v))
        _lib.check(self._lib, self._ctx, rc, "sph_get_option")
        return int(v.value)

    def sync(self):
        self._call("sph_sync")

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.sph_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- reference methods ------------------------------------------------
    def build_solver(self):
        """particle_system.py:214-221."""
        solver_type = self.cfg.get_cfg("simulationMethod")
        if solver_type == 0:
            from .WCSPH import WCSPHSolver
            return WCSPHSolver(self)
        if solver_type == 4:
            from .DFSPH import DFSPHSolver          # (on a slab rank the loops are driven by distributed.SlabSolver)
            return DFSPHSolver(self)
        raise NotImplementedError(f"Solver type {solver_type} has not been implemented.")

    def update_grid_id(self):
        """particle_system.py:311-320."""
        self._call("sph_update_grid_id")

    def prefix_sum(self):
        """self.prefix_sum_executor.run(self.grid_particles_num)  (particle_system.py:374)."""
        self._call("sph_prefix_sum")

    def counting_sort(self):
        """particle_system.py:322-369."""
        self._call("sph_counting_sort")

    def initialize_particle_system(self):
        """particle_system.py:372-375."""
        self.update_grid_id()
        self.prefix_sum()
        self.counting_sort()

    def add_particles(self, object_id, new_particles_num, new_particles_positions, new_particles_velocity,
                      new_particle_density, new_particle_pressure, new_particles_material, new_particles_is_dynamic,
                      new_particles_color):
        """particle_system.py:237-284: rows [particle_num, particle_num + n) of every field take the new particles
        with add_particle's derived values (:222-235: x_0 = x, m_V = m_V0, m = m_V0 * density, acceleration left as
        it is), then particle_num += n.  The fields are fixed-size (`particle_max_num` rows, :101-113): the
        reference writes past their end when more is added than the scene file announced; here that raises.
        Host-side read-modify-write of whole arrays: scene set-up, not a per-step path."""
        n = int(new_particles_num)
        p0 = int(self.particle_num[None])
        if n < 0 or p0 + n > self.particle_max_num:
            raise ValueError(f"add_particles: {p0} + {n} particles exceed particle_max_num = {self.particle_max_num} "
                             "(the fields are sized from the scene file, particle_system.py:52-83)")
        if n == 0:
            return
        if self.slab is not None:
            raise ValueError("add_particles on a slab rank: build the scene before it is cut")
        pos = np.asarray(new_particles_positions, dtype=np.float32).reshape(n, self.dim)
        dens = np.asarray(new_particle_density, dtype=np.float32).reshape(n)
        rows = {
            "object_id": np.full(n, int(object_id), dtype=np.int32),
            "x": pos, "x_0": pos,
            "v": np.asarray(new_particles_velocity, dtype=np.float32).reshape(n, self.dim),
            "density": dens,
            "m_V": np.full(n, self.m_V0, dtype=np.float32),
            "m": (np.float32(self.m_V0) * dens).astype(np.float32),
            "pressure": np.asarray(new_particle_pressure, dtype=np.float32).reshape(n),
            "material": np.asarray(new_particles_material, dtype=np.int32).reshape(n),
            "is_dynamic": np.asarray(new_particles_is_dynamic, dtype=np.int32).reshape(n),
            "color": np.asarray(new_particles_color, dtype=np.int32).reshape(n, 3),
        }
        # (the persistent id of a new row is the id that row already holds: pid is a permutation of the rows at all
        # times -- identity before the first sort -- so x_0 / colour, which are keyed by it, land where the row reads them)
        for name, val in rows.items():
            f = getattr(self, name)
            a = f.to_numpy()
            a[p0:p0 + n] = val
            f.from_numpy(a)
        self.particle_num[None] = p0 + n

    def add_cube(self, object_id, lower_corner, cube_size, material, is_dynamic, color=(0, 0, 0), density=None,
                 pressure=None, velocity=None):
        """particle_system.py:458-495: a lattice of particle_diameter spacing from lower_corner (np.arange per axis,
        'ij' meshgrid, x slowest), constant material / is_dynamic / colour / density (1000 by default) / pressure (0)
        / velocity (0), appended through add_particles."""
        pos = _scene.cube_positions(lower_corner, cube_size, self.particle_diameter, self.dim)
        n = pos.shape[0]
        vel = np.zeros_like(pos) if velocity is None else np.tile(np.asarray(velocity, dtype=np.float32), (n, 1))
        self.add_particles(object_id, n, pos, vel,
                           np.full(n, density if density is not None else 1000.0, dtype=np.float32),
                           np.full(n, pressure if pressure is not None else 0.0, dtype=np.float32),
                           np.full(n, material, dtype=np.int32), np.full(n, is_dynamic, dtype=np.int32),
                           np.tile(np.asarray(color, dtype=np.int32), (n, 1)))
        return n

    def compute_cube_particle_num(self, start, end):
        return _scene.compute_cube_particle_num(start, end, self.particle_diameter, self.dim)

    def dump(self, obj_id):
        """particle_system.py:409-418."""
        mask = (self.object_id.to_numpy() == obj_id).nonzero()
        return {"position": self.x.to_numpy()[mask], "velocity": self.v.to_numpy()[mask]}

    # ---- state checkpoint (the reference has none) ----------------------
    _STATE_FIELDS = ("object_id", "x", "x_0", "v", "acceleration", "m_V", "m", "density", "pressure", "material",
                     "color", "is_dynamic", "pid")

    def save_state(self, path: str, **meta):
        """Every per-particle array in the current (cell-sorted) order + rigid_rest_cm, as one .npz.  Restoring it
        into a ParticleSystem built from the same scene continues the run exactly where it stopped."""
        data = {}
        for f in self._STATE_FIELDS:
            try:
                data[f] = getattr(self, f).to_numpy()
            except _lib.SphError:
                # a slab rank right after a fused sph_slab_forces: the interior accelerations were consumed in the sweep's
                # finish and never written out (sph_download refuses stale values).  The field is dead across steps --
                # every step rewrites it before reading it -- so the checkpoint carries zeros.
                if f != "acceleration":
                    raise
                data[f] = np.zeros((self.count() if self.slab is not None else self.particle_max_num, 3), dtype=np.float32)
        if self.num_rigid_bodies > 0:
            data["rigid_rest_cm"] = self.rigid_rest_cm.to_numpy()
        data["particle_max_num"] = np.int64(self.particle_max_num)
        for k, v in meta.items():
            data["meta_" + k] = np.asarray(v)
        np.savez_compressed(path, **data)

    def load_state(self, path: str):
        z = np.load(path)
        if int(z["particle_max_num"]) != self.particle_max_num:
            raise ValueError(f"checkpoint holds {int(z['particle_max_num'])} particles, this scene {self.particle_max_num}")
        self.pid.from_numpy(z["pid"])                  # first: x_0 / color are keyed by the persistent id
        for f in self._STATE_FIELDS:
            if f != "pid":
                getattr(self, f).from_numpy(z[f])
        if "rigid_rest_cm" in z.files and self.num_rigid_bodies > 0:
            self.rigid_rest_cm.from_numpy(z["rigid_rest_cm"])
        return {k[5:]: z[k] for k in z.files if k.startswith("meta_")}

    def copy_to_vis_buffer(self, invisible_objects=[]):
        """particle_system.py:392-407 (host copy; there is no GGUI here)."""
        assert self.GGUI
        oid = self.object_id.to_numpy()
        x = self.x.to_numpy()
        col = self.color.to_numpy()
        if len(invisible_objects) != 0:
            self.x_vis_buffer[:] = 0.0
            self.color_vis_buffer[:] = 0.0
        vis = ~np.isin(oid, list(invisible_objects)) & np.isin(oid, list(self.object_collection.keys()))
        self.x_vis_buffer[vis] = x[vis]
        self.color_vis_buffer[vis] = col[vis] / 255.0
