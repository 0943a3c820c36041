"""ctypes front-end of the CPU oracle (oracle/sph_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of sph_oracle.c.  Imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the
product package.

The per-particle arrays live in NumPy (this module owns them) and are handed to
C as raw pointers, laid out exactly like the reference's Taichi fields
(/root/reference/particle_system.py:101-113): vectors are [N,3] f32 row-major,
scalars [N], ints i32.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_F3 = C.c_float * 3
_I3 = C.c_int32 * 3
_pf = C.POINTER(C.c_float)
_pi = C.POINTER(C.c_int32)

_STATE_ARRAYS_F = ["x", "x_0", "v", "acceleration", "m_V", "m", "density", "pressure"]
_STATE_ARRAYS_I = ["object_id", "material", "color", "is_dynamic"]
_VEC = {"x", "x_0", "v", "acceleration", "color"}


class _State(C.Structure):
    _fields_ = [
        ("N", C.c_int32), ("grid_num", _I3), ("G", C.c_int32), ("n_objects", C.c_int32),
        ("grid_size", C.c_float), ("support_radius", C.c_float), ("particle_diameter", C.c_float),
        ("m_V0", C.c_float), ("density_0", C.c_float), ("stiffness", C.c_float),
        ("exponent", C.c_float), ("viscosity", C.c_float), ("surface_tension", C.c_float),
        ("dt", C.c_float), ("g", _F3), ("domain_size", _F3), ("padding", C.c_float), ("wall_hi", _F3),
        ("k_w", C.c_float), ("k_dw", C.c_float), ("visc_d_nu", C.c_float), ("visc_eps", C.c_float),
        ("omp_threads", C.c_int32), ("rigid_sums_f64", C.c_int32),
        ("object_id", _pi), ("x", _pf), ("x_0", _pf), ("v", _pf), ("acceleration", _pf),
        ("m_V", _pf), ("m", _pf), ("density", _pf), ("pressure", _pf),
        ("material", _pi), ("color", _pi), ("is_dynamic", _pi),
        ("object_id_buffer", _pi), ("x_buffer", _pf), ("x_0_buffer", _pf), ("v_buffer", _pf),
        ("acceleration_buffer", _pf), ("m_V_buffer", _pf), ("m_buffer", _pf),
        ("density_buffer", _pf), ("pressure_buffer", _pf), ("material_buffer", _pi),
        ("color_buffer", _pi), ("is_dynamic_buffer", _pi),
        ("grid_ids", _pi), ("grid_ids_buffer", _pi), ("grid_ids_new", _pi),
        ("grid_particles_num", _pi), ("grid_particles_num_temp", _pi),
        ("rigid_rest_cm", _pf),
        ("pid", _pi), ("pid_buffer", _pi),
        # DFSPH (simulationMethod 4)
        ("simulation_method", C.c_int32), ("fluid_particle_num", C.c_int32), ("enable_divergence_solver", C.c_int32),
        ("m_max_iterations_v", C.c_int32), ("m_max_iterations", C.c_int32), ("m_eps", C.c_float),
        ("max_error_V", C.c_double), ("max_error", C.c_double),
        ("last_iterations_v", C.c_int32), ("last_iterations", C.c_int32),
        ("last_avg_err_v", C.c_double), ("last_avg_err", C.c_double),
        ("dfsph_factor", _pf), ("density_adv", _pf), ("dfsph_factor_buffer", _pf), ("density_adv_buffer", _pf),
        ("perturb", C.c_int32),      # test switch: bit 0 reversed neighbour traversal, bit 1 integer Tait exponent by multiplication
    ]


def build(force: bool = False, timing: bool = False) -> str:
    """Compile oracle/libsph_oracle.so with gcc (oracle/Makefile).  timing=True: the -O3 -march=native build that
    bench.py times as cpu_baseline (always rebuilt on the host it runs on; never used as a checker)."""
    name = "libsph_oracle_timing.so" if timing else "libsph_oracle.so"
    so = os.path.join(_HERE, name)
    src = os.path.join(_HERE, "sph_oracle.c")
    if force or timing or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-s", "-B", name], check=True)
    return so


_LIB_TIMING = None


def lib(timing: bool = False):
    global _LIB, _LIB_TIMING
    if timing:
        if _LIB_TIMING is None:
            _LIB_TIMING = _bind(C.CDLL(build(timing=True)))
        return _LIB_TIMING
    if _LIB is None:
        _LIB = _bind(C.CDLL(build()))
    return _LIB


def _bind(L):
    ps = C.POINTER(_State)
    for name in ("update_grid_id", "prefix_sum", "counting_sort", "initialize_particle_system",
                 "compute_static_boundary_volume", "compute_moving_boundary_volume",
                 "compute_densities", "compute_non_pressure_forces", "compute_pressure_forces",
                 "advect", "substep", "dfsph_compute_densities", "dfsph_compute_non_pressure_forces",
                 "dfsph_compute_factor", "dfsph_compute_density_change", "dfsph_compute_density_adv",
                 "dfsph_divergence_solver_iteration_kernel", "dfsph_pressure_solve_iteration_kernel",
                 "dfsph_divergence_solve", "dfsph_pressure_solve", "dfsph_predict_velocity", "dfsph_advect",
                 "dfsph_substep"):
        f = getattr(L, "oracle_" + name)
        f.argtypes = [ps]
        f.restype = None
    L.oracle_dfsph_compute_density_error.argtypes = [ps, C.c_float]
    L.oracle_dfsph_compute_density_error.restype = C.c_float
    L.oracle_dfsph_multiply_time_step.argtypes = [ps, C.c_float]
    L.oracle_dfsph_multiply_time_step.restype = None
    for name in ("dfsph_divergence_solver_iteration", "dfsph_pressure_solve_iteration"):
        f = getattr(L, "oracle_" + name)
        f.argtypes = [ps]
        f.restype = C.c_double
    L.oracle_enforce_boundary_3D.argtypes = [ps, C.c_int32]
    L.oracle_compute_rigid_rest_cm.argtypes = [ps, C.c_int32]
    L.oracle_solve_constraints.argtypes = [ps, C.c_int32, _pf]
    L.oracle_solve_rigid_body.argtypes = [ps, _pi, C.c_int32]
    L.oracle_step.argtypes = [ps, _pi, C.c_int32, C.c_int32, C.POINTER(C.c_double)]
    L.oracle_initialize.argtypes = [ps, _pi, C.c_int32]
    L.oracle_cubic_kernel.argtypes = [ps, C.c_float]
    L.oracle_cubic_kernel.restype = C.c_float
    L.oracle_cubic_kernel_derivative.argtypes = [ps, _pf, _pf]
    L.oracle_polar_rotation.argtypes = [_pf, _pf]
    L.oracle_sizeof_state.restype = C.c_int32
    L.oracle_max_threads.restype = C.c_int32
    assert L.oracle_sizeof_state() == C.sizeof(_State), "OracleState layout mismatch"
    return L


def kernel_constants(support_radius: float, viscosity: float = 0.01, dim: int = 3):
    """f64 constant folding exactly as the reference's Python scope does it
    (sph_base.py:27-35, 50-57; WCSPH.py:104,113), to be cast to f32 at use."""
    k = 8 / np.pi
    k /= support_radius ** dim
    k_dw = 6.0 * (8 / np.pi) / support_radius ** dim
    d = 2 * (dim + 2)
    return dict(k_w=k, k_dw=k_dw, visc_d_nu=d * viscosity, visc_eps=0.01 * support_radius ** 2)


class Oracle:
    """One simulation state + the reference's per-kernel methods.

    params: dict with particle_radius, domain_size[3], density_0, stiffness,
    exponent, dt, g[3]  (values as the scene JSON gives them); simulation_method 4 selects the DFSPH
    step (DFSPH.py) with its solver knobs at the reference's defaults (DFSPH.py:12-20).
    arrays: dict of the reference's per-particle arrays (any missing one is
    zero-initialised).  rigid_body_ids: ids in ps.object_id_rigid_body;
    dynamic_ids: those with isDynamic.
    """

    def __init__(self, params: dict, arrays: dict, n_objects: int = 1,
                 rigid_body_ids=(), dynamic_ids=(), omp_threads: int = 1, rigid_sums_f64: bool = False,
                 timing_build: bool = False, perturb: int = 0):
        L = lib(timing=timing_build)   # timing_build: bench.py's cpu_baseline only -- never a checker
        self.L = L
        N = int(np.asarray(arrays["x"]).shape[0])
        self.N = N
        r = float(params["particle_radius"])
        d = 2 * r                                  # particle_system.py:36
        h = r * 4.0                                # particle_system.py:37
        dom = np.asarray(params["domain_size"], dtype=np.float64)
        grid_num = np.ceil(dom / h).astype(int)    # particle_system.py:44
        self.grid_num = grid_num
        G = int(grid_num[0] * grid_num[1] * grid_num[2])
        self.G = G
        s = _State()
        s.N = N
        s.grid_num = _I3(*[int(v) for v in grid_num])
        s.G = G
        s.n_objects = int(n_objects)
        s.grid_size = h
        s.support_radius = h
        s.particle_diameter = d
        s.m_V0 = 0.8 * d ** 3                      # particle_system.py:38
        s.density_0 = float(params["density_0"])
        s.stiffness = float(params["stiffness"])
        s.exponent = float(params["exponent"])
        s.viscosity = float(params.get("viscosity", 0.01))
        s.surface_tension = float(params.get("surface_tension", 0.01))
        s.dt = float(params["dt"])
        s.g = _F3(*[float(v) for v in params["g"]])
        s.domain_size = _F3(*[float(v) for v in dom])
        s.padding = h
        s.wall_hi = _F3(*[float(v) - h for v in dom])
        kc = kernel_constants(h, s.viscosity)
        s.k_w, s.k_dw, s.visc_d_nu, s.visc_eps = kc["k_w"], kc["k_dw"], kc["visc_d_nu"], kc["visc_eps"]
        s.omp_threads = int(omp_threads)
        s.rigid_sums_f64 = int(bool(rigid_sums_f64))
        s.perturb = int(perturb)     # test switch (see sph_oracle.c): ulp-level perturbations of the same formulas
        self.a = {}

        def alloc(name, dtype, vec):
            shape = (N, 3) if vec else (N,)
            src = arrays.get(name)
            arr = np.zeros(shape, dtype=dtype) if src is None else np.ascontiguousarray(src, dtype=dtype).reshape(shape).copy()
            self.a[name] = arr
            setattr(s, name, arr.ctypes.data_as(_pf if dtype == np.float32 else _pi))
            buf = np.zeros(shape, dtype=dtype)
            self.a[name + "_buffer"] = buf
            setattr(s, name + "_buffer", buf.ctypes.data_as(_pf if dtype == np.float32 else _pi))

        for n in _STATE_ARRAYS_F:
            alloc(n, np.float32, n in _VEC)
        for n in _STATE_ARRAYS_I:
            alloc(n, np.int32, n in _VEC)
        if arrays.get("pid") is None:
            arrays = dict(arrays, pid=np.arange(N, dtype=np.int32))
        alloc("pid", np.int32, False)
        for n, size in (("grid_ids", N), ("grid_ids_buffer", N), ("grid_ids_new", N),
                        ("grid_particles_num", G), ("grid_particles_num_temp", G)):
            arr = np.zeros(size, dtype=np.int32)
            self.a[n] = arr
            setattr(s, n, arr.ctypes.data_as(_pi))
        # DFSPH state (particle_system.py:115-117, 134-135) and solver knobs (DFSPH.py:12-20)
        s.simulation_method = int(params.get("simulation_method", 0))
        mat = self.a["material"]
        s.fluid_particle_num = int(params.get("fluid_particle_num", int((mat == 1).sum())))
        s.enable_divergence_solver = int(params.get("enable_divergence_solver", True))
        s.m_max_iterations_v = int(params.get("m_max_iterations_v", 100))
        s.m_max_iterations = int(params.get("m_max_iterations", 100))
        s.m_eps = float(params.get("m_eps", 1e-5))
        s.max_error_V = float(params.get("max_error_V", 0.1))
        s.max_error = float(params.get("max_error", 0.05))
        for n in ("dfsph_factor", "density_adv", "dfsph_factor_buffer", "density_adv_buffer"):
            arr = np.zeros(N, dtype=np.float32)
            self.a[n] = arr
            setattr(s, n, arr.ctypes.data_as(_pf))
        self.a["rigid_rest_cm"] = np.zeros((max(int(n_objects), 1), 3), dtype=np.float32)
        s.rigid_rest_cm = self.a["rigid_rest_cm"].ctypes.data_as(_pf)
        self.s = s
        self.rigid_body_ids = np.asarray(list(rigid_body_ids), dtype=np.int32)
        self.dynamic_ids = np.asarray(list(dynamic_ids), dtype=np.int32)

    # -- accessors -----------------------------------------------------
    def __getitem__(self, name):
        return self.a[name]

    def snapshot(self, names=None):
        names = names or (_STATE_ARRAYS_F + _STATE_ARRAYS_I + ["pid", "grid_ids", "grid_particles_num"])
        return {n: self.a[n].copy() for n in names}

    def by_pid(self, name):
        """Array `name` re-ordered by persistent particle id."""
        out = np.empty_like(self.a[name])
        out[self.a["pid"]] = self.a[name]
        return out

    def set_threads(self, n):
        self.s.omp_threads = int(n)

    def _p(self):
        return C.byref(self.s)

    def _ids(self, arr):
        return arr.ctypes.data_as(_pi), int(arr.size)

    # -- the reference's methods ---------------------------------------
    def update_grid_id(self): self.L.oracle_update_grid_id(self._p())
    def prefix_sum(self): self.L.oracle_prefix_sum(self._p())
    def counting_sort(self): self.L.oracle_counting_sort(self._p())
    def initialize_particle_system(self): self.L.oracle_initialize_particle_system(self._p())
    def compute_static_boundary_volume(self): self.L.oracle_compute_static_boundary_volume(self._p())
    def compute_moving_boundary_volume(self): self.L.oracle_compute_moving_boundary_volume(self._p())
    def compute_densities(self): self.L.oracle_compute_densities(self._p())
    def compute_non_pressure_forces(self): self.L.oracle_compute_non_pressure_forces(self._p())
    def compute_pressure_forces(self): self.L.oracle_compute_pressure_forces(self._p())
    def advect(self): self.L.oracle_advect(self._p())
    def substep(self): self.L.oracle_substep(self._p())
    # DFSPH.py method surface
    def compute_DFSPH_factor(self): self.L.oracle_dfsph_compute_factor(self._p())
    def compute_density_change(self): self.L.oracle_dfsph_compute_density_change(self._p())
    def compute_density_adv(self): self.L.oracle_dfsph_compute_density_adv(self._p())
    def compute_density_error(self, offset): return float(self.L.oracle_dfsph_compute_density_error(self._p(), float(offset)))
    def multiply_time_step(self, time_step): self.L.oracle_dfsph_multiply_time_step(self._p(), float(time_step))
    def divergence_solver_iteration_kernel(self): self.L.oracle_dfsph_divergence_solver_iteration_kernel(self._p())
    def pressure_solve_iteration_kernel(self): self.L.oracle_dfsph_pressure_solve_iteration_kernel(self._p())
    def divergence_solver_iteration(self): return float(self.L.oracle_dfsph_divergence_solver_iteration(self._p()))
    def pressure_solve_iteration(self): return float(self.L.oracle_dfsph_pressure_solve_iteration(self._p()))
    def divergence_solve(self): self.L.oracle_dfsph_divergence_solve(self._p()); return self.s.last_iterations_v
    def pressure_solve(self): self.L.oracle_dfsph_pressure_solve(self._p()); return self.s.last_iterations
    def predict_velocity(self): self.L.oracle_dfsph_predict_velocity(self._p())
    def dfsph_advect(self): self.L.oracle_dfsph_advect(self._p())
    def dfsph_substep(self): self.L.oracle_dfsph_substep(self._p())

    def enforce_boundary_3D(self, particle_type): self.L.oracle_enforce_boundary_3D(self._p(), int(particle_type))
    def compute_rigid_rest_cm(self, object_id): self.L.oracle_compute_rigid_rest_cm(self._p(), int(object_id))

    def solve_constraints(self, object_id):
        R = np.zeros(9, dtype=np.float32)
        self.L.oracle_solve_constraints(self._p(), int(object_id), R.ctypes.data_as(_pf))
        return R.reshape(3, 3)

    def solve_rigid_body(self):
        p, n = self._ids(self.dynamic_ids)
        self.L.oracle_solve_rigid_body(self._p(), p, n)

    def initialize(self):
        p, n = self._ids(self.rigid_body_ids)
        self.L.oracle_initialize(self._p(), p, n)

    def step(self, n_steps: int = 1):
        """n_steps x SPHBase.step(); returns per-phase ms [sort, neighbour, force, integrate]."""
        ms = (C.c_double * 4)(0, 0, 0, 0)
        p, n = self._ids(self.dynamic_ids)
        self.L.oracle_step(self._p(), p, n, int(n_steps), ms)
        return list(ms)

    def cubic_kernel(self, r):
        return float(self.L.oracle_cubic_kernel(self._p(), float(r)))

    def cubic_kernel_derivative(self, r):
        rin = np.asarray(r, dtype=np.float32)
        out = np.zeros(3, dtype=np.float32)
        self.L.oracle_cubic_kernel_derivative(self._p(), rin.ctypes.data_as(_pf), out.ctypes.data_as(_pf))
        return out


def polar_rotation(A):
    A = np.ascontiguousarray(A, dtype=np.float32).reshape(9)
    R = np.zeros(9, dtype=np.float32)
    lib().oracle_polar_rotation(A.ctypes.data_as(_pf), R.ctypes.data_as(_pf))
    return R.reshape(3, 3)


def usable_cpus():
    """CPUs this process can actually run on: the scheduler affinity capped by the cgroup CPU quota.  (On the GPU
    boxes os.cpu_count() says 256 while cpu.max grants 16: more OpenMP threads than that only fight for time slices --
    measured there, 262 k particles: 27 ms/step with 16 threads, 98 with 128, 1163 with 256.)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                n = min(n, max(1, int(-(-float(quota) // period))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def max_threads():
    """OpenMP threads worth using here (never more than usable_cpus())."""
    return usable_cpus()
