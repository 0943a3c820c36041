"""ctypes binding of libsph_hip.so (C ABI: include/sph_hip.h).

The product path has no fallback: if the shared library is missing (and cannot
be built) or no gfx950 device is visible, this raises.
"""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build

_F3 = C.c_float * 3
_I3 = C.c_int32 * 3


class SphParams(C.Structure):
    """Mirror of `struct SphParams` (include/sph_hip.h)."""
    _fields_ = [
        ("n_particles", C.c_int32), ("capacity", C.c_int32), ("grid_num", _I3), ("cell_origin", _I3),
        ("n_objects", C.c_int32), ("cold_capacity", C.c_int32),
        ("grid_size", C.c_float), ("support_radius", C.c_float), ("particle_diameter", C.c_float),
        ("m_V0", C.c_float), ("density_0", C.c_float), ("stiffness", C.c_float), ("exponent", C.c_float),
        ("viscosity", C.c_float), ("surface_tension", C.c_float), ("dt", C.c_float),
        ("g", _F3), ("domain_size", _F3), ("padding", C.c_float), ("wall_hi", _F3),
        ("k_w", C.c_float), ("k_dw", C.c_float), ("visc_d_nu", C.c_float), ("visc_eps", C.c_float),
    ]


class SphTimings(C.Structure):
    _fields_ = [("sort_ms", C.c_double), ("neighbour_ms", C.c_double), ("force_ms", C.c_double),
                ("integrate_ms", C.c_double), ("halo_ms", C.c_double), ("total_ms", C.c_double),
                ("steps", C.c_int64)]


class SphStats(C.Structure):
    _fields_ = [("targets", C.c_int64), ("list_entries", C.c_int64), ("max_list", C.c_int32),
                ("list_overflow_targets", C.c_int32), ("lds_overflow_targets", C.c_int32),
                ("max_cell_occupancy", C.c_int32), ("nonempty_cells", C.c_int32), ("polar_fallbacks", C.c_int32)]


class SphDfsphParams(C.Structure):
    _fields_ = [("enable_divergence_solver", C.c_int32), ("m_max_iterations_v", C.c_int32),
                ("m_max_iterations", C.c_int32), ("fluid_particle_num", C.c_int32), ("m_eps", C.c_float),
                ("reserved_", C.c_float), ("max_error_V", C.c_double), ("max_error", C.c_double)]


class SphDfsphStats(C.Structure):
    _fields_ = [("iterations_v", C.c_int32), ("iterations", C.c_int32), ("avg_density_err_v", C.c_double),
                ("avg_density_err", C.c_double), ("total_iterations_v", C.c_int64), ("total_iterations", C.c_int64),
                ("steps", C.c_int64)]


# enum SphField
F_OBJECT_ID, F_X, F_X_0, F_V, F_ACCELERATION, F_M_V, F_M, F_DENSITY, F_PRESSURE, F_MATERIAL, F_COLOR, \
    F_IS_DYNAMIC, F_GRID_IDS, F_GRID_PARTICLES_NUM, F_PID, F_RIGID_REST_CM, F_DFSPH_FACTOR, F_DENSITY_ADV = range(18)
# enum SphOption
OPT_GATHER_IMPL, OPT_TIMING, OPT_FUSED_STEP, OPT_BRICK_SHAPE, OPT_NO_DYNAMIC_SOLIDS, OPT_DEBUG_ABLATE, \
    OPT_SLAB_DROP_OUTSIDE, OPT_UNIFORM_FLUID, OPT_UNIFORM_FLUID_STATE, OPT_SORT_BY_PID, OPT_KERNEL_VARIANT, \
    OPT_RIGID_BATCH, OPT_EXACT_MATH, OPT_DF_RUNAHEAD, OPT_RIGID_SUMS_FROM_X0, OPT_PURE_FLUID_INSTANCE, OPT_BRICK_RECORDS, OPT_DF_FUSE_ERROR = range(18)
VAR_GROUPS, VAR_GAT_LDS, VAR_GAT_LDS4, VAR_FORCE_BF, VAR_DEEP, VAR_MFMA = 1, 2, 4, 8, 16, 32
VAR_DEFAULT = VAR_GROUPS | VAR_FORCE_BF | VAR_DEEP

ABI_VERSION = 5

# every symbol include/sph_hip.h declares: (name, restype, argtypes)
_ctx = C.c_void_p
SYMBOLS = [
    ("sph_abi_version", C.c_int32, []),
    ("sph_device_count", C.c_int32, []),
    ("sph_create", C.c_int32, [C.POINTER(SphParams), C.c_int32, C.c_void_p, C.POINTER(_ctx)]),
    ("sph_destroy", C.c_int32, [_ctx]),
    ("sph_last_error", C.c_char_p, [_ctx]),
    ("sph_set_option", C.c_int32, [_ctx, C.c_int32, C.c_int32]),
    ("sph_get_option", C.c_int32, [_ctx, C.c_int32, C.POINTER(C.c_int32)]),
    ("sph_set_params", C.c_int32, [_ctx, C.POINTER(SphParams)]),
    ("sph_set_dt", C.c_int32, [_ctx, C.c_float]),
    ("sph_set_particle_count", C.c_int32, [_ctx, C.c_int32]),
    ("sph_upload", C.c_int32, [_ctx, C.c_int32, C.c_void_p, C.c_size_t]),
    ("sph_download", C.c_int32, [_ctx, C.c_int32, C.c_void_p, C.c_size_t]),
    ("sph_update_grid_id", C.c_int32, [_ctx]),
    ("sph_prefix_sum", C.c_int32, [_ctx]),
    ("sph_counting_sort", C.c_int32, [_ctx]),
    ("sph_initialize_particle_system", C.c_int32, [_ctx]),
    ("sph_compute_boundary_volume", C.c_int32, [_ctx, C.c_int32]),
    ("sph_compute_densities", C.c_int32, [_ctx]),
    ("sph_compute_non_pressure_forces", C.c_int32, [_ctx]),
    ("sph_compute_pressure_forces", C.c_int32, [_ctx]),
    ("sph_advect", C.c_int32, [_ctx]),
    ("sph_enforce_boundary_3D", C.c_int32, [_ctx, C.c_int32]),
    ("sph_compute_rigid_rest_cm", C.c_int32, [_ctx, C.c_int32]),
    ("sph_solve_constraints", C.c_int32, [_ctx, C.c_int32, C.POINTER(C.c_float)]),
    ("sph_compute_com", C.c_int32, [_ctx, C.c_int32, C.POINTER(C.c_float)]),
    ("sph_step", C.c_int32, [_ctx, C.c_int32, C.POINTER(C.c_int32), C.c_int32]),
    ("sph_sync", C.c_int32, [_ctx]),
    ("sph_get_timings", C.c_int32, [_ctx, C.POINTER(SphTimings)]),
    ("sph_reset_timings", C.c_int32, [_ctx]),
    ("sph_get_stats", C.c_int32, [_ctx, C.POINTER(SphStats)]),
    ("sph_get_particle_count", C.c_int32, [_ctx, C.POINTER(C.c_int32)]),
    ("sph_layer_offsets", C.c_int32, [_ctx, C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_int32)]),
    ("sph_layer_offsets_begin", C.c_int32, [_ctx, C.POINTER(C.c_int32), C.c_int32]),
    ("sph_layer_offsets_end", C.c_int32, [_ctx, C.POINTER(C.c_int32), C.c_int32]),
    ("sph_select_range", C.c_int32, [_ctx, C.c_int32, C.c_int32]),
    ("sph_truncate", C.c_int32, [_ctx, C.c_int32]),
    ("sph_pack_range", C.c_int32, [_ctx, C.c_int32, C.c_int32, C.c_void_p]),
    ("sph_append_records", C.c_int32, [_ctx, C.c_void_p, C.c_int32]),
    ("sph_set_target_layers", C.c_int32, [_ctx, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    ("sph_slab_set_window", C.c_int32, [_ctx, C.c_int32, C.c_int32]),
    ("sph_slab_pack", C.c_int32, [_ctx, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    ("sph_slab_advance", C.c_int32, [_ctx, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                     C.POINTER(C.c_int32), C.c_int32, C.c_int32]),
    ("sph_slab_forces", C.c_int32, [_ctx, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                    C.c_int32, C.c_int32, C.c_void_p]),
    ("sph_slab_wait_pack", C.c_int32, [_ctx]),
    ("sph_slab_density", C.c_int32, [_ctx]),
    ("sph_sort", C.c_int32, [_ctx]),
    ("sph_sweeps", C.c_int32, [_ctx]),
    ("sph_rigid_partial_sums", C.c_int32, [_ctx, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    ("sph_rigid_apply_sums", C.c_int32, [_ctx, C.c_int32, C.c_void_p, C.c_int32]),
    ("sph_upload_rest_positions", C.c_int32, [_ctx, C.c_void_p, C.c_void_p, C.c_int32]),
    ("sph_dfsph_set_params", C.c_int32, [_ctx, C.POINTER(SphDfsphParams)]),
    ("sph_dfsph_get_stats", C.c_int32, [_ctx, C.POINTER(SphDfsphStats)]),
    ("sph_dfsph_compute_densities", C.c_int32, [_ctx]),
    ("sph_dfsph_compute_DFSPH_factor", C.c_int32, [_ctx]),
    ("sph_dfsph_compute_density_change", C.c_int32, [_ctx]),
    ("sph_dfsph_compute_density_adv", C.c_int32, [_ctx]),
    ("sph_dfsph_compute_density_error", C.c_int32, [_ctx, C.c_float, C.POINTER(C.c_float)]),
    ("sph_dfsph_multiply_time_step", C.c_int32, [_ctx, C.c_float]),
    ("sph_dfsph_divergence_solver_iteration_kernel", C.c_int32, [_ctx]),
    ("sph_dfsph_pressure_solve_iteration_kernel", C.c_int32, [_ctx]),
    ("sph_dfsph_divergence_solve", C.c_int32, [_ctx]),
    ("sph_dfsph_pressure_solve", C.c_int32, [_ctx]),
    ("sph_dfsph_compute_non_pressure_forces", C.c_int32, [_ctx]),
    ("sph_dfsph_predict_velocity", C.c_int32, [_ctx]),
    ("sph_dfsph_advect", C.c_int32, [_ctx]),
    ("sph_dfsph_step", C.c_int32, [_ctx, C.c_int32, C.POINTER(C.c_int32), C.c_int32]),
    ("sph_dfsph_compute_density_error_range", C.c_int32, [_ctx, C.c_float, C.c_int32, C.c_int32, C.POINTER(C.c_double)]),
    ("sph_copy_velocity_records", C.c_int32, [_ctx, C.c_int32, C.c_int32, C.c_void_p, C.c_int32]),
    ("sph_comm_last_error", C.c_char_p, []),
    ("sph_comm_available", C.c_int32, []),
    ("sph_comm_unique_id", C.c_int32, [C.c_void_p]),
    ("sph_comm_create", C.c_int32, [_ctx, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    ("sph_comm_destroy", C.c_int32, [C.c_void_p]),
    ("sph_slab_announce", C.c_int32, [_ctx, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    ("sph_slab_incoming", C.c_int32, [_ctx, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("sph_slab_exchange", C.c_int32, [_ctx, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                      C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32]),
    ("sph_comm_swap", C.c_int32, [_ctx, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                  C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]),
    ("sph_comm_all_reduce", C.c_int32, [_ctx, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    ("sph_comm_sync", C.c_int32, [_ctx, C.c_void_p]),
    ("sph_comm_halo_time", C.c_int32, [_ctx, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    ("sph_comm_info", C.c_int32, [_ctx, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
]

_LIB = None


class SphError(RuntimeError):
    pass


def profiling_variant() -> bool:
    """SPH_HIP_LIB_VARIANT=profile: load the profiling build (section ablation compiled in) instead of the product library.
    Set by bench.py --ablate / --ablate-mask before the first load; never by the product."""
    return os.environ.get("SPH_HIP_LIB_VARIANT", "") == "profile"


def experiment_variant() -> str:
    """SPH_HIP_LIB_VARIANT=x-<tag>: load libsph_hip_x-<tag>.so -- the SAME sources built with other compiler flags by
    tools/sched_sweep.py (an A/B aid of the measurement tools, never set by the product; the file must exist, it is never built here)."""
    v = os.environ.get("SPH_HIP_LIB_VARIANT", "")
    return v if v.startswith("x-") else ""


def library_path() -> str:
    if experiment_variant():
        return os.path.join(os.path.dirname(_build.LIB), f"libsph_hip_{experiment_variant()}.so")
    return _build.LIB_PROFILE if profiling_variant() else _build.LIB


def _preload_shared_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME
    as /opt/rocm's); whichever copy is mapped first serves both torch and libsph_hip.so, and torch
    cannot initialise on the system copy.  So when torch is installed, map its copy first -- then the
    import order of torch and this package no longer matters.  Without torch the system runtime is used."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load(build_if_missing: bool = True):
    """dlopen libsph_hip.so, binding every symbol of the header.  Raises if absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if build_if_missing and not experiment_variant() and _build.stale(profiling_variant()):
        try:
            _build.build(profile=profiling_variant())
        except Exception as e:  # no hipcc on this box: use the prebuilt file if any
            if not os.path.exists(path):
                raise SphError(f"libsph_hip.so is missing and could not be built: {e}") from e
    if not os.path.exists(path):
        raise SphError(f"{path} not found: run `python -m sph_taichi_amd.build` (needs hipcc)")
    _preload_shared_hip_runtime()
    lib = C.CDLL(path)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if lib.sph_abi_version() != ABI_VERSION:
        raise SphError(f"libsph_hip ABI {lib.sph_abi_version()} != binding {ABI_VERSION}")
    _LIB = lib
    return lib


def check(lib, ctx, rc: int, what: str):
    if rc != 0:
        msg = lib.sph_last_error(ctx)
        raise SphError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
