/*
 * sph_hip.h -- C ABI of libsph_hip.so: the MI355X (gfx950) WCSPH step (and, at the end of the file,
 * the DFSPH step on the same neighbour machinery).
 *
 * The reference (erizmr/SPH_Taichi) has no FFI layer: its device code is
 * Taichi @ti.kernel Python.  This header is the boundary a maintainer binds
 * (ctypes) to replace each kernel of the hot path.  Every
 * entry point names the reference kernel it replaces (paths relative to the
 * reference root).  Plain pointers and sizes only; no torch / C++ types.
 *
 * Contract
 *  - One opaque context per GPU.  The context owns every device buffer and
 *    (unless a stream is handed to sph_create) one HIP stream.  A context is driven by
 *    one host thread at a time.
 *  - Compute entry points ENQUEUE on the context's stream and return;
 *    sph_download / sph_sync / sph_get_timings / sph_layer_offsets synchronise (and the DFSPH solver loops,
 *    once per iteration, like the reference's compute_density_error).
 *  - Every function returns 0 on success, a negative SPH_E_* code or a positive
 *    hipError_t otherwise; sph_last_error(ctx) gives the message.  Nothing
 *    throws.
 *  - Array layout at the boundary is the reference's (particle_system.py:
 *    101-113): vectors [N,3] f32 row-major, scalars [N] f32, ints [N] i32,
 *    color [N,3] i32.  The SoA/float4 packing inside is private.
 *  - Particle order: after sph_counting_sort / sph_step every array is in the
 *    reference's cell-sorted order with the STABLE intra-cell order a serial
 *    run of particle_system.py:322-330 produces.
 */
#ifndef SPH_HIP_H
#define SPH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPH_ABI_VERSION 5

typedef struct SphContext SphContext;

/* Scalars of ParticleSystem.__init__ (particle_system.py:17-46), SPHBase.__init__
 * (sph_base.py:8-21) and WCSPHSolver.__init__ (WCSPH.py:6-16).  Constants that
 * the reference folds in Python f64 before they enter a kernel (k_w, k_dw,
 * visc_d_nu, visc_eps) are passed already folded, as f32. */
typedef struct SphParams {
    int32_t n_particles;        /* particle_max_num              particle_system.py:83 */
    int32_t capacity;           /* >= n_particles; slack for halo/migration (multi-GPU) */
    int32_t grid_num[3];        /* LOCAL grid dims               particle_system.py:44 */
    int32_t cell_origin[3];     /* global cell coord of local cell (0,0,0); 0 on one GPU */
    int32_t n_objects;          /* len(rigid_rest_cm)            particle_system.py:93 */
    int32_t cold_capacity;      /* rows of the pid-indexed x_0 / color store; 0 = capacity (slabs: global N) */
    float grid_size;            /* = support_radius              particle_system.py:43 */
    float support_radius;       /* 4 r                           particle_system.py:37 */
    float particle_diameter;    /* 2 r                           particle_system.py:36 */
    float m_V0;                 /* 0.8 d^3                       particle_system.py:38 */
    float density_0;            /* sph_base.py:18 */
    float stiffness;            /* WCSPH.py:13 */
    float exponent;             /* WCSPH.py:10 */
    float viscosity;            /* sph_base.py:15 */
    float surface_tension;      /* WCSPH.py:15 */
    float dt;                   /* WCSPH.py:16 */
    float g[3];                 /* sph_base.py:13 */
    float domain_size[3];       /* particle_system.py:22 */
    float padding;              /* particle_system.py:46 */
    float wall_hi[3];           /* domain_size - padding, folded in f64 (sph_base.py:153-170) */
    float k_w;                  /* 8/pi/h^3          sph_base.py:33-35 */
    float k_dw;                 /* 6*8/pi/h^3        sph_base.py:57 */
    float visc_d_nu;            /* 2(dim+2)*viscosity WCSPH.py:104,112 */
    float visc_eps;             /* 0.01 h^2          WCSPH.py:113 */
} SphParams;

/* Per-particle arrays of particle_system.py:101-113, 138 (+ the grid counter
 * array :96 and a persistent particle id that the reference does not have). */
enum SphField {
    SPH_F_OBJECT_ID = 0,          /* i32 [N]   */
    SPH_F_X = 1,                  /* f32 [N,3] */
    SPH_F_X_0 = 2,                /* f32 [N,3] */
    SPH_F_V = 3,                  /* f32 [N,3] */
    SPH_F_ACCELERATION = 4,       /* f32 [N,3] */
    SPH_F_M_V = 5,                /* f32 [N]   */
    SPH_F_M = 6,                  /* f32 [N]   */
    SPH_F_DENSITY = 7,            /* f32 [N]   */
    SPH_F_PRESSURE = 8,           /* f32 [N]   */
    SPH_F_MATERIAL = 9,           /* i32 [N]   */
    SPH_F_COLOR = 10,             /* i32 [N,3] */
    SPH_F_IS_DYNAMIC = 11,        /* i32 [N]   */
    SPH_F_GRID_IDS = 12,          /* i32 [N]   particle_system.py:138 (download only) */
    SPH_F_GRID_PARTICLES_NUM = 13,/* i32 [G]   particle_system.py:96  (download only) */
    SPH_F_PID = 14,               /* i32 [N]   persistent id (default: index at creation); < cold_capacity */
    SPH_F_RIGID_REST_CM = 15,     /* f32 [n_objects,3]  particle_system.py:93 */
    SPH_F_DFSPH_FACTOR = 16,      /* f32 [N]   particle_system.py:116 (simulationMethod 4).  Like the reference's, */
    SPH_F_DENSITY_ADV = 17,       /* f32 [N]   particle_system.py:117  the values are dead across steps: they are
                                   *           NOT carried through the sort (every step recomputes them first). */
    SPH_F_COUNT_ = 18
};

enum SphError {
    SPH_OK = 0,
    SPH_E_INVALID = -1,      /* bad argument / size mismatch */
    SPH_E_NO_DEVICE = -2,    /* no HIP device / wrong architecture */
    SPH_E_NOMEM = -3,
    SPH_E_STATE = -4         /* call order violated (e.g. sort before grid ids) */
};

/* Neighbour-gather implementation: 0 = per-particle cell walk through L1/L2,
 * 1 = LDS-staged cell bricks (default). */
enum SphOption {
    SPH_OPT_GATHER_IMPL = 0,
    SPH_OPT_TIMING = 1,        /* k > 0 = record per-phase HIP events inside every k-th step (sph_step, sph_dfsph_step, the slab
                                  calls); SphTimings then holds sums over the timed steps.  An event is a packet of its own
                                  between two kernels: k = 1 costs the 1.75 M step 4 %, k = 8 half a percent */
    SPH_OPT_FUSED_STEP = 2,    /* 1 (default) = sph_step uses the fused density+EOS / force kernels */
    SPH_OPT_BRICK_SHAPE = 3,   /* how the LDS-brick sweeps cut the grid: 0 (default) = 4 x 2 cell columns times a height chosen
                                  per brick from the cell histogram (at most 4 layers, at most 256 targets, shell within the
                                  LDS tile); 1 = fixed 4 x 2 x 4 bricks (the partition of ABI <= 2 builds, kept for A/B) */
    SPH_OPT_NO_DYNAMIC_SOLIDS = 4 /* 1 = the caller guarantees no dynamic solid particle exists (slab ranks
                                  cannot know this locally); skips the per-step device count */,
    SPH_OPT_DEBUG_ABLATE = 5,  /* profiling BUILD only (-DSPH_PROFILE: libsph_hip_profile.so): bit mask of sweep sections to skip (results
                                  are then wrong); the production library refuses any value but 0 */
    SPH_OPT_SLAB_DROP_OUTSIDE = 6 /* slab ranks: a particle whose x cell layer is outside the local grid is hashed to
                                  a virtual cell G (sorted behind every real cell) instead of being clamped; the
                                  host then truncates the particle set with sph_truncate */,
    SPH_OPT_UNIFORM_FLUID = 7  /* fused force sweep with ONE neighbour gather per pair instead of two.  Exact, but only
                                  valid when every fluid particle has the same mass m and m_V == m_V0 (what the
                                  reference's add_particle produces for scenes whose fluid blocks share one density).
                                  -1 (default) = check on the device whenever m / m_V / material were uploaded or the
                                  particle set changed, and use it when it holds; 0 = never; 1 = check once, then the
                                  caller vouches for later arrivals (slab ranks: migrating particles of the same scene) */,
    SPH_OPT_UNIFORM_FLUID_STATE = 8 /* read-only (sph_get_option): -1 not decided yet, 0 general sweep, 1 one-gather sweep */,
    SPH_OPT_SORT_BY_PID = 9    /* 1 = inside a cell, order by persistent id instead of by previous index.  The reference
                                  order (previous index) is kept on a single GPU; slab ranks running DFSPH need an order
                                  that the owner of a boundary band and the neighbour holding it as ghosts agree on, so
                                  that ghost velocities can be refreshed record for record */,
    SPH_OPT_KERNEL_VARIANT = 10 /* A/B switch for the brick sweeps of the fused WCSPH step: bit mask of
                                  SPH_VAR_* below.  Every combination computes the same sums (list order and rounding
                                  apart); -1 = the library's default */,
    SPH_OPT_RIGID_BATCH = 11   /* 1 (default) = solve_rigid_body() (sph_base.py:247-260) of ALL dynamic bodies in three launches
                                  inside sph_step / sph_dfsph_step, the per-body solid wall passes replayed per particle;
                                  0 = body by body (4 launches each).  Same results bit for bit.  (One launch behind grid-wide
                                  barriers was built too and is slower: commit 98934a4.) */,
    SPH_OPT_DF_RUNAHEAD = 13,  /* DFSPH solver loops: 1 = the host enqueues Jacobi iteration k + 1 before it waits for iteration k's
                                  convergence test (made on the device either way); a body enqueued past convergence leaves at
                                  once and the GPU never waits for the host.  0 (default) = enqueue, wait, decide: a host round
                                  trip per iteration, as in the reference's loop -- measured 2 % FASTER at 1.75 M particles
                                  (2.906 vs 2.965 ms per step): the empty launches of the one speculative body per solve cost
                                  more than the bubbles.  Same iteration counts, same results. */
    SPH_OPT_EXACT_MATH = 12    /* A/B of the fast-math choice (never the default): 1 = the brick sweeps of the fused WCSPH step
                                  (density + EOS, force) evaluate r.norm(), r / (|r| h), x / y with IEEE sqrt and divide
                                  as the reference's f32 expressions do, instead of v_rsq_f32 / v_rcp_f32 (~1 ulp).
                                  profiles/r04_parity_fastmath_ab.json holds the two builds side by side.
                                  RESTRICTION: exact-math instances exist for the uniform-fluid step only (GM_DENSITY_EOS
                                  inline / lean and GM_FORCE_FUSED_U); scenes with any solid run the general sweeps, whose
                                  pair physics keeps v_rsq / v_rcp, and so do the cell-walk fallbacks.  sph_get_option
                                  reports the EFFECTIVE state: 1 only while the context's step actually runs those
                                  instances (SPH_OPT_UNIFORM_FLUID_STATE == 1), else 0. */,
    SPH_OPT_RIGID_SUMS_FROM_X0 = 14, /* 1 = the centre-of-mass sums of sph_compute_rigid_rest_cm and sph_rigid_partial_sums read the REST
                                  positions x_0 instead of x.  In the reference x_0 == x when initialize() computes
                                  rigid_rest_cm (sph_base.py:80-90); a RESTART (positions given, x_0 from the scene file) must not
                                  take the displaced body's centre for the rest centre.  The Python layer sets it around the
                                  rest-cm computation of a restarted ParticleSystem and clears it again; default 0. */
    SPH_OPT_PURE_FLUID_INSTANCE = 15 /* 1 (default) = a single context found (on the device) to hold no solid particle at all runs the
                                  density sweep's pure-fluid instance (m_V_j = m_V0 from a register; bit-identical results);
                                  0 = always the general instance (the A/B switch; was the SPH_DISABLE_PURE_FLUID environment
                                  variable in ABI 4); 2 = the CALLER vouches that the whole scene holds no solid particle at all
                                  (slab ranks: arrivals are never checked on the device; the Python layer sets it when the scene
                                  file has no rigid block and no rigid body) */,
    SPH_OPT_BRICK_RECORDS = 16 /* 1 (default) = the list-WRITING brick sweep (density) leaves each brick's column tables (what step A of
                                  the sweep computes from the cell array: 512 bytes per brick) in HBM, and the list-READING sweeps
                                  over the same partition and target ranges (the fused force sweep; the ~18 sweeps of a DFSPH step)
                                  load them with one 16-byte read per lane instead of recomputing them; 0 = every sweep recomputes
                                  (A/B).  Same tables, same results bit for bit. */,
    SPH_OPT_DF_FUSE_ERROR = 17 /* 1 (default) = inside the DFSPH solver loops (sph_dfsph_divergence_solve / _pressure_solve) the density-change
                                  / density-advection sweep of an iteration also reduces compute_density_error()'s sum (DFSPH.py:224-230)
                                  over its own targets -- one f64 partial per brick, added up in brick-list order by the convergence
                                  test -- instead of a streaming kernel that re-reads every particle; 0 = that kernel (A/B).  The same
                                  per-particle f32 terms either way; the f64 grouping differs (both deterministic). */
};
#define SPH_VAR_GROUPS 1   /* density: all nine runs filtered first (masks in registers), hits emitted centre run / edge runs / corner
                              runs, each group by descending hit count */
/* (bit 2 was SPH_VAR_RING, the per-lane LDS ring of hit masks with ONE balanced emission loop per lane: built, parity-green,
   slower than GROUPS -- profiles/archive/r03f_variants_partition_x_emission.json, DESIGN.md 4.5 -- and removed after commit 1bbd9b5;
   a mask with that bit set is refused) */
#define SPH_VAR_MFMA 32     /* density (with GROUPS): the candidate filter on the MATRIX pipe -- v_mfma_f32_16x16x4_f32 tests 16 candidate
                              rows against 16 target columns; a wave's 64 targets are four tiles of 16 consecutive targets, the
                              4-bit row pieces reach their target's lane through a v_permlane32/16_swap transpose (round 5:
                              the A/B the verdict asked for; DESIGN_HISTORY.md).  Same hits, same lists, same sums. */
#define SPH_VAR_GAT_LDS 2   /* (round 6) uniform-fluid force sweep: the neighbour's SECOND record (v, p / rho^2) is staged in LDS beside the first
                              instead of gathered through the L1 per pair; tile of 1,408 shell records x 32 B, three workgroups per CU;
                              the step's bricks are cut for that tile.  Implies FORCE_BF | DEEP.  Same sums (the partition, hence the
                              list order, differs from the default's where a shell exceeds the smaller tile) */
#define SPH_VAR_GAT_LDS4 4  /* the same with a tile of 1,200 records: four workgroups per CU, more bricks cut short */
#define SPH_VAR_FORCE_BF 8 /* force sweep: branch-free fluid pair term, buffer addressing for list and gather */
#define SPH_VAR_DEEP 16    /* list-reading sweeps: list entries loaded a whole round (3 pairs) before they are decoded */
/* (r04: bit 64 was SPH_VAR_PERSIST -- both sweeps as PERSISTENT workgroups, grid = the chip's resident slots, bricks taken by
   per-XCD TICKETS with the next ticket fetched while computing: built, parity-green, slower like round 2's static walk (density
   +17 %, force +25 %: profiles/r04c_variants_persistent_tickets.json, DESIGN_HISTORY.md) and removed after commit e7af346) */
/* 0 = the baseline: run-by-run emission in the reference's (dx, dy) order, plain list loop in the force sweep.
 * Default = GROUPS | FORCE_BF | DEEP.  (Rounds 1-2 also carried PAD, MICRO -- now always on -- and 2PHASE, MIRROR,
 * SORTED: measured, superseded and removed; their tables are profiles/archive/r02c, r02o, r02p.) */

/* ms accumulated by sph_step since the last sph_reset_timings (HIP events on
 * the context's stream).  sort = K1+K2+K3 (initialize_particle_system),
 * neighbour = K4+K5 (boundary volume + density/EOS), force = K6+K7,
 * integrate = K8+K9+K10, halo = multi-GPU exchange packing. */
typedef struct SphTimings {
    double sort_ms, neighbour_ms, force_ms, integrate_ms, halo_ms, total_ms;
    int64_t steps;
} SphTimings;

int32_t sph_abi_version(void);
int32_t sph_device_count(void);

/* ParticleSystem.__init__ field allocation (particle_system.py:91-145). `stream`
 * may be NULL (the context creates its own) or a hipStream_t to enqueue on. */
int32_t sph_create(const SphParams* params, int32_t device, void* stream, SphContext** out);
int32_t sph_destroy(SphContext* ctx);
const char* sph_last_error(const SphContext* ctx);
int32_t sph_set_option(SphContext* ctx, int32_t option, int32_t value);
int32_t sph_get_option(const SphContext* ctx, int32_t option, int32_t* value);
/* Replace the solver scalars (WCSPHSolver.__init__, WCSPH.py:6-16); the sizes in
 * `params` (n_particles, capacity, grid_num, cell_origin, n_objects) must be unchanged. */
int32_t sph_set_params(SphContext* ctx, const SphParams* params);
int32_t sph_set_dt(SphContext* ctx, float dt);                        /* solver.dt[None] = ..  sph_base.py:20-21 */
int32_t sph_set_particle_count(SphContext* ctx, int32_t n);           /* owned+ghost count (multi-GPU) */

/* ti.field.from_numpy / to_numpy / _add_particles (particle_system.py:260-284,
 * 409-418).  `host` is borrowed for the duration of the call. */
int32_t sph_upload(SphContext* ctx, int32_t field, const void* host, size_t bytes);
int32_t sph_download(SphContext* ctx, int32_t field, void* host, size_t bytes);

/* --- K1-K3: neighbour structure (particle_system.py:311-375) --- */
int32_t sph_update_grid_id(SphContext* ctx);       /* update_grid_id          :311-320 */
int32_t sph_prefix_sum(SphContext* ctx);           /* PrefixSumExecutor.run   :374 (scan_single_buffer.py:108-146) */
int32_t sph_counting_sort(SphContext* ctx);        /* counting_sort           :322-369 */
int32_t sph_initialize_particle_system(SphContext* ctx); /* :372-375 */

/* --- K4-K10: solver kernels --- */
int32_t sph_compute_boundary_volume(SphContext* ctx, int32_t dynamic); /* sph_base.py:91-98 (0) / :106-113 (1) */
int32_t sph_compute_densities(SphContext* ctx);             /* WCSPH.py:33-43 */
int32_t sph_compute_non_pressure_forces(SphContext* ctx);   /* WCSPH.py:128-140 */
int32_t sph_compute_pressure_forces(SphContext* ctx);       /* WCSPH.py:70-85 */
int32_t sph_advect(SphContext* ctx);                        /* WCSPH.py:143-149 */
int32_t sph_enforce_boundary_3D(SphContext* ctx, int32_t particle_type); /* sph_base.py:149-179 */
int32_t sph_compute_rigid_rest_cm(SphContext* ctx, int32_t object_id);   /* sph_base.py:87-89 */
/* solve_constraints (sph_base.py:200-222).  R_out (9 floats, row-major) may be
 * NULL; when non-NULL the call synchronises, like the reference's return. */
int32_t sph_solve_constraints(SphContext* ctx, int32_t object_id, float* R_out);
/* compute_com_kernel (sph_base.py:195-197); synchronises. */
int32_t sph_compute_com(SphContext* ctx, int32_t object_id, float* cm_out);

/* SPHBase.step() x n_steps (sph_base.py:263-271) entirely on the device.
 * dynamic_ids = objectIds of dynamic RigidBodies (sph_base.py:249-250). */
int32_t sph_step(SphContext* ctx, int32_t n_steps, const int32_t* dynamic_ids, int32_t n_dynamic);

int32_t sph_sync(SphContext* ctx);
int32_t sph_get_timings(SphContext* ctx, SphTimings* out);
int32_t sph_reset_timings(SphContext* ctx);

/* Neighbourhood statistics of the LAST density sweep of the LDS-brick path (no reference counterpart; the
 * reference's for_all_neighbors, particle_system.py:378-385, has no capacity limits to report on).  They say how
 * far a flow state is from the rest lattice and whether any target left the fast path.  Synchronises. */
typedef struct SphStats {
    int64_t targets;               /* fluid particles that gathered */
    int64_t list_entries;          /* sum of their neighbour-list lengths (superset filter: includes the particle
                                      itself and pairs within 1e-4 h beyond h) */
    int32_t max_list;              /* longest list */
    int32_t list_overflow_targets; /* lists longer than the hand-off capacity (95): the force sweep walks the cells */
    int32_t lds_overflow_targets;  /* targets of bricks whose shell did not fit the LDS tile: both sweeps walk the cells */
    int32_t max_cell_occupancy;    /* particles in the fullest cell */
    int32_t nonempty_cells;
    int32_t polar_fallbacks;       /* solve_constraints() calls since sph_create whose polar rotation (sph_base.py:212) left the Newton
                                      iteration for the Jacobi-SVD form: degenerate bodies (flat = rank-2 A, rods, reflections) */
} SphStats;
int32_t sph_get_stats(SphContext* ctx, SphStats* out);

/* --- multi-GPU slab support (no reference counterpart) -------------
 * Cells are flattened x-slowest (particle_system.py:294), so after the sort every
 * set of x-layers is ONE contiguous index range.  A slab rank exchanges such ranges
 * with its x-neighbours as packed records (48 B/particle: `count` xm float4s, then
 * `count` vf float4s, then `count` aux float4s) through device buffers it owns
 * (torch tensors + torch.distributed/RCCL on the host side). */
#define SPH_RECORD_BYTES 48
int32_t sph_get_particle_count(SphContext* ctx, int32_t* n);
/* out[k] = number of particles in local x-layers < layers[k] (valid after
 * sph_prefix_sum / the sort).  Synchronises. */
int32_t sph_layer_offsets(SphContext* ctx, const int32_t* layers, int32_t n, int32_t* out);
/* Split form: _begin enqueues the copies (pinned host memory + an event) right behind the sort and returns;
 * _end waits for that event only, so kernels enqueued in between (sph_sweeps) keep the GPU busy. n <= 16. */
int32_t sph_layer_offsets_begin(SphContext* ctx, const int32_t* layers, int32_t n);
int32_t sph_layer_offsets_end(SphContext* ctx, int32_t* out, int32_t n);
/* Keep only [first, first+count) of the current order as the particle set (drops
 * ghosts); must be followed by a sort before any sweep. */
int32_t sph_select_range(SphContext* ctx, int32_t first, int32_t count);
/* Shrink the particle set to its first n records without invalidating the neighbour structure
 * (drops the particles the sort moved into the virtual cell, SPH_OPT_SLAB_DROP_OUTSIDE). */
int32_t sph_truncate(SphContext* ctx, int32_t n);
int32_t sph_pack_range(SphContext* ctx, int32_t first, int32_t count, void* device_dst);
/* Append `count` packed records after the current particles (count += n). */
int32_t sph_append_records(SphContext* ctx, const void* device_src, int32_t count);
/* Restrict the TARGETS of the density (+EOS) and force sweeps to local x layers [lo, hi) (candidates are never
 * restricted).  A slab rank sets density = owned + first ghost layer, force = owned: the outer ghost layer only
 * serves as neighbours.  Default: all layers. */
int32_t sph_set_target_layers(SphContext* ctx, int32_t density_lo, int32_t density_hi, int32_t force_lo, int32_t force_hi);
/* Re-cut (load balance, SURVEY 8e "re-cut every K steps"): move the context's window of global x layers to
 * [origin_x, origin_x + nx), nx <= the grid_num[0] given to sph_create (which is the allocation).  Takes effect at
 * the next sort: records outside the new window fall into the virtual cell and are dropped like any stray.  Resets
 * the target layers to all layers (call sph_set_target_layers afterwards). */
int32_t sph_slab_set_window(SphContext* ctx, int32_t origin_x, int32_t nx);
/* One slab step's device work in two calls (same effect as the individual calls, fewer host round trips):
 *   sph_slab_pack : pack [firstL, firstL+nL) into dstL and [firstR, ...) into dstR, then synchronise (the
 *                   buffers are handed to the transport next);
 *   sph_slab_advance : select [keep_first, keep_first+keep_count), append the two received ranges, sort,
 *                   enqueue the offset read-back for `layers` (sph_layer_offsets_begin), enqueue the sweeps
 *                   (if do_sweeps).  Follow with sph_layer_offsets_end + sph_truncate. */
int32_t sph_slab_pack(SphContext* ctx, int32_t firstL, int32_t nL, void* dstL, int32_t firstR, int32_t nR, void* dstR);
int32_t sph_slab_advance(SphContext* ctx, int32_t keep_first, int32_t keep_count, const void* srcL, int32_t nL,
                         const void* srcR, int32_t nR, const int32_t* layers, int32_t n_layers, int32_t do_sweeps);
/* Exchange hidden behind compute.  sph_slab_advance(..., do_sweeps = 2) runs only the boundary-volume and density
 * sweeps after the sort.  sph_slab_forces then enqueues: force sweep of the boundary layers [bl_lo,bl_hi) and
 * [br_lo,br_hi); the two halo packers (records [firstL,+nL) / [firstR,+nR) advanced by this step's Euler + wall
 * update, written to dstL / dstR, arrays untouched); an event; the force sweep of the remaining owned layers; the
 * in-place advect.  sph_slab_wait_pack blocks until the packers are done -- the caller starts the exchange while
 * the interior force sweep is still running.  (With the advect fused into the interior sweep's finish the interior
 * particles' `acceleration` field is not materialised by this call: only the boundary sets' is, for the packers;
 * sph_download(SPH_F_ACCELERATION) then fails with SPH_E_STATE until something writes every acceleration again:
 * sph_compute_non_pressure_forces (+ sph_compute_pressure_forces), sph_step, or an upload.) */
int32_t sph_slab_forces(SphContext* ctx, int32_t bl_lo, int32_t bl_hi, int32_t br_lo, int32_t br_hi,
                        int32_t firstL, int32_t nL, void* dstL, int32_t firstR, int32_t nR, void* dstR);
int32_t sph_slab_wait_pack(SphContext* ctx);
int32_t sph_slab_density(SphContext* ctx);   /* moving boundary volume + density/EOS sweep (what do_sweeps = 2 runs) */
/* The sort of sph_step (no acceleration permutation): hash + scan + scatter. */
int32_t sph_sort(SphContext* ctx);
/* One step's sweeps without the sort: boundary volume, density+EOS, force, advect +
 * fluid walls (the slab driver sorts and exchanges halos itself). */
int32_t sph_sweeps(SphContext* ctx);

/* ---- shape matching across slabs (sph_base.py:182-222 with the sums split over ranks) ----
 * A body's particles may live on several ranks.  Each rank adds up ITS OWN particles of `object_id` (indices
 * [first, first+count) of the current order = its owned range), the 16 sums of all ranks are added (one
 * all-reduce of 16 doubles), and every rank moves all local particles of the body -- ghosts included -- with the
 * same cm and rotation.  sums (DEVICE pointers, doubles): [0] sum m, [1..3] sum m x, [4..6] sum m q,
 * [7..15] sum m x (x) q row-major, with m = m_V0 * density, q = x_0 - rigid_rest_cm[object]; so
 * A = sum m (x - cm)(x) q = [7..15] - cm (x) [4..6] exactly as sph_base.py:206-210.
 * sph_rigid_partial_sums enqueues on the context's stream (sph_sync before handing the buffer to the
 * communication library).  sph_rigid_apply_sums: mode 0 = rest centre of mass (sph_base.py:87-89), mode 1 =
 * solve_constraints (cm, polar rotation, x = cm + R q).  x_0 lives in a persistent-id-indexed table; ranks that
 * can receive a body's particles later load the whole body's rest positions with sph_upload_rest_positions. */
int32_t sph_rigid_partial_sums(SphContext* ctx, int32_t object_id, int32_t first, int32_t count, double* dev_sums16);
int32_t sph_rigid_apply_sums(SphContext* ctx, int32_t object_id, const double* dev_sums16, int32_t mode);
int32_t sph_upload_rest_positions(SphContext* ctx, const int32_t* pid, const float* x0, int32_t n);

/* ---- the slab exchange itself: RCCL point-to-point over xGMI, enqueued on the device (sph_comm.hip) ----
 * SURVEY 8(b) sph_exchange_halo / sph_migrate.  One SphComm per context (= per GPU / rank).  The host gets the 128-byte
 * RCCL unique id on rank 0 (sph_comm_unique_id), distributes it by whatever it has (torch.distributed, MPI, a file)
 * and every rank calls sph_comm_create.  left / right below = the x-neighbours' ranks, -1 at a domain end (a rank may
 * name itself: the loop-back used by the one-GPU tests).  librccl is dlopen'ed at the first call.
 * Per step:  sph_slab_announce   after the sort, when the layer offsets -- hence the record counts of THIS step's
 *                                messages -- are known: sends them ahead (non-blocking);
 *            sph_slab_incoming   the counts the neighbours announced (waits for the tiny message: normally long there),
 *                                so that the host can size its receive buffers;
 *            sph_slab_exchange   ncclGroupStart; ncclSend / ncclRecv x <= 4; ncclGroupEnd on the communication stream,
 *                                which first waits for the halo packers' event of sph_slab_forces (after_packers = 1;
 *                                the interior force sweep keeps running on the main stream) or for the main stream
 *                                (after_packers = 0); the main stream then waits for the exchange.  Nothing blocks the host.
 * sph_comm_swap: fixed-size exchange (DFSPH ghost-velocity refresh); sph_comm_all_reduce: in-place sum of n doubles
 * (dtype 0) or int64 (dtype 1) in device memory; both are ordered after the main stream's work and before what it
 * does next.  sph_comm_halo_time: accumulated duration of the payload exchanges on the communication stream. */
typedef struct SphComm SphComm;
const char* sph_comm_last_error(void);
/* 0 if librccl (or the library named by SPH_RCCL_LIB) can be opened and exports what this file binds: lets the ranks of a
 * job agree on the transport BEFORE anybody enters the collective sph_comm_create. */
int32_t sph_comm_available(void);
int32_t sph_comm_unique_id(uint8_t* out128);
int32_t sph_comm_create(SphContext* ctx, const uint8_t* id128, int32_t rank, int32_t world, SphComm** out);
int32_t sph_comm_destroy(SphComm* comm);
int32_t sph_slab_announce(SphContext* ctx, SphComm* comm, int32_t left, int32_t right, int32_t n_to_left, int32_t n_to_right);
int32_t sph_slab_incoming(SphContext* ctx, SphComm* comm, int32_t* n_from_left, int32_t* n_from_right);
int32_t sph_slab_exchange(SphContext* ctx, SphComm* comm, int32_t left, int32_t right, const void* send_left, int32_t n_to_left,
                          const void* send_right, int32_t n_to_right, void* recv_left, int32_t n_from_left, void* recv_right,
                          int32_t n_from_right, int32_t after_packers);
int32_t sph_comm_swap(SphContext* ctx, SphComm* comm, int32_t left, int32_t right, const void* send_left, int64_t bytes_to_left,
                      const void* send_right, int64_t bytes_to_right, void* recv_left, int64_t bytes_from_left, void* recv_right,
                      int64_t bytes_from_right);
int32_t sph_comm_all_reduce(SphContext* ctx, SphComm* comm, void* dev, int32_t n, int32_t dtype);
int32_t sph_comm_sync(SphContext* ctx, SphComm* comm);
int32_t sph_comm_halo_time(SphContext* ctx, SphComm* comm, double* ms, int64_t* exchanges);
/* rank / world of the communicator as the communication library reports them (ncclCommUserRank / ncclCommCount) */
int32_t sph_comm_info(SphContext* ctx, SphComm* comm, int32_t* rank, int32_t* world);

/* ======================================================================================
 * DFSPH (simulationMethod 4): DFSPHSolver of /root/reference/DFSPH.py on the same neighbour
 * machinery.  The density sweep writes every fluid particle's neighbour list once per step;
 * all later sweeps of the step (factor, density change / advection, both Jacobi solvers,
 * non-pressure forces) read those lists.  The solver loops run inside the library.  The
 * reference's loop has the host between two iterations (compute_density_error returns a
 * float that Python compares with eta); here the test is made on the device with the same
 * arithmetic and the host reads its verdict (sph_api.hip: df_solve_loop).  With
 * SPH_OPT_DF_RUNAHEAD 1 the host enqueues iteration k + 1 before it waits for iteration
 * k's verdict and sweeps enqueued past convergence leave at once; the iteration counts
 * are the reference's either way.
 * ==================================================================================== */
typedef struct SphDfsphParams {
    int32_t enable_divergence_solver; /* DFSPH.py:12 */
    int32_t m_max_iterations_v;       /* DFSPH.py:14 */
    int32_t m_max_iterations;         /* DFSPH.py:15 */
    int32_t fluid_particle_num;       /* particle_system.py:57-62: divisor of the average density error */
    float m_eps;                      /* DFSPH.py:17 */
    float reserved_;
    double max_error_V;               /* DFSPH.py:19 (Python-scope floats: f64) */
    double max_error;                 /* DFSPH.py:20 */
} SphDfsphParams;

typedef struct SphDfsphStats {
    int32_t iterations_v;             /* what DFSPH.py:258 prints for the last divergence solve */
    int32_t iterations;               /* what DFSPH.py:353 prints for the last pressure solve */
    double avg_density_err_v;
    double avg_density_err;
    int64_t total_iterations_v;       /* solver iterations run since sph_create (incl. the unconditional first one) */
    int64_t total_iterations;
    int64_t steps;
} SphDfsphStats;

int32_t sph_dfsph_set_params(SphContext* ctx, const SphDfsphParams* params);
int32_t sph_dfsph_get_stats(SphContext* ctx, SphDfsphStats* out);
int32_t sph_dfsph_compute_densities(SphContext* ctx);              /* DFSPH.py:37-47 */
int32_t sph_dfsph_compute_DFSPH_factor(SphContext* ctx);           /* DFSPH.py:116-154 */
int32_t sph_dfsph_compute_density_change(SphContext* ctx);         /* DFSPH.py:157-197 */
int32_t sph_dfsph_compute_density_adv(SphContext* ctx);            /* DFSPH.py:200-221 */
int32_t sph_dfsph_compute_density_error(SphContext* ctx, float offset, float* out); /* DFSPH.py:224-230; synchronises */
int32_t sph_dfsph_multiply_time_step(SphContext* ctx, float time_step);            /* DFSPH.py:233-237 on dfsph_factor */
int32_t sph_dfsph_divergence_solver_iteration_kernel(SphContext* ctx);             /* DFSPH.py:285-321 */
int32_t sph_dfsph_pressure_solve_iteration_kernel(SphContext* ctx);                /* DFSPH.py:356-394 */
int32_t sph_dfsph_divergence_solve(SphContext* ctx);               /* DFSPH.py:240-283 (loop included) */
int32_t sph_dfsph_pressure_solve(SphContext* ctx);                 /* DFSPH.py:324-354 (loop included) */
int32_t sph_dfsph_compute_non_pressure_forces(SphContext* ctx);    /* DFSPH.py:49-112 */
int32_t sph_dfsph_predict_velocity(SphContext* ctx);               /* DFSPH.py:388-394 */
int32_t sph_dfsph_advect(SphContext* ctx);                         /* DFSPH.py:100-107 */
/* n_steps x SPHBase.step() with DFSPHSolver.substep (sph_base.py:263-271, DFSPH.py:400-408).  Phase timings:
 * neighbour = boundary volume + density + factor, force = both solvers + non-pressure forces + predict_velocity. */
int32_t sph_dfsph_step(SphContext* ctx, int32_t n_steps, const int32_t* dynamic_ids, int32_t n_dynamic);
/* DFSPH across slabs: a rank's share of compute_density_error over its OWNED records [first, first+count) (f64, to be
 * all-reduced; synchronises), and the velocity records (v + flags, 16 B each) of a contiguous range copied to
 * (to_context = 0) or from (1) a caller-owned device buffer -- each Jacobi sweep is followed by a refresh of the ghost
 * layers' velocities from their owners. */
int32_t sph_dfsph_compute_density_error_range(SphContext* ctx, float offset, int32_t first, int32_t count, double* out);
int32_t sph_copy_velocity_records(SphContext* ctx, int32_t first, int32_t count, void* device_buf, int32_t to_context);

#ifdef __cplusplus
}
#endif
#endif /* SPH_HIP_H */
