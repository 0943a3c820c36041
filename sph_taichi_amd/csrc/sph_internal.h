// sph_internal.h -- private to libsph_hip.so (gfx950 only).
//
// Data layout in HBM (all particle arrays are in the CURRENT cell-sorted order
// unless marked "cold"):
//   xm [cap] float4 = (x, y, z, m_V)                  hot, ping-pong through the sort
//   vf [cap] float4 = (vx, vy, vz, bits(flags))       hot, ping-pong
//   aux[cap] float4 = (m, density, pressure, bits(pid)) hot, ping-pong
//   eos[cap] float4 = (p/rho^2, m/rho_raw, m, rho)    written by density+EOS, read by force
//                     (DFSPH: (dfsph_factor, density_adv, m, rho))
//   glist[24 << glist_shift bytes] u16, gcnt[cap] u8  neighbour lists: entry r of particle i at (r >> 2) << glist_shift | i * 8 | (r & 3) * 2
//   acc[cap] float4 = (ax, ay, az, 0)
//   key[cap] int    = grid_ids                        ping-pong
//   x0_cold [3*cap] f32, color_cold [3*cap] i32       indexed by pid, never moved
//   cell_end[G] int = inclusive prefix of the cell histogram (= the reference's
//                     grid_particles_num after PrefixSumExecutor.run)
// flags = material (bits 0..7) | is_dynamic (bit 8) | object_id (bits 9..31).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/sph_hip.h"

#define SPH_MATERIAL_SOLID 0  // particle_system.py:30
#define SPH_MATERIAL_FLUID 1  // particle_system.py:31
#define SPH_MAX_TIMED_STEPS 128
#define SPH_GLIST_ROWS 96  // list entries allocated per particle (24 groups of four); lists up to LISTCAP = 95 entries (developed dam-break flows reach 58, profiles/archive/r03i)
#define SPH_DF_ERR_BLOCKS 512
#define SPH_VAR_PURE_INTERNAL 64   // not a user bit (sph_set_option refuses masks above 63): the pure-fluid instance of the density sweep
#define SPH_VAR_DEFAULT (SPH_VAR_GROUPS | SPH_VAR_FORCE_BF | SPH_VAR_DEEP)  // SPH_OPT_KERNEL_VARIANT when the caller does not choose: the fastest rows of profiles/archive/r03f_variants_partition_x_emission.json (density) and r02g_variants_force.json (force)

struct DevView {
    int N, G;
    int nx, ny, nz;
    int ox, oy, oz;  // cell_origin (multi-GPU slabs)
    int tgt_lo, tgt_hi;  // local x layers whose particles are targets of this sweep ...
    int tgt_lo2, tgt_hi2;  // ... plus an optional second range (slab mode: both boundary sets in one launch)
    int sort_by_pid;   // intra-cell order by persistent id (SPH_OPT_SORT_BY_PID)
    int drop_outside;  // slab mode: x layer outside the local grid -> virtual cell G
    int exp_int;     // Tait exponent as a small integer (1..32) when it is one, else 0 (WCSPH.py:76)
    // DFSPH solver loops (round 5): the sweeps of a Jacobi iteration that the host enqueued AHEAD of the previous
    // iteration's convergence test leave at once when that test (made on the device) has closed the solve
    double* df_bpart;      // non-null: the density-change / -advection sweep leaves compute_density_error()'s partial sum of each brick here
    const unsigned* gate;  // null outside the solver loops
    unsigned gate_epoch;   // *gate == gate_epoch: this solve has converged
#ifdef SPH_PROFILE
    int ablate;      // profiling build only: bit0 skip phase 2, bit1 skip list write-out, bit2 skip phase 1, ... (sph_gather.hip)
    unsigned long long* prof_ts;  // profiling build only: 8 words per hardware block of the brick sweeps (SPH_TS)
#endif
    float grid_size, h, inv_h, d2, m_V0, rho0, stiffness, exponent, sigma, dt;
    float gx, gy, gz;
    float pad;
    float k_w, k_dw, visc_d_nu, visc_eps;
    float w_zero, w_d;  // W(0), W(d)
    float m_eps;        // DFSPH.py:17
    float m_u;          // the common fluid particle mass (uniform-fluid force path)
    double fx_scale;    // 2^S of the fixed-point shape-matching sums (sph_integrate.hip)
    float whx, why, whz;  // upper wall planes (domain_size - padding), for the advect fused into the force sweep
    int fuse_advect;    // GM_FORCE_FUSED_U finish also integrates its fluid targets (WCSPH.py:143-149 + fluid walls)
    int write_sg;       // density finish also writes the stg (/ gat) records of the one-gather sweeps
    int write_k;        // density-change / -advection finish also writes k_j into kbuf
    int store_acc;      // GM_FORCE_FUSED_U finish with the fused advect: also store the acceleration (0: a step inside
                        // sph_step(n) that is not the last -- nothing can read the field before the next step rewrites it)
    int rigid_from_x0;  // the centre-of-mass sums read x_0 instead of x (SPH_OPT_RIGID_SUMS_FROM_X0: rest cm of a restart)
    float4* xm;
    float4* vf;
    float4* aux;
    float4* eos;
    float2* eos2; // lean target record (p, rho) of the uniform-fluid WCSPH step (same memory as eos)
    float4* stg;  // (x, y, z, U): U = m/rho_raw (fluid, > 0) or -m_V (solid)      } uniform-fluid force path:
    float4* gat;  // (vx, vy, vz, p/rho^2) (fluid) or (v, 1 if dynamic else 0) (solid) } staged / gathered records
    float* kbuf;  // DFSPH: k_j = b_j * dfsph_factor_j (same memory as gat)
    float4* acc;
    long long* acc_fx;  // [3 N] two-way coupling reactions on dynamic rigid particles, 2^32 fixed point (couple_scatter / k_fold_coupling)
    int* key;
    int* cell_end;
    const float* x0_cold;
    int* polar_fb;   // counter of solve_constraints() calls that took the Jacobi-SVD fallback (context's dyn_count[3])
    float* rigid_rest_cm;
};

struct CellIdx16 { int v[16]; };   // up to 16 cell indices whose scanned values the host wants back (slab layer offsets); -1 = unused

struct SphContext {
    SphParams p;
    int device;
    hipStream_t stream;
    bool own_stream;
    int N;    // current particle count
    int cap;  // capacity
    int cold_cap;  // rows of x0_cold / color_cold
    int G;
    int cur;  // which ping-pong set is current
    int* h_pinned;       // 16 ints of pinned host memory for sph_layer_offsets_begin/_end
    hipEvent_t ev_off;   // recorded behind those copies
    hipEvent_t ev_pack;  // recorded behind the halo packers of sph_slab_forces
    hipEvent_t ev_fork;  // main stream -> side stream hand-off in sph_slab_forces
    hipStream_t side;    // slab mode: boundary force sweep + halo packers run here, concurrently with the interior sweep
    bool use_side;       // launchers enqueue on `side` (with their own brick list) while this is set
    int2* brick_list2;   // [brick_cap] brick list of launches on the side stream
    int* brick_count2;
    int off_zero_mask;   // which of the pending offsets are layer 0 (no copy needed)
    int off_stamp;       // stamp of the last sort that delivered layer offsets (h_pinned[17] shows it once they are there)
    bool off_stamp_pending;  // sph_layer_offsets_end waits for the stamp (spinning on mapped memory), not for ev_off
    bool off_in_sort;    // sph_slab_advance: the sort that follows delivers the layer offsets itself (k_unstable_place reads the
    CellIdx16 off_ix;    //   scanned cells into the mapped buffer and ev_off is recorded right behind it) -- no launch of their own
    int tgt_layers[4];  // density lo/hi, force lo/hi (slab mode); default 0..nx
    int nx_alloc;       // grid_num[0] at sph_create: what the cell arrays and brick lists are sized for (sph_slab_set_window)
    int in_off;  // first live record of the current set (non-zero only between sph_select_range and the next sort)
    float4* xm[2];
    float4* vf[2];
    float4* aux[2];
    int* key[2];
    float4* eos;
    float2* eos2; // lean target record (p, rho) of the uniform-fluid WCSPH step (same memory as eos)
    float4* stg;
    float4* gat;
    float4* acc;
    float4* acc_tmp;
    long long* acc_fx;   // [3 cap] fixed-point accumulators of the coupling reactions
    int* cell_end;     // [G+1] the CURRENT cell array (one of cell_buf[])
    int* cell_buf[2];  // two cell arrays: while one serves the sweeps, the scatter zeroes the other for the next histogram
    int cell_cur;
    bool brick_count_zero;  // brick_count is zero (set by the hash kernel) and no list has been built into it since
    bool next_cells_zero;  // cell_buf[cell_cur ^ 1] is all zero (no memset needed before the next histogram)
    int* rank_off;     // [cap] arbitrary intra-cell offset from the histogram atomics
    int* idx_unstable; // [cap]
    unsigned long long* scan_status;  // per scan tile: launch epoch << 32 | tile total (k_scan_fused)
    unsigned scan_epoch;
    int* scan_err;     // raised by a scan tile that never saw a predecessor's total (a bounded wait, never a hang)
    unsigned short* glist;  // neighbour lists handed from the density to the force sweep: SPH_GLIST_ROWS / 4 entry groups of
                            // 2^glist_shift bytes, entry r of particle i at (r >> 2) << glist_shift | i * 8 | (r & 3) * 2
    int glist_shift;        // smallest shift with cap * 8 <= 2^shift; 0 = no lists (32-bit offsets would not reach: cell walk)
    unsigned char* gcnt;    // [cap] list lengths (255 = take the global cell walk)
    int4* brick_rec;        // [brick_cap][32] per-brick column tables (LDS slot -> global, segment start, target start, target offset) left by
                            // the list-writing sweep for the readers of its lists (k_gather_brick step A); null: not allocated
    bool brec_valid;        // brick_rec describes the current lists for the partition / target ranges in brec_key
    int brec_key[5];
    int opt_brick_rec;      // SPH_OPT_BRICK_RECORDS (default 1)
    int2* brick_list;       // [brick_cap] bricks of the sweep being launched: (column group, first z layer | height << 16)
    int* brick_count;       // device counter
    int brick_cap;
    int scan_blocks;
    float* x0_cold;    // [3*cap]
    int* color_cold;   // [3*cap]
    float* rigid_rest_cm;  // [n_objects*3]
    int* dyn_list;     // [cap] sorted-order indices of dynamic rigid particles
    int* dyn_count;    // device counter
    int n_dyn_host;    // number of dynamic rigid particles (constant; counted at upload)
    int rigid_fx_exp;  // S of the fixed-point shape-matching sums: set with n_dyn_host from the count, the largest body density and the domain size
    double* rigid_part;   // [rigid_part_blocks][16] per-block partial sums of the shape-matching reductions
    int rigid_part_blocks;
    float* rigid_R;    // [12] cm[3] + R[9]
    void* stage;       // upload/download staging, cap*16 bytes (>= G*4)
    size_t stage_bytes;
    bool have_keys, have_prefix, sorted;
    bool lists_valid;   // glist/gcnt describe the CURRENT positions and order (written by a list-writing brick sweep)
    bool gcnt_written;  // gcnt was written by a brick density sweep since the last sort (sph_get_stats reports list lengths only then)
    int stg_kind;       // what stg / gat hold for the current positions: 0 nothing, 1 the WCSPH records of
                        // GM_DENSITY_EOS, 2 the DFSPH record (x, y, z, +m_V fluid / -m_V solid) of GM_DF_DENSITY
    int k_kind;         // gat-as-float holds k_j = b_j * factor_j: 0 no, 1 b = density_adv, 2 b = density_adv - 1
    bool bricks_valid;  // brick_list/brick_count describe the current order for the target ranges in bricks_key
    int bricks_key[5];  // partition id (footprint, cut rule, limits), tgt_lo, tgt_hi, tgt_lo2, tgt_hi2
    double* h_df_err;   // pinned, device-visible: result of compute_density_error
    struct DfSlot { float err; int converged; double avg; }* h_df_slot;  // [4] pinned, device-visible: the convergence test of a solver iteration
    unsigned* df_gate;  // device word: epoch of the last solve that converged (DevView::gate)
    unsigned df_epoch;  // epoch of the running solve (0: none -- the gate is closed to everybody else)
    unsigned df_epoch_done;  // epoch of the last finished solve
    hipEvent_t ev_df[4];
    double* df_part;    // [SPH_DF_ERR_BLOCKS] per-workgroup partial sums
    double* df_bpart;   // [brick_cap] per-BRICK partial sums of the density error, written by the refresh sweep of a solver iteration (SPH_OPT_DF_FUSE_ERROR)
    int df_collect;     // the sweep being enqueued is such a refresh sweep (sph_view hands df_bpart to the kernel)
    bool df_bpart_valid;  // ... and it went through the brick kernel: the convergence test adds up df_bpart instead of re-reading the particles
    int opt_df_fuse_err;  // SPH_OPT_DF_FUSE_ERROR (default 1)
    SphDfsphParams df;  // DFSPH solver knobs
    SphDfsphStats df_stats;
    double* df_err;     // device accumulator of compute_density_error
    // options
    int timing_phase;  // steps since the last timed one (SPH_OPT_TIMING k)
    int opt_gather_impl, opt_timing, opt_fused, opt_brick_shape, opt_no_dynamic, opt_ablate, opt_drop_outside;
    int opt_sort_by_pid;
    int opt_rigid_batch;  // SPH_OPT_RIGID_BATCH: 1 (default) = solve_rigid_body() of all bodies in three launches
    int opt_df_runahead;  // SPH_OPT_DF_RUNAHEAD: 1 = DFSPH solver bodies are enqueued one ahead of the convergence test
    int opt_exact_math;   // SPH_OPT_EXACT_MATH: 1 = IEEE divide / sqrt instances of the brick sweeps (A/B of the fast-math choice)
    int opt_rigid_x0;     // SPH_OPT_RIGID_SUMS_FROM_X0: the rest-cm sums read x_0 (set by the host around the rest cm of a restart)
    int opt_pure_instance;  // SPH_OPT_PURE_FLUID_INSTANCE: 1 (default) = a solid-free single context may run the pure-fluid density instance
    int opt_variant;     // SPH_OPT_KERNEL_VARIANT (bit mask of SPH_VAR_*)
    int fuse_advect;     // set around the force launch of sph_step when the advect can ride in its finish
    int skip_acc;        // set by sph_step for every step but the last of a call: the fused force finish keeps its acceleration to itself
    bool acc_partial;    // sph_slab_forces with the interior advect fused: the interior targets' accelerations were consumed in the
                         // force finish and never written out; sph_download(ACCELERATION) refuses until something writes them all
    bool aux_stale;      // density / pressure of the fluid live in eos2 (written by the lean density finish), not yet in aux:
                         // sph_ensure_aux folds them in before anything reads aux.y / aux.z or a reference-API sort moves the records
    int opt_uniform;     // SPH_OPT_UNIFORM_FLUID: -1 auto, 0 off, 1 check once
    int uniform_state;   // -1 unknown, 0 the precondition fails, 1 holds (m_uniform valid)
    int pure_fluid;      // (with uniform_state == 1) the check found no solid particle at all among pure_fluid_n particles:
    int pure_fluid_n;    //   every m_V is m_V0 bit for bit (single context only: a slab rank's arrivals are never checked)
    float m_uniform;
    // timing
    hipEvent_t ev[SPH_MAX_TIMED_STEPS][5];
    int ev_used;
    bool slab_ev_open;  // slab mode: events [0..2] of ev[ev_used] are recorded, [3..4] follow in sph_slab_forces
    SphTimings tm;
    char err[512];
};

DevView sph_view(const SphContext* c);
static inline hipStream_t sph_stream(const SphContext* c) { return c->use_side ? c->side : c->stream; }
// particle positions / order / flags changed: neighbour lists and the non-empty-brick list are stale
#define SPH_BRICK_HEAVY 160  // targets from which a brick counts as heavy (a full 4x2x4 brick at rest has 256)
static inline void sph_invalidate_lists(SphContext* c) { c->lists_valid = false; c->bricks_valid = false; c->brec_valid = false; c->stg_kind = 0; c->k_kind = 0; }
int sph_fail(SphContext* c, int code, const char* what);
// the particle SET changed (records appended / dropped / re-selected): whatever a device-side check established about it is void
static inline void sph_forget_pure_fluid(SphContext* c) { c->pure_fluid = 0; c->pure_fluid_n = -1; }
// after a host synchronisation: did a device-side error flag rise since the last look (k_scan_fused's bounded wait)?  Marks the
// sort invalid and returns SPH_E_STATE with the message set; 0 otherwise.
int sph_check_device_flags(SphContext* c);

#define SPH_HIP(ctx, expr)                                                        \
    do {                                                                          \
        hipError_t e__ = (expr);                                                  \
        if (e__ != hipSuccess) {                                                  \
            snprintf((ctx)->err, sizeof((ctx)->err), "%s: %s (%s:%d)", #expr,     \
                     hipGetErrorString(e__), __FILE__, __LINE__);                 \
            return (int)e__;                                                      \
        }                                                                         \
    } while (0)

#define SPH_LAUNCH_CHECK(ctx) SPH_HIP(ctx, hipGetLastError())

// ---- launch entry points implemented in the .hip files --------------------
int sphk_hash_histogram(SphContext* c);
int sphk_scan(SphContext* c);
int sphk_sort_scatter(SphContext* c, bool sort_acc);
int sphk_rigid_partial16(SphContext* c, int object_id, int first, int count, double* out);
int sphk_rigid_apply16(SphContext* c, int object_id, const double* sums, int mode);
int sphk_scatter_rest(SphContext* c, const int* pid_dev, const float* x0_dev, int n);
struct BrickListArgs;
int sphk_brick_list_prepare(SphContext* c, BrickListArgs* a);  // sph_gather.hip: what the sort's place kernel needs to build the step's brick list
int sphk_gather(SphContext* c, int mode);
int sphk_gather_layers(SphContext* c, int mode, int lo, int hi, int lo2, int hi2);  // brick sweep, targets in x layers [lo,hi) u [lo2,hi2)
int sphk_pack_advected(SphContext* c, int first, int count, void* dst);
int sphk_eos(SphContext* c);
int sph_ensure_aux(SphContext* c);  // materialise density / pressure in aux if the lean density finish left them in eos2
int sphk_stats(SphContext* c, SphStats* out);  // synchronises
int sphk_df_convergence_test(SphContext* c, float offset, double eta, int slot);  // device-side compute_density_error + test
int sphk_check_uniform_fluid(SphContext* c);  // sets uniform_state / m_uniform (synchronises)
int sphk_df_density_error(SphContext* c, float offset, float* out_host);
int sphk_df_density_error_range(SphContext* c, float offset, int first, int count, double* out_host);
int sphk_df_scale_factor(SphContext* c, float s);
int sphk_df_predict_velocity(SphContext* c);
int sphk_df_advect(SphContext* c, bool fused_fluid_walls);
int sphk_advect(SphContext* c, bool fused_fluid_walls);
int sphk_advect_dyn_list(SphContext* c);  // dynamic rigid particles only
int sphk_fold_coupling_range(SphContext* c, int first, int count);  // ... of a record range, on the current sweep stream
int sphk_fold_coupling(SphContext* c);    // acc of the dynamic rigid particles += their fixed-point reaction sums (which are zeroed)
int sphk_advect_range(SphContext* c, int first, int count);
int sphk_enforce_boundary(SphContext* c, int particle_type);
int sphk_rigid_com(SphContext* c, int object_id, bool to_rest);
int sphk_rigid_solve(SphContext* c, int object_id);
int sphk_rigid_solve_all(SphContext* c, const int* ids, int n_ids, bool advect_first);  // every dynamic body + the solid wall passes, batched
int sphk_extract(SphContext* c, int field, void* dst);
int sphk_insert(SphContext* c, int field, const void* src);

enum GatherMode {
    GM_BVOL_STATIC = 0,
    GM_BVOL_DYNAMIC = 1,
    GM_DENSITY = 2,       // WCSPH.py:33-43 only
    GM_DENSITY_EOS = 3,   // + EOS (WCSPH.py:74-76), eos record, acc init of solids (fused step)
    GM_NONPRESSURE = 4,   // WCSPH.py:128-140
    GM_PRESSURE = 5,      // WCSPH.py:77-85 (second loop; EOS done by sphk_eos)
    GM_FORCE_FUSED = 6,   // K6 + K7 in one sweep (needs GM_DENSITY_EOS before)
    // ---- DFSPH (DFSPH.py); eos record = (dfsph_factor, density_adv, m, density) ----
    GM_DF_DENSITY = 7,         // DFSPH.py:37-47 (+ neighbour lists for every later sweep of the step)
    GM_DF_FACTOR = 8,          // DFSPH.py:116-154
    GM_DF_DENSITY_CHANGE = 9,  // DFSPH.py:157-197
    GM_DF_DENSITY_ADV = 10,    // DFSPH.py:200-221
    GM_DF_DIV_ITER = 11,       // DFSPH.py:285-321
    GM_DF_PRESSURE_ITER = 12,  // DFSPH.py:356-394
    GM_DF_NONPRESSURE = 13,    // DFSPH.py:49-97
    GM_FORCE_FUSED_U = 14,     // GM_FORCE_FUSED for fluids of one common particle mass: one gather per pair
    GM_DF_DIV_ITER_U = 15,     // GM_DF_DIV_ITER / GM_DF_PRESSURE_ITER with the neighbour's k_j = b_j * factor_j read from a
    GM_DF_PRESSURE_ITER_U = 16 //   4-byte array (written by the preceding density-change / -advection sweep)
};

#ifdef __HIPCC__
// ---------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ int sph_flags_material(int f) { return f & 0xFF; }
__device__ __forceinline__ int sph_flags_dynamic(int f) { return (f >> 8) & 1; }
__device__ __forceinline__ int sph_flags_object(int f) { return (int)((unsigned)f >> 9); }
__device__ __forceinline__ bool sph_is_fluid(int f) { return (f & 0xFF) == SPH_MATERIAL_FLUID; }
__device__ __forceinline__ bool sph_is_static_rigid(int f) { return (f & 0x1FF) == 0; }
__device__ __forceinline__ bool sph_is_dynamic_rigid(int f) { return (f & 0x1FF) == 0x100; }

// particle_system.py:287-289 pos_to_index: (pos / grid_size).cast(int), f32
// IEEE division (hipcc's default is the correctly rounded divide), truncation.
// Out-of-domain coordinates are clamped (UB in the reference).
__device__ __forceinline__ int sph_cell_coord(float p, float grid_size, int origin, int n) {
    int c = (int)(p / grid_size) - origin;
    c = c < 0 ? 0 : c;
    c = c > n - 1 ? n - 1 : c;
    return c;
}
// particle_system.py:292-294 flatten_grid_index (z fastest)
__device__ __forceinline__ int sph_flatten(const DevView& d, int cx, int cy, int cz) {
    return (cx * d.ny + cy) * d.nz + cz;
}

// wave64 inclusive scan (the wavefront primitive replacing scan_single_buffer.py:4-29's
// 32-lane shfl_up ladder)
__device__ __forceinline__ int sph_wave_inclusive_scan(int v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int n = __shfl_up(v, off, 64);
        if (lane >= off) v += n;
    }
    return v;
}

// WCSPH.py:76 ti.pow(x, exponent).  Every scene of the reference has an integer exponent (7): x^7 is four
// multiplies (~2 ulp) -- cheaper and an order of magnitude more accurate than exp2(e * log2 x) on the hardware
// transcendentals, whose error is amplified by stiffness * (x^e - 1) at x ~ 1.  exp_int is a kernel argument, so the
// loop is scalar control flow.  Non-integer exponents keep the transcendental form (FAST) or libm's powf.
template <bool FAST>
__device__ __forceinline__ float sph_tait_pow(const DevView& d, float x) {
    if (d.exp_int > 0) {
        float r = 1.0f, b = x;
        for (int e = d.exp_int; e; e >>= 1) {
            if (e & 1) r *= b;
            b *= b;
        }
        return r;
    }
    return FAST ? __builtin_amdgcn_exp2f(d.exponent * __builtin_amdgcn_logf(x)) : powf(x, d.exponent);
}

// sph_base.py:149-179 enforce_boundary_3D body for one particle
__device__ __forceinline__ void wall_collide(const DevView& d, const float hi[3], float4& xm, float4& vf) {
    const float pos[3] = {xm.x, xm.y, xm.z};
    float x[3] = {xm.x, xm.y, xm.z};
    float n[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (pos[a] > hi[a]) { n[a] += 1.0f; x[a] = hi[a]; }
        if (pos[a] <= d.pad) { n[a] += -1.0f; x[a] = d.pad; }
    }
    xm.x = x[0]; xm.y = x[1]; xm.z = x[2];
    const float len = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    if (len > 1e-6f) {
        // sph_base.py:118-123 simulate_collisions, c_f = 0.5
        const float vx_ = n[0] / len, vy_ = n[1] / len, vz_ = n[2] / len;
        const float vd = vf.x * vx_ + vf.y * vy_ + vf.z * vz_;
        vf.x -= (1.0f + 0.5f) * vd * vx_;
        vf.y -= (1.0f + 0.5f) * vd * vy_;
        vf.z -= (1.0f + 0.5f) * vd * vz_;
    }
}


// WCSPH.py:143-149 advect; FLUID_WALLS additionally applies
// enforce_boundary_3D(material_fluid) (sph_base.py:270-271) to fluid particles in
// the same pass (it only touches fluid, so it commutes with the rigid solve).
// one particle's symplectic-Euler update (+ fluid wall pass); shared by the in-place kernel and the packer
template <bool FLUID_WALLS>
__device__ __forceinline__ void advect_one(const DevView& d, const float hi[3], float4& xm, float4& vf, const float4 a) {
    const int fl = __float_as_int(vf.w);
    if (!sph_flags_dynamic(fl)) return;
    vf.x += d.dt * a.x; vf.y += d.dt * a.y; vf.z += d.dt * a.z;
    xm.x += d.dt * vf.x; xm.y += d.dt * vf.y; xm.z += d.dt * vf.z;
    if (FLUID_WALLS && sph_is_fluid(fl)) wall_collide(d, hi, xm, vf);
}


#endif  // __HIPCC__
