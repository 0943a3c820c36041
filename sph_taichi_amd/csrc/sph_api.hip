// sph_api.hip -- the extern "C" surface declared in include/sph_hip.h: context
// lifetime, field upload/download, one entry point per reference kernel, and
// sph_step = SPHBase.step() (sph_base.py:263-271) looped on the device.
#include <sched.h>
#include "sph_internal.h"

#define SCAN_TILE 2048

int sphk_init_pid(SphContext* c);
int sphk_build_dyn_list(SphContext* c);

static thread_local char g_err[256] = "";

int sph_fail(SphContext* c, int code, const char* what) {
    if (c) snprintf(c->err, sizeof(c->err), "%s", what);
    else snprintf(g_err, sizeof(g_err), "%s", what);
    return code;
}

DevView sph_view(const SphContext* c) {
    DevView d;
    const SphParams& p = c->p;
    d.N = c->N; d.G = c->G;
    d.nx = p.grid_num[0]; d.ny = p.grid_num[1]; d.nz = p.grid_num[2];
    d.tgt_lo = 0; d.tgt_hi = p.grid_num[0]; d.tgt_lo2 = d.tgt_hi2 = 0;
#ifdef SPH_PROFILE
    d.ablate = c->opt_ablate;
    d.prof_ts = (c->opt_ablate & (3 << 28)) ? reinterpret_cast<unsigned long long*>(c->stage) : nullptr;  // (the staging buffer: 16 B per particle of capacity, 64 B per hardware block needed)
#endif
    d.drop_outside = c->opt_drop_outside;
    d.sort_by_pid = c->opt_sort_by_pid;
    d.ox = p.cell_origin[0]; d.oy = p.cell_origin[1]; d.oz = p.cell_origin[2];
    d.grid_size = p.grid_size; d.h = p.support_radius; d.inv_h = 1.0f / p.support_radius;
    d.d2 = p.particle_diameter * p.particle_diameter;  // WCSPH.py:96
    d.m_V0 = p.m_V0; d.rho0 = p.density_0; d.stiffness = p.stiffness; d.exponent = p.exponent;
    d.exp_int = (p.exponent >= 1.0f && p.exponent <= 32.0f && p.exponent == (float)(int)p.exponent) ? (int)p.exponent : 0;
    d.sigma = p.surface_tension; d.dt = p.dt;
    d.gx = p.g[0]; d.gy = p.g[1]; d.gz = p.g[2];
    d.pad = p.padding;
    d.k_w = p.k_w; d.k_dw = p.k_dw; d.visc_d_nu = p.visc_d_nu; d.visc_eps = p.visc_eps;
    d.w_zero = p.k_w * (6.0f * 0.0f - 6.0f * 0.0f + 1.0f);  // W(0), sph_base.py:41
    {   // W(d): q = d/h (sph_base.py:36-44)
        const float q = p.particle_diameter / p.support_radius;
        if (q <= 0.5f) d.w_d = p.k_w * (6.0f * q * q * q - 6.0f * q * q + 1.0f);
        else if (q <= 1.0f) d.w_d = p.k_w * 2.0f * (1.0f - q) * (1.0f - q) * (1.0f - q);
        else d.w_d = 0.0f;
    }
    const int o = c->in_off;
    d.xm = c->xm[c->cur] + o; d.vf = c->vf[c->cur] + o; d.aux = c->aux[c->cur] + o; d.key = c->key[c->cur] + o;
    d.eos = c->eos; d.eos2 = reinterpret_cast<float2*>(c->eos); d.acc = c->acc + o; d.acc_fx = c->acc_fx + 3 * (size_t)o; d.cell_end = c->cell_end;
    d.x0_cold = c->x0_cold; d.rigid_rest_cm = c->rigid_rest_cm; d.polar_fb = c->dyn_count ? c->dyn_count + 3 : nullptr;
    d.m_eps = c->df.m_eps;
    d.stg = c->stg; d.gat = c->gat; d.kbuf = reinterpret_cast<float*>(c->gat);
    d.m_u = c->m_uniform; d.write_sg = 0; d.write_k = 0;
    d.whx = p.wall_hi[0]; d.why = p.wall_hi[1]; d.whz = p.wall_hi[2];
    d.fuse_advect = c->fuse_advect;
    d.df_bpart = c->df_collect ? c->df_bpart : nullptr;
    d.gate = (c->df_epoch && c->opt_df_runahead) ? c->df_gate : nullptr;   // (without run-ahead no body is ever enqueued past convergence)
    d.gate_epoch = c->df_epoch;
    d.fx_scale = ldexp(1.0, c->rigid_fx_exp);
    d.store_acc = !(c->fuse_advect && c->skip_acc);
    d.rigid_from_x0 = 0;   // set by the two launchers the option speaks of (sphk_rigid_com to_rest, sphk_rigid_partial16)
    return d;
}

// out[k] = number of records in local x layers [0, layers[k])  (the scanned cell table at a layer boundary)
__global__ void k_read_layer_offsets(const int* __restrict__ cell_end, const int* __restrict__ layers, int n,
                                     int per_layer, int* __restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int L = layers[k];
    out[k] = L > 0 ? cell_end[(size_t)L * per_layer - 1] : 0;
}

__global__ void k_read_cells(const int* __restrict__ cell_end, CellIdx16 ix, int* __restrict__ out) {
    const int k = threadIdx.x;
    if (k < 16 && ix.v[k] >= 0) out[k] = cell_end[ix.v[k]];
}

// Every entry that synchronises with the stream looks here afterwards (sph_sync, sph_download, sph_get_stats, sph_get_timings,
// the DFSPH solver loops): a scan tile whose bounded wait ran out has substituted 0 for a predecessor's total, so the cell
// table of that sort -- and everything computed from it since -- is wrong.  The flag lives in mapped host memory.
int sph_check_device_flags(SphContext* c) {
    if (c->h_pinned && ((volatile int*)c->h_pinned)[16]) {
        ((volatile int*)c->h_pinned)[16] = 0;
        c->sorted = c->have_prefix = c->have_keys = false;
        sph_invalidate_lists(c);
        return sph_fail(c, SPH_E_STATE, "prefix sum: a scan tile never saw a predecessor's total (k_scan_fused's bounded wait ran out); "
                                        "the cell table of that sort and every result since are invalid: re-upload the state and sort again");
    }
    return 0;
}

extern "C" {

int32_t sph_abi_version(void) { return SPH_ABI_VERSION; }

int32_t sph_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* sph_last_error(const SphContext* ctx) { return ctx ? ctx->err : g_err; }

static int alloc_dev(SphContext* c, void** p, size_t bytes) {
    hipError_t e = hipMalloc(p, bytes ? bytes : 16);
    if (e != hipSuccess) {
        snprintf(c->err, sizeof(c->err), "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return SPH_E_NOMEM;
    }
    e = hipMemsetAsync(*p, 0, bytes ? bytes : 16, c->stream);
    return e == hipSuccess ? 0 : (int)e;
}

int32_t sph_create(const SphParams* params, int32_t device, void* stream, SphContext** out) {
    if (!params || !out) return sph_fail(nullptr, SPH_E_INVALID, "sph_create: null argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return sph_fail(nullptr, SPH_E_NO_DEVICE, "sph_create: no HIP device visible (libsph_hip needs an MI355X/gfx950 GPU)");
    if (device < 0 || device >= ndev) return sph_fail(nullptr, SPH_E_INVALID, "sph_create: bad device index");
    if (params->n_particles < 0 || params->capacity < params->n_particles || params->grid_num[0] <= 0 ||
        params->grid_num[1] <= 0 || params->grid_num[2] <= 0 || params->n_objects < 0)
        return sph_fail(nullptr, SPH_E_INVALID, "sph_create: bad sizes");
    if ((long long)params->grid_num[0] * params->grid_num[1] * params->grid_num[2] > 0x7ffff000LL)
        return sph_fail(nullptr, SPH_E_INVALID, "sph_create: grid too large for i32 cell ids");
    if (hipSetDevice(device) != hipSuccess) return sph_fail(nullptr, SPH_E_NO_DEVICE, "hipSetDevice failed");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        snprintf(g_err, sizeof(g_err), "sph_create: device arch %s is not gfx950", prop.gcnArchName);
        return SPH_E_NO_DEVICE;
    }
    SphContext* c = new SphContext();
    memset(c, 0, sizeof(*c));
    c->p = *params;
    c->device = device;
    c->N = params->n_particles;
    c->cap = params->capacity > 0 ? params->capacity : 1;
    c->G = params->grid_num[0] * params->grid_num[1] * params->grid_num[2];
    c->nx_alloc = params->grid_num[0];
    c->tgt_layers[0] = c->tgt_layers[2] = 0;
    c->tgt_layers[1] = c->tgt_layers[3] = params->grid_num[0];
    c->opt_gather_impl = 1;
    c->opt_fused = 1;
    c->opt_timing = 0;
    c->timing_phase = 0;
    c->opt_brick_shape = 0;
    c->opt_rigid_batch = 1;
    c->opt_df_fuse_err = 1;
    c->df_collect = 0;
    c->df_bpart_valid = false;
    c->opt_df_runahead = 0;   // measured (r05): running ahead costs 2 % more than the bubbles it removes
    c->opt_exact_math = 0;
    c->opt_rigid_x0 = 0;
    c->opt_pure_instance = 1;
    c->opt_brick_rec = 1;
    if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
    else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
            delete c;
            return sph_fail(nullptr, SPH_E_NO_DEVICE, "hipStreamCreate failed");
        }
        c->own_stream = true;
    }
    const size_t cap = (size_t)c->cap;
    c->scan_blocks = (c->G + 1 + SCAN_TILE - 1) / SCAN_TILE;  // + the virtual cell G of slab mode
    int rc = 0;
    for (int s = 0; s < 2 && !rc; ++s) {
        rc = rc ? rc : alloc_dev(c, (void**)&c->xm[s], cap * 16);
        rc = rc ? rc : alloc_dev(c, (void**)&c->vf[s], cap * 16);
        rc = rc ? rc : alloc_dev(c, (void**)&c->aux[s], cap * 16);
        rc = rc ? rc : alloc_dev(c, (void**)&c->key[s], cap * 4);
    }
    rc = rc ? rc : alloc_dev(c, (void**)&c->eos, cap * 16);
    rc = rc ? rc : alloc_dev(c, (void**)&c->stg, cap * 16);
    rc = rc ? rc : alloc_dev(c, (void**)&c->gat, cap * 16);
    rc = rc ? rc : alloc_dev(c, (void**)&c->acc, cap * 16);
    rc = rc ? rc : alloc_dev(c, (void**)&c->acc_tmp, cap * 16);
    rc = rc ? rc : alloc_dev(c, (void**)&c->acc_fx, cap * 3 * sizeof(long long));  // all zero except between a coupling sweep and its fold
    rc = rc ? rc : alloc_dev(c, (void**)&c->cell_buf[0], (size_t)c->scan_blocks * SCAN_TILE * 4);
    rc = rc ? rc : alloc_dev(c, (void**)&c->cell_buf[1], (size_t)c->scan_blocks * SCAN_TILE * 4);
    if (!rc) { c->cell_cur = 0; c->cell_end = c->cell_buf[0]; c->next_cells_zero = false; }
    rc = rc ? rc : alloc_dev(c, (void**)&c->rank_off, cap * 4);
    rc = rc ? rc : alloc_dev(c, (void**)&c->idx_unstable, cap * 4);
    rc = rc ? rc : alloc_dev(c, (void**)&c->scan_status, (size_t)(c->scan_blocks + 1) * 8);
    {   // four consecutive entries of a particle form one 8-byte word; a group of entries spans a power of two of bytes
        int sh = 3;
        while (((size_t)1 << sh) < cap * 8) ++sh;
        const bool reach = (((unsigned long long)(SPH_GLIST_ROWS / 4 + 4)) << sh) <= (1ull << 32);   // (+ the readers' look-ahead)
        c->glist_shift = reach ? sh : 0;
        rc = rc ? rc : alloc_dev(c, (void**)&c->glist, reach ? ((size_t)(SPH_GLIST_ROWS / 4) << sh) : 16);
    }
    rc = rc ? rc : alloc_dev(c, (void**)&c->gcnt, cap);
    // bricks have a 4x2-column footprint and a height the list builder chooses (k_brick_list): at worst one per z layer
    c->brick_cap = ((params->grid_num[0] + 3) / 4) * ((params->grid_num[1] + 1) / 2) * params->grid_num[2] + 8;
    rc = rc ? rc : alloc_dev(c, (void**)&c->brick_list, (size_t)c->brick_cap * 8);
    rc = rc ? rc : alloc_dev(c, (void**)&c->brick_count, 16);
    rc = rc ? rc : alloc_dev(c, (void**)&c->brick_rec, (size_t)c->brick_cap * 32 * sizeof(int4));   // 512 B per brick slot
    rc = rc ? rc : alloc_dev(c, (void**)&c->brick_list2, (size_t)c->brick_cap * 8);
    rc = rc ? rc : alloc_dev(c, (void**)&c->brick_count2, 16);
    const size_t cold = params->cold_capacity > 0 ? (size_t)params->cold_capacity : cap;
    if (cold < cap) { sph_destroy(c); return sph_fail(nullptr, SPH_E_INVALID, "sph_create: cold_capacity < capacity"); }
    c->cold_cap = (int)cold;
    rc = rc ? rc : alloc_dev(c, (void**)&c->x0_cold, cold * 12);
    rc = rc ? rc : alloc_dev(c, (void**)&c->color_cold, cold * 12);
    rc = rc ? rc : alloc_dev(c, (void**)&c->rigid_rest_cm, (size_t)(params->n_objects > 0 ? params->n_objects : 1) * 12);
    rc = rc ? rc : alloc_dev(c, (void**)&c->dyn_list, cap * 4);
    rc = rc ? rc : alloc_dev(c, (void**)&c->dyn_count, 16);
    c->rigid_part_blocks = (c->cap + 255) / 256 + 1;
    rc = rc ? rc : alloc_dev(c, (void**)&c->rigid_part, (size_t)c->rigid_part_blocks * 16 * sizeof(double));
    rc = rc ? rc : alloc_dev(c, (void**)&c->rigid_R, 16 * sizeof(float));
    rc = rc ? rc : alloc_dev(c, (void**)&c->df_err, sizeof(double));
    rc = rc ? rc : alloc_dev(c, (void**)&c->df_part, SPH_DF_ERR_BLOCKS * sizeof(double));
    rc = rc ? rc : alloc_dev(c, (void**)&c->df_bpart, (size_t)c->brick_cap * sizeof(double));
    if (!rc && hipHostMalloc((void**)&c->h_df_err, sizeof(double), hipHostMallocMapped) != hipSuccess) rc = SPH_E_NOMEM;
    if (!rc && hipHostMalloc((void**)&c->h_df_slot, 4 * sizeof(*c->h_df_slot), hipHostMallocMapped) != hipSuccess) rc = SPH_E_NOMEM;
    rc = rc ? rc : alloc_dev(c, (void**)&c->df_gate, 16);
    for (int k = 0; k < 4 && !rc; ++k)
        if (hipEventCreateWithFlags(&c->ev_df[k], hipEventDisableTiming) != hipSuccess) rc = SPH_E_NOMEM;
    c->stage_bytes = cap * 16 > (size_t)c->G * 4 ? cap * 16 : (size_t)c->G * 4;
    if (c->stage_bytes < 65536) c->stage_bytes = 65536;  // (also the scratch of sph_get_stats' partial rows)
    rc = rc ? rc : alloc_dev(c, &c->stage, c->stage_bytes);
    if (!rc && hipHostMalloc((void**)&c->h_pinned, 32 * sizeof(int), hipHostMallocMapped) != hipSuccess) rc = SPH_E_NOMEM;
    if (!rc) {   // [16]: raised (from the device, through the mapping) by a scan tile whose bounded wait ran out; read by sph_sync
        memset(c->h_pinned, 0, 32 * sizeof(int));
        int* dev = nullptr;
        if (hipHostGetDevicePointer((void**)&dev, c->h_pinned, 0) != hipSuccess) rc = SPH_E_NOMEM;
        else c->scan_err = dev + 16;
    }
    if (!rc && hipEventCreateWithFlags(&c->ev_off, hipEventDisableTiming) != hipSuccess) rc = SPH_E_NOMEM;
    if (!rc && hipEventCreateWithFlags(&c->ev_pack, hipEventDisableTiming) != hipSuccess) rc = SPH_E_NOMEM;
    if (!rc && hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess) rc = SPH_E_NOMEM;
    if (!rc && hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess) rc = SPH_E_NOMEM;
    if (!rc) rc = sphk_init_pid(c);
    for (int s = 0; s < SPH_MAX_TIMED_STEPS && !rc; ++s)
        for (int k = 0; k < 5 && !rc; ++k)
            if (hipEventCreate(&c->ev[s][k]) != hipSuccess) rc = SPH_E_NOMEM;
    if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = SPH_E_NO_DEVICE;
    if (rc) {
        snprintf(g_err, sizeof(g_err), "sph_create failed: %s", c->err);
        sph_destroy(c);
        return rc;
    }
    c->n_dyn_host = -1;  // unknown until material / is_dynamic are uploaded
    c->rigid_fx_exp = 30;
    sph_invalidate_lists(c);
    c->opt_uniform = -1; c->uniform_state = -1; c->m_uniform = 0.0f;  // SPH_OPT_UNIFORM_FLUID: auto
    c->opt_variant = SPH_VAR_DEFAULT;
    if (const char* e = getenv("SPH_KERNEL_VARIANT")) c->opt_variant = atoi(e) & 31;  // A/B aid: the default mask of every context of this process
    memset(&c->df_stats, 0, sizeof(c->df_stats));
    c->df.enable_divergence_solver = 1; c->df.m_max_iterations_v = 100; c->df.m_max_iterations = 100;  // DFSPH.py:12-20
    c->df.fluid_particle_num = 0; c->df.m_eps = 1e-5f; c->df.reserved_ = 0.0f; c->df.max_error_V = 0.1; c->df.max_error = 0.05;
    *out = c;
    return 0;
}

int32_t sph_destroy(SphContext* c) {
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    void* ptrs[] = {c->xm[0], c->xm[1], c->vf[0], c->vf[1], c->aux[0], c->aux[1], c->key[0], c->key[1], c->eos, c->stg, c->gat, c->acc,
                    c->acc_tmp, c->cell_buf[0], c->cell_buf[1], c->rank_off, c->idx_unstable, c->scan_status, c->x0_cold, c->color_cold,
                    c->rigid_rest_cm, c->dyn_list, c->dyn_count, c->acc_fx, c->rigid_part, c->rigid_R, c->df_err, c->df_part, c->df_bpart, c->stage, c->glist, c->gcnt, c->brick_list, c->brick_count, c->brick_list2, c->brick_count2, c->brick_rec};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (int s = 0; s < SPH_MAX_TIMED_STEPS; ++s)
        for (int k = 0; k < 5; ++k) if (c->ev[s][k]) (void)hipEventDestroy(c->ev[s][k]);
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    if (c->h_df_err) (void)hipHostFree(c->h_df_err);
    if (c->h_df_slot) (void)hipHostFree(c->h_df_slot);
    if (c->df_gate) (void)hipFree(c->df_gate);
    for (int k = 0; k < 4; ++k) if (c->ev_df[k]) (void)hipEventDestroy(c->ev_df[k]);
    if (c->ev_off) (void)hipEventDestroy(c->ev_off);
    if (c->ev_pack) (void)hipEventDestroy(c->ev_pack);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

int32_t sph_set_option(SphContext* c, int32_t option, int32_t value) {
    if (!c) return SPH_E_INVALID;
    switch (option) {
        case SPH_OPT_GATHER_IMPL: if (value < 0 || value > 1) return sph_fail(c, SPH_E_INVALID, "gather impl must be 0 or 1"); c->opt_gather_impl = value; c->uniform_state = -1; return 0;
        case SPH_OPT_TIMING: c->opt_timing = value > 0 ? value : 0; c->timing_phase = 0; return 0;
        case SPH_OPT_FUSED_STEP: c->opt_fused = value ? 1 : 0; return 0;
        case SPH_OPT_BRICK_SHAPE: if (value < 0 || value > 1) return sph_fail(c, SPH_E_INVALID, "brick shape must be 0 (adaptive height) or 1 (fixed 4x2x4)"); c->opt_brick_shape = value; sph_invalidate_lists(c); return 0;
        case SPH_OPT_NO_DYNAMIC_SOLIDS: c->opt_no_dynamic = value ? 1 : 0; c->n_dyn_host = -1; c->uniform_state = -1; return 0;
        case SPH_OPT_DEBUG_ABLATE:
#ifndef SPH_PROFILE
            if (value != 0) return sph_fail(c, SPH_E_INVALID, "section ablation exists only in the profiling build (python -m sph_taichi_amd.build --profile; bench.py --ablate loads it)");
#endif
            c->opt_ablate = value;
            return 0;
        case SPH_OPT_SLAB_DROP_OUTSIDE: c->opt_drop_outside = value ? 1 : 0; c->uniform_state = -1; return 0;
        case SPH_OPT_SORT_BY_PID: c->opt_sort_by_pid = value ? 1 : 0; return 0;
        case SPH_OPT_RIGID_BATCH: c->opt_rigid_batch = value ? 1 : 0; return 0;
        case SPH_OPT_EXACT_MATH: c->opt_exact_math = value ? 1 : 0; sph_invalidate_lists(c); return 0;
        case SPH_OPT_DF_RUNAHEAD: c->opt_df_runahead = value ? 1 : 0; return 0;
        case SPH_OPT_KERNEL_VARIANT:
            if (value < -1 || value > 63 || (value > 0 && (value & 6) == 6)) return sph_fail(c, SPH_E_INVALID, "kernel variant must be -1 (default) or a mask of SPH_VAR_* (GAT_LDS and GAT_LDS4 exclude each other)");
            c->opt_variant = value < 0 ? SPH_VAR_DEFAULT : value;
            sph_invalidate_lists(c);
            return 0;
        case SPH_OPT_UNIFORM_FLUID: if (value < -1 || value > 1) return sph_fail(c, SPH_E_INVALID, "uniform-fluid option must be -1, 0 or 1"); c->opt_uniform = value; c->uniform_state = -1; return 0;
        case SPH_OPT_RIGID_SUMS_FROM_X0: c->opt_rigid_x0 = value ? 1 : 0; return 0;
        case SPH_OPT_PURE_FLUID_INSTANCE:
            if (value < 0 || value > 2) return sph_fail(c, SPH_E_INVALID, "pure-fluid instance option must be 0, 1 or 2");
            c->opt_pure_instance = value; sph_invalidate_lists(c); return 0;
        case SPH_OPT_BRICK_RECORDS: c->opt_brick_rec = value ? 1 : 0; sph_invalidate_lists(c); return 0;
        case SPH_OPT_DF_FUSE_ERROR: c->opt_df_fuse_err = value ? 1 : 0; return 0;
    }
    return sph_fail(c, SPH_E_INVALID, "unknown option");
}

int32_t sph_get_option(const SphContext* c, int32_t option, int32_t* value) {
    if (!c || !value) return SPH_E_INVALID;
    switch (option) {
        case SPH_OPT_GATHER_IMPL: *value = c->opt_gather_impl; return 0;
        case SPH_OPT_TIMING: *value = c->opt_timing; return 0;
        case SPH_OPT_FUSED_STEP: *value = c->opt_fused; return 0;
        case SPH_OPT_BRICK_SHAPE: *value = c->opt_brick_shape; return 0;
        case SPH_OPT_NO_DYNAMIC_SOLIDS: *value = c->opt_no_dynamic; return 0;
        case SPH_OPT_DEBUG_ABLATE: *value = c->opt_ablate; return 0;
        case SPH_OPT_SLAB_DROP_OUTSIDE: *value = c->opt_drop_outside; return 0;
        case SPH_OPT_UNIFORM_FLUID: *value = c->opt_uniform; return 0;
        case SPH_OPT_SORT_BY_PID: *value = c->opt_sort_by_pid; return 0;
        case SPH_OPT_RIGID_BATCH: *value = c->opt_rigid_batch; return 0;
        // the EFFECTIVE state: exact-math instances exist for the uniform-fluid step only (sph_hip.h); a scene found not to
        // be one runs the general sweeps with v_rsq / v_rcp whatever was requested, and an A/B must be able to see that
        case SPH_OPT_EXACT_MATH: *value = (c->opt_exact_math && c->uniform_state != 0) ? 1 : 0; return 0;
        case SPH_OPT_KERNEL_VARIANT: *value = c->opt_variant; return 0;
        case SPH_OPT_DF_RUNAHEAD: *value = c->opt_df_runahead; return 0;
        case SPH_OPT_UNIFORM_FLUID_STATE: *value = c->uniform_state; return 0;
        case SPH_OPT_RIGID_SUMS_FROM_X0: *value = c->opt_rigid_x0; return 0;
        case SPH_OPT_PURE_FLUID_INSTANCE: *value = c->opt_pure_instance; return 0;
        case SPH_OPT_BRICK_RECORDS: *value = c->opt_brick_rec; return 0;
        case SPH_OPT_DF_FUSE_ERROR: *value = c->opt_df_fuse_err; return 0;
    }
    return SPH_E_INVALID;
}

int32_t sph_set_params(SphContext* c, const SphParams* p) {
    if (!c || !p) return SPH_E_INVALID;
    if (p->capacity != c->p.capacity || p->n_objects != c->p.n_objects || p->cold_capacity != c->p.cold_capacity ||
        memcmp(p->grid_num, c->p.grid_num, sizeof(p->grid_num)) != 0 ||
        memcmp(p->cell_origin, c->p.cell_origin, sizeof(p->cell_origin)) != 0)
        return sph_fail(c, SPH_E_INVALID, "sph_set_params: sizes differ from sph_create");
    const int n = c->p.n_particles;
    c->p = *p;
    c->p.n_particles = n;
    return 0;
}

int32_t sph_set_dt(SphContext* c, float dt) {
    if (!c) return SPH_E_INVALID;
    c->p.dt = dt;
    return 0;
}

int32_t sph_set_particle_count(SphContext* c, int32_t n) {
    if (!c || n < 0 || n > c->cap) return sph_fail(c, SPH_E_INVALID, "particle count out of range");
    c->N = n;
    c->n_dyn_host = -1;
    sph_forget_pure_fluid(c);
    sph_invalidate_lists(c);
    c->aux_stale = false;
    c->have_keys = c->have_prefix = false;
    return 0;
}

static size_t field_bytes(const SphContext* c, int field) {
    switch (field) {
        case SPH_F_X: case SPH_F_X_0: case SPH_F_V: case SPH_F_ACCELERATION: case SPH_F_COLOR: return (size_t)c->N * 12;
        case SPH_F_GRID_PARTICLES_NUM: return (size_t)c->G * 4;
        case SPH_F_RIGID_REST_CM: return (size_t)c->p.n_objects * 12;
        default: return (size_t)c->N * 4;
    }
}

int32_t sph_upload(SphContext* c, int32_t field, const void* host, size_t bytes) {
    if (!c || !host) return SPH_E_INVALID;
    if (field < 0 || field >= SPH_F_COUNT_ || field == SPH_F_GRID_IDS || field == SPH_F_GRID_PARTICLES_NUM)
        return sph_fail(c, SPH_E_INVALID, "sph_upload: field is not uploadable");
    if (bytes != field_bytes(c, field)) return sph_fail(c, SPH_E_INVALID, "sph_upload: size mismatch");
    SPH_HIP(c, hipSetDevice(c->device));
    if (bytes == 0) return 0;
    if (field == SPH_F_DENSITY || field == SPH_F_PRESSURE) { int rc0 = sph_ensure_aux(c); if (rc0) return rc0; }
    if (field == SPH_F_ACCELERATION) c->acc_partial = false;
    if (field == SPH_F_RIGID_REST_CM) {
        SPH_HIP(c, hipMemcpyAsync(c->rigid_rest_cm, host, bytes, hipMemcpyHostToDevice, c->stream));
        SPH_HIP(c, hipStreamSynchronize(c->stream));
        return 0;
    }
    SPH_HIP(c, hipMemcpyAsync(c->stage, host, bytes, hipMemcpyHostToDevice, c->stream));
    int rc = sphk_insert(c, field, c->stage);
    if (rc) return rc;
    SPH_HIP(c, hipStreamSynchronize(c->stream));  // host buffer is only borrowed for the call
    if (field == SPH_F_MATERIAL || field == SPH_F_IS_DYNAMIC || field == SPH_F_DENSITY) c->n_dyn_host = -1;  // (density: the scale of the rigid sums)
    if (field == SPH_F_MATERIAL || field == SPH_F_M || field == SPH_F_M_V) c->uniform_state = -1;
    if (field == SPH_F_X) c->have_keys = c->have_prefix = false;
    return 0;
}

int32_t sph_download(SphContext* c, int32_t field, void* host, size_t bytes) {
    if (!c || !host) return SPH_E_INVALID;
    if (field < 0 || field >= SPH_F_COUNT_) return sph_fail(c, SPH_E_INVALID, "sph_download: bad field");
    if (bytes != field_bytes(c, field)) return sph_fail(c, SPH_E_INVALID, "sph_download: size mismatch");
    SPH_HIP(c, hipSetDevice(c->device));
    if (bytes == 0) return 0;
    if (field == SPH_F_ACCELERATION && c->acc_partial)
        return sph_fail(c, SPH_E_STATE, "sph_download(acceleration): the last sph_slab_forces consumed the interior particles' accelerations "
                                        "in its fused finish and did not write them out (stale values would be returned); call "
                                        "sph_compute_non_pressure_forces + sph_compute_pressure_forces, or run with SPH_OPT_FUSED_STEP 0");
    const void* src = c->stage;
    if (field == SPH_F_GRID_PARTICLES_NUM) src = c->cell_end;
    else if (field == SPH_F_RIGID_REST_CM) src = c->rigid_rest_cm;
    else {
        int rc = (field == SPH_F_DENSITY || field == SPH_F_PRESSURE) ? sph_ensure_aux(c) : 0;
        rc = rc ? rc : sphk_extract(c, field, c->stage);
        if (rc) return rc;
    }
    SPH_HIP(c, hipMemcpyAsync(host, src, bytes, hipMemcpyDeviceToHost, c->stream));
    SPH_HIP(c, hipStreamSynchronize(c->stream));
    return sph_check_device_flags(c);
}

// number (and current-order list) of dynamic rigid particles; refreshed lazily
static int refresh_dyn(SphContext* c) {
    if (c->n_dyn_host >= 0) return 0;
    if (c->opt_no_dynamic) { c->n_dyn_host = 0; return 0; }  // the host vouches: no dynamic solids anywhere
    int rc = sphk_build_dyn_list(c);
    if (rc) return rc;
    int nr[2] = {0, 0};
    SPH_HIP(c, hipMemcpyAsync(nr, c->dyn_count, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    SPH_HIP(c, hipStreamSynchronize(c->stream));
    c->n_dyn_host = nr[0];
    // Scale of the fixed-point shape-matching sums: n terms of at most m_V0 rho_max L^2 (L = twice the domain's largest
    // extent, a generous bound for |x|, |x - cm| and |x_0 - cm_rest|) must stay below 2^62.
    float rho_max;
    memcpy(&rho_max, &nr[1], sizeof(float));
    double L = 1.0;
    for (int a = 0; a < 3; ++a) L = fmax(L, 2.0 * fabs((double)c->p.domain_size[a]) + 1.0);
    const double bound = fmax((double)nr[0], 1.0) * fmax((double)c->p.m_V0 * fmax((double)rho_max, 1.0), 1e-30) * L * L;
    int e = 0;
    (void)frexp(bound, &e);          // bound < 2^e
    c->rigid_fx_exp = 62 - e;
    if (c->rigid_fx_exp > 80) c->rigid_fx_exp = 80;
    if (c->rigid_fx_exp < -200) c->rigid_fx_exp = -200;
    return 0;
}

#define ENTER(c)                                  \
    if (!(c)) return SPH_E_INVALID;               \
    SPH_HIP((c), hipSetDevice((c)->device));

int32_t sph_update_grid_id(SphContext* c) {
    ENTER(c);
    int rc = sphk_hash_histogram(c);
    if (rc) return rc;
    c->have_keys = true;
    c->have_prefix = false;
    return 0;
}

int32_t sph_prefix_sum(SphContext* c) {
    ENTER(c);
    if (!c->have_keys) return sph_fail(c, SPH_E_STATE, "sph_prefix_sum before sph_update_grid_id");
    int rc = sphk_scan(c);
    if (rc) return rc;
    c->have_prefix = true;
    return 0;
}

static int counting_sort(SphContext* c, bool sort_acc) {
    if (!c->have_prefix) return sph_fail(c, SPH_E_STATE, "sph_counting_sort before sph_prefix_sum");
    int rc = refresh_dyn(c);
    if (rc) return rc;
    rc = sphk_sort_scatter(c, sort_acc);
    if (rc) return rc;
    c->in_off = 0;         // the scatter writes the other set from record 0
    c->have_keys = false;  // the histogram offsets are consumed
    c->sorted = true;
    return 0;
}

int32_t sph_counting_sort(SphContext* c) {
    ENTER(c);
    int rc = sph_ensure_aux(c);  // the reference's sort carries density and pressure along (particle_system.py:332-369)
    return rc ? rc : counting_sort(c, true);
}

int32_t sph_initialize_particle_system(SphContext* c) {
    ENTER(c);
    int rc = sph_ensure_aux(c);
    rc = rc ? rc : sph_update_grid_id(c);
    rc = rc ? rc : sph_prefix_sum(c);
    rc = rc ? rc : counting_sort(c, true);
    return rc;
}

static int need_sorted(SphContext* c, const char* who) {
    if (!c->sorted || !c->have_prefix) {
        snprintf(c->err, sizeof(c->err), "%s needs the neighbour structure: call sph_initialize_particle_system first", who);
        return SPH_E_STATE;
    }
    return 0;
}

int32_t sph_compute_boundary_volume(SphContext* c, int32_t dynamic) {
    ENTER(c);
    int rc = need_sorted(c, "sph_compute_boundary_volume");
    rc = rc ? rc : refresh_dyn(c);
    return rc ? rc : sphk_gather(c, dynamic ? GM_BVOL_DYNAMIC : GM_BVOL_STATIC);
}

int32_t sph_compute_densities(SphContext* c) {
    ENTER(c);
    int rc = need_sorted(c, "sph_compute_densities");
    return rc ? rc : sphk_gather(c, GM_DENSITY);
}

int32_t sph_compute_non_pressure_forces(SphContext* c) {
    ENTER(c);
    int rc = need_sorted(c, "sph_compute_non_pressure_forces");
    rc = rc ? rc : sph_ensure_aux(c);
    rc = rc ? rc : sphk_gather(c, GM_NONPRESSURE);
    if (!rc) c->acc_partial = false;   // every particle's acceleration has been (re)written
    return rc;
}

int32_t sph_compute_pressure_forces(SphContext* c) {
    ENTER(c);
    int rc = need_sorted(c, "sph_compute_pressure_forces");
    rc = rc ? rc : sph_ensure_aux(c);
    rc = rc ? rc : sphk_eos(c);                      // WCSPH.py:71-76
    return rc ? rc : sphk_gather(c, GM_PRESSURE);    // WCSPH.py:77-85
}

int32_t sph_advect(SphContext* c) {
    ENTER(c);
    return sphk_advect(c, false);
}

int32_t sph_enforce_boundary_3D(SphContext* c, int32_t particle_type) {
    ENTER(c);
    int rc = refresh_dyn(c);
    return rc ? rc : sphk_enforce_boundary(c, particle_type);
}

int32_t sph_compute_rigid_rest_cm(SphContext* c, int32_t object_id) {
    ENTER(c);
    if (object_id < 0 || object_id >= c->p.n_objects) return sph_fail(c, SPH_E_INVALID, "object id out of range");
    int rc = refresh_dyn(c);
    return rc ? rc : sphk_rigid_com(c, object_id, true);
}

int32_t sph_solve_constraints(SphContext* c, int32_t object_id, float* R_out) {
    ENTER(c);
    if (object_id < 0 || object_id >= c->p.n_objects) return sph_fail(c, SPH_E_INVALID, "object id out of range");
    int rc = refresh_dyn(c);
    rc = rc ? rc : sphk_rigid_solve(c, object_id);
    if (rc) return rc;
    if (R_out) {
        float tmp[12] = {0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1};
        if (c->n_dyn_host > 0) {
            SPH_HIP(c, hipMemcpyAsync(tmp, c->rigid_R, sizeof(tmp), hipMemcpyDeviceToHost, c->stream));
            SPH_HIP(c, hipStreamSynchronize(c->stream));
        }
        memcpy(R_out, tmp + 3, 9 * sizeof(float));
    }
    return 0;
}

int32_t sph_compute_com(SphContext* c, int32_t object_id, float* cm_out) {
    ENTER(c);
    if (!cm_out) return SPH_E_INVALID;
    int rc = refresh_dyn(c);
    rc = rc ? rc : sphk_rigid_com(c, object_id, false);
    if (rc) return rc;
    SPH_HIP(c, hipMemcpyAsync(cm_out, c->rigid_R, 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    SPH_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
}

// SPH_OPT_TIMING k > 0: is the step that starts now one of the timed ones?  (An event record is a packet of its own
// between two kernels: five per step cost the C3' step 4 % -- k = 8 keeps the phase means and costs 0.5 %.)
static bool sph_timed_step(SphContext* c) {
    if (c->opt_timing <= 0) return false;
    const bool timed = c->timing_phase == 0;
    c->timing_phase = (c->timing_phase + 1) % c->opt_timing;
    return timed;
}

static int harvest_events(SphContext* c) {
    if (c->ev_used == 0) return 0;
    SPH_HIP(c, hipEventSynchronize(c->ev[c->ev_used - 1][4]));
    for (int s = 0; s < c->ev_used; ++s) {
        float ms[4];
        for (int k = 0; k < 4; ++k) SPH_HIP(c, hipEventElapsedTime(&ms[k], c->ev[s][k], c->ev[s][k + 1]));
        c->tm.sort_ms += ms[0];
        c->tm.neighbour_ms += ms[1];
        c->tm.force_ms += ms[2];
        c->tm.integrate_ms += ms[3];
        c->tm.total_ms += ms[0] + ms[1] + ms[2] + ms[3];
        c->tm.steps += 1;
    }
    c->ev_used = 0;
    return 0;
}

// uniform-fluid force path: decide (once per change of the particle set) whether its precondition holds
static int uniform_fluid(SphContext* c) {
    // (a slab rank with dynamic solids computes forces on its first ghost layer too, whose neighbours in the outer
    // ghost layer get no stg / gat records from the density sweep: general path there)
    if (c->opt_uniform == 0 || c->opt_gather_impl != 1 || (c->opt_drop_outside && !c->opt_no_dynamic)) {
        c->uniform_state = 0;
        return 0;
    }
    if (c->uniform_state >= 0) return 0;
    return sphk_check_uniform_fluid(c);
}

// sweeps of one step after the sort; ev (nullable) = the step's 5 events, ev[1] already recorded
static int step_sweeps(SphContext* c, hipEvent_t* ev, const int32_t* dynamic_ids, int32_t n_dynamic) {
    int rc = 0;
    bool fused_advect = false;
    // compute_moving_boundary_volume()                     sph_base.py:265
    if (c->n_dyn_host > 0) { rc = sphk_gather(c, GM_BVOL_DYNAMIC); if (rc) return rc; }
    if (c->opt_fused) {
        rc = uniform_fluid(c);
        rc = rc ? rc : sphk_gather(c, GM_DENSITY_EOS);      // WCSPH.py:153 (+ EOS of :74-76)
        if (rc) return rc;
        if (ev) SPH_HIP(c, hipEventRecord(ev[2], c->stream));
        // With the one-gather force sweep no workgroup reads another particle's xm / vf (neighbours come from the
        // stg / gat copies), so each fluid target is integrated right in the sweep's finish and the streaming advect
        // kernel disappears.  Dynamic rigid particles collect coupling reactions from many workgroups during the sweep:
        // they are integrated afterwards, by a kernel over their (short) list.
        const bool fuse = c->uniform_state == 1 && c->stg_kind == 1 && c->lists_valid && !c->opt_drop_outside &&
                          c->opt_gather_impl == 1 && c->N > 0;
        c->fuse_advect = fuse ? 1 : 0;
        rc = sphk_gather(c, GM_FORCE_FUSED);                // WCSPH.py:154-155 (+ :156 and sph_base.py:270-271 when fused)
        c->fuse_advect = 0;
        if (rc) return rc;
        fused_advect = fuse;
    } else {
        rc = sphk_gather(c, GM_DENSITY);
        if (rc) return rc;
        if (ev) SPH_HIP(c, hipEventRecord(ev[2], c->stream));
        rc = sphk_gather(c, GM_NONPRESSURE);
        rc = rc ? rc : sphk_eos(c);
        rc = rc ? rc : sphk_gather(c, GM_PRESSURE);
        if (rc) return rc;
    }
    if (ev) SPH_HIP(c, hipEventRecord(ev[3], c->stream));
    // advect (WCSPH.py:156) + enforce_boundary_3D(fluid) (sph_base.py:270-271) in one pass
    // solve_rigid_body()                                   sph_base.py:247-260
    if (fused_advect) return sphk_rigid_solve_all(c, dynamic_ids, n_dynamic, true);   // (the dynamic solids' advect rides in its first kernel)
    rc = sphk_advect(c, true);
    if (rc) return rc;
    if (c->n_dyn_host > 0) { rc = sphk_rigid_solve_all(c, dynamic_ids, n_dynamic, false); if (rc) return rc; }
    return 0;
}

int32_t sph_set_target_layers(SphContext* c, int32_t dlo, int32_t dhi, int32_t flo, int32_t fhi) {
    if (!c) return SPH_E_INVALID;
    const int nx = c->p.grid_num[0];
    if (dlo < 0 || dhi > nx || dlo > dhi || flo < 0 || fhi > nx || flo > fhi) return sph_fail(c, SPH_E_INVALID, "sph_set_target_layers: bad range");
    c->tgt_layers[0] = dlo; c->tgt_layers[1] = dhi; c->tgt_layers[2] = flo; c->tgt_layers[3] = fhi;
    return 0;
}

int32_t sph_slab_set_window(SphContext* c, int32_t origin_x, int32_t nx) {
    ENTER(c);
    if (nx <= 0 || nx > c->nx_alloc) return sph_fail(c, SPH_E_INVALID, "sph_slab_set_window: nx exceeds the allocation of sph_create");
    c->p.cell_origin[0] = origin_x;
    c->p.grid_num[0] = nx;
    c->G = nx * c->p.grid_num[1] * c->p.grid_num[2];
    c->scan_blocks = (c->G + 1 + SCAN_TILE - 1) / SCAN_TILE;  // <= the allocation's
    c->tgt_layers[0] = c->tgt_layers[2] = 0;
    c->tgt_layers[1] = c->tgt_layers[3] = nx;
    // every cell id changes meaning: nothing derived from the old window survives, and the spare cell array was
    // zeroed for the old padded size only
    c->next_cells_zero = false;
    sph_invalidate_lists(c);
    c->have_keys = c->have_prefix = c->sorted = false;
    c->n_dyn_host = -1;
    return 0;
}

// hash + scan + scatter without touching aux: for callers that run a density sweep right behind it (nothing can read the
// stale density / pressure copies the scatter moves in between)
static int sort_no_fold(SphContext* c) {
    int rc = sph_update_grid_id(c);
    rc = rc ? rc : sph_prefix_sum(c);
    return rc ? rc : counting_sort(c, false);
}

int32_t sph_sort(SphContext* c) {
    ENTER(c);
    // (after the lean density finish, density / pressure live in eos2 in the OLD order; a stand-alone sort must
    // fold them into aux first, like the reference's counting_sort carries them along, or a later download reads values
    // from the last fold)
    int rc = sph_ensure_aux(c);
    return rc ? rc : sort_no_fold(c);
}

int32_t sph_sweeps(SphContext* c) {
    ENTER(c);
    int rc = need_sorted(c, "sph_sweeps");
    rc = rc ? rc : refresh_dyn(c);
    if (rc) return rc;
    return step_sweeps(c, nullptr, nullptr, 0);
}

int32_t sph_step(SphContext* c, int32_t n_steps, const int32_t* dynamic_ids, int32_t n_dynamic) {
    ENTER(c);
    if (n_steps < 0 || n_dynamic < 0 || (n_dynamic > 0 && !dynamic_ids)) return sph_fail(c, SPH_E_INVALID, "sph_step: bad arguments");
    int rc = refresh_dyn(c);
    if (rc) return rc;
    if (n_steps > 0) c->acc_partial = false;   // the call's last step writes every acceleration out
    for (int it = 0; it < n_steps; ++it) {
        hipEvent_t* ev = nullptr;
        const bool timing = sph_timed_step(c);  // SPH_OPT_TIMING k: every k-th step carries the five events
        if (timing) {
            if (c->ev_used == SPH_MAX_TIMED_STEPS) { rc = harvest_events(c); if (rc) return rc; }
            ev = c->ev[c->ev_used];
            SPH_HIP(c, hipEventRecord(ev[0], c->stream));
        }
        // ps.initialize_particle_system()                      sph_base.py:264
        rc = sph_update_grid_id(c);
        rc = rc ? rc : sph_prefix_sum(c);
        rc = rc ? rc : counting_sort(c, false);  // acceleration is dead here: every particle's a is rewritten below
        if (rc) return rc;
        if (timing) SPH_HIP(c, hipEventRecord(ev[1], c->stream));
        c->skip_acc = it + 1 < n_steps;  // only the acceleration of the call's last step can ever be read
        rc = step_sweeps(c, ev, dynamic_ids, n_dynamic);
        c->skip_acc = 0;
        if (rc) return rc;
        if (timing) { SPH_HIP(c, hipEventRecord(ev[4], c->stream)); c->ev_used++; }
    }
    return 0;
}

// ---- multi-GPU slab support ---------------------------------------------------
int32_t sph_get_particle_count(SphContext* c, int32_t* n) {
    if (!c || !n) return SPH_E_INVALID;
    *n = c->N;
    return 0;
}

int32_t sph_layer_offsets(SphContext* c, const int32_t* layers, int32_t n, int32_t* out) {
    ENTER(c);
    if (!layers || !out || n < 0) return SPH_E_INVALID;
    if (!c->have_prefix) return sph_fail(c, SPH_E_STATE, "sph_layer_offsets needs the prefix sum");
    const int per_layer = c->p.grid_num[1] * c->p.grid_num[2];
    for (int k = 0; k < n; ++k)
        if (layers[k] < 0 || layers[k] > c->p.grid_num[0]) return sph_fail(c, SPH_E_INVALID, "layer out of range");
    if (n > 8 && (size_t)n * 8 <= c->stage_bytes) {
        // many layers (the per-layer histogram of a re-cut): list up, one kernel, offsets back -- not n copies
        int* dl = (int*)c->stage;
        int* dout = dl + n;
        SPH_HIP(c, hipMemcpyAsync(dl, layers, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_read_layer_offsets, dim3((n + 255) / 256), dim3(256), 0, c->stream, c->cell_end, dl, n, per_layer, dout);
        SPH_LAUNCH_CHECK(c);
        SPH_HIP(c, hipMemcpyAsync(out, dout, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
        SPH_HIP(c, hipStreamSynchronize(c->stream));
        return 0;
    }
    for (int k = 0; k < n; ++k) {
        const int L = layers[k];
        if (L == 0) out[k] = 0;
        else SPH_HIP(c, hipMemcpyAsync(&out[k], c->cell_end + (size_t)L * per_layer - 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    }
    SPH_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
}

// which cells hold the record counts below the given x layers (sets off_zero_mask)
static int layer_offset_cells(SphContext* c, const int32_t* layers, int32_t n, CellIdx16* ix) {
    const int per_layer = c->p.grid_num[1] * c->p.grid_num[2];
    c->off_zero_mask = 0;
    for (int k = 0; k < 16; ++k) ix->v[k] = -1;
    for (int k = 0; k < n; ++k) {
        const int L = layers[k];
        if (L < 0 || L > c->p.grid_num[0]) return sph_fail(c, SPH_E_INVALID, "layer out of range");
        if (L == 0) { c->off_zero_mask |= 1 << k; continue; }
        ix->v[k] = L * per_layer - 1;
    }
    return 0;
}

int32_t sph_layer_offsets_begin(SphContext* c, const int32_t* layers, int32_t n) {
    ENTER(c);
    c->off_stamp_pending = false;   // (this path delivers through k_read_cells + ev_off)
    if (!layers || n < 0 || n > 16) return SPH_E_INVALID;
    if (!c->have_prefix) return sph_fail(c, SPH_E_STATE, "sph_layer_offsets_begin needs the prefix sum");
    CellIdx16 ix;
    int rc0 = layer_offset_cells(c, layers, n, &ix);
    if (rc0) return rc0;
    // one tiny kernel writes all offsets straight into mapped pinned memory (n separate device-to-host copies
    // used to sit on the stream between the sort and the density sweep)
    int* dev_out = nullptr;
    SPH_HIP(c, hipHostGetDevicePointer((void**)&dev_out, c->h_pinned, 0));
    hipLaunchKernelGGL(k_read_cells, dim3(1), dim3(64), 0, c->stream, c->cell_end, ix, dev_out);
    SPH_LAUNCH_CHECK(c);
    SPH_HIP(c, hipEventRecord(c->ev_off, c->stream));
    return 0;
}

int32_t sph_layer_offsets_end(SphContext* c, int32_t* out, int32_t n) {
    ENTER(c);
    if (!out || n < 0 || n > 16) return SPH_E_INVALID;
    if (c->off_stamp_pending) {
        // the offsets came with the sort (sph_slab_advance): spin on the stamp its place kernel writes behind them.  Bounded:
        // if the stream has run dry without the stamp (a failed launch), say so instead of spinning for ever.
        c->off_stamp_pending = false;
        volatile int* hp = c->h_pinned;
        long spins = 0;
        while (__atomic_load_n(&hp[17], __ATOMIC_ACQUIRE) != c->off_stamp) {
            ++spins;
            // (several ranks may share the host's cores -- the one-GPU tests run 2-3 of them: past the first few microseconds a
            // waiting rank gives its core away instead of burning it)
            if (spins > 4096) sched_yield();
            if ((spins & 0xffff) == 0) {
                const hipError_t q = hipStreamQuery(c->stream);
                if (q == hipSuccess && __atomic_load_n(&hp[17], __ATOMIC_ACQUIRE) != c->off_stamp)
                    return sph_fail(c, SPH_E_STATE, "sph_layer_offsets_end: the sort finished without delivering the layer offsets");
                if (q != hipSuccess && q != hipErrorNotReady) SPH_HIP(c, q);   // a failed launch / device fault: the stamp will never come
            }
        }
    } else {
        SPH_HIP(c, hipEventSynchronize(c->ev_off));
    }
    for (int k = 0; k < n; ++k) out[k] = (c->off_zero_mask >> k) & 1 ? 0 : c->h_pinned[k];
    return 0;
}

int32_t sph_select_range(SphContext* c, int32_t first, int32_t count) {
    ENTER(c);
    if (first < 0 || count < 0 || first + count > c->in_off + c->N) return sph_fail(c, SPH_E_INVALID, "sph_select_range: out of range");
    c->in_off += first;
    c->N = count;
    sph_forget_pure_fluid(c);
    sph_invalidate_lists(c);
    c->aux_stale = false;  // (eos2 is indexed from the old first record)
    c->have_keys = c->have_prefix = c->sorted = false;
    c->n_dyn_host = -1;
    return 0;
}

int32_t sph_truncate(SphContext* c, int32_t n) {
    if (!c || n < 0 || n > c->N) return sph_fail(c, SPH_E_INVALID, "sph_truncate: out of range");
    if (n != c->N && !c->opt_no_dynamic) c->n_dyn_host = -1;  // the sort's list skipped the dropped strays: recount
    if (n != c->N) sph_forget_pure_fluid(c);   // (a truncate + append that restores N may bring solids in)
    c->N = n;
    return 0;
}

int32_t sph_pack_range(SphContext* c, int32_t first, int32_t count, void* dst) {
    ENTER(c);
    if (first < 0 || count < 0 || first + count > c->N || (count > 0 && !dst)) return sph_fail(c, SPH_E_INVALID, "sph_pack_range: out of range");
    if (count == 0) return 0;
    const size_t b = (size_t)count * 16, o = (size_t)c->in_off + first;
    char* d = (char*)dst;
    SPH_HIP(c, hipMemcpyAsync(d, c->xm[c->cur] + o, b, hipMemcpyDeviceToDevice, c->stream));
    SPH_HIP(c, hipMemcpyAsync(d + b, c->vf[c->cur] + o, b, hipMemcpyDeviceToDevice, c->stream));
    SPH_HIP(c, hipMemcpyAsync(d + 2 * b, c->aux[c->cur] + o, b, hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

int32_t sph_append_records(SphContext* c, const void* src, int32_t count) {
    ENTER(c);
    if (count < 0 || (count > 0 && !src)) return SPH_E_INVALID;
    if (c->in_off + c->N + count > c->cap) return sph_fail(c, SPH_E_NOMEM, "sph_append_records: capacity exceeded");
    if (count == 0) return 0;
    const size_t b = (size_t)count * 16, o = (size_t)c->in_off + c->N;
    const char* s = (const char*)src;
    SPH_HIP(c, hipMemcpyAsync(c->xm[c->cur] + o, s, b, hipMemcpyDeviceToDevice, c->stream));
    SPH_HIP(c, hipMemcpyAsync(c->vf[c->cur] + o, s + b, b, hipMemcpyDeviceToDevice, c->stream));
    SPH_HIP(c, hipMemcpyAsync(c->aux[c->cur] + o, s + 2 * b, b, hipMemcpyDeviceToDevice, c->stream));
    c->N += count;
    if (c->opt_uniform != 1) c->uniform_state = -1;  // arrivals are unchecked unless the caller vouches for them
    sph_forget_pure_fluid(c);   // ... and vouching is for a uniform fluid MASS only: arrivals may be solids
    sph_invalidate_lists(c);
    c->have_keys = c->have_prefix = c->sorted = false;
    c->n_dyn_host = -1;
    return 0;
}

int32_t sph_sync(SphContext* c) {
    ENTER(c);
    SPH_HIP(c, hipStreamSynchronize(c->stream));
    return sph_check_device_flags(c);
}

int32_t sph_get_timings(SphContext* c, SphTimings* out) {
    ENTER(c);
    if (!out) return SPH_E_INVALID;
    int rc = harvest_events(c);
    rc = rc ? rc : sph_check_device_flags(c);   // (harvest_events waited for the last timed step)
    if (rc) return rc;
    *out = c->tm;
    return 0;
}

int32_t sph_get_stats(SphContext* c, SphStats* out) {
    ENTER(c);
    if (!out) return SPH_E_INVALID;
    if (!c->have_prefix) return sph_fail(c, SPH_E_STATE, "sph_get_stats needs a sorted state (run a step first)");
    int rc = sphk_stats(c, out);   // synchronises
    return rc ? rc : sph_check_device_flags(c);
}

int32_t sph_reset_timings(SphContext* c) {
    ENTER(c);
    int rc = harvest_events(c);
    if (rc) return rc;
    memset(&c->tm, 0, sizeof(c->tm));
    return 0;
}

int32_t sph_slab_pack(SphContext* c, int32_t firstL, int32_t nL, void* dstL, int32_t firstR, int32_t nR, void* dstR) {
    ENTER(c);
    int rc = sph_pack_range(c, firstL, nL, dstL);
    rc = rc ? rc : sph_pack_range(c, firstR, nR, dstR);
    if (rc) return rc;
    SPH_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
}

int32_t sph_slab_advance(SphContext* c, int32_t keep_first, int32_t keep_count, const void* srcL, int32_t nL,
                         const void* srcR, int32_t nR, const int32_t* layers, int32_t n_layers, int32_t do_sweeps) {
    ENTER(c);
    // SPH_OPT_TIMING in slab mode: the five events of one step are split over this call (sort, density) and the
    // following sph_slab_forces (force, advect); the halo exchange itself is host-side and not on this stream
    hipEvent_t* ev = nullptr;
    c->slab_ev_open = false;
    if (do_sweeps == 2 && sph_timed_step(c)) {
        if (c->ev_used == SPH_MAX_TIMED_STEPS) { int rh = harvest_events(c); if (rh) return rh; }
        ev = c->ev[c->ev_used];
        SPH_HIP(c, hipEventRecord(ev[0], c->stream));
    }
    // do_sweeps != 0: a density sweep follows at once and rewrites every fluid particle's density / pressure -- the stale
    // copies the scatter moves are never observable (the received records' neither: their owner's sweep is as old as
    // ours).  do_sweeps == 0: the caller may read the fields next, so they are folded into aux first -- before
    // the selection, which re-bases the records the lean (p, rho) array is indexed by.
    int rc = do_sweeps == 0 ? sph_ensure_aux(c) : 0;
    rc = rc ? rc : sph_select_range(c, keep_first, keep_count);
    rc = rc ? rc : sph_append_records(c, srcL, nL);
    rc = rc ? rc : sph_append_records(c, srcR, nR);
    // the layer offsets ride in the sort: its place kernel -- the first kernel behind the scan -- copies the scanned cells into
    // the mapped buffer and writes the sort's stamp behind them, which sph_layer_offsets_end spins on (r06: k_read_cells was a
    // launch of its own between the scatter and the density sweep and an event record behind it, ~8 us on the stream of every
    // slab step; the host now also gets the offsets a scatter earlier)
    if (!rc && (!layers || n_layers < 0 || n_layers > 16)) rc = SPH_E_INVALID;
    if (!rc) rc = layer_offset_cells(c, layers, n_layers, &c->off_ix);
    if (!rc) { c->off_in_sort = true; c->off_stamp_pending = false; }
    rc = rc ? rc : sort_no_fold(c);
    c->off_in_sort = false;
    // an EMPTY record set launches no place kernel, so nobody writes the offsets or the stamp: every count below every layer is 0
    // (sph_layer_offsets_end then neither spins nor reads the mapped words of an earlier step)
    if (!rc && !c->off_stamp_pending) c->off_zero_mask = 0xffff;
    if (!rc && ev) SPH_HIP(c, hipEventRecord(ev[1], c->stream));
    if (!rc && do_sweeps == 1) rc = sph_sweeps(c);
    if (!rc && do_sweeps == 2) {  // boundary volume + density only; sph_slab_forces does the rest
        rc = refresh_dyn(c);
        rc = rc ? rc : uniform_fluid(c);
        if (!rc && c->n_dyn_host > 0) rc = sphk_gather(c, GM_BVOL_DYNAMIC);
        rc = rc ? rc : sphk_gather(c, GM_DENSITY_EOS);
        if (!rc && ev) { SPH_HIP(c, hipEventRecord(ev[2], c->stream)); c->slab_ev_open = true; }
    }
    return rc;
}

int32_t sph_slab_forces(SphContext* c, int32_t bl_lo, int32_t bl_hi, int32_t br_lo, int32_t br_hi,
                        int32_t firstL, int32_t nL, void* dstL, int32_t firstR, int32_t nR, void* dstR) {
    ENTER(c);
    int rc = need_sorted(c, "sph_slab_forces");
    if (rc) return rc;
    if (!c->opt_fused || c->opt_gather_impl != 1) return sph_fail(c, SPH_E_STATE, "sph_slab_forces needs the fused brick sweeps");
    if (firstL < 0 || nL < 0 || firstL + nL > c->N || firstR < 0 || nR < 0 || firstR + nR > c->N ||
        (nL > 0 && !dstL) || (nR > 0 && !dstR))
        return sph_fail(c, SPH_E_INVALID, "sph_slab_forces: bad pack range");
    const int f_lo = c->tgt_layers[2], f_hi = c->tgt_layers[3];
    // clip the boundary sets to the force-target layers and keep them disjoint
    bl_lo = bl_lo < f_lo ? f_lo : bl_lo; bl_hi = bl_hi > f_hi ? f_hi : bl_hi;
    br_hi = br_hi > f_hi ? f_hi : br_hi; br_lo = br_lo < bl_hi ? bl_hi : br_lo;
    if (bl_hi < bl_lo) bl_hi = bl_lo;
    if (br_lo > br_hi) br_lo = br_hi;
    // Boundary sets (incl. the ghost-side strips that exist when dynamic solids are force targets) in ONE launch, then
    // the two packers -- on the side stream, so the big interior sweep does not queue behind that small launch (a few
    // hundred workgroups cannot fill 256 CUs) but runs beside it.  Both sweeps write disjoint targets' accelerations.
    static const bool no_side_env = getenv("SPH_NO_SIDE_STREAM") != nullptr;  // debugging aid: everything on one stream
    // a rank without neighbours (world = 1, or nothing to pack and no boundary strip): no boundary launch, no packers,
    // so no fork to the side stream and no join either
    const bool no_boundary = bl_hi <= f_lo && br_lo >= f_hi && nL == 0 && nR == 0;
    const bool no_side = no_side_env || no_boundary;
    if (!no_side) {
        SPH_HIP(c, hipEventRecord(c->ev_fork, c->stream));
        SPH_HIP(c, hipStreamWaitEvent(c->side, c->ev_fork, 0));
        c->use_side = true;
    }
    rc = sphk_gather_layers(c, GM_FORCE_FUSED, f_lo, bl_hi, br_lo, f_hi);
    // the packed layers' dynamic solids: their coupling reactions all come from the boundary sets (which reach one
    // layer further in for that), so they are complete now and the packers can advance them
    rc = rc ? rc : sphk_fold_coupling_range(c, firstL, nL);
    rc = rc ? rc : sphk_fold_coupling_range(c, firstR, nR);
    rc = rc ? rc : sphk_pack_advected(c, firstL, nL, dstL);
    rc = rc ? rc : sphk_pack_advected(c, firstR, nR, dstR);
    c->use_side = false;
    if (rc) return rc;
    if (!no_boundary) SPH_HIP(c, hipEventRecord(c->ev_pack, no_side ? c->stream : c->side));
    // interior: overlaps with the exchange; with the one-gather sweep its finish integrates its own targets (see
    // step_sweeps), so that afterwards only the two boundary ranges -- exactly the packed ranges -- are left to advect
    // (ghost records are not advected at all: the exchange replaces them)
    const bool fuse = c->uniform_state == 1 && c->stg_kind == 1 && c->lists_valid && c->opt_no_dynamic;
    c->fuse_advect = fuse ? 1 : 0;
    // (an interior target's acceleration is consumed by the advect in the same finish and by nobody else -- the packers
    // read the BOUNDARY sets' -- so, as inside sph_step, it is not written out: 16 B per particle and step less)
    c->skip_acc = fuse ? 1 : 0;
    rc = sphk_gather_layers(c, GM_FORCE_FUSED, bl_hi, br_lo, 0, 0);
    c->fuse_advect = 0;
    c->skip_acc = 0;
    c->acc_partial = fuse;
    if (rc) return rc;
    if (!no_boundary) SPH_HIP(c, hipStreamWaitEvent(c->stream, c->ev_pack, 0));  // the packers read what the advect overwrites
    hipEvent_t* ev = c->slab_ev_open ? c->ev[c->ev_used] : nullptr;
    if (ev) SPH_HIP(c, hipEventRecord(ev[3], c->stream));  // force = interior sweep (+ the wait for the side stream)
    if (c->n_dyn_host != 0) { rc = sphk_fold_coupling(c); if (rc) return rc; }  // both force launches have joined: reactions -> accelerations
    if (fuse) {
        // (in a slab only HALO+1 layers wide the two packed ranges overlap: advect their union once)
        const int a0 = firstL, a1 = firstL + nL, b0 = firstR, b1 = firstR + nR;
        if (nL > 0 && nR > 0 && b0 < a1) {
            const int lo = a0 < b0 ? a0 : b0, hi = a1 > b1 ? a1 : b1;
            rc = sphk_advect_range(c, lo, hi - lo);
        } else {
            rc = sphk_advect_range(c, firstL, nL);
            rc = rc ? rc : sphk_advect_range(c, firstR, nR);
        }
    } else {
        rc = sphk_advect(c, true);
    }
    if (!rc && ev) { SPH_HIP(c, hipEventRecord(ev[4], c->stream)); c->ev_used++; c->slab_ev_open = false; }
    return rc;
}

int32_t sph_slab_density(SphContext* c) {
    ENTER(c);
    int rc = need_sorted(c, "sph_slab_density");
    rc = rc ? rc : refresh_dyn(c);
    rc = rc ? rc : uniform_fluid(c);
    if (!rc && c->n_dyn_host > 0) rc = sphk_gather(c, GM_BVOL_DYNAMIC);
    return rc ? rc : sphk_gather(c, GM_DENSITY_EOS);
}

int32_t sph_rigid_partial_sums(SphContext* c, int32_t object_id, int32_t first, int32_t count, double* dev_sums16) {
    ENTER(c);
    if (object_id < 0 || object_id >= c->p.n_objects || !dev_sums16 || first < 0 || count < 0 || first + count > c->N)
        return sph_fail(c, SPH_E_INVALID, "sph_rigid_partial_sums: bad arguments");
    int rc = refresh_dyn(c);
    return rc ? rc : sphk_rigid_partial16(c, object_id, first, count, dev_sums16);
}

int32_t sph_rigid_apply_sums(SphContext* c, int32_t object_id, const double* dev_sums16, int32_t mode) {
    ENTER(c);
    if (object_id < 0 || object_id >= c->p.n_objects || !dev_sums16 || (mode != 0 && mode != 1))
        return sph_fail(c, SPH_E_INVALID, "sph_rigid_apply_sums: bad arguments");
    int rc = refresh_dyn(c);
    return rc ? rc : sphk_rigid_apply16(c, object_id, dev_sums16, mode);
}

int32_t sph_upload_rest_positions(SphContext* c, const int32_t* pid, const float* x0, int32_t n) {
    ENTER(c);
    if (n < 0 || (n > 0 && (!pid || !x0))) return sph_fail(c, SPH_E_INVALID, "sph_upload_rest_positions: bad arguments");
    // staged in chunks through the upload buffer: [pid i32 | x0 3 f32] = 16 bytes per entry
    const int chunk = (int)(c->stage_bytes / 16);
    for (int done = 0; done < n; done += chunk) {
        const int m = n - done < chunk ? n - done : chunk;
        int* dpid = (int*)c->stage;
        float* dx0 = (float*)((char*)c->stage + (size_t)m * 4);
        SPH_HIP(c, hipMemcpyAsync(dpid, pid + done, (size_t)m * 4, hipMemcpyHostToDevice, c->stream));
        SPH_HIP(c, hipMemcpyAsync(dx0, x0 + 3 * (size_t)done, (size_t)m * 12, hipMemcpyHostToDevice, c->stream));
        int rc = sphk_scatter_rest(c, dpid, dx0, m);
        if (rc) return rc;
        SPH_HIP(c, hipStreamSynchronize(c->stream));
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// DFSPH (DFSPH.py)
// ---------------------------------------------------------------------------------------------
int32_t sph_dfsph_set_params(SphContext* c, const SphDfsphParams* p) {
    if (!c || !p) return SPH_E_INVALID;
    if (p->m_max_iterations_v < 0 || p->m_max_iterations < 0 || p->fluid_particle_num < 0)
        return sph_fail(c, SPH_E_INVALID, "sph_dfsph_set_params: negative count");
    c->df = *p;
    return 0;
}

int32_t sph_dfsph_get_stats(SphContext* c, SphDfsphStats* out) {
    if (!c || !out) return SPH_E_INVALID;
    *out = c->df_stats;
    return 0;
}

#define DF_SWEEP(NAME, MODE)                         \
    int32_t NAME(SphContext* c) {                    \
        ENTER(c);                                    \
        int rc = need_sorted(c, #NAME);              \
        rc = rc ? rc : refresh_dyn(c);               \
        return rc ? rc : sphk_gather(c, MODE);       \
    }
DF_SWEEP(sph_dfsph_compute_densities, GM_DF_DENSITY)
DF_SWEEP(sph_dfsph_compute_DFSPH_factor, GM_DF_FACTOR)
DF_SWEEP(sph_dfsph_compute_density_change, GM_DF_DENSITY_CHANGE)
DF_SWEEP(sph_dfsph_compute_density_adv, GM_DF_DENSITY_ADV)
DF_SWEEP(sph_dfsph_divergence_solver_iteration_kernel, GM_DF_DIV_ITER)
DF_SWEEP(sph_dfsph_pressure_solve_iteration_kernel, GM_DF_PRESSURE_ITER)
DF_SWEEP(sph_dfsph_compute_non_pressure_forces, GM_DF_NONPRESSURE)
#undef DF_SWEEP

int32_t sph_dfsph_compute_density_error(SphContext* c, float offset, float* out) {
    ENTER(c);
    if (!out) return SPH_E_INVALID;
    return sphk_df_density_error(c, offset, out);
}

int32_t sph_dfsph_compute_density_error_range(SphContext* c, float offset, int32_t first, int32_t count, double* out) {
    ENTER(c);
    if (!out || first < 0 || count < 0 || first + count > c->N) return sph_fail(c, SPH_E_INVALID, "sph_dfsph_compute_density_error_range: bad range");
    return sphk_df_density_error_range(c, offset, first, count, out);
}

int32_t sph_copy_velocity_records(SphContext* c, int32_t first, int32_t count, void* device_buf, int32_t to_context) {
    ENTER(c);
    if (first < 0 || count < 0 || first + count > c->N || (count > 0 && !device_buf))
        return sph_fail(c, SPH_E_INVALID, "sph_copy_velocity_records: bad range");
    if (count == 0) return 0;
    float4* vf = c->vf[c->cur] + (size_t)c->in_off + first;
    if (to_context) SPH_HIP(c, hipMemcpyAsync(vf, device_buf, (size_t)count * 16, hipMemcpyDeviceToDevice, c->stream));
    else SPH_HIP(c, hipMemcpyAsync(device_buf, vf, (size_t)count * 16, hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

int32_t sph_dfsph_multiply_time_step(SphContext* c, float time_step) {
    ENTER(c);
    return sphk_df_scale_factor(c, time_step);
}

int32_t sph_dfsph_predict_velocity(SphContext* c) {
    ENTER(c);
    return sphk_df_predict_velocity(c);
}

int32_t sph_dfsph_advect(SphContext* c) {
    ENTER(c);
    return sphk_df_advect(c, false);
}

// The Jacobi loops of DFSPH.py:240-283 / 324-354 without a host round trip on the GPU's critical path (round 5; VERDICT r04
// "next" #5).  The reference's loop is  body; err = compute_density_error(); if err / n <= eta: break  with the host between
// two bodies; rounds 2-4 mirrored it: every iteration ended in a stream synchronisation and a read-back, the GPU idle until
// the host had looked at one f32 and enqueued the next body (~15 of them per step).  Now the convergence test is made ON THE
// DEVICE by the reduction's last kernel -- the same arithmetic: the f32 sum, divided by the fluid particle count and compared
// with eta in f64 -- which (a) writes (err, converged) into pinned host memory and (b) on convergence stamps the solve's epoch
// into a gate word.  The host enqueues body k + 1 BEFORE it waits for body k's event: the GPU always has the next body
// queued, and if body k closed the solve, body k + 1's sweeps find the gate stamped and leave at once (a few us of empty
// launches per solve instead of a bubble per iteration).  Iteration counts are the reference's by construction: a body
// changes state iff no earlier body of the solve converged, and the host counts exactly the bodies it would have run.
struct DfSolve { int sweep, refresh; float offset; double eta; int limit; };

static int df_enqueue_body(SphContext* c, const DfSolve& s, int k) {
    int rc = sphk_gather(c, s.sweep);                    // *_solver_iteration(): kernel,
    // (the refresh sweep also reduces the density error over its own targets, brick by brick, when it runs as a brick sweep)
    c->df_collect = c->opt_df_fuse_err;
    c->df_bpart_valid = false;
    rc = rc ? rc : sphk_gather(c, s.refresh);            //   compute_density_change() / compute_density_adv(),
    c->df_collect = 0;
    rc = rc ? rc : sphk_df_convergence_test(c, s.offset, s.eta, k & 3);  // compute_density_error() + the test
    if (rc) return rc;
    SPH_HIP(c, hipEventRecord(c->ev_df[k & 3], c->stream));
    return 0;
}

static int df_solve_loop(SphContext* c, const DfSolve& s, int* iterations, double* avg_err, int64_t* total) {
    c->df_epoch = c->df_epoch_done + 1u;
    if (c->df_epoch == 0u) {  // (wrapped: an old stamp could be mistaken for this solve's)
        SPH_HIP(c, hipMemsetAsync(c->df_gate, 0, sizeof(unsigned), c->stream));
        c->df_epoch = 1u;
    }
    int rc = df_enqueue_body(c, s, 0);
    int k = 0;
    double avg = 0.0;
    while (!rc) {
        const bool more = k + 1 < s.limit;
        const bool ahead = more && c->opt_df_runahead;
        if (ahead) rc = df_enqueue_body(c, s, k + 1);  // ahead of body k's test; a no-op on the device if that test closes the solve
        if (rc) break;
        if (hipEventSynchronize(c->ev_df[k & 3]) != hipSuccess) { rc = sph_fail(c, SPH_E_STATE, "DFSPH solver: event wait failed"); break; }
        if ((rc = sph_check_device_flags(c)) != 0) break;
        const volatile SphContext::DfSlot* slot = c->h_df_slot + (k & 3);
        *total += 1;
        avg = slot->avg;
        if (slot->converged) break;   // DFSPH.py:262-263 / 348-349
        k += 1;                       // m_iterations += 1
        if (!more) break;             // the while condition: m_iterations reached max(1, m_max_iterations)
        if (!ahead) rc = df_enqueue_body(c, s, k);
    }
    c->df_epoch_done = c->df_epoch;
    c->df_epoch = 0u;                 // the gate is closed to every kernel outside this loop
    *iterations = k;
    *avg_err = avg;
    return rc;
}

// DFSPH.py:240-283.  Host arithmetic in double, like the reference's Python floats.
int32_t sph_dfsph_divergence_solve(SphContext* c) {
    ENTER(c);
    int rc = need_sorted(c, "sph_dfsph_divergence_solve");
    rc = rc ? rc : refresh_dyn(c);
    if (rc) return rc;
    const double dt = (double)c->p.dt;
    const double inv_dt = 1 / dt;
    // (compute_density_change does not read the factor, so scaling it first changes nothing -- and lets that sweep
    // leave k_j = density_adv_j * factor_j behind for the Jacobi sweep)
    rc = sphk_df_scale_factor(c, (float)inv_dt);
    rc = rc ? rc : sphk_gather(c, GM_DF_DENSITY_CHANGE);
    if (rc) return rc;
    DfSolve s;
    s.sweep = GM_DF_DIV_ITER; s.refresh = GM_DF_DENSITY_CHANGE; s.offset = 0.0f;
    s.eta = 1.0 / dt * c->df.max_error_V * 0.01 * (double)c->p.density_0;
    s.limit = c->df.m_max_iterations_v > 1 ? c->df.m_max_iterations_v : 1;
    int it = 0;
    double avg = 0.0;
    rc = df_solve_loop(c, s, &it, &avg, &c->df_stats.total_iterations_v);
    if (rc) return rc;
    c->df_stats.iterations_v = it;
    c->df_stats.avg_density_err_v = avg;
    return sphk_df_scale_factor(c, (float)dt);
}

// DFSPH.py:324-354
int32_t sph_dfsph_pressure_solve(SphContext* c) {
    ENTER(c);
    int rc = need_sorted(c, "sph_dfsph_pressure_solve");
    rc = rc ? rc : refresh_dyn(c);
    if (rc) return rc;
    const double dt = (double)c->p.dt;
    const double inv_dt2 = 1 / (dt * dt);
    rc = sphk_df_scale_factor(c, (float)inv_dt2);          // (before compute_density_adv, see divergence_solve)
    rc = rc ? rc : sphk_gather(c, GM_DF_DENSITY_ADV);
    if (rc) return rc;
    DfSolve s;
    s.sweep = GM_DF_PRESSURE_ITER; s.refresh = GM_DF_DENSITY_ADV; s.offset = c->p.density_0;
    s.eta = c->df.max_error * 0.01 * (double)c->p.density_0;
    s.limit = c->df.m_max_iterations > 1 ? c->df.m_max_iterations : 1;
    int it = 0;
    double avg = 0.0;
    rc = df_solve_loop(c, s, &it, &avg, &c->df_stats.total_iterations);
    if (rc) return rc;
    c->df_stats.iterations = it;
    c->df_stats.avg_density_err = avg;
    return 0;
}

int32_t sph_dfsph_step(SphContext* c, int32_t n_steps, const int32_t* dynamic_ids, int32_t n_dynamic) {
    ENTER(c);
    if (n_steps < 0 || n_dynamic < 0 || (n_dynamic > 0 && !dynamic_ids)) return sph_fail(c, SPH_E_INVALID, "sph_dfsph_step: bad arguments");
    int rc = refresh_dyn(c);
    if (rc) return rc;
    if (n_steps > 0) c->acc_partial = false;
    for (int it = 0; it < n_steps; ++it) {
        hipEvent_t* ev = nullptr;
        const bool timing = sph_timed_step(c);  // SPH_OPT_TIMING k: every k-th step carries the five events
        if (timing) {
            if (c->ev_used == SPH_MAX_TIMED_STEPS) { rc = harvest_events(c); if (rc) return rc; }
            ev = c->ev[c->ev_used];
            SPH_HIP(c, hipEventRecord(ev[0], c->stream));
        }
        rc = sph_update_grid_id(c);                                  // sph_base.py:264
        rc = rc ? rc : sph_prefix_sum(c);
        rc = rc ? rc : counting_sort(c, false);
        if (rc) return rc;
        if (timing) SPH_HIP(c, hipEventRecord(ev[1], c->stream));
        if (c->n_dyn_host > 0) { rc = sphk_gather(c, GM_BVOL_DYNAMIC); if (rc) return rc; }  // sph_base.py:265
        // DFSPHSolver.substep (DFSPH.py:400-408)
        rc = sphk_gather(c, GM_DF_DENSITY);
        rc = rc ? rc : sphk_gather(c, GM_DF_FACTOR);
        if (rc) return rc;
        if (timing) SPH_HIP(c, hipEventRecord(ev[2], c->stream));
        if (c->df.enable_divergence_solver) { rc = sph_dfsph_divergence_solve(c); if (rc) return rc; }
        rc = sphk_gather(c, GM_DF_NONPRESSURE);
        rc = rc ? rc : sphk_df_predict_velocity(c);
        rc = rc ? rc : sph_dfsph_pressure_solve(c);
        if (rc) return rc;
        if (timing) SPH_HIP(c, hipEventRecord(ev[3], c->stream));
        rc = sphk_df_advect(c, true);                                // advect + enforce_boundary_3D(fluid)
        if (rc) return rc;
        if (c->n_dyn_host > 0) { rc = sphk_rigid_solve_all(c, dynamic_ids, n_dynamic, false); if (rc) return rc; }  // solve_rigid_body()  sph_base.py:247-260
        if (timing) { SPH_HIP(c, hipEventRecord(ev[4], c->stream)); c->ev_used++; }
        c->df_stats.steps++;
    }
    return 0;
}

#ifdef SPH_PROFILE
// profiling build only: the per-workgroup time line the brick sweeps left in the staging buffer (SPH_TS in sph_gather.hip)
int32_t sph_profile_read(SphContext* c, void* host, size_t bytes) {
    ENTER(c);
    if (!host || bytes > c->stage_bytes) return sph_fail(c, SPH_E_INVALID, "sph_profile_read: bad size");
    SPH_HIP(c, hipMemcpyAsync(host, c->stage, bytes, hipMemcpyDeviceToHost, c->stream));
    SPH_HIP(c, hipStreamSynchronize(c->stream));
    return 0;
}
#endif

int32_t sph_slab_wait_pack(SphContext* c) {
    ENTER(c);
    SPH_HIP(c, hipEventSynchronize(c->ev_pack));
    return 0;
}

}  // extern "C"
