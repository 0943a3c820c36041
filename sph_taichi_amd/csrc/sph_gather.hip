// sph_gather.hip -- the 27-cell neighbour sweeps K4..K7 of the hot path.
// Replaces particle_system.py:378-385 (for_all_neighbors) and the tasks it is
// instantiated with: sph_base.py:91-113 (boundary volume), WCSPH.py:19-43
// (density), WCSPH.py:88-140 (surface tension + viscosity), WCSPH.py:46-85
// (Tait EOS + pressure gradient + two-way coupling scatter).
//
// Two implementations of the same sweeps:
//  * k_gather_simple : one lane per target particle walks its 9 (x,y) cell
//    columns; the 3 z-cells of a column are ONE contiguous index range because
//    the flatten order is z-fastest (particle_system.py:294).  Candidates come
//    through L1/L2.  Same traversal order as the reference.  Used for small
//    target sets (rigid particles) and as the overflow path.
//  * k_gather_brick  : a 256-lane workgroup owns a BXxBYxBZ brick of cells,
//    stages the brick + one-cell shell (per column one contiguous segment) in
//    LDS, then every lane (a) filters its candidates into a private LDS index
//    list (cheap loop: ds_read_b128 + 7 VALU), (b) runs the expensive pair
//    physics over the list only, so lanes stay converged in the costly part.
//
// No MFMA: this is a bandwidth/VALU-bound gather, not a contraction.
#include "sph_internal.h"

#define TPB 256

// ---------------------------------------------------------------------------
// per-target state and pair physics, shared by both implementations
// ---------------------------------------------------------------------------
struct Target {
    float x, y, z, mV;
    float vx, vy, vz;
    int flags;
    float m, rho, p, dpi;  // own mass, density (clamped where the mode needs it), pressure, p/rho^2
    float s0;              // scalar accumulator (density / boundary volume)
    float ax, ay, az;      // non-pressure accumulator (starts at g)
    float px, py, pz;      // pressure accumulator
    float st_c;            // surface_tension / m_i
    float dpj_solid;       // p_i / rho0^2 (WCSPH.py:60)
};

template <int MODE>
__device__ __forceinline__ bool mode_needs_B() {
    return MODE != GM_DENSITY && MODE != GM_DENSITY_EOS;
}
template <int MODE>
__device__ __forceinline__ bool mode_needs_C() {
    return MODE == GM_NONPRESSURE || MODE == GM_PRESSURE || MODE == GM_FORCE_FUSED;
}

// is particle (flags) a gather target of this mode?
template <int MODE>
__device__ __forceinline__ bool target_gathers(int fl) {
    if (MODE == GM_BVOL_STATIC) return sph_is_static_rigid(fl);    // sph_base.py:93-94
    if (MODE == GM_BVOL_DYNAMIC) return sph_is_dynamic_rigid(fl);  // sph_base.py:108-109
    return sph_is_fluid(fl);  // WCSPH.py:36-37, 80-83, 138
}

// candidate record C = (p/rho^2, m/rho_raw, m, rho) built from the aux record
// (m, density, pressure, pid) for the stand-alone (API) modes
template <int MODE>
__device__ __forceinline__ float4 make_C_from_aux(const float4 aux) {
    float4 c;
    c.x = (MODE == GM_PRESSURE) ? aux.z / (aux.y * aux.y) : 0.0f;  // WCSPH.py:54
    c.y = (MODE == GM_NONPRESSURE) ? aux.x / aux.y : 0.0f;         // WCSPH.py:112
    c.z = aux.x;
    c.w = aux.y;
    return c;
}

template <int MODE>
__device__ __forceinline__ void target_init(const DevView& d, Target& t, int i, const float4 A, const float4 B) {
    t.x = A.x; t.y = A.y; t.z = A.z; t.mV = A.w;
    t.vx = B.x; t.vy = B.y; t.vz = B.z;
    t.flags = __float_as_int(B.w);
    t.s0 = 0.0f;
    t.ax = d.gx; t.ay = d.gy; t.az = d.gz;  // WCSPH.py:135-136 d_v = g
    t.px = t.py = t.pz = 0.0f;
    t.m = t.rho = t.p = t.dpi = t.st_c = t.dpj_solid = 0.0f;
    if (MODE == GM_NONPRESSURE || MODE == GM_PRESSURE) {
        const float4 aux = d.aux[i];
        t.m = aux.x; t.rho = aux.y; t.p = aux.z;
        t.dpi = aux.z / (aux.y * aux.y);  // WCSPH.py:49
        t.st_c = d.sigma / aux.x;         // WCSPH.py:100
        t.dpj_solid = aux.z / (d.rho0 * d.rho0);
    }
    if (MODE == GM_FORCE_FUSED) {
        const float4 e = d.eos[i];
        t.dpi = e.x; t.m = e.z; t.rho = e.w;
        t.p = e.x * (e.w * e.w);
        t.st_c = d.sigma / e.z;
        t.dpj_solid = t.p / (d.rho0 * d.rho0);
    }
    if (MODE == GM_BVOL_STATIC || MODE == GM_BVOL_DYNAMIC) t.s0 = d.w_zero;  // sph_base.py:95, 110
}

// Fast reciprocal / rsqrt (v_rsq_f32 / v_rcp_f32, ~1 ulp).  The reference's
// r.norm(), r / (r_norm * h), x / y become r2 * rsq(r2), r * (rsq * 1/h), x * rcp(y):
// same formulas, a few ulp apart, far inside the 1e-4 position budget; hipcc's
// correctly-rounded div/sqrt expansions (~10 VALU each) were the dominant cost.
__device__ __forceinline__ float sph_rsq(float x) { return x > 0.0f ? __builtin_amdgcn_rsqf(x) : 0.0f; }
__device__ __forceinline__ float sph_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// sph_base.py:23-44 cubic_kernel as a function of q = r/h (q <= 1 is guaranteed by the caller's r < h)
__device__ __forceinline__ float sph_W_q(const DevView& d, float q) {
    const float t = 1.0f - q;
    const float inner = d.k_w * ((6.0f * q - 6.0f) * q * q + 1.0f);
    const float outer = d.k_w * 2.0f * (t * t * t);
    return q <= 0.5f ? inner : outer;
}
// sph_base.py:46-68 cubic_kernel_derivative = coef * (rx, ry, rz); coef folds grad_q = r / (|r| h)
__device__ __forceinline__ float sph_gradW_coef(const DevView& d, float q, float r_norm, float rinv) {
    const float f = 1.0f - q;
    const float c = q <= 0.5f ? d.k_dw * q * (3.0f * q - 2.0f) : d.k_dw * (-f * f);
    return r_norm > 1e-5f ? c * (rinv * d.inv_h) : 0.0f;
}

// One accepted pair (i != j, r_norm = |x_i - x_j| < h).  gj = global (sorted)
// index of j, needed only for the coupling scatter.
template <int MODE>
__device__ __forceinline__ void pair_physics(const DevView& d, Target& t, float rx, float ry, float rz, float r2,
                                             float r_norm, float rinv, const float4 A, const float4 B,
                                             const float4 Cc, int gj) {
    const float q = r_norm * d.inv_h;
    if (MODE == GM_DENSITY || MODE == GM_DENSITY_EOS) {
        // WCSPH.py:19-30: fluid and solid neighbours add m_V_j * W identically
        t.s0 += A.w * sph_W_q(d, q);
        return;
    }
    const int fj = __float_as_int(B.w);
    if (MODE == GM_BVOL_STATIC || MODE == GM_BVOL_DYNAMIC) {
        if (sph_flags_material(fj) == SPH_MATERIAL_SOLID) t.s0 += sph_W_q(d, q);  // sph_base.py:100-103
        return;
    }
    const bool j_fluid = sph_is_fluid(fj);
    const float gc = sph_gradW_coef(d, q, r_norm, rinv);
    if (MODE == GM_NONPRESSURE || MODE == GM_FORCE_FUSED) {
        if (j_fluid) {
            // surface tension  WCSPH.py:93-102
            const float w = (r2 > d.d2) ? sph_W_q(d, q) : d.w_d;
            const float c = t.st_c * Cc.z * w;
            // viscosity  WCSPH.py:105-116
            const float v_xy = (t.vx - B.x) * rx + (t.vy - B.y) * ry + (t.vz - B.z) * rz;
            const float cv = d.visc_d_nu * Cc.y * v_xy * sph_rcp(r2 + d.visc_eps) * gc;
            const float k = cv - c;
            t.ax += k * rx; t.ay += k * ry; t.az += k * rz;
        }
        // solid neighbour: boundary_viscosity = 0.0 => contributes exactly 0 (WCSPH.py:117-125)
    }
    if (MODE == GM_PRESSURE || MODE == GM_FORCE_FUSED) {
        if (j_fluid) {
            // WCSPH.py:51-57
            const float c = -d.rho0 * A.w * (t.dpi + Cc.x) * gc;
            t.px += c * rx; t.py += c * ry; t.pz += c * rz;
        } else if (sph_flags_material(fj) == SPH_MATERIAL_SOLID) {
            // WCSPH.py:58-68 (Akinci 2012 boundary pressure + two-way coupling)
            const float c = -d.rho0 * A.w * (t.dpi + t.dpj_solid) * gc;
            const float fx = c * rx, fy = c * ry, fz = c * rz;
            t.px += fx; t.py += fy; t.pz += fz;
            if (sph_is_dynamic_rigid(fj)) {
                const float sc = d.rho0 * sph_rcp(Cc.w);
                float* a = reinterpret_cast<float*>(&d.acc[gj]);
                unsafeAtomicAdd(a + 0, -fx * sc);
                unsafeAtomicAdd(a + 1, -fy * sc);
                unsafeAtomicAdd(a + 2, -fz * sc);
            }
        }
    }
}

// write-back of one particle (target or not) for the given mode
template <int MODE>
__device__ __forceinline__ void target_finish(const DevView& d, Target& t, int i, bool gathered) {
    if (MODE == GM_BVOL_STATIC || MODE == GM_BVOL_DYNAMIC) {
        // sph_base.py:98, 113: m_V = 1/delta * 3.0   (only .w is written; .xyz are read concurrently)
        if (gathered) reinterpret_cast<float*>(&d.xm[i])[3] = 1.0f / t.s0 * 3.0f;
        return;
    }
    if (MODE == GM_DENSITY) {
        // WCSPH.py:39-43
        if (gathered) reinterpret_cast<float*>(&d.aux[i])[1] = (t.mV * d.w_zero + t.s0) * d.rho0;
        return;
    }
    if (MODE == GM_DENSITY_EOS) {
        float4 aux = d.aux[i];
        float4 e;
        if (gathered) {
            const float rho_raw = (t.mV * d.w_zero + t.s0) * d.rho0;  // WCSPH.py:39-43
            const float rho = fmaxf(rho_raw, d.rho0);                 // WCSPH.py:75
            const float p = d.stiffness * (powf(rho / d.rho0, d.exponent) - 1.0f);  // WCSPH.py:76
            e = make_float4(p / (rho * rho), aux.x / rho_raw, aux.x, rho);
            aux.y = rho; aux.z = p;
            d.aux[i] = aux;
        } else {
            e = make_float4(0.0f, 0.0f, aux.x, aux.y);
            // WCSPH.py:131-137 for non-fluid particles: static a = 0, dynamic rigid a = g
            const bool st = sph_is_static_rigid(t.flags);
            d.acc[i] = st ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(d.gx, d.gy, d.gz, 0.f);
        }
        d.eos[i] = e;
        return;
    }
    if (MODE == GM_NONPRESSURE) {
        // WCSPH.py:130-140
        if (sph_is_static_rigid(t.flags)) d.acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        else d.acc[i] = make_float4(t.ax, t.ay, t.az, 0.f);
        return;
    }
    if (MODE == GM_PRESSURE) {
        // WCSPH.py:77-85
        if (sph_is_static_rigid(t.flags)) d.acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        else if (gathered) {
            float4 a = d.acc[i];
            a.x += t.px; a.y += t.py; a.z += t.pz;
            d.acc[i] = a;
        }
        return;
    }
    if (MODE == GM_FORCE_FUSED) {
        // fluid: a = (g + non-pressure) + pressure  (WCSPH.py:140 then :85)
        if (gathered) d.acc[i] = make_float4(t.ax + t.px, t.ay + t.py, t.az + t.pz, 0.f);
        return;
    }
}

template <int MODE>
__device__ __forceinline__ float4 load_C_global(const DevView& d, int j) {
    if (MODE == GM_FORCE_FUSED) return d.eos[j];
    if (MODE == GM_NONPRESSURE || MODE == GM_PRESSURE) return make_C_from_aux<MODE>(d.aux[j]);
    return make_float4(0.f, 0.f, 0.f, 0.f);
}

// ---------------------------------------------------------------------------
// v0: per-target cell walk through global memory (reference traversal order)
// ---------------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ void gather_walk_global(const DevView& d, Target& t, int i) {
    const int c = d.key[i];
    const int cz = c % d.nz;
    const int cy = (c / d.nz) % d.ny;
    const int cx = c / (d.nz * d.ny);
    const int zlo = cz > 0 ? cz - 1 : 0;
    const int zhi = cz < d.nz - 1 ? cz + 1 : d.nz - 1;
    for (int dx = -1; dx <= 1; ++dx) {
        const int nx = cx + dx;
        if (nx < 0 || nx >= d.nx) continue;
        for (int dy = -1; dy <= 1; ++dy) {
            const int ny = cy + dy;
            if (ny < 0 || ny >= d.ny) continue;
            const int flo = sph_flatten(d, nx, ny, zlo);
            const int fhi = sph_flatten(d, nx, ny, zhi);
            const int beg = d.cell_end[flo > 0 ? flo - 1 : 0];  // particle_system.py:384 max(0, idx-1)
            const int end = d.cell_end[fhi];
            for (int j = beg; j < end; ++j) {
                if (j == i) continue;
                const float4 A = d.xm[j];
                const float rx = t.x - A.x, ry = t.y - A.y, rz = t.z - A.z;
                const float r2 = rx * rx + ry * ry + rz * rz;
                const float rinv = sph_rsq(r2);
                const float rn = r2 * rinv;
                if (rn < d.h) {  // particle_system.py:385
                    float4 B = make_float4(0.f, 0.f, 0.f, 0.f), Cc = B;
                    if (mode_needs_B<MODE>()) B = d.vf[j];
                    if (mode_needs_C<MODE>()) Cc = load_C_global<MODE>(d, j);
                    pair_physics<MODE>(d, t, rx, ry, rz, r2, rn, rinv, A, B, Cc, j);
                }
            }
        }
    }
}

// list == nullptr: all particles [0,N); else the n entries of list
template <int MODE>
__global__ __launch_bounds__(TPB) void k_gather_simple(DevView d, const int* __restrict__ list, int n) {
    const int tix = blockIdx.x * TPB + threadIdx.x;
    if (tix >= n) return;
    const int i = list ? list[tix] : tix;
    Target t;
    const float4 A = d.xm[i];
    const float4 B = d.vf[i];
    target_init<MODE>(d, t, i, A, B);
    const bool g = target_gathers<MODE>(t.flags);
    if (g) gather_walk_global<MODE>(d, t, i);
    target_finish<MODE>(d, t, i, g);
}

// ---------------------------------------------------------------------------
// v1: LDS-staged cell bricks
// ---------------------------------------------------------------------------
// LDS holds the shell's records as two float2 arrays, (x,y) and (z,m_V): with 8
// particles per cell the slots of lanes in different cells differ by multiples
// of 8, which is a 2-way bank conflict for ds_read_b128's interleaved lane
// groups but conflict-free for ds_read_b64.
// List entries are u16: (shell column << 11) | LDS slot, so CAP <= 2048 and
// NCOL <= 32; the global index of a slot is sColG[col] + slot - sColS[col].
// In the fused step the density sweep writes each target's list to HBM
// (glist[k*cap + i], gcnt[i]) and the force sweep -- same positions, same
// brick layout -- reads it back instead of filtering again.
#define SPH_CNT_WALK 255  // gcnt sentinel: this target must take the exact global cell walk

template <int MODE>
__host__ __device__ constexpr bool mode_writes_list() { return MODE == GM_DENSITY_EOS; }
template <int MODE>
__host__ __device__ constexpr bool mode_reads_list() { return MODE == GM_FORCE_FUSED; }

template <int BX_, int BY_, int BZ_, int CAP_, int LISTCAP_>
struct BrickCfg {
    static constexpr int BX = BX_, BY = BY_, BZ = BZ_, CAP = CAP_, LISTCAP = LISTCAP_;
    static constexpr int NCOL = (BX + 2) * (BY + 2);
    static constexpr int NZS = BZ + 3;  // cell-end entries per column (start + BZ+2 ends)
    static constexpr int PER = (CAP + TPB - 1) / TPB;  // staged records per lane
    // LDS carve (bytes, every offset a multiple of 16)
    static constexpr int OFF_X = 0;
    static constexpr int OFF_Y = OFF_X + CAP * 4;
    static constexpr int OFF_Z = OFF_Y + CAP * 4;
    static constexpr int OFF_W = OFF_Z + CAP * 4;
    static constexpr int OFF_CE = OFF_W + CAP * 4;
    static constexpr int OFF_COLG = OFF_CE + ((NCOL * NZS * 4 + 15) / 16) * 16;
    static constexpr int OFF_COLS = OFF_COLG + 64 * 4;
    static constexpr int OFF_TG = OFF_COLS + 80 * 4;
    static constexpr int OFF_TOFF = OFF_TG + 64 * 4;
    static constexpr int OFF_LIST = OFF_TOFF + 80 * 4;
    static constexpr int BYTES_NOLIST = OFF_LIST;
    static constexpr int BYTES_LIST = OFF_LIST + (LISTCAP + 1) * TPB * 2;  // +1 guard row for overflowing appends
    static_assert(NCOL <= 32, "column id must fit 5 bits of a list entry");
    static_assert(CAP <= 2048, "LDS slot must fit 11 bits of a list entry");
    static_assert(LISTCAP < SPH_CNT_WALK, "gcnt is a byte");
    static_assert(BYTES_LIST <= 54608, "three workgroups per CU (160 KiB LDS)");
};

template <int MODE, class CFG>
__global__ __launch_bounds__(TPB) void k_gather_brick(DevView d, int nbx, int nby, int nbz, int nbricks,
                                                      int bricks_per_xcd, unsigned short* __restrict__ glist,
                                                      unsigned char* __restrict__ gcnt, int cap) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sX = reinterpret_cast<float*>(smem + CFG::OFF_X);
    float* sY = reinterpret_cast<float*>(smem + CFG::OFF_Y);
    float* sZ = reinterpret_cast<float*>(smem + CFG::OFF_Z);
    float* sW = reinterpret_cast<float*>(smem + CFG::OFF_W);
    int* sCE = reinterpret_cast<int*>(smem + CFG::OFF_CE);      // [NCOL][NZS] raw cell_end values of the shell
    int* sColG = reinterpret_cast<int*>(smem + CFG::OFF_COLG);  // global start of the column segment
    int* sColS = reinterpret_cast<int*>(smem + CFG::OFF_COLS);  // LDS start of the column segment (+ total at [64])
    int* sTG = reinterpret_cast<int*>(smem + CFG::OFF_TG);      // global start of the column's targets
    int* sTOff = reinterpret_cast<int*>(smem + CFG::OFF_TOFF);  // target-number start of the column (+ total at [64])
    unsigned short* sList = reinterpret_cast<unsigned short*>(smem + CFG::OFF_LIST);  // absent when the list is read from HBM

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // XCD-aware brick order: hardware block b runs on XCD b%8; give each XCD a
    // contiguous run of bricks so neighbouring bricks share that XCD's L2.
    const int b = blockIdx.x;
    const int brick = (b & 7) * bricks_per_xcd + (b >> 3);
    if (brick >= nbricks) return;
    const int bzi = brick % nbz;
    const int byi = (brick / nbz) % nby;
    const int bxi = brick / (nbz * nby);
    const int cx0 = bxi * CFG::BX, cy0 = byi * CFG::BY, cz0 = bzi * CFG::BZ;
    const int cx1 = min(cx0 + CFG::BX, d.nx), cy1 = min(cy0 + CFG::BY, d.ny), cz1 = min(cz0 + CFG::BZ, d.nz);  // excl.
    const int sx0 = max(cx0 - 1, 0), sy0 = max(cy0 - 1, 0), sz0 = max(cz0 - 1, 0);
    const int sx1 = min(cx1, d.nx - 1), sy1 = min(cy1, d.ny - 1), sz1 = min(cz1, d.nz - 1);  // incl.
    const int ncy = sy1 - sy0 + 1;
    const int ncols = (sx1 - sx0 + 1) * ncy;
    const int nzs = sz1 - sz0 + 1;

    // ---- step A: per shell column, the cell-end table and the segment scan (wave 0) ----
    if (wave == 0) {
        int len = 0, tlen = 0, gstart = 0, tstart = 0;
        if (lane < ncols) {
            const int ix = sx0 + lane / ncy, iy = sy0 + lane % ncy;
            const int base = sph_flatten(d, ix, iy, sz0);
            gstart = d.cell_end[base > 0 ? base - 1 : 0];  // particle_system.py:384 max(0, idx-1): cell 0 quirk kept
            sCE[lane * CFG::NZS] = gstart;
#pragma unroll
            for (int k = 1; k < CFG::NZS; ++k)
                if (k <= nzs) sCE[lane * CFG::NZS + k] = d.cell_end[base + k - 1];
            len = sCE[lane * CFG::NZS + nzs] - gstart;
            if (ix >= cx0 && ix < cx1 && iy >= cy0 && iy < cy1) {
                // targets: cells cz0 .. cz1-1 of this column (true start, also for flat cell 0)
                tstart = (base + (cz0 - sz0) > 0) ? sCE[lane * CFG::NZS + (cz0 - sz0)] : 0;
                tlen = sCE[lane * CFG::NZS + (cz1 - sz0)] - tstart;
            }
        }
        const int incl = sph_wave_inclusive_scan(len, lane);
        const int tincl = sph_wave_inclusive_scan(tlen, lane);
        sColG[lane] = gstart;
        sColS[lane] = incl - len;
        sTG[lane] = tstart;
        sTOff[lane] = tincl - tlen;
        if (lane == 63) { sColS[64] = incl; sTOff[64] = tincl; }
    }
    __syncthreads();
    const int T = sTOff[64];
    const int total = sColS[64];
    if (T == 0) return;
    const bool overflow = total > CFG::CAP;

    // ---- step B: stage the shell's (x, y, z, m_V) records; all loads of a lane in flight together ----
    if (!overflow) {
        float4 buf[CFG::PER];
#pragma unroll
        for (int u = 0; u < CFG::PER; ++u) {
            const int idx = min(tid + u * TPB, total - 1);  // clamped: the load is always valid
            int col = 0;  // largest col with sColS[col] <= idx (entries >= ncols hold `total`)
#pragma unroll
            for (int step = 16; step > 0; step >>= 1)
                if (sColS[col + step] <= idx) col += step;
            buf[u] = d.xm[sColG[col] + (idx - sColS[col])];
        }
#pragma unroll
        for (int u = 0; u < CFG::PER; ++u) {
            const int idx = tid + u * TPB;
            if (idx < total) {
                sX[idx] = buf[u].x; sY[idx] = buf[u].y; sZ[idx] = buf[u].z; sW[idx] = buf[u].w;
            }
        }
    }
    __syncthreads();

    // ---- step C: targets ----
    for (int tn = tid; tn < T; tn += TPB) {
        int col = 0;
#pragma unroll
        for (int step = 16; step > 0; step >>= 1)
            if (sTOff[col + step] <= tn) col += step;
        const int gi = sTG[col] + (tn - sTOff[col]);
        Target t;
        const float4 Ai = d.xm[gi];
        const float4 Bi = d.vf[gi];
        target_init<MODE>(d, t, gi, Ai, Bi);
        const bool g = target_gathers<MODE>(t.flags);
        bool walk = g && overflow;
        int cnt = 0;
        const int li = sColS[col] + (gi - sColG[col]);  // own LDS slot
        if (g && !overflow && !mode_reads_list<MODE>() && !(d.ablate & 4)) {
            const int ix = sx0 + col / ncy, iy = sy0 + col % ncy;
            const int cz = d.key[gi] % d.nz;
            const int klo = (cz > 0 ? cz - 1 : 0) - sz0;  // first cell of the z-run, shell-relative
            const int khi = (cz < d.nz - 1 ? cz + 1 : d.nz - 1) - sz0;
            const float h2p = d.h * d.h * 1.000001f;  // superset filter; the exact r < h test is in phase 2
            // point-to-cell distances for culling (exact: a culled cell cannot hold a particle within h;
            // 0.1% slack on h^2 covers the ulp-level mismatch between float cell hashing and cell geometry)
            const float h2c = d.h * d.h * 1.001f;
            // offsets inside the own cell, clamped so that a particle hashed into a boundary cell from
            // outside the domain only makes the culling more conservative
            const float ux = fminf(fmaxf(t.x - (float)(ix + d.ox) * d.grid_size, 0.0f), d.grid_size),
                        uy = fminf(fmaxf(t.y - (float)(iy + d.oy) * d.grid_size, 0.0f), d.grid_size),
                        uz = fminf(fmaxf(t.z - (float)(cz + d.oz) * d.grid_size, 0.0f), d.grid_size);
            const float gzlo = uz * uz, gzhi = (d.grid_size - uz) * (d.grid_size - uz);
            char* lp = reinterpret_cast<char*>(sList) + tid * 2;
            char* const lp0 = lp;
            char* const lp_guard = lp + CFG::LISTCAP * TPB * 2;
            // phase 1: filter the 9 column runs into the private index list (4 LDS slots in flight)
            for (int dx = -1; dx <= 1; ++dx) {
                const int nx = ix + dx;
                if (nx < 0 || nx >= d.nx) continue;
                const float gx = dx == 0 ? 0.0f : (dx > 0 ? d.grid_size - ux : ux);
                for (int dy = -1; dy <= 1; ++dy) {
                    const int ny = iy + dy;
                    if (ny < 0 || ny >= d.ny) continue;
                    const float gy = dy == 0 ? 0.0f : (dy > 0 ? d.grid_size - uy : uy);
                    const float gxy2 = gx * gx + gy * gy;
                    if (gxy2 > h2c) continue;  // whole column out of reach
                    const int ncol = (nx - sx0) * ncy + (ny - sy0);
                    const int rel = sColS[ncol] - sColG[ncol];
                    // trim the lower / upper cell of the z-run when it is out of reach
                    const int kl = klo + ((cz > 0 && gxy2 + gzlo > h2c) ? 1 : 0);
                    const int kh = khi - ((cz < d.nz - 1 && gxy2 + gzhi > h2c) ? 1 : 0);
                    const int lo = sCE[ncol * CFG::NZS + kl] + rel;
                    const int hi = sCE[ncol * CFG::NZS + kh + 1] + rel;
                    const unsigned tag = (unsigned)ncol << 11;
#define SPH_FILTER(X_, Y_, Z_, jj)                                                   \
    {                                                                                \
        const float rx_ = t.x - (X_), ry_ = t.y - (Y_), rz_ = t.z - (Z_);            \
        const float r2_ = rx_ * rx_ + ry_ * ry_ + rz_ * rz_;                         \
        if (r2_ < h2p) {                                                             \
            *reinterpret_cast<unsigned short*>(lp < lp_guard ? lp : lp_guard) =      \
                (unsigned short)(tag | (unsigned)(jj));                              \
            lp += TPB * 2;                                                           \
        }                                                                            \
    }
                    int j = lo;
                    for (; j + 4 <= hi; j += 4) {
                        const float x0 = sX[j], x1 = sX[j + 1], x2 = sX[j + 2], x3 = sX[j + 3];
                        const float y0 = sY[j], y1 = sY[j + 1], y2 = sY[j + 2], y3 = sY[j + 3];
                        const float z0 = sZ[j], z1 = sZ[j + 1], z2 = sZ[j + 2], z3 = sZ[j + 3];
                        SPH_FILTER(x0, y0, z0, j) SPH_FILTER(x1, y1, z1, j + 1) SPH_FILTER(x2, y2, z2, j + 2)
                        SPH_FILTER(x3, y3, z3, j + 3)
                    }
                    for (; j < hi; ++j) {
                        const float x0 = sX[j], y0 = sY[j], z0 = sZ[j];
                        SPH_FILTER(x0, y0, z0, j)
                    }
#undef SPH_FILTER
                }
            }
            cnt = (int)(lp - lp0) / (TPB * 2);
            if (cnt > CFG::LISTCAP && !(d.ablate & 8)) walk = true;  // list overflow (extreme compression): exact slow path
        }
        if (mode_reads_list<MODE>() && g && !overflow) {
            cnt = gcnt[gi];
            if (cnt == SPH_CNT_WALK) { walk = true; cnt = 0; }
        }
        if (mode_writes_list<MODE>() && g) {
            gcnt[gi] = (unsigned char)(walk ? SPH_CNT_WALK : cnt);
            if (!walk && !(d.ablate & 2))
                for (int k = 0; k < cnt; ++k) glist[(size_t)k * cap + gi] = sList[k * TPB + tid];
        }
        if (g && !walk && !(d.ablate & 1)) {
            // phase 2: pair physics over the list; the next entry's records are prefetched
            unsigned e1 = 0, e2 = 0;  // entries k+1 and k+2
            if (mode_reads_list<MODE>()) {
                if (cnt > 0) e1 = glist[gi];
                if (cnt > 1) e2 = glist[(size_t)cap + gi];
            } else {
                if (cnt > 0) e1 = sList[tid];
                if (cnt > 1) e2 = sList[TPB + tid];
            }
            float4 An = make_float4(0.f, 0.f, 0.f, 0.f);
            float4 Bn = make_float4(0.f, 0.f, 0.f, 0.f), Cn = Bn;
            int gn = 0, jn = -1;
            if (cnt > 0) {
                jn = e1 & 2047;
                const int c2 = e1 >> 11;
                An = make_float4(sX[jn], sY[jn], sZ[jn], sW[jn]);
                gn = sColG[c2] + (jn - sColS[c2]);
                if (mode_needs_B<MODE>()) Bn = d.vf[gn];
                if (mode_needs_C<MODE>()) Cn = load_C_global<MODE>(d, gn);
            }
            for (int k = 0; k < cnt; ++k) {
                const float4 A = An;
                const float4 B = Bn, Cc = Cn;
                const int gj = gn, j = jn;
                if (k + 1 < cnt) {
                    jn = e2 & 2047;
                    const int c2 = e2 >> 11;
                    An = make_float4(sX[jn], sY[jn], sZ[jn], sW[jn]);
                    gn = sColG[c2] + (jn - sColS[c2]);
                    if (mode_needs_B<MODE>()) Bn = d.vf[gn];
                    if (mode_needs_C<MODE>()) Cn = load_C_global<MODE>(d, gn);
                    if (k + 2 < cnt)
                        e2 = mode_reads_list<MODE>() ? (unsigned)glist[(size_t)(k + 2) * cap + gi]
                                                     : (unsigned)sList[(k + 2) * TPB + tid];
                }
                const float rx = t.x - A.x, ry = t.y - A.y, rz = t.z - A.z;
                const float r2 = rx * rx + ry * ry + rz * rz;
                const float rinv = sph_rsq(r2);
                const float rn = r2 * rinv;
                if (rn < d.h && j != li)  // particle_system.py:385
                    pair_physics<MODE>(d, t, rx, ry, rz, r2, rn, rinv, A, B, Cc, gj);
            }
        }
        if (walk) {
            target_init<MODE>(d, t, gi, Ai, Bi);
            gather_walk_global<MODE>(d, t, gi);
        }
        target_finish<MODE>(d, t, gi, g);
    }
}

// EOS alone (first loop of compute_pressure_forces, WCSPH.py:71-76)
__global__ __launch_bounds__(TPB) void k_eos(DevView d) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= d.N) return;
    if (!sph_is_fluid(__float_as_int(d.vf[i].w))) return;
    float4 aux = d.aux[i];
    aux.y = fmaxf(aux.y, d.rho0);
    aux.z = d.stiffness * (powf(aux.y / d.rho0, d.exponent) - 1.0f);
    d.aux[i] = aux;
}

// ---------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------
typedef BrickCfg<4, 2, 4, 1664, 47> Cfg0;  // 4x2x4 cells: 1152 candidates / 256 targets at rest
typedef BrickCfg<2, 2, 8, 1664, 47> Cfg1;  // 2x2x8 cells: 1280 candidates / 256 targets at rest
typedef BrickCfg<2, 4, 4, 1664, 47> Cfg2;  // 2x4x4 cells
typedef BrickCfg<2, 2, 4, 1664, 47> Cfg3;  // 2x2x4 cells:  768 candidates / 128 targets at rest

template <int MODE>
static int launch_simple(SphContext* c, const int* list, int n) {
    if (n <= 0) return 0;
    DevView d = sph_view(c);
    hipLaunchKernelGGL(k_gather_simple<MODE>, dim3((n + TPB - 1) / TPB), dim3(TPB), 0, c->stream, d, list, n);
    SPH_LAUNCH_CHECK(c);
    return 0;
}

template <int MODE, class CFG>
static int launch_brick_cfg(SphContext* c) {
    DevView d = sph_view(c);
    const int nbx = (d.nx + CFG::BX - 1) / CFG::BX, nby = (d.ny + CFG::BY - 1) / CFG::BY,
              nbz = (d.nz + CFG::BZ - 1) / CFG::BZ;
    const int nbricks = nbx * nby * nbz;
    const int per_xcd = (nbricks + 7) / 8;
    const int bytes = mode_reads_list<MODE>() ? CFG::BYTES_NOLIST : CFG::BYTES_LIST;
    static bool attr_set = false;
    if (!attr_set) {
        SPH_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gather_brick<MODE, CFG>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        attr_set = true;
    }
    hipLaunchKernelGGL((k_gather_brick<MODE, CFG>), dim3(per_xcd * 8), dim3(TPB), bytes, c->stream, d, nbx, nby, nbz,
                       nbricks, per_xcd, c->glist, c->gcnt, c->cap);
    SPH_LAUNCH_CHECK(c);
    return 0;
}

template <int MODE>
static int launch_brick(SphContext* c) {
    switch (c->opt_brick_shape) {
        case 1: return launch_brick_cfg<MODE, Cfg1>(c);
        case 2: return launch_brick_cfg<MODE, Cfg2>(c);
        case 3: return launch_brick_cfg<MODE, Cfg3>(c);
        default: return launch_brick_cfg<MODE, Cfg0>(c);
    }
}

template <int MODE>
static int launch_sweep(SphContext* c) {
    if (c->N <= 0) return 0;
    if (c->opt_gather_impl == 0) return launch_simple<MODE>(c, nullptr, c->N);
    return launch_brick<MODE>(c);
}

int sphk_gather(SphContext* c, int mode) {
    switch (mode) {
        case GM_BVOL_STATIC: return launch_simple<GM_BVOL_STATIC>(c, nullptr, c->N);  // init only
        case GM_BVOL_DYNAMIC: return launch_simple<GM_BVOL_DYNAMIC>(c, c->dyn_list, c->n_dyn_host);
        case GM_DENSITY: return launch_sweep<GM_DENSITY>(c);
        case GM_DENSITY_EOS: return launch_sweep<GM_DENSITY_EOS>(c);
        case GM_NONPRESSURE: return launch_sweep<GM_NONPRESSURE>(c);
        case GM_PRESSURE: return launch_sweep<GM_PRESSURE>(c);
        case GM_FORCE_FUSED: return launch_sweep<GM_FORCE_FUSED>(c);
    }
    return sph_fail(c, SPH_E_INVALID, "unknown gather mode");
}

int sphk_eos(SphContext* c) {
    if (c->N <= 0) return 0;
    DevView d = sph_view(c);
    hipLaunchKernelGGL(k_eos, dim3((c->N + TPB - 1) / TPB), dim3(TPB), 0, c->stream, d);
    SPH_LAUNCH_CHECK(c);
    return 0;
}
