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
    t.m = t.rho = t.p = t.dpi = t.st_c = 0.0f;
    if (MODE == GM_NONPRESSURE || MODE == GM_PRESSURE) {
        const float4 aux = d.aux[i];
        t.m = aux.x; t.rho = aux.y; t.p = aux.z;
        t.dpi = aux.z / (aux.y * aux.y);  // WCSPH.py:49
        t.st_c = d.sigma / aux.x;         // WCSPH.py:100
    }
    if (MODE == GM_FORCE_FUSED) {
        const float4 e = d.eos[i];
        t.dpi = e.x; t.m = e.z; t.rho = e.w;
        t.p = e.x * (e.w * e.w);
        t.st_c = d.sigma / e.z;
    }
    if (MODE == GM_BVOL_STATIC || MODE == GM_BVOL_DYNAMIC) t.s0 = d.w_zero;  // sph_base.py:95, 110
}

// One accepted pair (i != j, r_norm = |x_i - x_j| < h).  gj = global (sorted)
// index of j, needed only for the coupling scatter.
template <int MODE>
__device__ __forceinline__ void pair_physics(const DevView& d, Target& t, float rx, float ry, float rz, float r2,
                                             float r_norm, const float4 A, const float4 B, const float4 Cc, int gj) {
    if (MODE == GM_DENSITY || MODE == GM_DENSITY_EOS) {
        // WCSPH.py:19-30: fluid and solid neighbours add m_V_j * W identically
        t.s0 += A.w * sph_W(d, r_norm);
        return;
    }
    const int fj = __float_as_int(B.w);
    if (MODE == GM_BVOL_STATIC || MODE == GM_BVOL_DYNAMIC) {
        if (sph_flags_material(fj) == SPH_MATERIAL_SOLID) t.s0 += sph_W(d, r_norm);  // sph_base.py:100-103
        return;
    }
    const bool j_fluid = sph_is_fluid(fj);
    if (MODE == GM_NONPRESSURE || MODE == GM_FORCE_FUSED) {
        if (j_fluid) {
            // surface tension  WCSPH.py:93-102
            const float w = (r2 > d.d2) ? sph_W(d, r_norm) : d.w_d;
            const float c = t.st_c * Cc.z;
            t.ax -= c * rx * w; t.ay -= c * ry * w; t.az -= c * rz * w;
            // viscosity  WCSPH.py:105-116
            const float v_xy = (t.vx - B.x) * rx + (t.vy - B.y) * ry + (t.vz - B.z) * rz;
            const float3 gw = sph_gradW(d, rx, ry, rz, r_norm);
            const float cv = d.visc_d_nu * Cc.y * v_xy / (r_norm * r_norm + d.visc_eps);
            t.ax += cv * gw.x; t.ay += cv * gw.y; t.az += cv * gw.z;
        }
        // solid neighbour: boundary_viscosity = 0.0 => contributes exactly 0 (WCSPH.py:117-125)
    }
    if (MODE == GM_PRESSURE || MODE == GM_FORCE_FUSED) {
        const float3 gw = sph_gradW(d, rx, ry, rz, r_norm);
        if (j_fluid) {
            // WCSPH.py:51-57
            const float c = -d.rho0 * A.w * (t.dpi + Cc.x);
            t.px += c * gw.x; t.py += c * gw.y; t.pz += c * gw.z;
        } else if (sph_flags_material(fj) == SPH_MATERIAL_SOLID) {
            // WCSPH.py:58-68 (Akinci 2012 boundary pressure + two-way coupling)
            const float dpj = t.p / (d.rho0 * d.rho0);
            const float c = -d.rho0 * A.w * (t.dpi + dpj);
            const float fx = c * gw.x, fy = c * gw.y, fz = c * gw.z;
            t.px += fx; t.py += fy; t.pz += fz;
            if (sph_is_dynamic_rigid(fj)) {
                const float sc = d.rho0 / Cc.w;
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
                const float rn = sqrtf(r2);
                if (rn < d.h) {  // particle_system.py:385
                    float4 B = make_float4(0.f, 0.f, 0.f, 0.f), Cc = B;
                    if (mode_needs_B<MODE>()) B = d.vf[j];
                    if (mode_needs_C<MODE>()) Cc = load_C_global<MODE>(d, j);
                    pair_physics<MODE>(d, t, rx, ry, rz, r2, rn, A, B, Cc, j);
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
template <int BX_, int BY_, int BZ_, int CAP_, int LISTCAP_, bool BC_LDS_>
struct BrickCfg {
    static constexpr int BX = BX_, BY = BY_, BZ = BZ_, CAP = CAP_, LISTCAP = LISTCAP_;
    static constexpr bool BC_LDS = BC_LDS_;
    static constexpr int NCOL = (BX + 2) * (BY + 2);  // <= 64 (one wave builds the column table)
    static constexpr int NZS = BZ + 3;                // cell-end entries per column
    // LDS carve (bytes, every offset a multiple of 16)
    static constexpr int OFF_A = 0;
    static constexpr int OFF_B = OFF_A + CAP * 16;
    static constexpr int OFF_C = OFF_B + (BC_LDS ? CAP * 16 : 0);
    static constexpr int OFF_G = OFF_C + (BC_LDS ? CAP * 16 : 0);  // global index of each staged candidate
    static constexpr int OFF_LIST = OFF_G + CAP * 4;
    static constexpr int OFF_CE = OFF_LIST + LISTCAP * TPB * 2;
    static constexpr int OFF_COLG = OFF_CE + ((NCOL * NZS * 4 + 15) / 16) * 16;
    static constexpr int OFF_COLS = OFF_COLG + 64 * 4;
    static constexpr int OFF_TG = OFF_COLS + 80 * 4;
    static constexpr int OFF_TOFF = OFF_TG + 64 * 4;
    static constexpr int BYTES = OFF_TOFF + 80 * 4;
    static_assert(NCOL <= 64, "column table is built by one wave");
    static_assert(CAP <= 65535, "LDS indices are stored as u16");
};

template <int MODE, class CFG>
__global__ __launch_bounds__(TPB) void k_gather_brick(DevView d, int nbx, int nby, int nbz, int nbricks,
                                                      int bricks_per_xcd) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* sA = reinterpret_cast<float4*>(smem + CFG::OFF_A);
    float4* sB = reinterpret_cast<float4*>(smem + CFG::OFF_B);
    float4* sC = reinterpret_cast<float4*>(smem + CFG::OFF_C);
    int* sG = reinterpret_cast<int*>(smem + CFG::OFF_G);
    unsigned short* sList = reinterpret_cast<unsigned short*>(smem + CFG::OFF_LIST);
    int* sCE = reinterpret_cast<int*>(smem + CFG::OFF_CE);      // [NCOL][NZS], LDS-relative candidate index
    int* sColG = reinterpret_cast<int*>(smem + CFG::OFF_COLG);  // global start of the column segment
    int* sColS = reinterpret_cast<int*>(smem + CFG::OFF_COLS);  // LDS start of the column segment (+ total)
    int* sTG = reinterpret_cast<int*>(smem + CFG::OFF_TG);      // global start of the column's targets
    int* sTOff = reinterpret_cast<int*>(smem + CFG::OFF_TOFF);  // target-number start of the column (+ total)

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

    // ---- step A: column table (wave 0) ----
    if (wave == 0) {
        int len = 0, tlen = 0, gstart = 0, tstart = 0;
        if (lane < ncols) {
            const int ix = sx0 + lane / ncy, iy = sy0 + lane % ncy;
            const int flo = sph_flatten(d, ix, iy, sz0);
            const int fhi = sph_flatten(d, ix, iy, sz1);
            gstart = d.cell_end[flo > 0 ? flo - 1 : 0];  // particle_system.py:384 (cell 0 quirk kept)
            len = d.cell_end[fhi] - gstart;
            if (ix >= cx0 && ix < cx1 && iy >= cy0 && iy < cy1) {
                const int tlo = sph_flatten(d, ix, iy, cz0);
                const int thi = sph_flatten(d, ix, iy, cz1 - 1);
                tstart = tlo > 0 ? d.cell_end[tlo - 1] : 0;
                tlen = d.cell_end[thi] - tstart;
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

    if (!overflow) {
        // cell-end table, LDS-relative: sCE[col][0] = segment start, sCE[col][k] = end of cell sz0+k-1
        for (int e = tid; e < ncols * (nzs + 1); e += TPB) {
            const int col = e / (nzs + 1), k = e % (nzs + 1);
            int v;
            if (k == 0) v = sColS[col];
            else {
                const int ix = sx0 + col / ncy, iy = sy0 + col % ncy;
                v = d.cell_end[sph_flatten(d, ix, iy, sz0 + k - 1)] - sColG[col] + sColS[col];
            }
            sCE[col * CFG::NZS + k] = v;
        }
        // ---- step B: stage candidate records, one column segment per wave pass ----
        for (int col = wave; col < ncols; col += TPB / 64) {
            const int g0 = sColG[col], s0 = sColS[col], len = sColS[col + 1 < ncols ? col + 1 : 64] - s0;
            for (int k = lane; k < len; k += 64) {
                sA[s0 + k] = d.xm[g0 + k];
                sG[s0 + k] = g0 + k;
                if (CFG::BC_LDS && mode_needs_B<MODE>()) sB[s0 + k] = d.vf[g0 + k];
                if (CFG::BC_LDS && mode_needs_C<MODE>()) sC[s0 + k] = load_C_global<MODE>(d, g0 + k);
            }
        }
    }
    __syncthreads();

    // ---- step C: targets ----
    for (int tn = tid; tn < T; tn += TPB) {
        int col = 0;
        while (tn >= sTOff[col + 1 < 64 ? col + 1 : 64] && col < ncols - 1) ++col;
        const int gi = sTG[col] + (tn - sTOff[col]);
        Target t;
        const float4 Ai = d.xm[gi];
        const float4 Bi = d.vf[gi];
        target_init<MODE>(d, t, gi, Ai, Bi);
        const bool g = target_gathers<MODE>(t.flags);
        if (g && overflow) gather_walk_global<MODE>(d, t, gi);
        if (g && !overflow) {
            const int ix = sx0 + col / ncy, iy = sy0 + col % ncy;
            const int cz = d.key[gi] % d.nz;
            const int li = sColS[col] + (gi - sColG[col]);  // own LDS slot
            const int klo = (cz > 0 ? cz - 1 : 0) - sz0;            // first cell of the z-run, shell-relative
            const int khi = (cz < d.nz - 1 ? cz + 1 : d.nz - 1) - sz0;
            const float h2p = d.h * d.h * 1.000001f;  // superset filter; the exact r < h test is in phase 2
            int cnt = 0;
            // phase 1: filter candidates into the private index list
            for (int dx = -1; dx <= 1; ++dx) {
                const int nx = ix + dx;
                if (nx < 0 || nx >= d.nx) continue;
                for (int dy = -1; dy <= 1; ++dy) {
                    const int ny = iy + dy;
                    if (ny < 0 || ny >= d.ny) continue;
                    const int ncol = (nx - sx0) * ncy + (ny - sy0);
                    const int lo = sCE[ncol * CFG::NZS + klo];
                    const int hi = sCE[ncol * CFG::NZS + khi + 1];
                    for (int j = lo; j < hi; ++j) {
                        const float4 A = sA[j];
                        const float rx = t.x - A.x, ry = t.y - A.y, rz = t.z - A.z;
                        const float r2 = rx * rx + ry * ry + rz * rz;
                        if (r2 < h2p && j != li) {
                            if (cnt < CFG::LISTCAP) {
                                sList[cnt * TPB + tid] = (unsigned short)j;
                                ++cnt;
                            } else {  // list full (extreme compression): do the pair now
                                const float rn = sqrtf(r2);
                                if (rn < d.h) {
                                    const int gj = sG[j];
                                    float4 B = make_float4(0.f, 0.f, 0.f, 0.f), Cc = B;
                                    if (mode_needs_B<MODE>()) B = CFG::BC_LDS ? sB[j] : d.vf[gj];
                                    if (mode_needs_C<MODE>()) Cc = CFG::BC_LDS ? sC[j] : load_C_global<MODE>(d, gj);
                                    pair_physics<MODE>(d, t, rx, ry, rz, r2, rn, A, B, Cc, gj);
                                }
                            }
                        }
                    }
                }
            }
            // phase 2: pair physics over the list
            for (int k = 0; k < cnt; ++k) {
                const int j = sList[k * TPB + tid];
                const float4 A = sA[j];
                const float rx = t.x - A.x, ry = t.y - A.y, rz = t.z - A.z;
                const float r2 = rx * rx + ry * ry + rz * rz;
                const float rn = sqrtf(r2);
                if (rn < d.h) {  // particle_system.py:385
                    const int gj = sG[j];
                    float4 B = make_float4(0.f, 0.f, 0.f, 0.f), Cc = B;
                    if (mode_needs_B<MODE>()) B = CFG::BC_LDS ? sB[j] : d.vf[gj];
                    if (mode_needs_C<MODE>()) Cc = CFG::BC_LDS ? sC[j] : load_C_global<MODE>(d, gj);
                    pair_physics<MODE>(d, t, rx, ry, rz, r2, rn, A, B, Cc, gj);
                }
            }
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
typedef BrickCfg<2, 2, 8, 2048, 48, false> Cfg0;  // 2x2x8 cells, A in LDS, B/C via L2   (~61 KB)
typedef BrickCfg<2, 2, 8, 1792, 48, true> Cfg1;   // 2x2x8 cells, A/B/C in LDS           (~120 KB)
typedef BrickCfg<4, 2, 4, 2048, 48, false> Cfg2;  // 4x2x4 cells, A in LDS
typedef BrickCfg<4, 4, 4, 2816, 48, false> Cfg3;  // 4x4x4 cells, A in LDS (two target rounds)

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
    static bool attr_set = false;
    if (!attr_set) {
        SPH_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gather_brick<MODE, CFG>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, CFG::BYTES));
        attr_set = true;
    }
    hipLaunchKernelGGL((k_gather_brick<MODE, CFG>), dim3(per_xcd * 8), dim3(TPB), CFG::BYTES, c->stream, d, nbx, nby,
                       nbz, nbricks, per_xcd);
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
