// sph_bricks.h -- the brick-list builder (one workgroup = four column groups), shared by the stand-alone kernel
// k_brick_list (sph_gather.hip: rebuilds for other target ranges) and the sort's place kernel (sph_sort.hip), which
// runs it in extra workgroups beside its own work: the list needs nothing but the scanned cell array, its 12 us are a
// chain of dependent phases on 304 workgroups, and as a launch of its own they sat on the step's critical path.
#pragma once
#include "sph_internal.h"

#define SPH_TPB 256
#define SPH_BRICK_BX 4
#define SPH_BRICK_BY 2
#define SPH_BRICK_BZ 4

// what the builder reads of a context (a whole DevView beside the scatter's own would cost that kernel SGPRs and, with
// them, resident workgroups)
struct BrickView {
    int nx, ny, nz;
    int tgt_lo, tgt_hi, tgt_lo2, tgt_hi2;  // x layers whose particles are targets of the sweeps the list is built for
    const int* cell_end;
};

struct BrickListArgs {
    BrickView d;
    int nbx, nby;
    int2* list;
    int* count;
    int list_cap, tmax, smax, fixed_bz;
    int nblocks;        // workgroups of the builder (0: no list wanted)
    unsigned lds_bytes; // dynamic LDS a builder workgroup needs
};

#ifdef __HIPCC__
template <int BX, int BY, int BZ>
__device__ __forceinline__ void sph_brick_list_block(const BrickView& d, int nbx, int nby, int2* __restrict__ list,
                                                     int* __restrict__ count, int list_cap, int tmax, int smax, int fixed_bz,
                                                     int block, int* sm_bl) {
    constexpr int NCOL = (BX + 2) * (BY + 2);
    __shared__ int s_cnt[SPH_TPB / 64][2];
    __shared__ int s_base[2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nz = d.nz;
    int* Sp = sm_bl + wave * 5 * (nz + 1);  // [nz + 1] shell records in layers [0, z) of the column group's shell columns
    int* Tp = Sp + (nz + 1);                // [nz + 1] targets in layers [0, z)
    int* Ex = Tp + (nz + 1);                // [nz] height of the brick that would start at layer z (0: no target there)
    int* bz = Ex + (nz + 1);                // bricks of this column group: first layer | height << 16
    int* bt = bz + (nz + 1);                //                              targets
    const int cg = block * (SPH_TPB / 64) + wave;
    const bool live = cg < nbx * nby;
    const int bxi = cg / nby, byi = cg % nby;
    const int cx0 = bxi * BX, cy0 = byi * BY;
    int carryS = 0, carryT = 0;
    if (live && lane == 0) { Sp[0] = 0; Tp[0] = 0; }
    for (int zb = 0; zb < nz; zb += 64) {
        const int z = zb + lane;
        // The columns' loads are issued in batches of (BY + 2) x 2 = 8 columns (16 loads in flight) before the first of a
        // batch is used: a loop that waited for each pair of loads made this kernel three times as long, all 48 at once
        // cost 96 VGPRs -- too many for a guest in the scatter kernel.
        int s_ = 0, t_ = 0;
#pragma unroll 1
        for (int c0 = 0; c0 < NCOL; c0 += 2 * (BY + 2)) {
            int hi_[2 * (BY + 2)], lo_[2 * (BY + 2)];
#pragma unroll
            for (int u = 0; u < 2 * (BY + 2); ++u) {
                const int c = c0 + u;
                const int ix = cx0 - 1 + c / (BY + 2), iy = cy0 - 1 + c % (BY + 2);
                const bool ok = live && z < nz && c < NCOL && ix >= 0 && ix < d.nx && iy >= 0 && iy < d.ny;
                const int f = ok ? (ix * d.ny + iy) * d.nz + z : 0;
                hi_[u] = ok ? d.cell_end[f] : 0;
                lo_[u] = (ok && f > 0) ? d.cell_end[f - 1] : 0;
            }
#pragma unroll
            for (int u = 0; u < 2 * (BY + 2); ++u) {
                const int c = c0 + u;
                const int ix = cx0 - 1 + c / (BY + 2), iy = cy0 - 1 + c % (BY + 2);
                const int n = hi_[u] - lo_[u];
                s_ += n;
                const bool tgt = ix >= cx0 && ix < cx0 + BX && iy >= cy0 && iy < cy0 + BY &&
                                 ((ix >= d.tgt_lo && ix < d.tgt_hi) || (ix >= d.tgt_lo2 && ix < d.tgt_hi2));
                t_ += tgt ? n : 0;
            }
        }
        const int si = sph_wave_inclusive_scan(s_, lane), ti = sph_wave_inclusive_scan(t_, lane);
        if (live && z < nz) { Sp[z + 1] = carryS + si; Tp[z + 1] = carryT + ti; }
        carryS += __shfl(si, 63, 64);
        carryT += __shfl(ti, 63, 64);
    }
    __syncthreads();
    // every layer: how high would a brick starting here be?
    if (live)
        for (int z = lane; z < nz; z += 64) {
            int e = 0;
            const int t0 = Tp[z];
            if (Tp[z + 1] != t0) {
                if (fixed_bz > 0) {
                    e = min(fixed_bz - z % fixed_bz, nz - z);
                } else {
                    const int s0 = Sp[max(z - 1, 0)];
                    e = 1;
                    while (z + e < nz && e < BZ && Tp[z + e + 1] != Tp[z + e] && Tp[z + e + 1] - t0 <= tmax &&
                           Sp[min(z + e + 2, nz)] - s0 <= smax)
                        ++e;
                }
            }
            Ex[z] = e;
        }
    __syncthreads();
    int nb = 0;
    if (live && lane == 0) {
        int z = 0;
        while (z < nz) {
            const int e = Ex[z];
            if (e == 0) { ++z; continue; }
            const int z0 = fixed_bz > 0 ? (z / fixed_bz) * fixed_bz : z;  // fixed partition: bricks aligned to multiples of fixed_bz
            const int z1 = z + e;
            bz[nb] = z0 | ((z1 - z0) << 16);
            bt[nb] = Tp[z1] - Tp[z0];
            ++nb;
            z = z1;
        }
    }
    nb = __shfl(nb, 0, 64);
    int nh = 0;
    for (int k0 = 0; k0 < nb; k0 += 64) {
        const int k = k0 + lane;
        nh += __popcll(__ballot(k < nb && bt[k] >= SPH_BRICK_HEAVY));  // (lane 0's LDS writes: same wave, program order)
    }
    if (lane == 0) { s_cnt[wave][0] = nh; s_cnt[wave][1] = nb - nh; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int th = 0, tl = 0;
        for (int w = 0; w < SPH_TPB / 64; ++w) { th += s_cnt[w][0]; tl += s_cnt[w][1]; }
        s_base[0] = th ? atomicAdd(&count[0], th) : 0;
        s_base[1] = tl ? atomicAdd(&count[1], tl) : 0;
    }
    __syncthreads();
    int oh = s_base[0], ol = s_base[1];
    for (int w = 0; w < wave; ++w) { oh += s_cnt[w][0]; ol += s_cnt[w][1]; }
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int k0 = 0; k0 < nb; k0 += 64) {
        const int k = k0 + lane;
        const bool ok = k < nb;
        const bool heavy = ok && bt[k] >= SPH_BRICK_HEAVY, light = ok && !heavy;
        const unsigned long long mh = __ballot(heavy), ml = __ballot(light);
        if (heavy) list[oh + __popcll(mh & below)] = make_int2(cg, bz[k]);
        if (light) list[list_cap - 1 - (ol + __popcll(ml & below))] = make_int2(cg, bz[k]);
        oh += __popcll(mh);
        ol += __popcll(ml);
    }
}

#endif  // __HIPCC__
