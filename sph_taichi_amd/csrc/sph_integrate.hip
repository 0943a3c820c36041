// sph_integrate.hip -- K8..K11 of the hot path: symplectic Euler (WCSPH.py:
// 143-149), domain walls (sph_base.py:118-123, 149-179), shape-matching rigid
// bodies (sph_base.py:87-89, 182-222), plus the field insert/extract kernels
// behind sph_upload / sph_download.  All streaming, HBM-bound.
#include "sph_internal.h"

#define TPB 256

struct WallHi { float v[3]; };

template <bool FLUID_WALLS>
__global__ __launch_bounds__(TPB) void k_advect(DevView d, WallHi hi) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= d.N) return;
    float4 vf = d.vf[i];
    if (!sph_flags_dynamic(__float_as_int(vf.w))) return;
    float4 xm = d.xm[i];
    advect_one<FLUID_WALLS>(d, hi.v, xm, vf, d.acc[i]);
    d.xm[i] = xm;
    d.vf[i] = vf;
}

// Precondition of the uniform-fluid force path: every fluid particle has the same mass and m_V == m_V0 (bitwise).
// out[0] = min, out[1] = max of the mass bit patterns (positive floats order like unsigned ints), out[2] = number of
// fluid particles whose m_V differs from m_V0, out[3] = number of fluid particles.
__global__ __launch_bounds__(TPB) void k_check_uniform(DevView d, unsigned* __restrict__ out) {
    unsigned lo = 0xFFFFFFFFu, hi = 0u;
    int bad = 0, nf = 0;
    for (int i = blockIdx.x * TPB + threadIdx.x; i < d.N; i += gridDim.x * TPB)  // few blocks: few same-address atomics
        if (sph_is_fluid(__float_as_int(d.vf[i].w))) {
            const unsigned mb = __float_as_uint(d.aux[i].x);
            lo = min(lo, mb); hi = max(hi, mb);
            bad += __float_as_uint(d.xm[i].w) != __float_as_uint(d.m_V0) ? 1 : 0;
            nf += 1;
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lo = min(lo, (unsigned)__shfl_down((int)lo, off, 64));
        hi = max(hi, (unsigned)__shfl_down((int)hi, off, 64));
        bad += __shfl_down(bad, off, 64);
        nf += __shfl_down(nf, off, 64);
    }
    if ((threadIdx.x & 63) == 0 && nf > 0) {
        atomicMin(&out[0], lo);
        atomicMax(&out[1], hi);
        atomicAdd(&out[2], (unsigned)bad);
        atomicAdd(&out[3], (unsigned)nf);
    }
}

// ---- DFSPH element-wise kernels (eos record = (dfsph_factor, density_adv, m, density)) ----
// DFSPH.py:388-394 predict_velocity
__global__ __launch_bounds__(TPB) void k_df_predict_velocity(DevView d) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= d.N) return;
    float4 vf = d.vf[i];
    const int fl = __float_as_int(vf.w);
    if (!(sph_flags_dynamic(fl) && sph_is_fluid(fl))) return;
    const float4 a = d.acc[i];
    vf.x += d.dt * a.x; vf.y += d.dt * a.y; vf.z += d.dt * a.z;
    d.vf[i] = vf;
}

// DFSPH.py:100-107 advect (only dynamic rigid particles integrate their acceleration here) + the fluid wall pass of
// sph_base.py:270-271, which commutes with the rigid solve in between
template <bool FLUID_WALLS>
__global__ __launch_bounds__(TPB) void k_df_advect(DevView d, WallHi hi) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= d.N) return;
    float4 vf = d.vf[i];
    const int fl = __float_as_int(vf.w);
    if (!sph_flags_dynamic(fl)) return;
    float4 xm = d.xm[i];
    if (sph_is_dynamic_rigid(fl)) {
        const float4 a = d.acc[i];
        vf.x += d.dt * a.x; vf.y += d.dt * a.y; vf.z += d.dt * a.z;
    }
    xm.x += d.dt * vf.x; xm.y += d.dt * vf.y; xm.z += d.dt * vf.z;
    if (FLUID_WALLS && sph_is_fluid(fl)) wall_collide(d, hi.v, xm, vf);
    d.xm[i] = xm;
    d.vf[i] = vf;
}

// DFSPH.py:233-237 multiply_time_step(dfsph_factor, s)
__global__ __launch_bounds__(TPB) void k_df_scale_factor(DevView d, float s) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= d.N) return;
    if (!sph_is_fluid(__float_as_int(d.vf[i].w))) return;
    float* e = reinterpret_cast<float*>(&d.eos[i]);
    e[0] *= s;
}

// DFSPH.py:224-230 compute_density_error: sum over fluid of density_0 * density_adv - offset.  Two stages, f64
// partials in a fixed order (the reference's f32 reduction order is scheduling-dependent anyway); the total lands
// in pinned host memory, so the host only waits for the stream.
__global__ __launch_bounds__(TPB) void k_df_density_error(DevView d, float offset, double* __restrict__ part, int first,
                                                          int last) {
    __shared__ double red[TPB / 64];
    double v = 0.0;
    for (int i = first + blockIdx.x * TPB + threadIdx.x; i < last; i += gridDim.x * TPB)
        if (sph_is_fluid(__float_as_int(d.vf[i].w))) v += (double)(d.rho0 * d.eos[i].y - offset);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < TPB / 64; ++w) t += red[w];
        part[blockIdx.x] = t;
    }
}

// ... and the solver loops' form (sph_api.hip: df_solve_loop): the same two stages behind the solve's gate, the second one
// making the reference's convergence test itself -- (f32 sum) / fluid_particle_num <= eta in f64, DFSPH.py:260-263 / 346-349 --
// and stamping the gate when it holds.
__global__ __launch_bounds__(TPB) void k_df_density_error_gated(DevView d, float offset, double* __restrict__ part) {
    if (d.gate && *d.gate == d.gate_epoch) return;
    __shared__ double red[TPB / 64];
    double v = 0.0;
    for (int i = blockIdx.x * TPB + threadIdx.x; i < d.N; i += gridDim.x * TPB)
        if (sph_is_fluid(__float_as_int(d.vf[i].w))) v += (double)(d.rho0 * d.eos[i].y - offset);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < TPB / 64; ++w) t += red[w];
        part[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(64) void k_df_convergence_test(const double* __restrict__ part, int n, int count, double eta,
                                                            SphContext::DfSlot* __restrict__ slot, unsigned* __restrict__ gate,
                                                            unsigned epoch) {
    if (*gate == epoch) return;
    double v = 0.0;
    for (int k = threadIdx.x; k < n; k += 64) v += part[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (threadIdx.x == 0) {
        const float err = (float)v;                          // compute_density_error returns an f32 (DFSPH.py:224-230)
        const double avg = (double)err / (double)count;      // Python-scope float arithmetic: f64
        const int conv = avg <= eta ? 1 : 0;
        slot->err = err;
        slot->avg = avg;
        slot->converged = conv;
        if (conv) *gate = epoch;
    }
}

// ... and the same test on the per-BRICK partials the refresh sweep itself left behind (SPH_OPT_DF_FUSE_ERROR: no streaming pass over
// the particles at all): entries [0, heavy) and (list_cap - light, list_cap) of the brick list, in that order
__global__ __launch_bounds__(1024) void k_df_convergence_test_bricks(const double* __restrict__ part, const int* __restrict__ brick_count,
                                                                     int list_cap, int count, double eta,
                                                                     SphContext::DfSlot* __restrict__ slot, unsigned* __restrict__ gate,
                                                                     unsigned epoch) {
    if (*gate == epoch) return;
    __shared__ double red[16];
    const int nbh = brick_count[0], n = nbh + brick_count[1];
    // ONE workgroup adds up ~16 k partials at 1.75 M particles: 1,024 lanes, four independent loads in flight per lane (a
    // 256-lane loop of dependent adds waited for ~60 loads one after the other: as slow as the streaming kernel it replaced)
    auto ld = [&](int k) -> double { return k < nbh ? part[k] : part[list_cap - 1 - (k - nbh)]; };
    double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
    int k = threadIdx.x;
    for (; k + 3 * 1024 < n; k += 4 * 1024) {
        const double a = ld(k), b = ld(k + 1024), c = ld(k + 2048), e = ld(k + 3072);
        v0 += a; v1 += b; v2 += c; v3 += e;
    }
    for (; k < n; k += 1024) v0 += ld(k);
    double v = (v0 + v1) + (v2 + v3);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += red[w];
        const float err = (float)t;                          // compute_density_error returns an f32 (DFSPH.py:224-230)
        const double avg = (double)err / (double)count;      // Python-scope float arithmetic: f64
        const int conv = avg <= eta ? 1 : 0;
        slot->err = err;
        slot->avg = avg;
        slot->converged = conv;
        if (conv) *gate = epoch;
    }
}

__global__ __launch_bounds__(64) void k_df_density_error_total(const double* __restrict__ part, int n,
                                                               double* __restrict__ out) {
    double v = 0.0;
    for (int k = threadIdx.x; k < n; k += 64) v += part[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (threadIdx.x == 0) *out = v;
}

// advect (+ fluid walls) of a contiguous record range (slab mode: the boundary layers, after their packers)
__global__ __launch_bounds__(TPB) void k_advect_range(DevView d, WallHi hi, int first, int count) {
    const int k = blockIdx.x * TPB + threadIdx.x;
    if (k >= count) return;
    const int i = first + k;
    float4 vf = d.vf[i];
    if (!sph_flags_dynamic(__float_as_int(vf.w))) return;
    float4 xm = d.xm[i];
    advect_one<true>(d, hi.v, xm, vf, d.acc[i]);
    d.xm[i] = xm;
    d.vf[i] = vf;
}

// advect of the dynamic rigid particles alone (the fluid was integrated inside the force sweep)
__global__ __launch_bounds__(TPB) void k_advect_list(DevView d, WallHi hi, const int* __restrict__ list, int n) {
    const int tix = blockIdx.x * TPB + threadIdx.x;
    if (tix >= n) return;
    const int i = list[tix];
    float4 vf = d.vf[i];
    if (!sph_is_dynamic_rigid(__float_as_int(vf.w))) return;
    float4 xm = d.xm[i];
    advect_one<false>(d, hi.v, xm, vf, d.acc[i]);
    d.xm[i] = xm;
    d.vf[i] = vf;
}

// Slab halo packer: records [first, first+count) AS THEY WILL BE after this step's advect, written to dst
// (count xm, then count vf, then count aux) without touching the arrays -- the interior force sweep that runs
// concurrently with the exchange still needs the old positions; k_advect later repeats the same update in place.
__global__ __launch_bounds__(TPB) void k_pack_advected(DevView d, WallHi hi, int first, int count,
                                                       float4* __restrict__ dst) {
    const int k = blockIdx.x * TPB + threadIdx.x;
    if (k >= count) return;
    const int i = first + k;
    float4 xm = d.xm[i], vf = d.vf[i];
    advect_one<true>(d, hi.v, xm, vf, d.acc[i]);
    dst[k] = xm;
    dst[count + k] = vf;
    dst[2 * count + k] = d.aux[i];
}

__global__ __launch_bounds__(TPB) void k_enforce_boundary(DevView d, WallHi hi, int particle_type,
                                                          const int* __restrict__ list, int n) {
    const int tix = blockIdx.x * TPB + threadIdx.x;
    if (tix >= n) return;
    const int i = list ? list[tix] : tix;
    float4 vf = d.vf[i];
    const int fl = __float_as_int(vf.w);
    if (!(sph_flags_material(fl) == particle_type && sph_flags_dynamic(fl))) return;
    float4 xm = d.xm[i];
    wall_collide(d, hi.v, xm, vf);
    d.xm[i] = xm;
    d.vf[i] = vf;
}

// ---- rigid bodies -----------------------------------------------------------
// Shape matching sums as a deterministic two-stage reduction: every block writes its partial sums to
// part[block][16] ([0] = sum m, [1..3] = sum m x, [4..12] = A row-major); consumers add the partials up in block
// order.  No atomics, so the result does not depend on scheduling (the reference's f32 atomic sums do).
#define RIGID_PART 16
// The sums are EXACT and therefore independent of the order in which the list of dynamic-rigid particles happens to
// be filled (the scatter appends to it with an atomic): every term is rounded once to 64-bit fixed point
// (d.fx_scale = 2^S, S chosen on the host from the number of dynamic particles, their largest density and the domain
// size so that the sum cannot overflow) and integers add associatively.  cm and R are then bit-reproducible from run
// to run, which a checkpoint restart with dynamic bodies needs (the reference's serial run is deterministic too,
// sph_base.py:200-222).
typedef long long fx_t;
__device__ __forceinline__ fx_t to_fx(const DevView& d, double v) { return __double2ll_rn(v * d.fx_scale); }
__device__ __forceinline__ fx_t wave_sum(fx_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// block-wide sum of NV values per thread -> part[blockIdx][off + k]
template <int NV>
__device__ __forceinline__ void block_store_partials(const fx_t (&s)[NV], fx_t* __restrict__ part, int off) {
    __shared__ fx_t red[TPB / 64][NV];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const fx_t w = wave_sum(s[k]);
        if (lane == 0) red[wave][k] = w;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        fx_t t = 0;
#pragma unroll
        for (int w = 0; w < TPB / 64; ++w) t += red[w][threadIdx.x];
        part[(size_t)blockIdx.x * RIGID_PART + off + threadIdx.x] = t;
    }
}

// total of partial component k over nblk blocks, computed by every thread that asks (small nblk, L2-resident)
__device__ __forceinline__ void sum_partials(const DevView& d, const fx_t* __restrict__ part, int nblk, int off, int nv, double* out,
                                             double* s_tmp) {
    // wave 0 reduces: a lane takes whole ROWS (blocks), all components of a row loaded together -- component by component
    // the loads formed a chain of nv dependent round trips, which is what these tiny kernels' 14 us consisted of (r04h trace)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave == 0) {
        fx_t acc[RIGID_PART] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // (nv <= 16: the slab path's one-pass sums use all)
        for (int bIdx = lane; bIdx < nblk; bIdx += 64) {
            const fx_t* row = part + (size_t)bIdx * RIGID_PART + off;
#pragma unroll
            for (int k = 0; k < RIGID_PART; ++k)
                if (k < nv) acc[k] += row[k];
        }
#pragma unroll
        for (int k = 0; k < RIGID_PART; ++k)
            if (k < nv) {
                const fx_t t = wave_sum(acc[k]);
                if (lane == 0) s_tmp[k] = (double)t / d.fx_scale;
            }
    }
    __syncthreads();
    for (int k = 0; k < nv; ++k) out[k] = s_tmp[k];
    __syncthreads();
}

// sph_base.py:182-192 compute_com (mass = m_V0 * density): per-block partial sums
__global__ __launch_bounds__(TPB) void k_rigid_sum(DevView d, const int* __restrict__ list, int n, int object_id,
                                                   fx_t* __restrict__ part) {
    const int tix = blockIdx.x * TPB + threadIdx.x;
    fx_t s[4] = {0, 0, 0, 0};
    if (tix < n) {
        const int i = list[tix];
        const int fl = __float_as_int(d.vf[i].w);
        if (sph_is_dynamic_rigid(fl) && sph_flags_object(fl) == object_id) {
            float4 xm = d.xm[i];
            const float4 aux = d.aux[i];
            if (d.rigid_from_x0) {   // SPH_OPT_RIGID_SUMS_FROM_X0: the rest centre of a restarted body
                const int pid = __float_as_int(aux.w);
                xm.x = d.x0_cold[3 * pid]; xm.y = d.x0_cold[3 * pid + 1]; xm.z = d.x0_cold[3 * pid + 2];
            }
            const float mass = d.m_V0 * aux.y;
            s[0] = to_fx(d, mass); s[1] = to_fx(d, (double)(mass * xm.x)); s[2] = to_fx(d, (double)(mass * xm.y)); s[3] = to_fx(d, (double)(mass * xm.z));
        }
    }
    block_store_partials<4>(s, part, 0);
}

// sph_base.py:206-210: A = sum m (x - cm) (x_0 - cm_rest)^T, per-block partials
__global__ __launch_bounds__(TPB) void k_rigid_A(DevView d, const int* __restrict__ list, int n, int object_id,
                                                 fx_t* __restrict__ part, int nblk) {
    __shared__ double s_tmp[16];
    double tot[4];
    sum_partials(d, part, nblk, 0, 4, tot, s_tmp);
    const int tix = blockIdx.x * TPB + threadIdx.x;
    fx_t s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (tix < n) {
        const int i = list[tix];
        const int fl = __float_as_int(d.vf[i].w);
        if (sph_is_dynamic_rigid(fl) && sph_flags_object(fl) == object_id) {
            const float sum_m = (float)tot[0];  // f32 division like the reference's cm /= sum_m
            const float cm[3] = {(float)tot[1] / sum_m, (float)tot[2] / sum_m, (float)tot[3] / sum_m};
            const float4 xm = d.xm[i];
            const float4 aux = d.aux[i];
            const int pid = __float_as_int(aux.w);
            const float* rc = &d.rigid_rest_cm[3 * object_id];
            const float q[3] = {d.x0_cold[3 * pid] - rc[0], d.x0_cold[3 * pid + 1] - rc[1],
                                d.x0_cold[3 * pid + 2] - rc[2]};
            const float p[3] = {xm.x - cm[0], xm.y - cm[1], xm.z - cm[2]};
            const float w = d.m_V0 * aux.y;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) s[3 * a + b] = to_fx(d, (double)(w * (p[a] * q[b])));
        }
    }
    block_store_partials<9>(s, part, 4);
}

// Rotation factor of A = R S (ti.polar_decompose, sph_base.py:212: third-party
// Taichi SVD, det U = det V = +1 => closest proper rotation).  f64 Jacobi on
// A^T A; one lane.
__device__ void polar_rotation(const double A[3][3], float R_[9]) {
    double S[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            S[i][j] = 0;
            for (int k = 0; k < 3; ++k) S[i][j] += A[k][i] * A[k][j];
        }
    for (int sweep = 0; sweep < 60; ++sweep) {
        const double off = fabs(S[0][1]) + fabs(S[0][2]) + fabs(S[1][2]);
        if (off <= 1e-30 * (fabs(S[0][0]) + fabs(S[1][1]) + fabs(S[2][2])) + 1e-300) break;  // cyclic Jacobi: quadratic, ~6 sweeps
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (fabs(S[p][q]) < 1e-300) continue;
                const double theta = (S[q][q] - S[p][p]) / (2.0 * S[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < 3; ++k) {
                    const double skp = S[k][p], skq = S[k][q];
                    S[k][p] = c * skp - sn * skq; S[k][q] = sn * skp + c * skq;
                }
                for (int k = 0; k < 3; ++k) {
                    const double spk = S[p][k], sqk = S[q][k];
                    S[p][k] = c * spk - sn * sqk; S[q][k] = sn * spk + c * sqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - sn * vkq; V[k][q] = sn * vkp + c * vkq;
                }
            }
    }
    int idx[3] = {0, 1, 2};
    for (int i = 0; i < 2; ++i)
        for (int j = i + 1; j < 3; ++j)
            if (S[idx[j]][idx[j]] > S[idx[i]][idx[i]]) { const int t = idx[i]; idx[i] = idx[j]; idx[j] = t; }
    double Vs[3][3], U[3][3], sig0;
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) Vs[r][c] = V[r][idx[c]];
    { const double e = S[idx[0]][idx[0]]; sig0 = e > 0 ? sqrt(e) : 0.0; }
    const double detV = Vs[0][0] * (Vs[1][1] * Vs[2][2] - Vs[1][2] * Vs[2][1]) -
                        Vs[0][1] * (Vs[1][0] * Vs[2][2] - Vs[1][2] * Vs[2][0]) +
                        Vs[0][2] * (Vs[1][0] * Vs[2][1] - Vs[1][1] * Vs[2][0]);
    if (detV < 0) for (int r = 0; r < 3; ++r) Vs[r][2] = -Vs[r][2];
    if (sig0 <= 1e-300) {
        for (int i = 0; i < 9; ++i) R_[i] = 0.0f;
        return;
    }
    for (int c = 0; c < 2; ++c) {
        double u[3];
        for (int r = 0; r < 3; ++r) {
            u[r] = 0;
            for (int k = 0; k < 3; ++k) u[r] += A[r][k] * Vs[k][c];
        }
        double n = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        if (c == 1) {
            const double dd = u[0] * U[0][0] + u[1] * U[1][0] + u[2] * U[2][0];
            for (int r = 0; r < 3; ++r) u[r] -= dd * U[r][0];
            n = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        }
        if (n < 1e-300) {
            double a[3] = {1, 0, 0};
            if (c == 1) {
                if (fabs(U[0][0]) > 0.9) { a[0] = 0; a[1] = 1; }
                const double dd = a[0] * U[0][0] + a[1] * U[1][0] + a[2] * U[2][0];
                for (int r = 0; r < 3; ++r) u[r] = a[r] - dd * U[r][0];
            } else {
                for (int r = 0; r < 3; ++r) u[r] = a[r];
            }
            n = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        }
        for (int r = 0; r < 3; ++r) U[r][c] = u[r] / n;
    }
    U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
    U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
    U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double r = 0;
            for (int k = 0; k < 3; ++k) r += U[i][k] * Vs[j][k];
            R_[3 * i + j] = (float)r;
        }
}

// The same rotation by Higham's scaled Newton iteration X <- (g X + X^-T / g) / 2, g = (|X^-1|_F / |X|_F)^1/2: for a
// well-conditioned A with det A > 0 -- every body that is not degenerate -- the orthogonal polar factor IS the closest proper
// rotation the SVD form above constructs, and the iteration reaches it (to f64 rounding) in 5-7 steps of ~100 flops where
// the Jacobi form needs ~6 sweeps of three rotations with two square roots and two divisions each: 8 us -> 2 us on the one
// lane that computes it, a fifth of the whole rigid phase.  Anything else (det <= 0, nearly singular, no convergence)
// returns false and takes the Jacobi form.
__device__ bool polar_rotation_newton(const double A[3][3], float R_[9]) {
    double X[3][3], n2 = 0.0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) n2 += A[i][j] * A[i][j];
    if (!(n2 > 1e-280)) return false;
    const double s0 = 1.0 / sqrt(n2 * (1.0 / 3.0));  // singular values ~ 1 for a body of any size and mass
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) X[i][j] = A[i][j] * s0;
    for (int it = 0; it < 24; ++it) {
        double C[3][3];  // cofactor matrix: X^-T = C / det
        C[0][0] = X[1][1] * X[2][2] - X[1][2] * X[2][1]; C[0][1] = X[1][2] * X[2][0] - X[1][0] * X[2][2]; C[0][2] = X[1][0] * X[2][1] - X[1][1] * X[2][0];
        C[1][0] = X[0][2] * X[2][1] - X[0][1] * X[2][2]; C[1][1] = X[0][0] * X[2][2] - X[0][2] * X[2][0]; C[1][2] = X[0][1] * X[2][0] - X[0][0] * X[2][1];
        C[2][0] = X[0][1] * X[1][2] - X[0][2] * X[1][1]; C[2][1] = X[0][2] * X[1][0] - X[0][0] * X[1][2]; C[2][2] = X[0][0] * X[1][1] - X[0][1] * X[1][0];
        const double det = X[0][0] * C[0][0] + X[0][1] * C[0][1] + X[0][2] * C[0][2];
        if (!(det > 1e-6)) return false;  // (X is scaled: det ~ 1 for a rotation-dominated A; reflections and flat bodies leave here)
        double nx = 0.0, nc = 0.0;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) { nx += X[i][j] * X[i][j]; nc += C[i][j] * C[i][j]; }
        const double idet = 1.0 / det;
        const double g = sqrt(sqrt(nc) * idet / sqrt(nx));  // |X^-1|_F = |C|_F / det
        const double a = 0.5 * g, b = 0.5 * idet / g;
        double diff = 0.0;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                const double y = a * X[i][j] + b * C[i][j];
                diff += (y - X[i][j]) * (y - X[i][j]);
                X[i][j] = y;
            }
        if (diff <= 1e-30 * 3.0) {  // |X_k+1 - X_k|_F <= 1e-15 |R|_F: converged (quadratically: the step before was ~1e-8)
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) R_[3 * i + j] = (float)X[i][j];
            return true;
        }
    }
    return false;
}

// cm (and the polar rotation) from the summed partials; one lane per block computes them into LDS
// `fb` (block 0 only, may be null): counts the solves that left the Newton iteration for the Jacobi-SVD form -- degenerate bodies
// (flat: rank-2 A; a rod; reflected) -- so that a test can see the fallback fire (SphStats.polar_fallbacks).
__device__ __forceinline__ void rigid_cm_R(const double* tot, bool want_R, float* cmR /*[12] in LDS*/, int* fb = nullptr) {
    const float sum_m = (float)tot[0];  // f32 division like the reference's cm /= sum_m (0/0 = NaN for static bodies)
    cmR[0] = (float)tot[1] / sum_m; cmR[1] = (float)tot[2] / sum_m; cmR[2] = (float)tot[3] / sum_m;
    if (!want_R) return;
    double A[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) A[i][j] = (double)(float)tot[4 + 3 * i + j];
    float R[9];
    if (!polar_rotation_newton(A, R)) {
        polar_rotation(A, R);
        if (fb) atomicAdd(fb, 1);
    }
    bool all_small = true;
    for (int i = 0; i < 9; ++i)
        if (!(fabsf(R[i]) < 1e-6f)) all_small = false;
    if (all_small) {  // sph_base.py:214-215
        for (int i = 0; i < 9; ++i) R[i] = 0.0f;
        R[0] = R[4] = R[8] = 1.0f;
    }
    for (int i = 0; i < 9; ++i) cmR[3 + i] = R[i];
}

// mode 0: cm -> rigid_rest_cm[object_id] (sph_base.py:87-89) and out[0..2]; mode 2: cm -> out only
__global__ __launch_bounds__(64) void k_rigid_cm_only(DevView d, const fx_t* __restrict__ part, int nblk, int object_id,
                                                      int mode, float* __restrict__ out) {
    __shared__ double s_tmp[16];
    __shared__ float cmR[12];
    double tot[4];
    sum_partials(d, part, nblk, 0, 4, tot, s_tmp);
    if (threadIdx.x == 0) {
        rigid_cm_R(tot, false, cmR);
        for (int k = 0; k < 3; ++k) {
            out[k] = cmR[k];
            if (mode == 0) d.rigid_rest_cm[3 * object_id + k] = cmR[k];
        }
    }
}

// sph_base.py:212-221: R = polar(A) (+ identity fallback), x = cm + R (x_0 - cm_rest).  Every block derives cm and R
// from the partials itself (a 3x3 f64 Jacobi is cheaper than another launch); block 0 also publishes them.
__global__ __launch_bounds__(TPB) void k_rigid_apply(DevView d, const int* __restrict__ list, int n, int object_id,
                                                     const fx_t* __restrict__ part, int nblk, float* __restrict__ out) {
    __shared__ double s_tmp[16];
    __shared__ float cmR[12];
    double tot[13];
    sum_partials(d, part, nblk, 0, 13, tot, s_tmp);
    if (threadIdx.x == 0) {
        rigid_cm_R(tot, true, cmR, blockIdx.x == 0 ? d.polar_fb : nullptr);
        if (blockIdx.x == 0)
            for (int k = 0; k < 12; ++k) out[k] = cmR[k];
    }
    __syncthreads();
    const int tix = blockIdx.x * TPB + threadIdx.x;
    if (tix >= n) return;
    const int i = list[tix];
    const int fl = __float_as_int(d.vf[i].w);
    if (!(sph_is_dynamic_rigid(fl) && sph_flags_object(fl) == object_id)) return;
    const int pid = __float_as_int(d.aux[i].w);
    const float* rc = &d.rigid_rest_cm[3 * object_id];
    const float q[3] = {d.x0_cold[3 * pid] - rc[0], d.x0_cold[3 * pid + 1] - rc[1], d.x0_cold[3 * pid + 2] - rc[2]};
    float4 xm = d.xm[i];
    float x[3] = {xm.x, xm.y, xm.z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float goal = cmR[a] + (cmR[3 + 3 * a] * q[0] + cmR[3 + 3 * a + 1] * q[1] + cmR[3 + 3 * a + 2] * q[2]);
        const float corr = (goal - x[a]) * 1.0f;
        x[a] += corr;
    }
    xm.x = x[0]; xm.y = x[1]; xm.z = x[2];
    d.xm[i] = xm;
}

// ---- solve_rigid_body() for ALL dynamic bodies in three launches (sph_base.py:247-260) ----
// The reference runs, body by body, solve_constraints(body) and then enforce_boundary_3D(solid) over EVERY dynamic solid
// particle.  The wall pass is a pure function of one particle's (x, v) -- but not idempotent: it clamps the position once,
// yet a particle sitting exactly on a LOW wall (x == padding satisfies `pos <= padding`, sph_base.py:158) has its velocity
// reflected again by every later pass.  So the passes cannot be merged into one; they can be REPLAYED per particle: a
// particle of the b-th body (0-based) has seen b passes when its body is solved and sees n - b afterwards; any other
// dynamic solid sees all n.  The kernels below apply those passes in registers, so the three phases (sums, A, apply)
// run once over the list of dynamic rigid particles with per-body rows of partial sums, instead of 4 launches per body,
// and give bit for bit what the sequence gives (the sums are exact fixed-point integers, the per-particle terms the
// same expressions): tests/test_gpu_parity.py::test_batched_rigid_solve_equals_the_body_by_body_sequence.
#define SPH_MAX_BATCH_BODIES 16
struct BodyIds { int n; int id[SPH_MAX_BATCH_BODIES]; };

__device__ __forceinline__ int body_slot(const BodyIds& b, int object_id) {
    int s = -1;
#pragma unroll
    for (int k = 0; k < SPH_MAX_BATCH_BODIES; ++k)
        if (k < b.n && b.id[k] == object_id) s = k;
    return s;
}
__device__ __forceinline__ void wall_passes(const DevView& d, const float hi[3], float4& xm, float4& vf, int passes) {
    for (int p = 0; p < passes; ++p) wall_collide(d, hi, xm, vf);
}

// block-wide sum of NV values per thread -> part[blockIdx][off + k] of one body's row block; callable in a loop
template <int NV>
__device__ __forceinline__ void block_store_partials_at(const fx_t (&s)[NV], fx_t* __restrict__ part_body, int off, fx_t (*red)[16]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const fx_t w = wave_sum(s[k]);
        if (lane == 0) red[wave][k] = w;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        fx_t t = 0;
#pragma unroll
        for (int w = 0; w < TPB / 64; ++w) t += red[w][threadIdx.x];
        part_body[(size_t)blockIdx.x * RIGID_PART + off + threadIdx.x] = t;
    }
    __syncthreads();
}

// totals of components [0, nv) of every body's partial rows -> s_tot[body][k] (wave w takes bodies w, w + 4, ...)
__device__ __forceinline__ void sum_partials_all(const DevView& d, const fx_t* __restrict__ part, int nblk, int nbodies, int nv,
                                                 double (*s_tot)[16]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int b = wave; b < nbodies; b += TPB / 64) {
        const fx_t* pb = part + (size_t)b * nblk * RIGID_PART;
        fx_t acc[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // (a lane takes whole rows: all loads of a row in flight together)
        for (int bIdx = lane; bIdx < nblk; bIdx += 64) {
            const fx_t* row = pb + (size_t)bIdx * RIGID_PART;
#pragma unroll
            for (int k = 0; k < 13; ++k)
                if (k < nv) acc[k] += row[k];
        }
#pragma unroll
        for (int k = 0; k < 13; ++k)
            if (k < nv) {
                const fx_t t = wave_sum(acc[k]);
                if (lane == 0) s_tot[b][k] = (double)t / d.fx_scale;
            }
    }
    __syncthreads();
}

// ADVECT: the symplectic-Euler update of the dynamic rigid particles (WCSPH.py:143-149) rides in this kernel -- it is
// what would otherwise be a launch of its own (k_advect_list) right before it, over the same list
template <bool ADVECT>
__device__ __forceinline__ void rigid_phase_sum(const DevView& d, const WallHi& hi, const int* __restrict__ list, int n, const BodyIds& ids,
                                                fx_t* __restrict__ part, int nblk) {
    __shared__ fx_t red[TPB / 64][16];
    const int tix = blockIdx.x * TPB + threadIdx.x;
    int slot = -1;
    fx_t v[4] = {0, 0, 0, 0};
    if (tix < n) {
        const int i = list[tix];
        float4 vf = d.vf[i];
        const int fl = __float_as_int(vf.w);
        float4 xm = d.xm[i];
        if (ADVECT && sph_is_dynamic_rigid(fl)) {
            advect_one<false>(d, hi.v, xm, vf, d.acc[i]);
            d.xm[i] = xm;
            d.vf[i] = vf;
        }
        if (sph_is_dynamic_rigid(fl) && (slot = body_slot(ids, sph_flags_object(fl))) >= 0) {
            wall_passes(d, hi.v, xm, vf, slot);
            const float mass = d.m_V0 * d.aux[i].y;
            v[0] = to_fx(d, mass); v[1] = to_fx(d, (double)(mass * xm.x)); v[2] = to_fx(d, (double)(mass * xm.y)); v[3] = to_fx(d, (double)(mass * xm.z));
        }
    }
    for (int b = 0; b < ids.n; ++b) {
        const fx_t s[4] = {slot == b ? v[0] : 0, slot == b ? v[1] : 0, slot == b ? v[2] : 0, slot == b ? v[3] : 0};
        block_store_partials_at<4>(s, part + (size_t)b * nblk * RIGID_PART, 0, red);
    }
}
template <bool ADVECT>
__global__ __launch_bounds__(TPB) void k_rigid_sum_all(DevView d, WallHi hi, const int* __restrict__ list, int n, BodyIds ids,
                                                       fx_t* __restrict__ part, int nblk) {
    rigid_phase_sum<ADVECT>(d, hi, list, n, ids, part, nblk);
}

__device__ __forceinline__ void rigid_phase_A(const DevView& d, const WallHi& hi, const int* __restrict__ list, int n, const BodyIds& ids,
                                              fx_t* __restrict__ part, int nblk) {
    __shared__ fx_t red[TPB / 64][16];
    __shared__ double s_tot[SPH_MAX_BATCH_BODIES][16];
    sum_partials_all(d, part, nblk, ids.n, 4, s_tot);
    const int tix = blockIdx.x * TPB + threadIdx.x;
    int slot = -1;
    fx_t v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (tix < n) {
        const int i = list[tix];
        float4 vf = d.vf[i];
        const int fl = __float_as_int(vf.w);
        const int object_id = sph_flags_object(fl);
        if (sph_is_dynamic_rigid(fl) && (slot = body_slot(ids, object_id)) >= 0) {
            float4 xm = d.xm[i];
            wall_passes(d, hi.v, xm, vf, slot);
            const double* tot = s_tot[slot];
            const float sum_m = (float)tot[0];  // f32 division like the reference's cm /= sum_m
            const float cm[3] = {(float)tot[1] / sum_m, (float)tot[2] / sum_m, (float)tot[3] / sum_m};
            const float4 aux = d.aux[i];
            const int pid = __float_as_int(aux.w);
            const float* rc = &d.rigid_rest_cm[3 * object_id];
            const float q[3] = {d.x0_cold[3 * pid] - rc[0], d.x0_cold[3 * pid + 1] - rc[1], d.x0_cold[3 * pid + 2] - rc[2]};
            const float p[3] = {xm.x - cm[0], xm.y - cm[1], xm.z - cm[2]};
            const float w = d.m_V0 * aux.y;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c) v[3 * a + c] = to_fx(d, (double)(w * (p[a] * q[c])));
        }
    }
    for (int b = 0; b < ids.n; ++b) {
        fx_t s[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) s[k] = slot == b ? v[k] : 0;
        block_store_partials_at<9>(s, part + (size_t)b * nblk * RIGID_PART, 4, red);
    }
}
__global__ __launch_bounds__(TPB) void k_rigid_A_all(DevView d, WallHi hi, const int* __restrict__ list, int n, BodyIds ids,
                                                     fx_t* __restrict__ part, int nblk) {
    rigid_phase_A(d, hi, list, n, ids, part, nblk);
}

__device__ __forceinline__ void rigid_phase_apply(const DevView& d, const WallHi& hi, const int* __restrict__ list, int n, const BodyIds& ids,
                                                  const fx_t* __restrict__ part, int nblk, float* __restrict__ out) {
    __shared__ double s_tot[SPH_MAX_BATCH_BODIES][16];
    __shared__ float cmR[SPH_MAX_BATCH_BODIES][12];
    sum_partials_all(d, part, nblk, ids.n, 13, s_tot);
    if ((int)threadIdx.x < ids.n) {
        rigid_cm_R(s_tot[threadIdx.x], true, cmR[threadIdx.x], blockIdx.x == 0 ? d.polar_fb : nullptr);
        if (blockIdx.x == 0 && (int)threadIdx.x == ids.n - 1)  // what the body-by-body sequence leaves behind: the last body's
            for (int k = 0; k < 12; ++k) out[k] = cmR[threadIdx.x][k];
    }
    __syncthreads();
    const int tix = blockIdx.x * TPB + threadIdx.x;
    if (tix >= n) return;
    const int i = list[tix];
    float4 vf = d.vf[i];
    const int fl = __float_as_int(vf.w);
    if (!sph_is_dynamic_rigid(fl)) return;
    const int object_id = sph_flags_object(fl);
    const int slot = body_slot(ids, object_id);
    float4 xm = d.xm[i];
    if (slot >= 0) {
        wall_passes(d, hi.v, xm, vf, slot);
        const int pid = __float_as_int(d.aux[i].w);
        const float* rc = &d.rigid_rest_cm[3 * object_id];
        const float q[3] = {d.x0_cold[3 * pid] - rc[0], d.x0_cold[3 * pid + 1] - rc[1], d.x0_cold[3 * pid + 2] - rc[2]};
        const float* cr = cmR[slot];
        float x[3] = {xm.x, xm.y, xm.z};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float goal = cr[a] + (cr[3 + 3 * a] * q[0] + cr[3 + 3 * a + 1] * q[1] + cr[3 + 3 * a + 2] * q[2]);
            const float corr = (goal - x[a]) * 1.0f;
            x[a] += corr;
        }
        xm.x = x[0]; xm.y = x[1]; xm.z = x[2];
        wall_passes(d, hi.v, xm, vf, ids.n - slot);
    } else {
        wall_passes(d, hi.v, xm, vf, ids.n);  // a dynamic solid that is no shape-matched body (a dynamic RigidBlock)
    }
    d.xm[i] = xm;
    d.vf[i] = vf;
}
__global__ __launch_bounds__(TPB) void k_rigid_apply_all(DevView d, WallHi hi, const int* __restrict__ list, int n, BodyIds ids,
                                                         const fx_t* __restrict__ part, int nblk, float* __restrict__ out) {
    rigid_phase_apply(d, hi, list, n, ids, part, nblk, out);
}

// (r04: the three phases in ONE launch behind grid-wide barriers -- k_rigid_all_fused, commit 98934a4 -- were bit-identical
// and slower, 0.0418 vs 0.0390 ms: an agent-scope release on this multi-XCD part writes the L2 back, which costs what a
// kernel boundary costs.  profiles/r04f_with_bodies_one_launch_vs_three.json)

// ---- the same solve with the sums split over slabs (include/sph_hip.h: sph_rigid_partial_sums) ----
// per-block partials of the 16 one-pass sums over the dynamic-rigid particles of `object_id` with index in [first,last)
__global__ __launch_bounds__(TPB) void k_rigid_sum16(DevView d, const int* __restrict__ list, int n, int object_id,
                                                     int first, int last, fx_t* __restrict__ part) {
    const int tix = blockIdx.x * TPB + threadIdx.x;
    fx_t s[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (tix < n) {
        const int i = list[tix];
        const int fl = __float_as_int(d.vf[i].w);
        if (i >= first && i < last && sph_is_dynamic_rigid(fl) && sph_flags_object(fl) == object_id) {
            const float4 xm = d.xm[i];
            const float4 aux = d.aux[i];
            const int pid = __float_as_int(aux.w);
            const float* rc = &d.rigid_rest_cm[3 * object_id];
            const float mass = d.m_V0 * aux.y;
            // (SPH_OPT_RIGID_SUMS_FROM_X0: x_0 stands in for x -- only sums 0..3, the rest centre of a restart, are used then)
            const float x[3] = {d.rigid_from_x0 ? d.x0_cold[3 * pid] : xm.x, d.rigid_from_x0 ? d.x0_cold[3 * pid + 1] : xm.y,
                                d.rigid_from_x0 ? d.x0_cold[3 * pid + 2] : xm.z};
            const float q[3] = {d.x0_cold[3 * pid] - rc[0], d.x0_cold[3 * pid + 1] - rc[1],
                                d.x0_cold[3 * pid + 2] - rc[2]};
            s[0] = to_fx(d, mass);
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                s[1 + a] = to_fx(d, (double)(mass * x[a]));  // same f32 product as k_rigid_sum
                s[4 + a] = to_fx(d, (double)mass * (double)q[a]);
#pragma unroll
                for (int b = 0; b < 3; ++b) s[7 + 3 * a + b] = to_fx(d, (double)mass * (double)x[a] * (double)q[b]);
            }
        }
    }
    block_store_partials<16>(s, part, 0);
}

__global__ __launch_bounds__(64) void k_rigid_total16(DevView d, const fx_t* __restrict__ part, int nblk, double* __restrict__ out) {
    __shared__ double s_tmp[16];
    double tot[16];
    sum_partials(d, part, nblk, 0, 16, tot, s_tmp);
    if (threadIdx.x < 16) out[threadIdx.x] = nblk > 0 ? s_tmp[threadIdx.x] : 0.0;
}

// mode 0: rest cm; mode 1: cm + polar rotation + goal positions for every local particle of the object
__global__ __launch_bounds__(TPB) void k_rigid_apply_sums(DevView d, const int* __restrict__ list, int n, int object_id,
                                                          const double* __restrict__ sums, int mode,
                                                          float* __restrict__ out) {
    __shared__ float cmR[12];
    if (threadIdx.x == 0) {
        double tot[13];
        for (int k = 0; k < 4; ++k) tot[k] = sums[k];
        const float sum_m = (float)tot[0];
        const double cm[3] = {(double)((float)tot[1] / sum_m), (double)((float)tot[2] / sum_m),
                              (double)((float)tot[3] / sum_m)};
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) tot[4 + 3 * a + b] = sums[7 + 3 * a + b] - cm[a] * sums[4 + b];
        rigid_cm_R(tot, mode == 1, cmR, blockIdx.x == 0 ? d.polar_fb : nullptr);
        if (blockIdx.x == 0) {
            for (int k = 0; k < (mode == 1 ? 12 : 3); ++k) out[k] = cmR[k];
            if (mode == 0)
                for (int k = 0; k < 3; ++k) d.rigid_rest_cm[3 * object_id + k] = cmR[k];
        }
    }
    __syncthreads();
    if (mode != 1) return;
    const int tix = blockIdx.x * TPB + threadIdx.x;
    if (tix >= n) return;
    const int i = list[tix];
    const int fl = __float_as_int(d.vf[i].w);
    if (!(sph_is_dynamic_rigid(fl) && sph_flags_object(fl) == object_id)) return;
    const int pid = __float_as_int(d.aux[i].w);
    const float* rc = &d.rigid_rest_cm[3 * object_id];
    const float q[3] = {d.x0_cold[3 * pid] - rc[0], d.x0_cold[3 * pid + 1] - rc[1], d.x0_cold[3 * pid + 2] - rc[2]};
    float4 xm = d.xm[i];
    float x[3] = {xm.x, xm.y, xm.z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float goal = cmR[a] + (cmR[3 + 3 * a] * q[0] + cmR[3 + 3 * a + 1] * q[1] + cmR[3 + 3 * a + 2] * q[2]);
        x[a] += (goal - x[a]) * 1.0f;
    }
    xm.x = x[0]; xm.y = x[1]; xm.z = x[2];
    d.xm[i] = xm;
}

__global__ __launch_bounds__(TPB) void k_scatter_rest(float* __restrict__ x0_cold, const int* __restrict__ pid,
                                                      const float* __restrict__ x0, int n, int cap) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const int p = pid[i];
    if (p < 0 || p >= cap) return;
    for (int k = 0; k < 3; ++k) x0_cold[3 * p + k] = x0[3 * i + k];
}

// list of dynamic rigid particles in the CURRENT order (used before the first sort)
__global__ __launch_bounds__(TPB) void k_build_dyn_list(DevView d, int* __restrict__ list, int* __restrict__ count) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= d.N) return;
    if (sph_is_dynamic_rigid(__float_as_int(d.vf[i].w))) {
        list[atomicAdd(count, 1)] = i;
        atomicMax(reinterpret_cast<unsigned*>(count) + 1, __float_as_uint(fabsf(d.aux[i].y)));  // largest body density (bits of a non-negative float order like integers)
    }
}

// ---- field insert / extract (sph_upload / sph_download) -------------------
__global__ __launch_bounds__(TPB) void k_extract(DevView d, int field, const int* __restrict__ color_cold,
                                                 void* __restrict__ out) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= d.N) return;
    float* of = reinterpret_cast<float*>(out);
    int* oi = reinterpret_cast<int*>(out);
    const int fl = __float_as_int(d.vf[i].w);
    const int pid = __float_as_int(d.aux[i].w);
    switch (field) {
        case SPH_F_OBJECT_ID: oi[i] = sph_flags_object(fl); break;
        case SPH_F_X: { const float4 v = d.xm[i]; of[3 * i] = v.x; of[3 * i + 1] = v.y; of[3 * i + 2] = v.z; } break;
        case SPH_F_X_0: for (int k = 0; k < 3; ++k) of[3 * i + k] = d.x0_cold[3 * pid + k]; break;
        case SPH_F_V: { const float4 v = d.vf[i]; of[3 * i] = v.x; of[3 * i + 1] = v.y; of[3 * i + 2] = v.z; } break;
        case SPH_F_ACCELERATION: { const float4 v = d.acc[i]; of[3 * i] = v.x; of[3 * i + 1] = v.y; of[3 * i + 2] = v.z; } break;
        case SPH_F_M_V: of[i] = d.xm[i].w; break;
        case SPH_F_M: of[i] = d.aux[i].x; break;
        case SPH_F_DENSITY: of[i] = d.aux[i].y; break;
        case SPH_F_PRESSURE: of[i] = d.aux[i].z; break;
        case SPH_F_MATERIAL: oi[i] = sph_flags_material(fl); break;
        case SPH_F_COLOR: for (int k = 0; k < 3; ++k) oi[3 * i + k] = color_cold[3 * pid + k]; break;
        case SPH_F_IS_DYNAMIC: oi[i] = sph_flags_dynamic(fl); break;
        case SPH_F_GRID_IDS: oi[i] = d.key[i]; break;
        case SPH_F_PID: oi[i] = pid; break;
        case SPH_F_DFSPH_FACTOR: of[i] = d.eos[i].x; break;
        case SPH_F_DENSITY_ADV: of[i] = d.eos[i].y; break;
        default: break;
    }
}

__global__ __launch_bounds__(TPB) void k_insert(DevView d, int field, float* __restrict__ x0_cold,
                                                int* __restrict__ color_cold, const void* __restrict__ in) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= d.N) return;
    const float* f = reinterpret_cast<const float*>(in);
    const int* n = reinterpret_cast<const int*>(in);
    float* xm = reinterpret_cast<float*>(&d.xm[i]);
    float* vf = reinterpret_cast<float*>(&d.vf[i]);
    float* aux = reinterpret_cast<float*>(&d.aux[i]);
    float* acc = reinterpret_cast<float*>(&d.acc[i]);
    const int pid = __float_as_int(aux[3]);
    int fl = __float_as_int(vf[3]);
    switch (field) {
        case SPH_F_OBJECT_ID: fl = (fl & 0x1FF) | (n[i] << 9); vf[3] = __int_as_float(fl); break;
        case SPH_F_X: xm[0] = f[3 * i]; xm[1] = f[3 * i + 1]; xm[2] = f[3 * i + 2]; break;
        case SPH_F_X_0: for (int k = 0; k < 3; ++k) x0_cold[3 * pid + k] = f[3 * i + k]; break;
        case SPH_F_V: vf[0] = f[3 * i]; vf[1] = f[3 * i + 1]; vf[2] = f[3 * i + 2]; break;
        case SPH_F_ACCELERATION: acc[0] = f[3 * i]; acc[1] = f[3 * i + 1]; acc[2] = f[3 * i + 2]; acc[3] = 0.f; break;
        case SPH_F_M_V: xm[3] = f[i]; break;
        case SPH_F_M: aux[0] = f[i]; break;
        case SPH_F_DENSITY: aux[1] = f[i]; break;
        case SPH_F_PRESSURE: aux[2] = f[i]; break;
        case SPH_F_MATERIAL: fl = (fl & ~0xFF) | (n[i] & 0xFF); vf[3] = __int_as_float(fl); break;
        case SPH_F_COLOR: for (int k = 0; k < 3; ++k) color_cold[3 * pid + k] = n[3 * i + k]; break;
        case SPH_F_IS_DYNAMIC: fl = (fl & ~0x100) | (n[i] ? 0x100 : 0); vf[3] = __int_as_float(fl); break;
        case SPH_F_PID: aux[3] = __int_as_float(n[i]); break;
        case SPH_F_DFSPH_FACTOR: reinterpret_cast<float*>(&d.eos[i])[0] = f[i]; break;
        case SPH_F_DENSITY_ADV: reinterpret_cast<float*>(&d.eos[i])[1] = f[i]; break;
        default: break;
    }
}

__global__ __launch_bounds__(TPB) void k_init_pid(float4* aux0, float4* aux1, int cap) {
    const int i = blockIdx.x * TPB + threadIdx.x;
    if (i >= cap) return;
    aux0[i] = make_float4(0.f, 0.f, 0.f, __int_as_float(i));
    aux1[i] = make_float4(0.f, 0.f, 0.f, __int_as_float(i));
}

// ---------------------------------------------------------------------------
static WallHi wall_hi(const SphContext* c) {
    WallHi w;
    for (int k = 0; k < 3; ++k) w.v[k] = c->p.wall_hi[k];
    return w;
}

int sphk_advect(SphContext* c, bool fused_fluid_walls) {
    if (c->N <= 0) return 0;
    DevView d = sph_view(c);
    const int nb = (c->N + TPB - 1) / TPB;
    if (fused_fluid_walls) hipLaunchKernelGGL(k_advect<true>, dim3(nb), dim3(TPB), 0, c->stream, d, wall_hi(c));
    else hipLaunchKernelGGL(k_advect<false>, dim3(nb), dim3(TPB), 0, c->stream, d, wall_hi(c));
    SPH_LAUNCH_CHECK(c);
    sph_invalidate_lists(c);
    return 0;
}

int sphk_advect_range(SphContext* c, int first, int count) {
    sph_invalidate_lists(c);
    if (count <= 0) return 0;
    DevView d = sph_view(c);
    hipLaunchKernelGGL(k_advect_range, dim3((count + TPB - 1) / TPB), dim3(TPB), 0, c->stream, d, wall_hi(c), first, count);
    SPH_LAUNCH_CHECK(c);
    return 0;
}

int sphk_advect_dyn_list(SphContext* c) {
    sph_invalidate_lists(c);
    if (c->n_dyn_host <= 0) return 0;
    DevView d = sph_view(c);
    hipLaunchKernelGGL(k_advect_list, dim3((c->n_dyn_host + TPB - 1) / TPB), dim3(TPB), 0, c->stream, d, wall_hi(c), c->dyn_list,
                       c->n_dyn_host);
    SPH_LAUNCH_CHECK(c);
    return 0;
}

int sphk_pack_advected(SphContext* c, int first, int count, void* dst) {
    if (count <= 0) return 0;
    DevView d = sph_view(c);
    hipLaunchKernelGGL(k_pack_advected, dim3((count + TPB - 1) / TPB), dim3(TPB), 0, sph_stream(c), d, wall_hi(c), first, count,
                       (float4*)dst);
    SPH_LAUNCH_CHECK(c);
    return 0;
}

int sphk_enforce_boundary(SphContext* c, int particle_type) {
    sph_invalidate_lists(c);
    if (c->N <= 0) return 0;
    DevView d = sph_view(c);
    if (particle_type == SPH_MATERIAL_SOLID) {
        // only dynamic solids can be hit: run over the dynamic-rigid list
        if (c->n_dyn_host <= 0) return 0;
        hipLaunchKernelGGL(k_enforce_boundary, dim3((c->n_dyn_host + TPB - 1) / TPB), dim3(TPB), 0, c->stream, d,
                           wall_hi(c), particle_type, c->dyn_list, c->n_dyn_host);
    } else {
        hipLaunchKernelGGL(k_enforce_boundary, dim3((c->N + TPB - 1) / TPB), dim3(TPB), 0, c->stream, d, wall_hi(c),
                           particle_type, (const int*)nullptr, c->N);
    }
    SPH_LAUNCH_CHECK(c);
    return 0;
}

// cm of object -> rigid_R[0..2] (and rigid_rest_cm[object] when to_rest)
int sphk_rigid_com(SphContext* c, int object_id, bool to_rest) {
    DevView d = sph_view(c);
    d.rigid_from_x0 = (to_rest && c->opt_rigid_x0) ? 1 : 0;
    const int n = c->n_dyn_host > 0 ? c->n_dyn_host : 0;
    const int nb = n > 0 ? (n + TPB - 1) / TPB : 1;
    if (nb > c->rigid_part_blocks) return sph_fail(c, SPH_E_NOMEM, "rigid partial-sum buffer too small");
    // n == 0 (no dynamic particle, e.g. the rest cm of a static body): one block of zeros -> 0/0 = NaN like the reference
    hipLaunchKernelGGL(k_rigid_sum, dim3(nb), dim3(TPB), 0, c->stream, d, c->dyn_list, n, object_id, (fx_t*)c->rigid_part);
    SPH_LAUNCH_CHECK(c);
    hipLaunchKernelGGL(k_rigid_cm_only, dim3(1), dim3(64), 0, c->stream, d, (const fx_t*)c->rigid_part, nb, object_id, to_rest ? 0 : 2,
                       c->rigid_R);
    SPH_LAUNCH_CHECK(c);
    return 0;
}

// solve_constraints (sph_base.py:200-222) without any host round trip: 3 launches, no atomics
int sphk_rigid_solve(SphContext* c, int object_id) {
    sph_invalidate_lists(c);
    if (c->n_dyn_host <= 0) return 0;
    DevView d = sph_view(c);
    const int n = c->n_dyn_host, nb = (n + TPB - 1) / TPB;
    if (nb > c->rigid_part_blocks) return sph_fail(c, SPH_E_NOMEM, "rigid partial-sum buffer too small");
    hipLaunchKernelGGL(k_rigid_sum, dim3(nb), dim3(TPB), 0, c->stream, d, c->dyn_list, n, object_id, (fx_t*)c->rigid_part);
    SPH_LAUNCH_CHECK(c);
    hipLaunchKernelGGL(k_rigid_A, dim3(nb), dim3(TPB), 0, c->stream, d, c->dyn_list, n, object_id, (fx_t*)c->rigid_part, nb);
    SPH_LAUNCH_CHECK(c);
    hipLaunchKernelGGL(k_rigid_apply, dim3(nb), dim3(TPB), 0, c->stream, d, c->dyn_list, n, object_id, (const fx_t*)c->rigid_part, nb,
                       c->rigid_R);
    SPH_LAUNCH_CHECK(c);
    return 0;
}

// solve_rigid_body() (sph_base.py:247-260) for the dynamic bodies `ids` in the reference's order: three launches for all
// of them when they fit the batch (<= 16 bodies, per-body rows of partials), else body by body.  advect_first: the
// advect of the dynamic rigid particles (what sphk_advect_dyn_list does) happens here too, inside the first kernel
int sphk_rigid_solve_all(SphContext* c, const int* ids, int n_ids, bool advect_first) {
    if (c->n_dyn_host <= 0 || n_ids <= 0) return advect_first ? sphk_advect_dyn_list(c) : 0;
    const int n = c->n_dyn_host, nb = (n + TPB - 1) / TPB;
    if (!c->opt_rigid_batch || n_ids > SPH_MAX_BATCH_BODIES || (long long)nb * n_ids > c->rigid_part_blocks) {
        if (advect_first) { int rc = sphk_advect_dyn_list(c); if (rc) return rc; }
        for (int k = 0; k < n_ids; ++k) {
            int rc = sphk_rigid_solve(c, ids[k]);
            rc = rc ? rc : sphk_enforce_boundary(c, SPH_MATERIAL_SOLID);
            if (rc) return rc;
        }
        return 0;
    }
    sph_invalidate_lists(c);
    DevView d = sph_view(c);
    BodyIds b;
    b.n = n_ids;
    for (int k = 0; k < SPH_MAX_BATCH_BODIES; ++k) b.id[k] = k < n_ids ? ids[k] : -1;
    if (advect_first) hipLaunchKernelGGL(k_rigid_sum_all<true>, dim3(nb), dim3(TPB), 0, c->stream, d, wall_hi(c), c->dyn_list, n, b, (fx_t*)c->rigid_part, nb);
    else hipLaunchKernelGGL(k_rigid_sum_all<false>, dim3(nb), dim3(TPB), 0, c->stream, d, wall_hi(c), c->dyn_list, n, b, (fx_t*)c->rigid_part, nb);
    SPH_LAUNCH_CHECK(c);
    hipLaunchKernelGGL(k_rigid_A_all, dim3(nb), dim3(TPB), 0, c->stream, d, wall_hi(c), c->dyn_list, n, b, (fx_t*)c->rigid_part, nb);
    SPH_LAUNCH_CHECK(c);
    hipLaunchKernelGGL(k_rigid_apply_all, dim3(nb), dim3(TPB), 0, c->stream, d, wall_hi(c), c->dyn_list, n, b, (const fx_t*)c->rigid_part, nb,
                       c->rigid_R);
    SPH_LAUNCH_CHECK(c);
    return 0;
}

int sphk_extract(SphContext* c, int field, void* dst) {
    if (c->N <= 0) return 0;
    DevView d = sph_view(c);
    hipLaunchKernelGGL(k_extract, dim3((c->N + TPB - 1) / TPB), dim3(TPB), 0, c->stream, d, field, c->color_cold, dst);
    SPH_LAUNCH_CHECK(c);
    return 0;
}

int sphk_insert(SphContext* c, int field, const void* src) {
    if (field == SPH_F_X || field == SPH_F_MATERIAL || field == SPH_F_IS_DYNAMIC) sph_invalidate_lists(c);
    if (c->N <= 0) return 0;
    DevView d = sph_view(c);
    hipLaunchKernelGGL(k_insert, dim3((c->N + TPB - 1) / TPB), dim3(TPB), 0, c->stream, d, field, c->x0_cold,
                       c->color_cold, src);
    SPH_LAUNCH_CHECK(c);
    return 0;
}

int sphk_init_pid(SphContext* c) {
    hipLaunchKernelGGL(k_init_pid, dim3((c->cap + TPB - 1) / TPB), dim3(TPB), 0, c->stream, c->aux[0], c->aux[1],
                       c->cap);
    SPH_LAUNCH_CHECK(c);
    return 0;
}

int sphk_build_dyn_list(SphContext* c) {
    SPH_HIP(c, hipMemsetAsync(c->dyn_count, 0, 2 * sizeof(int), c->stream));
    if (c->N > 0) {
        DevView d = sph_view(c);
        hipLaunchKernelGGL(k_build_dyn_list, dim3((c->N + TPB - 1) / TPB), dim3(TPB), 0, c->stream, d, c->dyn_list,
                           c->dyn_count);
        SPH_LAUNCH_CHECK(c);
    }
    return 0;
}

int sphk_rigid_partial16(SphContext* c, int object_id, int first, int count, double* out) {
    DevView d = sph_view(c);
    d.rigid_from_x0 = c->opt_rigid_x0;   // (the host sets the option around the mode-0 sums of a restart only)
    const int n = c->n_dyn_host;
    const int nb = n > 0 ? (n + TPB - 1) / TPB : 0;
    if (nb > c->rigid_part_blocks) return sph_fail(c, SPH_E_NOMEM, "rigid partial-sum buffer too small");
    if (nb > 0) {
        hipLaunchKernelGGL(k_rigid_sum16, dim3(nb), dim3(TPB), 0, c->stream, d, c->dyn_list, n, object_id, first,
                           first + count, (fx_t*)c->rigid_part);
        SPH_LAUNCH_CHECK(c);
    }
    hipLaunchKernelGGL(k_rigid_total16, dim3(1), dim3(64), 0, c->stream, d, (const fx_t*)c->rigid_part, nb, out);
    SPH_LAUNCH_CHECK(c);
    return 0;
}

int sphk_rigid_apply16(SphContext* c, int object_id, const double* sums, int mode) {
    sph_invalidate_lists(c);
    DevView d = sph_view(c);
    const int n = c->n_dyn_host;
    const int nb = (mode == 1 && n > 0) ? (n + TPB - 1) / TPB : 1;
    hipLaunchKernelGGL(k_rigid_apply_sums, dim3(nb), dim3(TPB), 0, c->stream, d, c->dyn_list, n, object_id, sums, mode,
                       c->rigid_R);
    SPH_LAUNCH_CHECK(c);
    return 0;
}

int sphk_scatter_rest(SphContext* c, const int* pid_dev, const float* x0_dev, int n) {
    hipLaunchKernelGGL(k_scatter_rest, dim3((n + TPB - 1) / TPB), dim3(TPB), 0, c->stream, c->x0_cold, pid_dev, x0_dev, n,
                       c->cold_cap);
    SPH_LAUNCH_CHECK(c);
    return 0;
}

// ---- DFSPH element-wise launchers ----
int sphk_df_predict_velocity(SphContext* c) {
    if (c->N <= 0) return 0;
    DevView d = sph_view(c);
    hipLaunchKernelGGL(k_df_predict_velocity, dim3((c->N + TPB - 1) / TPB), dim3(TPB), 0, c->stream, d);
    SPH_LAUNCH_CHECK(c);
    return 0;
}

int sphk_df_advect(SphContext* c, bool fused_fluid_walls) {
    if (c->N <= 0) return 0;
    DevView d = sph_view(c);
    const int nb = (c->N + TPB - 1) / TPB;
    if (fused_fluid_walls) hipLaunchKernelGGL(k_df_advect<true>, dim3(nb), dim3(TPB), 0, c->stream, d, wall_hi(c));
    else hipLaunchKernelGGL(k_df_advect<false>, dim3(nb), dim3(TPB), 0, c->stream, d, wall_hi(c));
    SPH_LAUNCH_CHECK(c);
    sph_invalidate_lists(c);
    return 0;
}

int sphk_df_scale_factor(SphContext* c, float s) {
    if (c->N <= 0) return 0;
    DevView d = sph_view(c);
    hipLaunchKernelGGL(k_df_scale_factor, dim3((c->N + TPB - 1) / TPB), dim3(TPB), 0, c->stream, d, s);
    SPH_LAUNCH_CHECK(c);
    c->k_kind = 0;  // k_j = b_j * factor_j is stale now
    return 0;
}

int sphk_df_density_error_range(SphContext* c, float offset, int first, int count, double* out_host) {
    double h = 0.0;
    if (count > 0) {
        DevView d = sph_view(c);
        int nb = (count + TPB - 1) / TPB;
        if (nb > SPH_DF_ERR_BLOCKS) nb = SPH_DF_ERR_BLOCKS;
        hipLaunchKernelGGL(k_df_density_error, dim3(nb), dim3(TPB), 0, c->stream, d, offset, c->df_part, first, first + count);
        SPH_LAUNCH_CHECK(c);
        double* dev_out = nullptr;
        SPH_HIP(c, hipHostGetDevicePointer((void**)&dev_out, c->h_df_err, 0));
        hipLaunchKernelGGL(k_df_density_error_total, dim3(1), dim3(64), 0, c->stream, c->df_part, nb, dev_out);
        SPH_LAUNCH_CHECK(c);
        SPH_HIP(c, hipStreamSynchronize(c->stream));
        h = *(volatile double*)c->h_df_err;
    }
    *out_host = h;
    return 0;
}

int sphk_df_convergence_test(SphContext* c, float offset, double eta, int slot) {
    DevView d = sph_view(c);
    SphContext::DfSlot* dev_slot = nullptr;
    SPH_HIP(c, hipHostGetDevicePointer((void**)&dev_slot, c->h_df_slot, 0));
    const int count = c->df.fluid_particle_num > 0 ? c->df.fluid_particle_num : 1;
    if (c->df_bpart_valid) {   // the refresh sweep ran as a brick sweep and left one partial per listed brick (SPH_OPT_DF_FUSE_ERROR)
        c->df_bpart_valid = false;
        hipLaunchKernelGGL(k_df_convergence_test_bricks, dim3(1), dim3(1024), 0, c->stream, c->df_bpart, c->brick_count, c->brick_cap, count,
                           eta, dev_slot + slot, c->df_gate, c->df_epoch);
        SPH_LAUNCH_CHECK(c);
        return 0;
    }
    int nb = (c->N + TPB - 1) / TPB;
    if (nb > SPH_DF_ERR_BLOCKS) nb = SPH_DF_ERR_BLOCKS;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(k_df_density_error_gated, dim3(nb), dim3(TPB), 0, c->stream, d, offset, c->df_part);
    SPH_LAUNCH_CHECK(c);
    hipLaunchKernelGGL(k_df_convergence_test, dim3(1), dim3(64), 0, c->stream, c->df_part, nb, count, eta, dev_slot + slot,
                       c->df_gate, c->df_epoch);
    SPH_LAUNCH_CHECK(c);
    return 0;
}

int sphk_df_density_error(SphContext* c, float offset, float* out_host) {
    double h = 0.0;
    int rc = sphk_df_density_error_range(c, offset, 0, c->N, &h);
    *out_host = (float)h;
    return rc;
}

int sphk_check_uniform_fluid(SphContext* c) {
    c->uniform_state = 0;
    c->pure_fluid = 0;
    c->pure_fluid_n = -1;
    c->m_uniform = 0.0f;
    if (c->N <= 0) return 0;
    DevView d = sph_view(c);
    unsigned* out = reinterpret_cast<unsigned*>(c->stage);  // 16 bytes of the staging buffer
    const unsigned init[4] = {0xFFFFFFFFu, 0u, 0u, 0u};
    unsigned h[4];
    SPH_HIP(c, hipMemcpyAsync(out, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
    const int nb_chk = (c->N + TPB - 1) / TPB < 512 ? (c->N + TPB - 1) / TPB : 512;
    hipLaunchKernelGGL(k_check_uniform, dim3(nb_chk), dim3(TPB), 0, c->stream, d, out);
    SPH_LAUNCH_CHECK(c);
    SPH_HIP(c, hipMemcpyAsync(h, out, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    SPH_HIP(c, hipStreamSynchronize(c->stream));
    if (h[3] > 0 && h[0] == h[1] && h[2] == 0) {
        memcpy(&c->m_uniform, &h[0], sizeof(float));
        c->uniform_state = c->m_uniform > 0.0f ? 1 : 0;
        // every particle is a fluid particle (and h[2] == 0 above: every fluid m_V is m_V0 bit for bit) => no solid at all.
        // (whether the launcher USES the verdict is SPH_OPT_PURE_FLUID_INSTANCE: the A/B switch of the instance it selects)
        c->pure_fluid = (c->uniform_state == 1 && h[3] == (unsigned)c->N) ? 1 : 0;
        c->pure_fluid_n = c->N;
    }
    return 0;
}
