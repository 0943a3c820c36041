// ubench_valu.hip -- issue rate of the instructions the gather sweeps are made of, on gfx950.
//
// Question ("What's weak" #2): does a wave64 VALU instruction occupy its SIMD for 2 cycles
// (SIMD-32, MI355X_MICROARCH.md) or for 4 (the model DESIGN.md r01 used)?  That decides whether the sweeps'
// SQ_INSTS_VALU count means 40 % or 80 % of the issue peak.
//
// Method: every wave runs `iters` trips of a 64-instruction block of ONE opcode on 8 independent register
// chains (so dependent-issue latency is hidden from 1 wave on), timed with s_memtime (= shader cycles) inside
// the wave and with HIP events outside.  Grids of 256 * k workgroups of 256 lanes put k waves on every SIMD
// (k = 1, 2, 4, 8).  cycles per wave-instruction per SIMD = wave's cycles / (instructions * k) when the k waves
// share the SIMD evenly; the event time gives the same figure chip-wide (1024 SIMDs) and the effective clock.
//
//   hipcc --offload-arch=gfx950 -O2 -o ubench_valu tools/ubench_valu.hip && ./ubench_valu
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum Op { OP_FMA, OP_FMA_DEP, OP_PK_FMA, OP_MUL, OP_ADD, OP_MAX, OP_MOV, OP_ALIGNBIT, OP_CNDMASK, OP_RSQ, OP_RCP, OP_EXP, OP_LOG,
          OP_FFBL, OP_AND, OP_LDS_B128, OP_LDS_B32, OP_LDS_B128_FMA4, OP_FILTER5, OP_COUNT };
static const char* op_name[OP_COUNT] = {
    "v_fma_f32 (8 chains)", "v_fma_f32 (1 dependent chain)", "v_pk_fma_f32 (8 chains)", "v_mul_f32", "v_add_f32", "v_max_f32",
    "v_mov_b32", "v_alignbit_b32", "v_cndmask_b32", "v_rsq_f32", "v_rcp_f32", "v_exp_f32", "v_log_f32", "v_ffbl_b32", "v_and_b32",
    "ds_read_b128 (conflict-free)", "ds_read_b32", "ds_read_b128 + 4 v_fma_f32", "filter: ds_read_b128 + sub + 3 fma + alignbit"};

typedef float float2v __attribute__((ext_vector_type(2)));
typedef float float4v __attribute__((ext_vector_type(4)));

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BLK8(STMT) STMT STMT STMT STMT STMT STMT STMT STMT

template <int OP>
__global__ __launch_bounds__(256) void k_bench(unsigned long long* cyc, float* sink, int iters) {
    __shared__ float4v lds[2048];
    const int tid = threadIdx.x;
    for (int i = tid; i < 2048; i += 256) lds[i] = float4v{(float)i, 1.0f, 2.0f, 3.0f};
    __syncthreads();
    float a0 = tid * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 0.999f, c = 1e-3f;
    float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = p0, p5 = p1, p6 = p2, p7 = p3, pb = {b, b}, pc = {c, c};
    unsigned u0 = tid, u1 = tid + 1, u2 = tid + 2, u3 = tid + 3, u4 = tid + 4, u5 = tid + 5, u6 = tid + 6, u7 = tid + 7;
    float4v q0 = {}, q1 = {}, q2 = {}, q3 = {};
    unsigned addr = (unsigned)(tid & 63) * 16u;  // consecutive 16-B records: conflict-free ds_read_b128
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (OP == OP_FMA) {
#define X(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
            BLK8(asm volatile(R8(X) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
#undef X
        } else if (OP == OP_FMA_DEP) {
            BLK8(asm volatile("v_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\n"
                              "v_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\nv_fma_f32 %0, %0, %1, %2\n"
                              : "+v"(a0) : "v"(b), "v"(c));)
        } else if (OP == OP_PK_FMA) {
#define X(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n"
            BLK8(asm volatile(R8(X) : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc));)
#undef X
        } else if (OP == OP_MUL || OP == OP_ADD || OP == OP_MAX) {
#define XM(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define XA(i) "v_add_f32 %" #i ", %" #i ", %8\n"
#define XX(i) "v_max_f32 %" #i ", %" #i ", %8\n"
            if (OP == OP_MUL) { BLK8(asm volatile(R8(XM) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
            if (OP == OP_ADD) { BLK8(asm volatile(R8(XA) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
            if (OP == OP_MAX) { BLK8(asm volatile(R8(XX) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
#undef XM
#undef XA
#undef XX
        } else if (OP == OP_MOV) {
            BLK8(asm volatile("v_mov_b32 %0, %1\nv_mov_b32 %1, %2\nv_mov_b32 %2, %3\nv_mov_b32 %3, %4\nv_mov_b32 %4, %5\nv_mov_b32 %5, %6\nv_mov_b32 %6, %7\nv_mov_b32 %7, %0\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (OP == OP_ALIGNBIT || OP == OP_FFBL || OP == OP_AND) {
#define XL(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 31\n"
#define XF(i) "v_ffbl_b32 %" #i ", %" #i "\n"
#define XN(i) "v_and_b32 %" #i ", %" #i ", %8\n"
            if (OP == OP_ALIGNBIT) { BLK8(asm volatile(R8(XL) : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(b));) }
            if (OP == OP_FFBL) { BLK8(asm volatile(R8(XF) : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7));) }
            if (OP == OP_AND) { BLK8(asm volatile(R8(XN) : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(b));) }
#undef XL
#undef XF
#undef XN
        } else if (OP == OP_CNDMASK) {
#define X(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
            BLK8(asm volatile(R8(X) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");)
#undef X
        } else if (OP == OP_RSQ || OP == OP_RCP || OP == OP_EXP || OP == OP_LOG) {
#define XS(i) "v_rsq_f32 %" #i ", %" #i "\n"
#define XR(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define XE(i) "v_exp_f32 %" #i ", %" #i "\n"
#define XG(i) "v_log_f32 %" #i ", %" #i "\n"
            if (OP == OP_RSQ) { BLK8(asm volatile(R8(XS) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
            if (OP == OP_RCP) { BLK8(asm volatile(R8(XR) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
            if (OP == OP_EXP) { BLK8(asm volatile(R8(XE) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
            if (OP == OP_LOG) { BLK8(asm volatile(R8(XG) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
#undef XS
#undef XR
#undef XE
#undef XG
        } else if (OP == OP_LDS_B128) {
            // 64 reads per trip, 16 in flight per wait
            for (int r = 0; r < 4; ++r) {
                BLK8(asm volatile("ds_read_b128 %0, %2\nds_read_b128 %1, %2 offset:1024\n" : "=v"(q0), "=v"(q1) : "v"(addr));)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        } else if (OP == OP_LDS_B32) {
            for (int r = 0; r < 4; ++r) {
                BLK8(asm volatile("ds_read_b32 %0, %2\nds_read_b32 %1, %2 offset:1024\n" : "=v"(a0), "=v"(a1) : "v"(addr));)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        } else if (OP == OP_LDS_B128_FMA4) {
            // per record: one ds_read_b128 and four v_fma_f32 (does the LDS pipe overlap VALU issue?)  64 records per trip
            for (int r = 0; r < 8; ++r) {
                asm volatile("ds_read_b128 %0, %4\nds_read_b128 %1, %4 offset:256\nds_read_b128 %2, %4 offset:512\nds_read_b128 %3, %4 offset:768\n"
                             : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3) : "v"(addr));
                asm volatile("ds_read_b128 %0, %1 offset:1024\n" : "=v"(q0) : "v"(addr));
                asm volatile("ds_read_b128 %0, %1 offset:1280\n" : "=v"(q1) : "v"(addr));
                asm volatile("ds_read_b128 %0, %1 offset:1536\n" : "=v"(q2) : "v"(addr));
                asm volatile("ds_read_b128 %0, %1 offset:1792\n" : "=v"(q3) : "v"(addr));
#define X(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
                asm volatile(R8(X) R8(X) R8(X) R8(X) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
#undef X
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        } else if (OP == OP_FILTER5) {
            // the density sweep's candidate test as compiled: per candidate 1 ds_read_b128 + v_sub + 3 v_fma + v_alignbit;
            // 8 candidates per batch, 64 per trip
            for (int r = 0; r < 8; ++r) {
                float4v c0, c1, c2, c3, c4, c5, c6, c7;
                asm volatile("ds_read_b128 %0, %8\nds_read_b128 %1, %8 offset:16\nds_read_b128 %2, %8 offset:32\nds_read_b128 %3, %8 offset:48\n"
                             "ds_read_b128 %4, %8 offset:64\nds_read_b128 %5, %8 offset:80\nds_read_b128 %6, %8 offset:96\nds_read_b128 %7, %8 offset:112\n"
                             "s_waitcnt lgkmcnt(0)\n"
                             : "=v"(c0), "=v"(c1), "=v"(c2), "=v"(c3), "=v"(c4), "=v"(c5), "=v"(c6), "=v"(c7) : "v"(addr));
#define T(Q) { float s = Q.w - a3; s = __builtin_fmaf(a0, Q.x, __builtin_fmaf(a1, Q.y, __builtin_fmaf(a2, Q.z, s))); \
               u0 = __builtin_amdgcn_alignbit(u0, __float_as_uint(s), 31); }
                T(c0) T(c1) T(c2) T(c3) T(c4) T(c5) T(c6) T(c7)
#undef T
                asm volatile("" : "+v"(u0));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((tid & 63) == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y +
              (float)(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7) + q0.x + q1.y + q2.z + q3.w;
    if (s == 123.456f) sink[tid] = s;
}

template <int OP>
static void run_op(unsigned long long* d_cyc, float* d_sink, FILE* out) {
    const int iters = 2000;
    const int insts_per_trip = 64;
    for (int k : {1, 2, 4, 8}) {
        const int grid = 256 * k;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_bench<OP>, dim3(grid), dim3(256), 0, 0, d_cyc, d_sink, 50);  // warm-up (clocks, code)
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_bench<OP>, dim3(grid), dim3(256), 0, 0, d_cyc, d_sink, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(grid * 4);
        CHECK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
        std::sort(h.begin(), h.end());
        const double med = (double)h[h.size() / 2], mx = (double)h.back();
        const double n_inst = (double)iters * insts_per_trip;
        // chip-wide: wave-instructions per second per SIMD, and the clock implied by the slowest wave
        const double clk_ghz = mx / (ms * 1e-3) / 1e9;
        const double inst_per_simd_per_s = n_inst * grid * 4 / 1024.0 / (ms * 1e-3);
        fprintf(out, "%-46s k=%d  wave cycles med %9.0f max %9.0f  cyc/inst/wave %6.2f  cyc/inst/SIMD(k waves) %5.2f  "
                     "event %7.3f ms  clock~%.2f GHz  chip: %6.1f Ginst/s/SIMD-> %5.2f cyc/inst\n",
                op_name[OP], k, med, mx, med / n_inst, med / n_inst / k, ms, clk_ghz, inst_per_simd_per_s / 1e9,
                clk_ghz * 1e9 / inst_per_simd_per_s);
        fflush(out);
        CHECK(hipEventDestroy(e0));
        CHECK(hipEventDestroy(e1));
    }
}

int main(int argc, char** argv) {
    FILE* out = stdout;
    unsigned long long* d_cyc;
    float* d_sink;
    CHECK(hipMalloc(&d_cyc, 256 * 8 * 4 * 8));
    CHECK(hipMalloc(&d_sink, 256 * 4));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    fprintf(out, "# %s  CUs %d  clockRate %d kHz; 64 instructions x 2000 trips per wave; 256*k workgroups of 4 waves\n", prop.gcnArchName,
            prop.multiProcessorCount, prop.clockRate);
    fprintf(out, "# (filter / LDS+fma rows: 'instruction' = one candidate record = 1 ds_read_b128 + its VALU)\n");
    run_op<OP_FMA>(d_cyc, d_sink, out);
    run_op<OP_FMA_DEP>(d_cyc, d_sink, out);
    run_op<OP_PK_FMA>(d_cyc, d_sink, out);
    run_op<OP_MUL>(d_cyc, d_sink, out);
    run_op<OP_ADD>(d_cyc, d_sink, out);
    run_op<OP_MAX>(d_cyc, d_sink, out);
    run_op<OP_MOV>(d_cyc, d_sink, out);
    run_op<OP_ALIGNBIT>(d_cyc, d_sink, out);
    run_op<OP_CNDMASK>(d_cyc, d_sink, out);
    run_op<OP_AND>(d_cyc, d_sink, out);
    run_op<OP_FFBL>(d_cyc, d_sink, out);
    run_op<OP_RSQ>(d_cyc, d_sink, out);
    run_op<OP_RCP>(d_cyc, d_sink, out);
    run_op<OP_EXP>(d_cyc, d_sink, out);
    run_op<OP_LOG>(d_cyc, d_sink, out);
    run_op<OP_LDS_B128>(d_cyc, d_sink, out);
    run_op<OP_LDS_B32>(d_cyc, d_sink, out);
    run_op<OP_LDS_B128_FMA4>(d_cyc, d_sink, out);
    run_op<OP_FILTER5>(d_cyc, d_sink, out);
    return 0;
}
