// Issue-rate probe (gfx950, wave64): what one wave-instruction of the filter's opcodes costs a SIMD.
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
// Every op is inline asm on 8 independent register chains, 8 waves per SIMD resident, so neither latency nor the
// vectoriser decides the figure.  Cycles are relative to the measured v_mov_b32 (the shader clock under load is not
// known to the probe).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
template <int OP>
__global__ __launch_bounds__(256) void k(int* out, int a, int b, int iters) {
    int x[8]; float f[8]; f2 p[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { x[u] = threadIdx.x * (u + 1) + a; f[u] = (float)x[u]; p[u] = f2{f[u], f[u] + 1.f}; }
    int A = a * 65537 + (int)threadIdx.x, B = b * 3 + (int)threadIdx.x;
    float fa = (float)a * 1e-3f, fb = (float)b;
    f2 pa = f2{fa, fa}, pb = f2{fb, fb};
    asm volatile("" : "+v"(A), "+v"(B), "+v"(fa), "+v"(fb), "+v"(pa), "+v"(pb));
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#define S_MOV(u) asm volatile("v_mov_b32 %0, %1" : "=v"(x[u]) : "v"(A));
#define S_FMA(u) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[u]) : "v"(fa), "v"(fb));
#define S_FMAC(u) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(f[u]) : "v"(fa), "v"(fb));
#define S_SUB(u) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(f[u]) : "v"(fa));
#define S_PKFMA(u) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[u]) : "v"(pa), "v"(pb));
#define S_PKADD(u) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[u]) : "v"(pa));
#define S_ALIGN(u) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(x[u]) : "v"(A));
#define S_DOT2C(u) asm volatile("v_dot2c_i32_i16 %0, %1, %2" : "+v"(x[u]) : "v"(A), "v"(B));
#define S_LSHLOR(u) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(x[u]) : "v"(A));
#define S_ADD3(u) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x[u]) : "v"(A), "v"(B));
#define S_BFE(u) asm volatile("v_bfe_u32 %0, %0, %1, 1" : "+v"(x[u]) : "v"(A));
#define S_CMPADDC(u) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(x[u]) : "v"(fa), "v"(f[u]) : "vcc");
#define S_SQRT(u) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[u]));
#define S_MUL(u) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[u]) : "v"(fa));
#define S_MAX(u) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[u]) : "v"(fa));
            if (OP == 0) { REP8(S_MOV) }
            if (OP == 1) { REP8(S_FMA) }
            if (OP == 2) { REP8(S_FMAC) }
            if (OP == 3) { REP8(S_SUB) }
            if (OP == 4) { REP8(S_PKFMA) }
            if (OP == 5) { REP8(S_PKADD) }
            if (OP == 6) { REP8(S_ALIGN) }
            if (OP == 7) { REP8(S_DOT2C) }
            if (OP == 8) { REP8(S_LSHLOR) }
            if (OP == 9) { REP8(S_ADD3) }
            if (OP == 10) { REP8(S_BFE) }
            if (OP == 11) { REP8(S_CMPADDC) }
            if (OP == 12) { REP8(S_SQRT) }
            if (OP == 13) { REP8(S_MUL) }
            if (OP == 14) { REP8(S_MAX) }
        }
    }
    int s = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += x[u] + (int)f[u] + (int)p[u].x + (int)p[u].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP> float run(int* out, int iters) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(256 * 8), dim3(256), 0, 0, out, 3, 5, 10);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(256 * 8), dim3(256), 0, 0, out, 3, 5, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    int* out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    const int iters = 20000;
    const char* nm[] = {"v_mov_b32", "v_fma_f32", "v_fmac_f32", "v_sub_f32", "v_pk_fma_f32 (2 lanes' worth)", "v_pk_add_f32", "v_alignbit_b32",
                        "v_dot2c_i32_i16", "v_lshl_or_b32", "v_add3_u32", "v_bfe_u32", "v_cmp_lt_f32 + v_addc_co_u32 (pair)", "v_sqrt_f32", "v_mul_f32", "v_max_f32"};
    float ms[15] = {run<0>(out, iters), run<1>(out, iters), run<2>(out, iters), run<3>(out, iters), run<4>(out, iters), run<5>(out, iters), run<6>(out, iters),
                    run<7>(out, iters), run<8>(out, iters), run<9>(out, iters), run<10>(out, iters), run<11>(out, iters), run<12>(out, iters), run<13>(out, iters), run<14>(out, iters)};
    for (int i = 0; i < 15; ++i) printf("%-40s %8.3f ms   %.2f x v_mov_b32\n", nm[i], ms[i], ms[i] / ms[0]);
    return 0;
}
