// issue-rate probe: v_dot2_i32_i16 against v_fma_f32 and v_alignbit_b32 (wave64, gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256) void k(int* out, int a, int b, int iters) {
    int x[8];
    float f[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { x[u] = threadIdx.x * (u + 1) + a; f[u] = (float)x[u]; }
    const s2 A = __builtin_bit_cast(s2, a * 65537 + (int)threadIdx.x), B = __builtin_bit_cast(s2, b * 3 + (int)threadIdx.x);
    const float fa = (float)a * 1e-3f, fb = (float)b;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (OP == 0) f[u] = fmaf(f[u], fa, fb);
            if (OP == 1) x[u] = __builtin_amdgcn_sdot2(__builtin_bit_cast(s2, x[u]), A, x[u], false);      // VOP2 dot2c (acc in place)
            if (OP == 2) x[u] = __builtin_amdgcn_sdot2(A, B, x[u], false) ^ 0;                                // same, constant multiplicands
            if (OP == 3) x[u] = __builtin_amdgcn_alignbit(x[u], x[(u + 1) & 7], 31);
            if (OP == 4) { int t = __builtin_amdgcn_sdot2(__builtin_bit_cast(s2, x[u]), A, b, false); x[u] = __builtin_amdgcn_sdot2(__builtin_bit_cast(s2, t), B, t, false); } // VOP3P form (C != D) + dot2c
        }
    }
    int s = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += x[u] + (int)f[u];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP> float run(int* out, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(256 * 8), dim3(256), 0, 0, out, 3, 5, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(256 * 8), dim3(256), 0, 0, out, 3, 5, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    int* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    const int iters = 20000;
    const double winst = 256.0 * 8 * 4 * iters * 32;  // wave-instructions of the measured op
    const char* nm[] = {"v_fma_f32", "v_dot2c_i32_i16 (x*A+x)", "v_dot2c (A*B+x)", "v_alignbit_b32", "dot2 + dot2c pair"};
    float ms[5] = {run<0>(out, iters), run<1>(out, iters), run<2>(out, iters), run<3>(out, iters), run<4>(out, iters)};
    for (int i = 0; i < 5; ++i) {
        const double w = winst * (i == 4 ? 2 : 1);
        printf("%-28s %8.3f ms  %.2f cycles per wave-instruction per SIMD (2.4 GHz, 1024 SIMDs)\n", nm[i], ms[i], ms[i] * 1e-3 * 2.4e9 * 1024 / w);
    }
    return 0;
}
