// Measures VALU issue rate on gfx950: plain v_fma_f32 vs v_pk_fma_f32 vs integer, at 1..8 waves per SIMD; modes 3-5:
// the plain fma loop with only lanes 0..31, only the even lanes, only lanes 0..15 active (does a wave64 instruction
// whose upper half is masked off issue in fewer cycles?).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float float2_ __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(64) k(float* out, int iters, float s) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float2_ p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    float2_ ss = {s, s};
    if (MODE == 3 && threadIdx.x >= 32) iters = 0;
    if (MODE == 4 && (threadIdx.x & 1)) iters = 0;
    if (MODE == 5 && threadIdx.x >= 16) iters = 0;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0 || MODE >= 3) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                a0 = __builtin_fmaf(a0, s, a1); a1 = __builtin_fmaf(a1, s, a2); a2 = __builtin_fmaf(a2, s, a3); a3 = __builtin_fmaf(a3, s, a4);
                a4 = __builtin_fmaf(a4, s, a5); a5 = __builtin_fmaf(a5, s, a6); a6 = __builtin_fmaf(a6, s, a7); a7 = __builtin_fmaf(a7, s, a0);
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                p0 = __builtin_elementwise_fma(p0, ss, p1); p1 = __builtin_elementwise_fma(p1, ss, p2); p2 = __builtin_elementwise_fma(p2, ss, p3); p3 = __builtin_elementwise_fma(p3, ss, p4);
                p4 = __builtin_elementwise_fma(p4, ss, p5); p5 = __builtin_elementwise_fma(p5, ss, p6); p6 = __builtin_elementwise_fma(p6, ss, p7); p7 = __builtin_elementwise_fma(p7, ss, p0);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) { // integer xor/shift mix
                unsigned u0 = __float_as_uint(a0), u1 = __float_as_uint(a1), u2 = __float_as_uint(a2), u3 = __float_as_uint(a3);
                u0 ^= u1 >> 2; u1 ^= u2 << 1; u2 ^= u3 >> 3; u3 ^= u0 << 4; u0 += u2; u1 += u3; u2 ^= u0; u3 ^= u1;
                a0 = __uint_as_float(u0); a1 = __uint_as_float(u1); a2 = __uint_as_float(u2); a3 = __uint_as_float(u3);
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}
int main() {
    float* out; hipMalloc(&out, 8192 * 64 * 4 * 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int mode = 0; mode < 6; ++mode)
        for (int w : {1, 2, 4, 8}) {
            int grid = 256 * 4 * w;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(64), 0, 0, out, iters, 0.999f);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(64), 0, 0, out, iters, 0.999f);
                else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(64), 0, 0, out, iters, 0.999f);
                else if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(64), 0, 0, out, iters, 0.999f);
                else if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(grid), dim3(64), 0, 0, out, iters, 0.999f);
                else hipLaunchKernelGGL(k<5>, dim3(grid), dim3(64), 0, 0, out, iters, 0.999f);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double instr_per_wave = (double)iters * 64.0 * (mode == 2 ? 1.5 : 1.0);  // 8x8 fma, or 8x12 int ops
            double cyc = ms * 1e-3 * 2.4e9;
            printf("mode %d waves/SIMD %d: %.3f ms, %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", mode, w, ms, cyc / (instr_per_wave * w));
        }
    return 0;
}
