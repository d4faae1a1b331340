// Does SALU work share the SIMD's issue slot with VALU work?  Loops of 64 v_fma, 64 s_add, or both interleaved.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ void __launch_bounds__(64) k(float* out, int iters, float s) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    unsigned x0 = blockIdx.x, x1 = blockIdx.x * 3u + 1u, x2 = gridDim.x + 2u, x3 = (unsigned)iters;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (MODE != 1) {
                a0 = __builtin_fmaf(a0, s, a1); a1 = __builtin_fmaf(a1, s, a2); a2 = __builtin_fmaf(a2, s, a3); a3 = __builtin_fmaf(a3, s, a0);
            }
            if (MODE != 0) {
                // wave-uniform integer work: the compiler keeps it on the scalar ALU
                x0 += x1; x1 ^= x2; x2 += x3; x3 ^= x0;
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + (float)(x0 ^ x1 ^ x2 ^ x3);
}
int main() {
    float* out; (void)hipMalloc(&out, 8192 * 64 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    const char* names[3] = {"64 v_fma", "64 s_alu", "64 v_fma + 64 s_alu"};
    for (int mode = 0; mode < 3; ++mode)
        for (int w : {1, 4, 8}) {
            int grid = 256 * 4 * w;
            for (int rep = 0; rep < 2; ++rep) {
                (void)hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(64), 0, 0, out, iters, 0.999f);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(64), 0, 0, out, iters, 0.999f);
                else hipLaunchKernelGGL(k<2>, dim3(grid), dim3(64), 0, 0, out, iters, 0.999f);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            }
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * w);
            printf("%-22s waves/SIMD %d: %.3f ms, %.0f cycles per loop body per wave-slot (at 2.4 GHz)\n", names[mode], w, ms, cyc);
        }
    return 0;
}
