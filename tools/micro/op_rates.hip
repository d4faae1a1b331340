// Issue cost on gfx950 of the instruction kinds the shading code is made of, in cycles per wave64 instruction per
// SIMD at 1, 2 and 4 waves per SIMD: v_fma_f32, v_pk_fma_f32, v_pk_mul_f32 + v_pk_add_f32, v_rcp_f32, v_sqrt_f32,
// v_cmp + v_cndmask, and the full-precision division sequence the compiler emits for a / b.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float F2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(64) k(float* out, int iters, float s) {
    float a0 = threadIdx.x + 1.5f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    F2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    F2 ss = {s, s};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (MODE == 0) {
                a0 = __builtin_fmaf(a0, s, a1); a1 = __builtin_fmaf(a1, s, a2); a2 = __builtin_fmaf(a2, s, a3); a3 = __builtin_fmaf(a3, s, a4);
                a4 = __builtin_fmaf(a4, s, a5); a5 = __builtin_fmaf(a5, s, a6); a6 = __builtin_fmaf(a6, s, a7); a7 = __builtin_fmaf(a7, s, a0);
            } else if (MODE == 1) {
                p0 = __builtin_elementwise_fma(p0, ss, p1); p1 = __builtin_elementwise_fma(p1, ss, p2); p2 = __builtin_elementwise_fma(p2, ss, p3); p3 = __builtin_elementwise_fma(p3, ss, p4);
                p4 = __builtin_elementwise_fma(p4, ss, p5); p5 = __builtin_elementwise_fma(p5, ss, p6); p6 = __builtin_elementwise_fma(p6, ss, p7); p7 = __builtin_elementwise_fma(p7, ss, p0);
            } else if (MODE == 2) { // 4 pk_mul + 4 pk_add
                p0 = p0 * ss; p1 = p1 + p0; p2 = p2 * ss; p3 = p3 + p2; p4 = p4 * ss; p5 = p5 + p4; p6 = p6 * ss; p7 = p7 + p6;
            } else if (MODE == 3) { // 8 v_rcp_f32
                a0 = __builtin_amdgcn_rcpf(a0); a1 = __builtin_amdgcn_rcpf(a1); a2 = __builtin_amdgcn_rcpf(a2); a3 = __builtin_amdgcn_rcpf(a3);
                a4 = __builtin_amdgcn_rcpf(a4); a5 = __builtin_amdgcn_rcpf(a5); a6 = __builtin_amdgcn_rcpf(a6); a7 = __builtin_amdgcn_rcpf(a7);
            } else if (MODE == 4) { // 8 v_sqrt_f32
                a0 = __builtin_amdgcn_sqrtf(a0); a1 = __builtin_amdgcn_sqrtf(a1); a2 = __builtin_amdgcn_sqrtf(a2); a3 = __builtin_amdgcn_sqrtf(a3);
                a4 = __builtin_amdgcn_sqrtf(a4); a5 = __builtin_amdgcn_sqrtf(a5); a6 = __builtin_amdgcn_sqrtf(a6); a7 = __builtin_amdgcn_sqrtf(a7);
            } else if (MODE == 5) { // 4 v_cmp + 4 v_cndmask
                a0 = a1 < s ? a2 : a0; a2 = a3 < s ? a4 : a2; a4 = a5 < s ? a6 : a4; a6 = a7 < s ? a0 : a6;
            } else if (MODE == 6) { // 2 rcp mixed with 6 fma
                a0 = __builtin_amdgcn_rcpf(a0); a1 = __builtin_fmaf(a1, s, a2); a2 = __builtin_fmaf(a2, s, a3); a3 = __builtin_fmaf(a3, s, a4);
                a4 = __builtin_amdgcn_rcpf(a4); a5 = __builtin_fmaf(a5, s, a6); a6 = __builtin_fmaf(a6, s, a7); a7 = __builtin_fmaf(a7, s, a0);
            } else { // 8 IEEE divisions
                a0 = a0 / a1; a1 = a1 / a2; a2 = a2 / a3; a3 = a3 / a4; a4 = a4 / a5; a5 = a5 / a6; a6 = a6 / a7; a7 = a7 / a0;
            }
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}
template <int MODE>
void run(float* out, const char* what, double per_iter) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    for (int w : {1, 2, 4}) {
        int grid = 256 * 4 * w;
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, out, iters, 0.999f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        printf("%-34s waves/SIMD %d: %8.3f ms, %6.2f cycles per item per SIMD (2.4 GHz)\n", what, w, ms, ms * 1e-3 * 2.4e9 / ((double)iters * per_iter * w));
    }
}
int main() {
    float* out; hipMalloc(&out, 8192 * 64 * 4);
    run<0>(out, "v_fma_f32", 64);
    run<1>(out, "v_pk_fma_f32", 64);
    run<2>(out, "v_pk_mul_f32 / v_pk_add_f32", 64);
    run<3>(out, "v_rcp_f32", 64);
    run<4>(out, "v_sqrt_f32", 64);
    run<5>(out, "v_cmp + v_cndmask (pair = 1 item)", 32);
    run<6>(out, "2 rcp + 6 fma (8 items)", 64);
    run<7>(out, "a / b (IEEE sequence, 1 item)", 64);
    return 0;
}
