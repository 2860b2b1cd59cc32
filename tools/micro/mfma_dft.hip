// Micro-benchmark behind DESIGN.md section 5 ("MFMA not used"): issue rate of the exact-fp32 matrix instruction against
// the vector FMA on one CU, to price a radix-8 DFT done as a matrix product.  A complex 8-point DFT of 16 lines is
// [16 x 16 real] x [16 x 16 real] = four v_mfma_f32_16x16x4_f32 (K = 16) = 4 x 2048 FLOP for what the vector butterfly does in
// 56 instructions x 64 lanes / (16 lines x 8 points): the matrix form spends 64 FLOP per point, the butterfly ~7.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_dft mfma_dft.hip ; run: ./mfma_dft
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float4v __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters) {
    float4v acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
    const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int i = 0; i < iters; ++i) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc3, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc0[0] + acc1[1] + acc2[2] + acc3[3];
}
__global__ __launch_bounds__(256) void fma_loop(float* out, int iters) {
    float x0 = threadIdx.x, x1 = 1.f, x2 = 2.f, x3 = 3.f, x4 = 4.f, x5 = 5.f, x6 = 6.f, x7 = 7.f;
    const float a = 1.0000001f, b = 1e-9f;
    for (int i = 0; i < iters; ++i) {
        x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
        x4 = fmaf(x4, a, b); x5 = fmaf(x5, a, b); x6 = fmaf(x6, a, b); x7 = fmaf(x7, a, b);
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
int main() {
    float* out; hipMalloc(&out, 256 * 1024 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 100000;
    for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
        for (int which = 0; which < 2; ++which) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (which == 0) hipLaunchKernelGGL(mfma_loop, dim3(256 * waves_per_simd), dim3(256), 0, 0, out, iters);
                else hipLaunchKernelGGL(fma_loop, dim3(256 * waves_per_simd), dim3(256), 0, 0, out, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double n_inst = (double)iters * (which == 0 ? 4 : 8);
            const double flop_per_inst = which == 0 ? 2.0 * 16 * 16 * 4 : 2.0 * 64;
            const double waves = 256.0 * waves_per_simd * 4;
            printf("%s, %d wave(s) per SIMD: %.2f ns per wave-instruction, %.1f TFLOP/s chip-wide\n", which == 0 ? "v_mfma_f32_16x16x4_f32" : "v_fma_f32             ",
                   waves_per_simd, 1e6 * best / n_inst / waves_per_simd, n_inst * flop_per_inst * waves / (best * 1e-3) / 1e12);
        }
    }
    return 0;
}
