// Micro-benchmark: VALU issue rate for packed / scalar fp32 FMA streams at the multislice kernel's occupancy
// (blocks of 704 threads = 11 waves per CU).  hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <bool PACKED> __global__ __launch_bounds__(704) void k(float* out, int iters) {
    v2f a[8];
    float s[16];
    for (int i = 0; i < 8; ++i) a[i] = (v2f){(float)threadIdx.x * 1e-3f + i, 1.0f};
    for (int i = 0; i < 16; ++i) s[i] = threadIdx.x * 1e-3f + i;
    const v2f m = {1.0001f, 0.9999f}, c = {1e-6f, -1e-6f};
    for (int it = 0; it < iters; ++it) {
        if (PACKED) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(a[i], m, c);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) s[i] = fmaf(s[i], 1.0001f, 1e-6f);
        }
    }
    float acc = 0;
    for (int i = 0; i < 8; ++i) acc += a[i].x + a[i].y;
    for (int i = 0; i < 16; ++i) acc += s[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
    float* d; hipMalloc(&d, 256 * 704 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int blocks : {32, 256}) for (int packed = 0; packed < 2; ++packed) for (int threads : {64, 256, 704}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (packed) hipLaunchKernelGGL(k<true>, dim3(blocks), dim3(threads), 0, 0, d, iters);
            else hipLaunchKernelGGL(k<false>, dim3(blocks), dim3(threads), 0, 0, d, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double instr_per_wave = (double)iters * 64;           // 64 VALU instructions per iteration either way
        double waves_per_simd = (threads / 64 + 3) / 4.0;
        printf("blocks %3d threads %3d %s: %.3f ms -> %.2f ns per instr per wave, x%.2f waves/SIMD(max) => %.2f ns per SIMD issue slot\n", blocks, threads,
               packed ? "v_pk_fma_f32" : "v_fma_f32   ", ms, ms * 1e6 / instr_per_wave, waves_per_simd, ms * 1e6 / instr_per_wave / ((threads / 64 + 3) / 4));
    }
    return 0;
}
