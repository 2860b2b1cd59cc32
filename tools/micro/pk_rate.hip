// Issue interval of v_fma_f32, v_pk_fma_f32, v_add_f32 and v_pk_add_f32 for one wave alone on its SIMD and for 2-4 waves per
// SIMD (inline asm, 8 independent dependency chains per wave): does packing two fp32 operations into one instruction help a
// wave that is bound by its own issue interval rather than by the SIMD's throughput?
// build: hipcc --offload-arch=gfx950 -O3 -o pk_rate pk_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP8(M) M(x0) M(x1) M(x2) M(x3) M(x4) M(x5) M(x6) M(x7)
template <int MODE> __global__ __launch_bounds__(256) void k(float* out, int iters) {
    v2f x0 = {(float)threadIdx.x, 1.f}, x1 = {1.f, 2.f}, x2 = {2.f, 3.f}, x3 = {3.f, 4.f}, x4 = {4.f, 5.f}, x5 = {5.f, 6.f}, x6 = {6.f, 7.f}, x7 = {7.f, 8.f};
    const v2f a = {1.0000001f, 0.9999999f}, b = {1e-9f, 2e-9f};
    for (int i = 0; i < iters; ++i) {
#define S_FMA(V) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(V.x) : "v"(a.x), "v"(b.x));
#define P_FMA(V) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(V) : "v"(a), "v"(b));
#define S_ADD(V) asm volatile("v_add_f32 %0, %0, %1" : "+v"(V.x) : "v"(b.x));
#define P_ADD(V) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(V) : "v"(b));
        if (MODE == 0) { REP8(S_FMA) }
        if (MODE == 1) { REP8(P_FMA) }
        if (MODE == 2) { REP8(S_ADD) }
        if (MODE == 3) { REP8(P_ADD) }
    }
    v2f s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}
int main() {
    float* out; hipMalloc(&out, 256 * 4096 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 200000;
    const char* names[4] = {"v_fma_f32   ", "v_pk_fma_f32", "v_add_f32   ", "v_pk_add_f32"};
    for (int wps = 1; wps <= 4; ++wps)
        for (int mode = 0; mode < 4; ++mode) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256 * wps), dim3(256), 0, 0, out, iters);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256 * wps), dim3(256), 0, 0, out, iters);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256 * wps), dim3(256), 0, 0, out, iters);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256 * wps), dim3(256), 0, 0, out, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("%d wave(s)/SIMD  %s: %.2f ns per instruction per wave, %.2f ns per instruction per SIMD\n", wps, names[mode], 1e6 * best / iters / 8,
                   1e6 * best / iters / 8 / wps);
        }
    return 0;
}
