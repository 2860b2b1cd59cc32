// Micro-benchmark behind DESIGN.md section 5 ("why one position is not split over several CUs"): the cost of ONE exchange
// step of a 2-D FFT split over G compute units of one XCD.  A 72 x 72 complex field (41.5 KB) is owned row-wise by G
// workgroups; a row->column ownership change means every workgroup publishes its share (41.5/G KB) and reads 1/G of
// every other workgroup's share.  Protocol = the guide's R1 form (MI355X_MICROARCH.md, Workgroup dispatch ...): sc1
// stores, every storing wave's vmcnt(0), workgroup barrier, one lane adds to an agent-scope counter; consumers poll the
// counter with sc1 loads, barrier, then sc1 loads of the payload.  Groups are made of workgroups with equal
// blockIdx.x % 8 (same XCD under round-robin placement; speed only).  Reported: microseconds per exchange, G = 2, 4, 8,
// with 32 / 256 groups running at once (32 positions in flight / every CU busy).
// build: hipcc --offload-arch=gfx950 -O3 -o xcu_handoff xcu_handoff.hip ; run: ./xcu_handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int FIELD = 72 * 72;      // float2 elements

__device__ __forceinline__ void st_sc1(float2* p, float2 v) { __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y); }

template <int G>
__global__ __launch_bounds__(704) void handoff_kernel(float2* buf, unsigned* counters, int iters, unsigned long long* cycles_out, int n_groups) {
    // workgroup -> (group, member): members of a group are 8 blocks apart so that they share blockIdx.x % 8
    const int xcd = blockIdx.x % 8, slot = blockIdx.x / 8;
    const int member = slot % G, group = (slot / G) * 8 + xcd;
    if (group >= n_groups) return;
    float2* field = buf + (size_t)group * 2 * FIELD;               // two generations (ping-pong)
    unsigned* ctr = counters + (size_t)group * 64;                  // one 256-B line per group
    const int share = FIELD / G;
    float2 v = make_float2((float)threadIdx.x, (float)member);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        float2* gen = field + (it & 1) * FIELD;
        // publish my share (write-through stores), drain, barrier, signal
        for (int i = threadIdx.x; i < share; i += blockDim.x) __hip_atomic_store(&gen[member * share + i].x, v.x + it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                                              __hip_atomic_store(&gen[member * share + i].y, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)(it + 1) * G;
            int spins = 0;      // bounded: a protocol bug must not hang the GPU
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && ++spins < 2000000) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        // read 1/G of everybody's share (the transposed ownership)
        float2 acc = make_float2(0.f, 0.f);
        for (int m = 0; m < G; ++m) {
            const float2* src = gen + m * share + member * (share / G);
            for (int i = threadIdx.x; i < share / G; i += blockDim.x) {
                const float x = __hip_atomic_load(&src[i].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const float y = __hip_atomic_load(&src[i].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                acc.x += x; acc.y += y;
            }
        }
        v.x += acc.x * 1e-30f; v.y += acc.y * 1e-30f;             // keep the loads alive
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cycles_out[blockIdx.x] = t1 - t0;
    if (v.x == 123456.f) buf[0] = v;
}

template <int G> int run(int n_groups, int iters) {
    float2* buf; unsigned* ctr; unsigned long long* cyc;
    const int blocks = n_groups * G;
    CHECK(hipMalloc(&buf, (size_t)n_groups * 2 * FIELD * sizeof(float2)));
    CHECK(hipMalloc(&ctr, (size_t)n_groups * 64 * sizeof(unsigned)));
    CHECK(hipMalloc(&cyc, blocks * sizeof(unsigned long long)));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemset(ctr, 0, (size_t)n_groups * 64 * sizeof(unsigned)));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(handoff_kernel<G>, dim3(blocks), dim3(704), 0, 0, buf, ctr, iters, cyc, n_groups);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("G=%d CUs per field, %3d fields in flight (%4d workgroups): %.2f us per exchange (publish %4.1f KB + gather)\n", G, n_groups, blocks,
           1e3f * best / iters, FIELD * 8.0 / G / 1024);
    hipFree(buf); hipFree(ctr); hipFree(cyc);
    return 0;
}

int main() {
    const int iters = 2000;
    if (run<2>(32, iters) || run<4>(32, iters) || run<8>(32, iters)) return 1;
    if (run<2>(128, iters) || run<4>(64, iters) || run<8>(32, iters)) return 1;      // every CU busy
    return 0;
}
