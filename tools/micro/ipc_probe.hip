// Probe for the peer-to-peer transport (adm_p2p.hip): what two PROCESSES that share one GPU can do with each other's
// device memory, without RCCL.  Forks two ranks BEFORE any HIP call (no exec afterwards), hands the IPC handles over a
// socket pair and reports, step by step:
//   1. hipIpcGetMemHandle / hipIpcOpenMemHandle on plain hipMalloc memory and on hipDeviceMallocUncached memory (flags);
//   2. a device-side flag written by a kernel of rank 0 into rank 1's flag buffer releases a kernel of rank 1 that was
//      already spinning (kernels of two processes run side by side);
//   3. data written by rank 0's kernel is read correctly by rank 1's kernel through the mapped pointer after the flag;
//   4. device-only ping-pong latency (signal kernel -> wait kernel across processes);
//   5. read bandwidth through a mapped pointer;
//   6. interprocess events (hipIpcGetEventHandle / hipIpcOpenEventHandle / hipStreamWaitEvent).
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/micro/ipc_probe tools/micro/ipc_probe.hip
#include <hip/hip_runtime.h>
#include <sys/socket.h>
#include <sys/wait.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x)                                                                                         \
    do {                                                                                              \
        hipError_t e_ = (x);                                                                          \
        if (e_ != hipSuccess) {                                                                       \
            printf("[rank %d] %s -> %s\n", g_rank, #x, hipGetErrorString(e_));                        \
            fflush(stdout);                                                                           \
            _exit(3);                                                                                 \
        }                                                                                             \
    } while (0)

static int g_rank = -1;

__global__ void fill_kernel(float* p, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + (float)(i & 1023);
}
__global__ void sum_kernel(const float* p, size_t n, double* out) {
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    atomicAdd(out, s);
}
__global__ void read_kernel(const float4* p, size_t n4, float* sink) {
    float a = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = p[i];
        a += v.x + v.y + v.z + v.w;
    }
    if (a == 123.456f) *sink = a;
}
__global__ void signal_kernel(unsigned long long* flag, unsigned long long v) {
    __threadfence_system();
    __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// spins until *flag >= v or ~timeout_ticks of the 100 MHz wall clock have passed; status: 1 = seen, 2 = timed out
__global__ void wait_kernel(unsigned long long* flag, unsigned long long v, unsigned long long timeout_ticks, int* status) {
    const unsigned long long t0 = wall_clock64();
    int st = 2;
    while (wall_clock64() - t0 < timeout_ticks) {
        if (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) >= v) { st = 1; break; }
        __builtin_amdgcn_s_sleep(8);
    }
    if (status) *status = st;
}

static void xsend(int fd, const void* p, size_t n) { if (write(fd, p, n) != (ssize_t)n) _exit(4); }
static void xrecv(int fd, void* p, size_t n) {
    size_t got = 0;
    while (got < n) { ssize_t k = read(fd, (char*)p + got, n - got); if (k <= 0) _exit(5); got += k; }
}
static void hbarrier(int fd) { char c = 1; xsend(fd, &c, 1); xrecv(fd, &c, 1); }

static int rank_main(int rank, int fd) {
    g_rank = rank;
    CK(hipSetDevice(0));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const size_t n = 64u << 20 >> 2;     // 64 MB of floats
    float* data = nullptr;
    CK(hipMalloc(&data, n * 4));
    unsigned long long* flags = nullptr;
    hipError_t eu = hipExtMallocWithFlags((void**)&flags, 4096, hipDeviceMallocUncached);
    printf("[rank %d] hipExtMallocWithFlags(uncached): %s\n", rank, hipGetErrorString(eu));
    bool uncached = eu == hipSuccess;
    if (!uncached) { (void)hipGetLastError(); CK(hipMalloc((void**)&flags, 4096)); }
    CK(hipMemset(flags, 0, 4096));
    hipIpcMemHandle_t hd, hf, pd, pf;
    CK(hipIpcGetMemHandle(&hd, data));
    hipError_t ef = hipIpcGetMemHandle(&hf, flags);
    printf("[rank %d] hipIpcGetMemHandle(flags, %s): %s\n", rank, uncached ? "uncached" : "plain", hipGetErrorString(ef));
    if (ef != hipSuccess && uncached) {
        (void)hipGetLastError();
        CK(hipFree(flags));
        CK(hipMalloc((void**)&flags, 4096));
        CK(hipMemset(flags, 0, 4096));
        CK(hipIpcGetMemHandle(&hf, flags));
        uncached = false;
        printf("[rank %d] flags fall back to plain hipMalloc\n", rank);
    }
    xsend(fd, &hd, sizeof hd); xsend(fd, &hf, sizeof hf);
    xrecv(fd, &pd, sizeof pd); xrecv(fd, &pf, sizeof pf);
    float* peer_data = nullptr;
    unsigned long long* peer_flags = nullptr;
    CK(hipIpcOpenMemHandle((void**)&peer_data, pd, hipIpcMemLazyEnablePeerAccess));
    CK(hipIpcOpenMemHandle((void**)&peer_flags, pf, hipIpcMemLazyEnablePeerAccess));
    printf("[rank %d] opened peer data %p flags %p (own %p %p)\n", rank, (void*)peer_data, (void*)peer_flags, (void*)data, (void*)flags);
    int* status = nullptr;
    double* acc = nullptr;
    CK(hipHostMalloc((void**)&status, 64, 0));
    CK(hipHostMalloc((void**)&acc, 64, 0));
    *status = 0; *acc = 0;
    hbarrier(fd);
    const unsigned long long T2S = 200000000ull;       // 2 s of a 100 MHz clock
    // ---- 2 + 3: rank 1 spins first; rank 0 fills its data 300 ms later and raises rank 1's flag[0] ----
    if (rank == 1) {
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(wait_kernel, dim3(1), dim3(1), 0, st, flags + 0, 1ull, T2S, status);
        hipLaunchKernelGGL(sum_kernel, dim3(1024), dim3(256), 0, st, (const float*)peer_data, n, acc);
        CK(hipStreamSynchronize(st));
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        double want = 0;
        for (size_t i = 0; i < n; ++i) want += 7.0 + (double)(i & 1023);
        printf("[rank 1] spin released: status %d (1 = flag seen, 2 = timeout) after %.1f ms; peer data sum %.6g, expected %.6g -> %s\n", *status, ms, *acc,
               want, (*acc == want) ? "OK" : "MISMATCH");
    } else {
        usleep(300000);
        hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, st, data, n, 7.0f);
        hipLaunchKernelGGL(signal_kernel, dim3(1), dim3(1), 0, st, peer_flags + 0, 1ull);
        CK(hipStreamSynchronize(st));
    }
    hbarrier(fd);
    // ---- 4: device-only ping-pong: flag[1] counts rounds; rank 0 raises odd values in rank 1's buffer, rank 1 even ones in rank 0's ----
    {
        const int rounds = 200;
        *status = 0;
        auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < rounds; ++k) {
            const unsigned long long v_mine = 2ull * k + 1 + rank, v_peer = rank == 0 ? 2ull * k + 2 : 2ull * k + 1;
            if (rank == 0) {
                hipLaunchKernelGGL(signal_kernel, dim3(1), dim3(1), 0, st, peer_flags + 1, v_mine);
                hipLaunchKernelGGL(wait_kernel, dim3(1), dim3(1), 0, st, flags + 1, v_peer, T2S, (int*)nullptr);
            } else {
                hipLaunchKernelGGL(wait_kernel, dim3(1), dim3(1), 0, st, flags + 1, v_peer, T2S, (int*)nullptr);
                hipLaunchKernelGGL(signal_kernel, dim3(1), dim3(1), 0, st, peer_flags + 1, v_mine);
            }
        }
        CK(hipStreamSynchronize(st));
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        printf("[rank %d] ping-pong: %d round trips in %.2f ms = %.1f us per round trip (2 signals + 2 waits)\n", rank, rounds, ms, 1e3 * ms / rounds);
    }
    hbarrier(fd);
    // ---- 5: bandwidth of reads through the mapped pointer vs own memory ----
    if (rank == 1) {
        hipEvent_t a, b;
        CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        float* sink = nullptr;
        CK(hipMalloc(&sink, 4));
        for (int which = 0; which < 2; ++which) {
            const float4* src = (const float4*)(which ? peer_data : data);
            hipLaunchKernelGGL(read_kernel, dim3(4096), dim3(256), 0, st, src, n / 4, sink);
            CK(hipEventRecord(a, st));
            for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(read_kernel, dim3(4096), dim3(256), 0, st, src, n / 4, sink);
            CK(hipEventRecord(b, st));
            CK(hipEventSynchronize(b));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, a, b));
            printf("[rank 1] read of 64 MB from %s memory: %.1f GB/s\n", which ? "PEER (mapped)" : "own", 10.0 * n * 4 / (ms * 1e-3) / 1e9);
        }
    }
    hbarrier(fd);
    // ---- 6: interprocess events ----
    {
        hipEvent_t ev;
        hipError_t e1 = hipEventCreateWithFlags(&ev, hipEventInterprocess | hipEventDisableTiming);
        printf("[rank %d] hipEventCreateWithFlags(interprocess): %s\n", rank, hipGetErrorString(e1));
        hipIpcEventHandle_t he, pe;
        memset(&he, 0, sizeof he);
        hipError_t e2 = e1 == hipSuccess ? hipIpcGetEventHandle(&he, ev) : e1;
        printf("[rank %d] hipIpcGetEventHandle: %s\n", rank, hipGetErrorString(e2));
        (void)hipGetLastError();
        int ok = e2 == hipSuccess, peer_ok = 0;
        xsend(fd, &ok, sizeof ok); xsend(fd, &he, sizeof he);
        xrecv(fd, &peer_ok, sizeof peer_ok); xrecv(fd, &pe, sizeof pe);
        if (ok && peer_ok) {
            hipEvent_t pev;
            hipError_t e3 = hipIpcOpenEventHandle(&pev, pe);
            printf("[rank %d] hipIpcOpenEventHandle: %s\n", rank, hipGetErrorString(e3));
            (void)hipGetLastError();
            int ok3 = e3 == hipSuccess, peer3 = 0;
            xsend(fd, &ok3, sizeof ok3); xrecv(fd, &peer3, sizeof peer3);
            if (ok3 && peer3) {
                // rank 0: fill with 9 after 200 ms, record; rank 1: stream waits for rank 0's event, then sums
                if (rank == 0) {
                    hbarrier(fd);
                    usleep(200000);
                    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, st, data, n, 9.0f);
                    hipError_t e4 = hipEventRecord(ev, st);
                    printf("[rank 0] hipEventRecord(interprocess): %s\n", hipGetErrorString(e4));
                    CK(hipStreamSynchronize(st));
                    hbarrier(fd);
                } else {
                    hbarrier(fd);
                    usleep(400000);      // (an IPC event that has not been recorded yet counts as complete: wait on the host side for the record)
                    *acc = 0;
                    auto t0 = std::chrono::steady_clock::now();
                    hipError_t e4 = hipStreamWaitEvent(st, pev, 0);
                    printf("[rank 1] hipStreamWaitEvent(peer event): %s\n", hipGetErrorString(e4));
                    hipLaunchKernelGGL(sum_kernel, dim3(1024), dim3(256), 0, st, (const float*)peer_data, n, acc);
                    CK(hipStreamSynchronize(st));
                    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                    double want = 0;
                    for (size_t i = 0; i < n; ++i) want += 9.0 + (double)(i & 1023);
                    printf("[rank 1] after the peer event (%.2f ms): sum %.6g expected %.6g -> %s\n", ms, *acc, want, *acc == want ? "OK" : "MISMATCH");
                    hbarrier(fd);
                }
            }
        }
    }
    hbarrier(fd);
    CK(hipIpcCloseMemHandle(peer_data));
    CK(hipIpcCloseMemHandle(peer_flags));
    hbarrier(fd);
    CK(hipFree(data));
    CK(hipFree(flags));
    printf("[rank %d] done\n", rank);
    fflush(stdout);
    return 0;
}

int main() {
    int sv[2];
    if (socketpair(AF_UNIX, SOCK_STREAM, 0, sv)) return 1;
    setvbuf(stdout, nullptr, _IOLBF, 0);
    pid_t kids[2];
    for (int r = 0; r < 2; ++r) {
        kids[r] = fork();            // before any HIP call in this process
        if (kids[r] == 0) {
            close(sv[1 - r]);
            _exit(rank_main(r, sv[r]));
        }
    }
    int rc = 0;
    for (int r = 0; r < 2; ++r) {
        int st = 0;
        waitpid(kids[r], &st, 0);
        if (!WIFEXITED(st) || WEXITSTATUS(st)) rc = 1;
        printf("rank %d exit status %d\n", r, WIFEXITED(st) ? WEXITSTATUS(st) : -1);
    }
    return rc;
}
