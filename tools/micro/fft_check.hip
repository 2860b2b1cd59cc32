// Unit check of the packed-fp32 butterflies and complex primitives of adorym_amd/csrc/adm_fft.h against a double-precision DFT.
// build (from tools/micro): hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o fft_check fft_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <complex>
#include "../../adorym_amd/csrc/adm_fft.h"
using namespace adm;
template <int R, bool INV> __global__ void k(cf* io) { cf a[R]; for (int i = 0; i < R; ++i) a[i] = io[threadIdx.x * R + i]; Dft<R, INV>::run(a); for (int i = 0; i < R; ++i) io[threadIdx.x * R + i] = a[i]; }
__global__ void kmul(cf* io) { cf a = io[0], b = io[1]; io[2] = cmul(a, b); io[3] = cmulc(a, b); io[4] = add_rot<false>(a, b); io[5] = add_rot<true>(a, b); io[6] = caxpy(a, 0.25f, b); }
template <int R, bool INV> double check() {
    std::vector<cf> h(R * 64); for (int i = 0; i < R * 64; ++i) h[i] = make_float2(sinf(0.37f * i) + 0.1f, cosf(0.11f * i * i));
    cf* d; hipMalloc(&d, h.size() * sizeof(cf)); hipMemcpy(d, h.data(), h.size() * sizeof(cf), hipMemcpyHostToDevice);
    hipLaunchKernelGGL((k<R, INV>), dim3(1), dim3(64), 0, 0, d);
    std::vector<cf> o(h.size()); hipMemcpy(o.data(), d, h.size() * sizeof(cf), hipMemcpyDeviceToHost);
    double err = 0;
    for (int t = 0; t < 64; ++t) for (int kk = 0; kk < R; ++kk) {
        std::complex<double> s = 0;
        for (int n = 0; n < R; ++n) s += std::complex<double>(h[t * R + n].x, h[t * R + n].y) * std::polar(1.0, (INV ? 2.0 : -2.0) * M_PI * n * kk / R);
        err = fmax(err, std::abs(s - std::complex<double>(o[t * R + kk].x, o[t * R + kk].y)));
    }
    hipFree(d); return err;
}
int main() {
    printf("Dft2 %.2e %.2e\n", check<2, false>(), check<2, true>());
    printf("Dft3 %.2e %.2e\n", check<3, false>(), check<3, true>());
    printf("Dft4 %.2e %.2e\n", check<4, false>(), check<4, true>());
    printf("Dft8 %.2e %.2e\n", check<8, false>(), check<8, true>());
    printf("Dft9 %.2e %.2e\n", check<9, false>(), check<9, true>());
    cf h[7] = {make_float2(1.5f, -2.f), make_float2(0.25f, 3.f)}; cf* d; hipMalloc(&d, sizeof(h)); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(kmul, dim3(1), dim3(1), 0, 0, d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    std::complex<double> a(1.5, -2), b(0.25, 3);
    auto p = [&](const char* n, cf g, std::complex<double> w) { printf("%s got (%g,%g) want (%g,%g)\n", n, g.x, g.y, w.real(), w.imag()); };
    p("cmul ", h[2], a * b); p("cmulc", h[3], a * std::conj(b)); p("a+(-i)b", h[4], a + std::complex<double>(0, -1) * b); p("a+(+i)b", h[5], a + std::complex<double>(0, 1) * b);
    p("a+.25b", h[6], a + 0.25 * b);
    return 0;
}
