// Small-radix complex DFT butterflies and the two-pass (N = R1*R2) line transform used by the
// multislice kernel.  Device-only, gfx950.  All loops are compile-time unrolled so the
// per-thread element arrays live in VGPRs.
//
// Conventions:  forward DFT  X[k] = sum_n x[n] exp(-2 pi i n k / N)   (torch.fft.fft2, norm=None)
//               inverse here is the UNNORMALISED conjugate transform; 1/N factors are folded
//               into the transfer-function registers by the caller.
#pragma once
#include <hip/hip_runtime.h>

namespace adm {

typedef float2 cf;

__device__ __forceinline__ cf cadd(cf a, cf b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cf csub(cf a, cf b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cf cmul(cf a, cf b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
// a * conj(b)
__device__ __forceinline__ cf cmulc(cf a, cf b) { return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }
__device__ __forceinline__ cf cscale(cf a, float s) { return make_float2(a.x * s, a.y * s); }
template <bool CONJ> __device__ __forceinline__ cf cmul_t(cf a, cf b) { return CONJ ? cmulc(a, b) : cmul(a, b); }
// multiply by -i (forward) or +i (inverse)
template <bool INV> __device__ __forceinline__ cf rot90(cf a) {
    return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}

template <int R, bool INV> struct Dft;

template <bool INV> struct Dft<2, INV> {
    static __device__ __forceinline__ void run(cf (&a)[2]) {
        cf t = a[0];
        a[0] = cadd(t, a[1]);
        a[1] = csub(t, a[1]);
    }
};

template <bool INV> struct Dft<3, INV> {
    static __device__ __forceinline__ void run(cf& a0, cf& a1, cf& a2) {
        const float h = 0.86602540378443864676f;  // sqrt(3)/2
        cf s = cadd(a1, a2);
        cf d = csub(a1, a2);
        cf m = make_float2(a0.x - 0.5f * s.x, a0.y - 0.5f * s.y);
        // forward: (-i h) d ; inverse: (+i h) d
        cf r = INV ? make_float2(-h * d.y, h * d.x) : make_float2(h * d.y, -h * d.x);
        a0 = cadd(a0, s);
        a1 = cadd(m, r);
        a2 = csub(m, r);
    }
    static __device__ __forceinline__ void run(cf (&a)[3]) { run(a[0], a[1], a[2]); }
};

template <bool INV> struct Dft<4, INV> {
    static __device__ __forceinline__ void run(cf& a0, cf& a1, cf& a2, cf& a3) {
        cf t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = rot90<INV>(csub(a1, a3));
        a0 = cadd(t0, t2);
        a1 = cadd(t1, t3);
        a2 = csub(t0, t2);
        a3 = csub(t1, t3);
    }
    static __device__ __forceinline__ void run(cf (&a)[4]) { run(a[0], a[1], a[2], a[3]); }
};

template <bool INV> struct Dft<8, INV> {
    static __device__ __forceinline__ void run(cf (&a)[8]) {
        const float c = 0.70710678118654752440f;
        cf e0 = a[0], e1 = a[2], e2 = a[4], e3 = a[6];
        cf o0 = a[1], o1 = a[3], o2 = a[5], o3 = a[7];
        Dft<4, INV>::run(e0, e1, e2, e3);
        Dft<4, INV>::run(o0, o1, o2, o3);
        // twiddles W8^k (forward exp(-i pi k/4), inverse conj)
        cf w1 = INV ? make_float2(c * (o1.x - o1.y), c * (o1.x + o1.y)) : make_float2(c * (o1.x + o1.y), c * (o1.y - o1.x));
        cf w2 = rot90<INV>(o2);
        cf w3 = INV ? make_float2(-c * (o3.x + o3.y), c * (o3.x - o3.y)) : make_float2(c * (o3.y - o3.x), -c * (o3.x + o3.y));
        a[0] = cadd(e0, o0);
        a[4] = csub(e0, o0);
        a[1] = cadd(e1, w1);
        a[5] = csub(e1, w1);
        a[2] = cadd(e2, w2);
        a[6] = csub(e2, w2);
        a[3] = cadd(e3, w3);
        a[7] = csub(e3, w3);
    }
};

template <bool INV> struct Dft<9, INV> {
    static __device__ __forceinline__ void run(cf (&a)[9]) {
        // n = 3 n1 + n2, k = k1 + 3 k2
        const float c1 = 0.76604444311897803520f, s1 = 0.64278760968653932632f;   // cos/sin(2pi/9)
        const float c2 = 0.17364817766693034885f, s2 = 0.98480775301220805937f;   // cos/sin(4pi/9)
        const float c4 = -0.93969262078590838405f, s4 = 0.34202014332566873304f;  // cos/sin(8pi/9)
        // stage 1: radix-3 over n1 for each n2 -> A[k1][n2] stored at a[3 k1 + n2]
        Dft<3, INV>::run(a[0], a[3], a[6]);
        Dft<3, INV>::run(a[1], a[4], a[7]);
        Dft<3, INV>::run(a[2], a[5], a[8]);
        // twiddle W9^(n2 k1): forward exp(-i..) = (c, -s); inverse (c, +s)
        const cf w1 = make_float2(c1, INV ? s1 : -s1);
        const cf w2 = make_float2(c2, INV ? s2 : -s2);
        const cf w4 = make_float2(c4, INV ? s4 : -s4);
        a[4] = cmul(a[4], w1);   // k1=1, n2=1
        a[5] = cmul(a[5], w2);   // k1=1, n2=2
        a[7] = cmul(a[7], w2);   // k1=2, n2=1
        a[8] = cmul(a[8], w4);   // k1=2, n2=2
        // stage 2: radix-3 over n2 for each k1 -> X[k1 + 3 k2] at a[3 k1 + k2]
        Dft<3, INV>::run(a[0], a[1], a[2]);
        Dft<3, INV>::run(a[3], a[4], a[5]);
        Dft<3, INV>::run(a[6], a[7], a[8]);
        // natural order: X[k1 + 3 k2] <- a[3 k1 + k2]  (3x3 transpose, register renaming only)
        cf t;
        t = a[1]; a[1] = a[3]; a[3] = t;
        t = a[2]; a[2] = a[6]; a[6] = t;
        t = a[5]; a[5] = a[7]; a[7] = t;
    }
};

}  // namespace adm
