// Small-radix complex DFT butterflies and complex arithmetic for the multislice kernels.  Device-only, gfx950.
// All loops are compile-time unrolled so the per-thread element arrays live in VGPRs.
//
// Conventions:  forward DFT  X[k] = sum_n x[n] exp(-2 pi i n k / N)   (torch.fft.fft2, norm=None)
//               inverse here is the UNNORMALISED conjugate transform; 1/N factors are folded
//               into the transfer-function registers by the caller.
//
// Packed fp32.  A complex number is an aligned VGPR pair, and gfx950's v_pk_{add,mul,fma}_f32 work on such a pair with
// per-half source selection (op_sel / op_sel_hi) and negation (neg_lo / neg_hi) for free.  Complex add / subtract, the
// multiplications by -+i that radix-4 / radix-8 butterflies need, "m +- i h d" of a radix-3 butterfly and a full complex
// multiply (2 instructions) are therefore one or two vector instructions instead of two to four.  A packed instruction
// occupies the SIMD twice as long as a scalar one, so at saturation the rates are equal -- but this kernel is not at
// saturation: its waves are bound by their own issue interval, which tools/micro/pk_rate.hip measures as the SAME for a
// packed and a scalar instruction when one to three waves share a SIMD (3.4 vs 3.8 ns alone; 2.0 ns per packed = 1.0 ns
// per operation against 1.7 ns per scalar instruction at three waves).  Halving the instruction count is what counts.
// (Round 1 found compiler SLP packing a loss: the auto-vectoriser shuffled halves with v_mov; here every swizzle rides on
// an instruction's modifiers.  The library is still built with -fno-slp-vectorize.)
#pragma once
#include <hip/hip_runtime.h>

namespace adm {

typedef float2 cf;
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f V(cf a) { return __builtin_bit_cast(v2f, a); }
__device__ __forceinline__ cf C(v2f a) { return __builtin_bit_cast(cf, a); }

__device__ __forceinline__ cf cadd(cf a, cf b) { return C(V(a) + V(b)); }
__device__ __forceinline__ cf csub(cf a, cf b) { return C(V(a) - V(b)); }
__device__ __forceinline__ cf cscale(cf a, float s) { return C(V(a) * s); }
// a + s*b, real s
__device__ __forceinline__ cf caxpy(cf a, float s, cf b) { return C(__builtin_elementwise_fma(V(b), (v2f){s, s}, V(a))); }

// a * b:  t = (a.x b.x, a.x b.y);  r = (-a.y b.y + t.x, a.y b.x + t.y)
// (ONE asm statement for the two instructions, and no temporary shared between consecutive products: between two asm
// statements of which the second reads or writes a register the first writes, the compiler's gfx950 hazard recogniser puts
// an s_nop -- it cannot see that a v_pk_* has no dst_sel forwarding hazard -- which was ~45 wasted issue slots per wave and
// propagation.)
__device__ __forceinline__ cf cmul(cf a, cf b) {
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
        : "=&v"(r) : "v"(V(a)), "v"(V(b)));
    return C(r);
}
// a * conj(b):  t = (a.x b.x, -a.x b.y);  r = (a.y b.y + t.x, a.y b.x + t.y)
__device__ __forceinline__ cf cmulc(cf a, cf b) {
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1] neg_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1]"
        : "=&v"(r) : "v"(V(a)), "v"(V(b)));
    return C(r);
}
// (z.y * c.x, z.x * c.y): the halves of z swapped on the way into a packed multiply; and the same added to g
__device__ __forceinline__ cf mul_swapped(cf z, v2f c) {
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(r) : "v"(V(z)), "v"(c));
    return C(r);
}
__device__ __forceinline__ cf fma_swapped(cf z, v2f c, cf g) {
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(V(z)), "v"(c), "v"(V(g)));
    return C(r);
}
template <bool CONJ> __device__ __forceinline__ cf cmul_t(cf a, cf b) { return CONJ ? cmulc(a, b) : cmul(a, b); }

// multiply by -i (forward) or +i (inverse)
template <bool INV> __device__ __forceinline__ cf rot90(cf a) {
    return INV ? make_float2(-a.y, a.x) : make_float2(a.y, -a.x);
}
// t + rot90<INV>(d) in one instruction: forward (t.x + d.y, t.y - d.x), inverse (t.x - d.y, t.y + d.x)
template <bool INV> __device__ __forceinline__ cf add_rot(cf t, cf d) {
    v2f r;
    if (INV) asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(V(t)), "v"(V(d)));
    else asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(V(t)), "v"(V(d)));
    return C(r);
}
// t - rot90<INV>(d)
template <bool INV> __device__ __forceinline__ cf sub_rot(cf t, cf d) { return add_rot<!INV>(t, d); }
// m + s * rot90<INV>(d), real s (both halves of sv hold s):  forward (m.x + s d.y, m.y - s d.x)
template <bool INV> __device__ __forceinline__ cf fma_rot(cf m, v2f sv, cf d) {
    v2f r;
    if (INV) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(V(d)), "v"(sv), "v"(V(m)));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(r) : "v"(V(d)), "v"(sv), "v"(V(m)));
    return C(r);
}

// The two irrational butterfly constants, as values (not literals), so that a caller can DITHER them: fl(sqrt(3)/2) and fl(1/sqrt 2)
// are both about 1.8e-8 (relative) below the numbers they stand for, in every butterfly of every pass of every slice, so a chain
// of forward / inverse transforms scales the components they touch by the same (1 - 3.6e-8) per slice and the error of a
// multislice sweep grows linearly with depth: at 256 slices the constants alone made the kernel's fp32 error 2.1-2.4x the
// reference's own (golden F17; with error-compensated constants -- one more multiply-add per use, +9 % kernel time -- 1.4x:
// profiles/r04/r04a_*).  fft_k_dithered(i) returns, for the i-th propagation of a sweep, the fp32 neighbour ABOVE the exact value
// every 4th (sqrt(3)/2: 3 x -1.795e-8 + 5.09e-8 = -0.3e-8 per four slices) or every 5th time (1/sqrt 2: 4 x -1.711e-8 + 6.72e-8),
// and the one below otherwise: every transform is still exact to fp32 rounding, the deficits cancel over a few slices instead
// of adding up, and it costs nothing.
struct FftK {
    float h;      // sqrt(3)/2
    float c;      // 1/sqrt(2)
};
__device__ __forceinline__ FftK fft_k_nominal() { return FftK{0.86602540378443864676f, 0.70710678118654752440f}; }
__device__ __forceinline__ FftK fft_k_dithered(int i) {
#ifdef ADM_NO_DITHER
    return fft_k_nominal();
#endif
    FftK k;
    k.h = ((i & 3) == 3) ? 0.866025447845458984375f : 0.86602538824081420898f;       // fl(h) + 1 ulp : fl(h)
    k.c = (i % 5 == 4) ? 0.707106828689575195312f : 0.70710676908493041992f;         // fl(c) + 1 ulp : fl(c)
    return k;
}

template <int R, bool INV> struct Dft;

template <bool INV> struct Dft<2, INV> {
    static __device__ __forceinline__ void run(cf (&a)[2], FftK = FftK()) {
        cf t = a[0];
        a[0] = cadd(t, a[1]);
        a[1] = csub(t, a[1]);
    }
};

template <bool INV> struct Dft<3, INV> {
    static __device__ __forceinline__ void run(cf& a0, cf& a1, cf& a2, float h = 0.86602540378443864676f) {
        const v2f hv = {h, h};
        const cf s = cadd(a1, a2);
        const cf d = csub(a1, a2);
        const cf m = caxpy(a0, -0.5f, s);
        a0 = cadd(a0, s);
        // forward: m -+ i h d ; inverse: m +- i h d
        a1 = fma_rot<INV>(m, hv, d);
        a2 = fma_rot<!INV>(m, hv, d);
    }
    static __device__ __forceinline__ void run(cf (&a)[3], FftK k = fft_k_nominal()) { run(a[0], a[1], a[2], k.h); }
};

template <bool INV> struct Dft<4, INV> {
    static __device__ __forceinline__ void run(cf& a0, cf& a1, cf& a2, cf& a3) {
        const cf t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), d = csub(a1, a3);
        a0 = cadd(t0, t2);
        a2 = csub(t0, t2);
        a1 = add_rot<INV>(t1, d);      // t1 + (-+i) d
        a3 = sub_rot<INV>(t1, d);
    }
    static __device__ __forceinline__ void run(cf (&a)[4], FftK = FftK()) { run(a[0], a[1], a[2], a[3]); }
};

template <bool INV> struct Dft<8, INV> {
    static __device__ __forceinline__ void run(cf (&a)[8], FftK k = fft_k_nominal()) {
        const float c = k.c;
        cf e0 = a[0], e1 = a[2], e2 = a[4], e3 = a[6];
        cf o0 = a[1], o1 = a[3], o2 = a[5], o3 = a[7];
        Dft<4, INV>::run(e0, e1, e2, e3);
        Dft<4, INV>::run(o0, o1, o2, o3);
        // twiddles W8^k (forward exp(-i pi k/4), inverse conj):
        //   W8^1 o1 = c (o1 + rot90<INV>(o1)),  W8^2 o2 = rot90<INV>(o2),  W8^3 o3 = -c (o3 + rot90<!INV>(o3))
        const cf s1 = add_rot<INV>(o1, o1);
        const cf s3 = add_rot<!INV>(o3, o3);
        a[0] = cadd(e0, o0);
        a[4] = csub(e0, o0);
        a[1] = caxpy(e1, c, s1);
        a[5] = caxpy(e1, -c, s1);
        a[2] = add_rot<INV>(e2, o2);
        a[6] = sub_rot<INV>(e2, o2);
        a[3] = caxpy(e3, -c, s3);
        a[7] = caxpy(e3, c, s3);
    }
};

template <bool INV> struct Dft<9, INV> {
    static __device__ __forceinline__ void run(cf (&a)[9], FftK k = fft_k_nominal()) {
        // n = 3 n1 + n2, k = k1 + 3 k2
        const float c1 = 0.76604444311897803520f, s1 = 0.64278760968653932632f;   // cos/sin(2pi/9)
        const float c2 = 0.17364817766693034885f, s2 = 0.98480775301220805937f;   // cos/sin(4pi/9)
        const float c4 = -0.93969262078590838405f, s4 = 0.34202014332566873304f;  // cos/sin(8pi/9)
        // stage 1: radix-3 over n1 for each n2 -> A[k1][n2] stored at a[3 k1 + n2]
        Dft<3, INV>::run(a[0], a[3], a[6], k.h);
        Dft<3, INV>::run(a[1], a[4], a[7], k.h);
        Dft<3, INV>::run(a[2], a[5], a[8], k.h);
        // twiddle W9^(n2 k1): forward exp(-i..) = (c, -s); inverse (c, +s)
        const cf w1 = make_float2(c1, INV ? s1 : -s1);
        const cf w2 = make_float2(c2, INV ? s2 : -s2);
        const cf w4 = make_float2(c4, INV ? s4 : -s4);
        a[4] = cmul(a[4], w1);   // k1=1, n2=1
        a[5] = cmul(a[5], w2);   // k1=1, n2=2
        a[7] = cmul(a[7], w2);   // k1=2, n2=1
        a[8] = cmul(a[8], w4);   // k1=2, n2=2
        // stage 2: radix-3 over n2 for each k1 -> X[k1 + 3 k2] at a[3 k1 + k2]
        Dft<3, INV>::run(a[0], a[1], a[2], k.h);
        Dft<3, INV>::run(a[3], a[4], a[5], k.h);
        Dft<3, INV>::run(a[6], a[7], a[8], k.h);
        // natural order: X[k1 + 3 k2] <- a[3 k1 + k2]  (3x3 transpose, register renaming only)
        cf t;
        t = a[1]; a[1] = a[3]; a[3] = t;
        t = a[2]; a[2] = a[6]; a[6] = t;
        t = a[5]; a[5] = a[7]; a[7] = t;
    }
};

}  // namespace adm
