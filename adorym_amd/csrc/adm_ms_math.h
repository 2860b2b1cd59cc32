// Device helpers of the multislice kernels (adm_multislice.hip): wave-local LDS ordering,
// the loss terms, and the branch-free sincos / exp used by the slice modulation.  Device-only, gfx950.
#pragma once
#include <hip/hip_runtime.h>
#include "adm_common.h"
#include "adm_fft.h"

namespace adm {

// Diagnostic build only (python adorym_amd/csrc/build.py --variant stamps -DADM_STAMPS; tools/stamps.py): per-wave shader-clock
// stamps of one slice step of workgroup 0.  In the production build the three macros expand to nothing.  Every stamp is
// itself a scalar-memory read plus a global store, so it perturbs what it measures (each costs the wave ~100 cycles and
// the stores of the 11 waves queue in the CU's store path): read phase ORDER and skew from it, not absolute store costs.
#ifdef ADM_STAMPS
#define ADM_STAMP_DECL bool stamp_on = false
#define ADM_STAMP_ON(cond) do { stamp_on = (cond); } while (0)
#define ADM_STAMP(i) do { if (stamp_on && (threadIdx.x & 63) == 0) g_stamps[(threadIdx.x >> 6) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ADM_STAMP_DECL
#define ADM_STAMP_ON(cond) do { } while (0)
#define ADM_STAMP(i) do { } while (0)
#endif

#ifdef ADM_SAFE_SYNC
#define WAVE_SYNC() __syncthreads()
#else
// LDS operations of one wave execute in order; this only stops the compiler from reordering.
#define WAVE_SYNC()                                        \
    do {                                                   \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                   \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)
#endif


// LDS layout of the P x P field: element (y, x) lives at  y*Q + posx(x),  posx(x) = (x / R2)*PA + (x % R2)*PB.
// (PA, PB, Q) = (R2, 1, N+1) is the plain padded row-major image; other triples permute the elements (rows may
// interleave as long as the map stays injective) to spread the stride-R2 accesses of the radix passes over the LDS
// banks.  P = 72: (1, 65, 73) is the best of the exhaustive conflict count of tools/lds/lds_layout_search.c over
// Q <= 116 (2889 LDS-array cycles per pass set against 3070 for the earlier (2, 9, 113) and 3777 for the plain padded
// image) and is also the smallest: 45.7 KB instead of 64.9 KB, which is what lets two workgroups share a CU.
template <int N, int R1, int R2> struct Layout {
    static constexpr int PA = R2, PB = 1, Q = N + 1;
};
template <> struct Layout<72, 8, 9> {
    static constexpr int PA = 1, PB = 65, Q = 73;
};
// P = 64 (BASELINE configs 1 and 2): the smallest of the best layouts of the same search under the instruction-time model
// (`lds_layout_search 64 8 8 100 1`): 4736 LDS cycles per pass set against 6912 for the plain padded image (8, 1, 65).
template <> struct Layout<64, 8, 8> {
    static constexpr int PA = 4, PB = 34, Q = 67;
};

template <int N, int R1, int R2> struct Geo {
    static constexpr int G = (R1 > R2) ? R1 : R2;      // threads per line
    static constexpr int LPW = 64 / G;                 // lines per wave
    static constexpr int NWAVES = (N + LPW - 1) / LPW;
    static constexpr int NT = NWAVES * 64;
    static constexpr int PA = Layout<N, R1, R2>::PA, PB = Layout<N, R1, R2>::PB, Q = Layout<N, R1, R2>::Q;
    // complex elements of one LDS field image: last row's origin + the largest in-row offset + 1
    static constexpr int FLD = (N - 1) * Q + ((N - 1) / R2) * PA + (R2 - 1) * PB + 1;
    // strides (in complex elements) of the two access patterns in the two roles
    static constexpr int ROW_P1_K = PA, ROW_P1_T = PB;           // element k*R2 + t of a row
    static constexpr int ROW_P2_K = PB, ROW_P2_T = PA;           // element t*R2 + k of a row
    static constexpr int COL_P1_K = R2 * Q, COL_P1_T = Q;        // element k*R2 + t of a column
    static constexpr int COL_P2_K = Q, COL_P2_T = R2 * Q;        // element t*R2 + k of a column
    static __device__ __forceinline__ int posx(int x) { return (x / R2) * PA + (x % R2) * PB; }
};

// ---- one line transform pass set (wave-local) ------------------------------------------------
// `base` points at the line's origin; element index -> address through the (KS, TS) strides above.

template <int R, int KS> __device__ __forceinline__ void ld_line(cf (&a)[R], const cf* base) {
#pragma unroll
    for (int k = 0; k < R; ++k) a[k] = base[k * KS];
}
template <int R, int KS> __device__ __forceinline__ void st_line(const cf (&a)[R], cf* base) {
#pragma unroll
    for (int k = 0; k < R; ++k) base[k * KS] = a[k];
}
// Workspace rows (stored wavefields, per-position tile gradients): one row = the R1 elements of each of the NT threads
// of a modulation step.  R1 even: the elements of a thread are stored in pairs, [k/2][tid][2], so that ONE 16-byte access
// per lane moves two of them -- the CU issues vector-memory instructions at a fixed cost per wave-instruction (~40
// cycles each when the 11 waves issue together, tools/stamps.py), so halving their number halves that phase.
// R1 odd: [k][tid].  tile_accumulate (adm_object.hip) addresses the tile gradients with ws_elem_offset().
__host__ __device__ __forceinline__ unsigned ws_elem_offset(int R1, int NT, int k, int tid) {
    return (R1 % 2 == 0) ? (unsigned)(((k >> 1) * NT + tid) * 2 + (k & 1)) : (unsigned)(k * NT + tid);
}
template <int R1, int K0 = 0, int K1 = R1> __device__ __forceinline__ void ws_store(float2* row, int NT, int tid, const cf (&v)[R1]) {
    if (R1 % 2 == 0) {
        float4* r4 = reinterpret_cast<float4*>(row);
#pragma unroll
        for (int k = K0; k < K1; k += 2) r4[(size_t)(k >> 1) * NT + tid] = make_float4(v[k].x, v[k].y, v[k + 1].x, v[k + 1].y);
    } else {
#pragma unroll
        for (int k = K0; k < K1; ++k) row[(size_t)k * NT + tid] = v[k];
    }
}
template <int R1, int K0 = 0, int K1 = R1> __device__ __forceinline__ void ws_load(const float2* row, int NT, int tid, cf (&v)[R1]) {
    if (R1 % 2 == 0) {
        const float4* r4 = reinterpret_cast<const float4*>(row);
#pragma unroll
        for (int k = K0; k < K1; k += 2) {
            const float4 q = r4[(size_t)(k >> 1) * NT + tid];
            v[k] = make_float2(q.x, q.y);
            v[k + 1] = make_float2(q.z, q.w);
        }
    } else {
#pragma unroll
        for (int k = K0; k < K1; ++k) v[k] = row[(size_t)k * NT + tid];
    }
}

// position p = k1*R2 + k2 of a scrambled spectrum holds frequency k1 + R1*k2
template <int R1, int R2> __device__ __forceinline__ int freq_of_pos(int p) { return p / R2 + R1 * (p % R2); }

__device__ __forceinline__ cf conjf2(cf a) { return make_float2(a.x, -a.y); }

// Per-pixel loss term and the factor g with dL/dPsi = g * Psi (adorym/forward_model.py:88-103):
//   LSQ      term = (|Psi| - t)^2,                         g = grad_scale * (|Psi| - t) / |Psi|     (0 at |Psi| = 0)
//   Poisson  term = |Psi|^2 pm - t pm log(|Psi|^2 pm),     g = grad_scale * pm * (1 - t / |Psi|^2)
// grad_scale = 2 / (minibatch * Py * Px) in both cases.
__device__ __forceinline__ float loss_term(float mag, float t, const MsParams& p, float& g) {
    if (p.loss_type == 0) {
        const float diff = mag - t;
        g = (mag > 0.f) ? p.grad_scale * diff / mag : 0.f;
        return diff * diff;
    }
    const float inten = mag * mag;
    g = p.grad_scale * p.poisson_mult * (1.f - t / inten);
    return inten * p.poisson_mult - t * p.poisson_mult * logf(inten * p.poisson_mult);
}

// Branch-free single-precision sin/cos: 3-term Cody-Waite reduction by pi/2 (exact product steps via
// fma, good for |x| < ~1e5) + Cephes minimax polynomials on [-pi/4, pi/4] (~1 ulp).  ocml's sincosf
// is equally accurate but costs several hundred instructions and dozens of branches per call, which
// made the slice modulation as expensive as the FFTs.
__device__ __forceinline__ void sincos_fast(float x, float& sn, float& cs) {
    const float q = rintf(x * 0.63661977236758134308f);
    float r = fmaf(q, -1.57079601287841796875f, x);
    r = fmaf(q, -3.1391647326017846353e-7f, r);
    r = fmaf(q, -5.3903025299577647655e-15f, r);
    const int iq = (int)q;
    const float r2 = r * r;
    float sp = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    sp = fmaf(sp, r2, -1.6666654611e-1f);
    sp = fmaf(sp * r2, r, r);
    float cp = fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    cp = fmaf(cp, r2, 4.166664568298827e-2f);
    cp = fmaf(cp * r2, r2, fmaf(r2, -0.5f, 1.0f));
    const bool swap = (iq & 1) != 0;
    const float s0 = swap ? cp : sp;
    const float c0 = swap ? sp : cp;
    sn = (iq & 2) ? -s0 : s0;
    cs = ((iq + 1) & 2) ? -c0 : c0;
}
// exp(x) ~1 ulp: 2^(x*log2e) with the product's rounding error and the low part of log2(e)
// re-injected to first order; v_exp_f32 itself is a 1-ulp instruction.
__device__ __forceinline__ float exp_fast(float x) {
    const float L2E = 1.44269502162933349609375f;
    const float t = x * L2E;
    float e = fmaf(x, L2E, -t);
    e = fmaf(x, 1.925963033500e-8f, e);
    const float r = __builtin_amdgcn_exp2f(t);
    return fmaf(r, e * 0.69314718055994530942f, r);
}

__device__ __forceinline__ float loss_term_nz(float mag, float t, const MsParams& p, float& g) {
    // multi-mode variant: pred = sqrt(sum_m |Psi_m|^2); like the reference there is no guard at pred = 0
    if (p.loss_type == 0) {
        const float diff = mag - t;
        g = p.grad_scale * diff / mag;
        return diff * diff;
    }
    const float inten = mag * mag;
    g = p.grad_scale * p.poisson_mult * (1.f - t / inten);
    return inten * p.poisson_mult - t * p.poisson_mult * logf(inten * p.poisson_mult);
}

// exp(-k1*beta) * (cos, sin)(-sigma*k1*delta) for the R1 pixels of a thread (adorym/wrappers.py:600-608): the same
// X-ray phase shifts per slice are tiny (k1*delta ~ 1e-2): when every lane of the wave is inside [-pi/4, pi/4] the
// range reduction and quadrant selection of sincos_fast are skipped (same polynomials => bit-identical results).  The
// wave-uniform test is taken ONCE for the R1 elements, so the elements are independent straight-line chains: with one
// branch per element (round 1) the modulation phase of a slice step ran at ~9 cycles per instruction.
// a[k] <- a[k] * m_k (CONJ: a[k] * conj(m_k)).
template <int R1, bool CONJ> __device__ __forceinline__ void modulate(cf (&a)[R1], const float2 (&db)[R1], float k1, float sigma) {
    bool big = false;
#pragma unroll
    for (int k = 0; k < R1; ++k) big = big || (fabsf(sigma * k1 * db[k].x) > 0.78539816f);
    const bool small = (__builtin_amdgcn_ballot_w64(big) == 0);
    // groups of four elements: enough independent chains to cover the ALU latency without holding the
    // temporaries of all R1 elements at once (this point is the register-pressure peak of the slice loop)
    constexpr int GRP = 4;
#pragma unroll
    for (int k0 = 0; k0 < R1; k0 += GRP) {
#pragma unroll
        for (int k = k0; k < k0 + GRP && k < R1; ++k) {
            const float phi = -sigma * k1 * db[k].x;
            float sn, cs;
            if (small) {
                const float r2 = phi * phi;
                float sp = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
                sp = fmaf(sp, r2, -1.6666654611e-1f);
                sn = fmaf(sp * r2, phi, phi);
                float cp = fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
                cp = fmaf(cp, r2, 4.166664568298827e-2f);
                cs = fmaf(cp * r2, r2, fmaf(r2, -0.5f, 1.0f));
            } else {
                sincos_fast(phi, sn, cs);
            }
            const float e = exp_fast(-k1 * db[k].y);
            a[k] = cmul_t<CONJ>(a[k], make_float2(e * cs, e * sn));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// The factor modulate<> multiplies with, for ONE voxel -- what adm_rotate_fwd stores per rotated-frame voxel when the
// plan caches the slice transmissions.  Always the general sincos path: for |phi| <= pi/4 its reduction is the identity
// (q = 0, r = fma(0, c, phi) = phi), so the value is bit-identical to either path of modulate<>.
__device__ __forceinline__ float2 slice_transmission(float2 db, float k1, float sigma) {
    float sn, cs;
    sincos_fast(-sigma * k1 * db.x, sn, cs);
    const float e = exp_fast(-k1 * db.y);
    return make_float2(e * cs, e * sn);
}

}  // namespace adm
