// Multislice forward + magnitude loss + hand-derived adjoint, one workgroup per probe position.
//
// Replaces (reference paths): adorym/propagate.py:195-280 (multislice_propagate_batch),
// adorym/forward_model.py:313-375 (tile gather, norm), :88-93 (LSQ loss) and the autograd backward
// (adorym/wrappers.py:322).  Math: SURVEY.md section 3.4.
//
// Design (gfx950):
//   * the complex Py x Px wavefield of one position lives in LDS (pitch P+1 complex => both
//     row-wise and column-wise lane mappings are bank-conflict-free for ds_read/write_b64);
//   * every 1-D transform of length N = R1*R2 is two register-resident radix passes; a wave owns
//     LPW = 64/G whole lines (G = max(R1,R2) threads per line), so the R1 -> R2 exchange is
//     wave-local (no workgroup barrier); only the row<->column ownership change needs __syncthreads:
//     two barriers per slice-to-slice propagation;
//   * forward transforms are decimation-in-frequency and leave the spectrum in a digit-scrambled
//     order, inverse transforms consume that order, so (last forward pass, H multiply, first inverse
//     pass) and (last inverse pass, slice modulation, first forward pass) each run in registers;
//   * the thread -> (ky, kx) mapping is static, so each thread keeps its R2 transfer-function
//     values (pre-divided by Py*Px) in VGPRs for the whole kernel: H costs no memory traffic;
//   * post-modulation wavefields psi'_s are stashed to HBM in thread-native order (perfectly
//     coalesced 8-B/lane stores) and read back in the reverse sweep;
//   * delta/beta tile slices are read straight from the slice-major rotated object [Z][Yp][Xp][2];
//     tile gradients are stored per position in the same thread-native order and overlap-added by a
//     separate streaming kernel (tile_accumulate_kernel): no atomics in the slice loop.
#include <hip/hip_runtime.h>
#include "adm_common.h"
#include "adm_fft.h"

namespace adm {

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

template <int N, int R1, int R2> struct Geo {
    static constexpr int G = (R1 > R2) ? R1 : R2;      // threads per line
    static constexpr int LPW = 64 / G;                 // lines per wave
    static constexpr int NWAVES = (N + LPW - 1) / LPW;
    static constexpr int NT = NWAVES * 64;
    static constexpr int PITCH = N + 1;                // complex elements
};


// ---- one line transform pass set (wave-local) ------------------------------------------------
// `fld` points at element 0 of the line, `ES` = element stride along the line (in complex units).

// pass 1 forward: a[n1] holds x[n1*R2 + t]; radix-R1, twiddle, store A'[k1] at (k1*R2 + t)
template <int N, int R1, int R2>
__device__ __forceinline__ void p1_fwd_store(cf (&a)[R1], const cf (&tw)[R1], cf* fld, int es, int t) {
    Dft<R1, false>::run(a);
#pragma unroll
    for (int k = 1; k < R1; ++k) a[k] = cmul(a[k], tw[k]);
#pragma unroll
    for (int k = 0; k < R1; ++k) fld[(k * R2 + t) * es] = a[k];
}
template <int N, int R1, int R2>
__device__ __forceinline__ void p1_load(cf (&a)[R1], const cf* fld, int es, int t) {
#pragma unroll
    for (int k = 0; k < R1; ++k) a[k] = fld[(k * R2 + t) * es];
}
// pass 1 inverse: a[k1] holds A'[k1] (position k1*R2+t); untwiddle, inverse radix-R1 -> x[n1*R2+t]
template <int N, int R1, int R2>
__device__ __forceinline__ void p1_inv(cf (&a)[R1], const cf (&tw)[R1]) {
#pragma unroll
    for (int k = 1; k < R1; ++k) a[k] = cmulc(a[k], tw[k]);
    Dft<R1, true>::run(a);
}
template <int R2> __device__ __forceinline__ void p2_load(cf (&b)[R2], const cf* fld, int es, int t) {
#pragma unroll
    for (int k = 0; k < R2; ++k) b[k] = fld[(t * R2 + k) * es];
}
template <int R2> __device__ __forceinline__ void p2_store(const cf (&b)[R2], cf* fld, int es, int t) {
#pragma unroll
    for (int k = 0; k < R2; ++k) fld[(t * R2 + k) * es] = b[k];
}

// position p = k1*R2 + k2 of a scrambled spectrum holds frequency k1 + R1*k2
template <int R1, int R2> __device__ __forceinline__ int freq_of_pos(int p) { return p / R2 + R1 * (p % R2); }

template <int N, int R1, int R2> struct Ctx {
    using GE = Geo<N, R1, R2>;
    cf* fld;          // LDS field [N][PITCH]
    int line, t;      // line owned in the current role (row for x passes, column for y passes), thread in line
    bool act1, act2;  // pass-1 role (t < R2) / pass-2 role (t < R1) active
    cf tw[R1];        // W_N^(t*k1)
    cf hs[R2];        // H[ky = t + R1*k2][kx(line)] / N^2   (pass-2 role)
};

// forward x passes starting from registers `a` (row-role, pass-1), ends with the scrambled x spectrum in LDS
template <int N, int R1, int R2>
__device__ __forceinline__ void x_fwd(Ctx<N, R1, R2>& c, cf (&a)[R1]) {
    using GE = Geo<N, R1, R2>;
    cf* row = c.fld + c.line * GE::PITCH;
    if (c.act1) p1_fwd_store<N, R1, R2>(a, c.tw, row, 1, c.t);
    WAVE_SYNC();
    if (c.act2) {
        cf b[R2];
        p2_load<R2>(b, row, 1, c.t);
        Dft<R2, false>::run(b);
        p2_store<R2>(b, row, 1, c.t);
    }
}
// inverse x passes, ends with the real-space row elements in registers `a`
template <int N, int R1, int R2>
__device__ __forceinline__ void x_inv(Ctx<N, R1, R2>& c, cf (&a)[R1]) {
    using GE = Geo<N, R1, R2>;
    cf* row = c.fld + c.line * GE::PITCH;
    if (c.act2) {
        cf b[R2];
        p2_load<R2>(b, row, 1, c.t);
        Dft<R2, true>::run(b);
        p2_store<R2>(b, row, 1, c.t);
    }
    WAVE_SYNC();
    if (c.act1) {
        p1_load<N, R1, R2>(a, row, 1, c.t);
        p1_inv<N, R1, R2>(a, c.tw);
    }
}
// forward y pass 1 (column role)
template <int N, int R1, int R2>
__device__ __forceinline__ void y_fwd_p1(Ctx<N, R1, R2>& c) {
    using GE = Geo<N, R1, R2>;
    cf* col = c.fld + c.line;
    if (c.act1) {
        cf a[R1];
        p1_load<N, R1, R2>(a, col, GE::PITCH, c.t);
        p1_fwd_store<N, R1, R2>(a, c.tw, col, GE::PITCH, c.t);
    }
    WAVE_SYNC();
}
template <int N, int R1, int R2>
__device__ __forceinline__ void y_inv_p1(Ctx<N, R1, R2>& c) {
    using GE = Geo<N, R1, R2>;
    cf* col = c.fld + c.line;
    WAVE_SYNC();
    if (c.act1) {
        cf a[R1];
        p1_load<N, R1, R2>(a, col, GE::PITCH, c.t);
        p1_inv<N, R1, R2>(a, c.tw);
#pragma unroll
        for (int k = 0; k < R1; ++k) col[(k * R2 + c.t) * GE::PITCH] = a[k];
    }
}

// psi <- IFFT2( Hmul * FFT2(psi) ), psi in registers `a` (row role) on entry and exit.
// CONJ: multiply by conj(hs) (adjoint).  `hs` already carries the 1/N^2 of the inverse.
template <int N, int R1, int R2, bool CONJ>
__device__ __forceinline__ void convolve(Ctx<N, R1, R2>& c, cf (&a)[R1], const cf (&hs)[R2]) {
    using GE = Geo<N, R1, R2>;
    x_fwd<N, R1, R2>(c, a);
    __syncthreads();
    y_fwd_p1<N, R1, R2>(c);
    if (c.act2) {
        cf* col = c.fld + c.line;
        cf b[R2];
        p2_load<R2>(b, col, GE::PITCH, c.t);
        Dft<R2, false>::run(b);
#pragma unroll
        for (int k = 0; k < R2; ++k) b[k] = cmul_t<CONJ>(b[k], hs[k]);
        Dft<R2, true>::run(b);
        p2_store<R2>(b, col, GE::PITCH, c.t);
    }
    y_inv_p1<N, R1, R2>(c);
    __syncthreads();
    x_inv<N, R1, R2>(c, a);
}

// unnormalised 2-D transform of registers `a` up to (and including) the last y pass, result left in
// registers b[k2] of the pass-2 column role = spectrum at (ky = t + R1*k2, kx = freq_of_pos(line)).
// INV selects the inverse (conjugate) transform.  Because forward/inverse butterflies are separate
// template instances, the inverse direction is implemented with the conjugation identity
// IDFT(x) = conj(DFT(conj(x))) applied by the caller.
template <int N, int R1, int R2>
__device__ __forceinline__ void fft2_to_regs(Ctx<N, R1, R2>& c, cf (&a)[R1], cf (&b)[R2]) {
    using GE = Geo<N, R1, R2>;
    x_fwd<N, R1, R2>(c, a);
    __syncthreads();
    y_fwd_p1<N, R1, R2>(c);
    if (c.act2) {
        p2_load<R2>(b, c.fld + c.line, GE::PITCH, c.t);
        Dft<R2, false>::run(b);
    }
}
// the matching unnormalised inverse, from registers b[k2] back to real-space registers a
template <int N, int R1, int R2>
__device__ __forceinline__ void ifft2_from_regs(Ctx<N, R1, R2>& c, cf (&b)[R2], cf (&a)[R1]) {
    using GE = Geo<N, R1, R2>;
    if (c.act2) {
        Dft<R2, true>::run(b);
        p2_store<R2>(b, c.fld + c.line, GE::PITCH, c.t);
    }
    y_inv_p1<N, R1, R2>(c);
    __syncthreads();
    x_inv<N, R1, R2>(c, a);
}

__device__ __forceinline__ cf conjf2(cf a) { return make_float2(a.x, -a.y); }

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

// exp(-k1*beta) * (cos, sin)(-sigma*k1*delta)      (adorym/wrappers.py:600-608)
__device__ __forceinline__ cf modulator(float2 db, float k1, float sigma) {
    const float e = exp_fast(-k1 * db.y);
    float sn, cs;
    sincos_fast(-sigma * k1 * db.x, sn, cs);
    return make_float2(e * cs, e * sn);
}

// sum of the (delta, beta) pairs of the slices of one modulation step for this thread's R1 pixels.
// BIN1 (binning == 1): pure loads, so the caller can issue them one step ahead and let the
// propagation hide their latency.
template <int R1, int R2, bool BIN1>
__device__ __forceinline__ void load_db(float2 (&db)[R1], const float2* __restrict__ base, size_t slice_stride, int step,
                                        int binning, int Z) {
    if (BIN1) {
        const float2* q = base + (size_t)step * slice_stride;
#pragma unroll
        for (int k = 0; k < R1; ++k) db[k] = q[k * R2];
    } else {
        const int s_lo = step * binning;
        const int s_hi = min(s_lo + binning, Z);
#pragma unroll
        for (int k = 0; k < R1; ++k) db[k] = make_float2(0.f, 0.f);
        for (int s = s_lo; s < s_hi; ++s) {
            const float2* q = base + (size_t)s * slice_stride;
#pragma unroll
            for (int k = 0; k < R1; ++k) {
                const float2 v = q[k * R2];
                db[k].x += v.x;
                db[k].y += v.y;
            }
        }
    }
}

template <int N, int R1, int R2, bool BIN1>
__global__ __launch_bounds__((Geo<N, R1, R2>::NT)) void ms_fwd_adj_kernel(MsParams p) {
    using GE = Geo<N, R1, R2>;
    __shared__ cf fld[N * GE::PITCH];
    __shared__ float red[GE::NWAVES];

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int li = lane / GE::G;
    Ctx<N, R1, R2> c;
    c.fld = fld;
    c.t = lane % GE::G;
    c.line = wave * GE::LPW + li;
    const bool line_ok = (li < GE::LPW) && (c.line < N);
    c.act1 = line_ok && (c.t < R2);
    c.act2 = line_ok && (c.t < R1);
    if (!line_ok) c.line = 0;   // keep addresses in range for inactive lanes
    const int b = blockIdx.x;

    // ---- static per-thread constants ----
#pragma unroll
    for (int k = 0; k < R1; ++k) c.tw[k] = p.twid[(c.t * k) % N];
    const int kx = freq_of_pos<R1, R2>(c.line);
    // 1/N^2 of the inverse transform is folded into H with ONE rounding per element (divide in
    // double): multiplying by fl(1/N^2) would scale every propagation by the same (1+eps) and the
    // bias would grow linearly with the number of slices.
    const double n2 = (double)(N * N);
#pragma unroll
    for (int k = 0; k < R2; ++k) {
        int ky = (c.t % R1) + R1 * k;
        cf h = p.h[ky * N + kx];
        c.hs[k] = make_float2((float)((double)h.x / n2), (float)((double)h.y / n2));
    }

    const int2 ps = p.pos[b];
    const int py = ps.x + p.pad_y0, px = ps.y + p.pad_x0;
    // element n1 of this thread: pixel (row = line, col = n1*R2 + t) of the tile
    const size_t slice_stride = (size_t)p.Yp * p.Xp;
    const size_t tile_off = (size_t)(py + c.line) * p.Xp + px + c.t;   // + n1*R2 + slice*slice_stride
    const bool do_grad = (p.want_grad != 0);
    float2* stash = p.stash + (size_t)b * p.n_steps * R1 * GE::NT + tid;
    float2* gtile = p.gtile + (size_t)b * p.n_steps * R1 * GE::NT + tid;

    cf a[R1];
#pragma unroll
    for (int k = 0; k < R1; ++k) a[k] = c.act1 ? p.probe[c.line * N + k * R2 + c.t] : make_float2(0.f, 0.f);

    // ================= forward sweep =================
    const float2* tile_base = p.obj_rot + tile_off;
    float2 db[R1];
    if (c.act1) load_db<R1, R2, BIN1>(db, tile_base, slice_stride, 0, p.binning, p.Z);
    for (int step = 0; step < p.n_steps; ++step) {
        if (c.act1) {
#pragma unroll
            for (int k = 0; k < R1; ++k) {
#ifndef ADM_ABL_NOMOD
                a[k] = cmul(a[k], modulator(db[k], p.k1, p.sigma));
#else
                a[k] = cmul(a[k], db[k]);
#endif
#ifndef ADM_ABL_NOSTASH
                if (do_grad) stash[(size_t)(step * R1 + k) * GE::NT] = a[k];
#endif
            }
            // next step's tile slice is requested before the propagation so its latency is hidden
            if (step + 1 < p.n_steps) load_db<R1, R2, BIN1>(db, tile_base, slice_stride, step + 1, p.binning, p.Z);
        }
#ifndef ADM_ABL_NOCONV
        if (step < p.n_steps - 1) convolve<N, R1, R2, false>(c, a, c.hs);
#endif
    }

    // ================= detector plane: loss and dL/dpsi_exit =================
    float lsum = 0.f;
    if (p.det_mode == ADM_DET_FRESNEL_) {
        cf hf[R2];
#pragma unroll
        for (int k = 0; k < R2; ++k) {
            int ky = (c.t % R1) + R1 * k;
            cf h = p.hfree[ky * N + kx];
            hf[k] = make_float2((float)((double)h.x / n2), (float)((double)h.y / n2));
        }
        convolve<N, R1, R2, false>(c, a, hf);
    }
    if (p.det_mode == ADM_DET_FARFIELD_) {
        // Psi = scale * F(psi)  (F forward, or inverse via conjugation when det_inverse)
        cf bb[R2];
        if (p.det_inverse) {
#pragma unroll
            for (int k = 0; k < R1; ++k) a[k] = conjf2(a[k]);
        }
        fft2_to_regs<N, R1, R2>(c, a, bb);
        if (c.act2) {
            const int mx = (kx + N / 2) % N;
#pragma unroll
            for (int k = 0; k < R2; ++k) {
                const int ky = c.t + R1 * k;
                const int my = (ky + N / 2) % N;
                const size_t di = ((size_t)b * N + my) * N + mx;
                cf psi = cscale(bb[k], p.det_scale);           // (conjugated when det_inverse; |.| unaffected)
                float mag = sqrtf(psi.x * psi.x + psi.y * psi.y);
                float diff = mag - p.target[di];
                lsum += diff * diff;
                if (p.pred) p.pred[di] = mag;
                float g = (mag > 0.f) ? p.grad_scale * diff / mag : 0.f;
                // adjoint of (scale * F): scale * F^H; with bb conjugated both ways the same code serves
                bb[k] = cscale(psi, g * p.det_scale);
            }
        }
        if (do_grad) {
            ifft2_from_regs<N, R1, R2>(c, bb, a);
            if (p.det_inverse) {
#pragma unroll
                for (int k = 0; k < R1; ++k) a[k] = conjf2(a[k]);
            }
        }
    } else {
        // near field / Fresnel: the detector field is `a` in real space
        if (c.act1) {
#pragma unroll
            for (int k = 0; k < R1; ++k) {
                const size_t di = ((size_t)b * N + c.line) * N + k * R2 + c.t;
                float mag = sqrtf(a[k].x * a[k].x + a[k].y * a[k].y);
                float diff = mag - p.target[di];
                lsum += diff * diff;
                if (p.pred) p.pred[di] = mag;
                float g = (mag > 0.f) ? p.grad_scale * diff / mag : 0.f;
                a[k] = cscale(a[k], g);
            }
        }
        if (do_grad && p.det_mode == ADM_DET_FRESNEL_) {
            cf hf[R2];
#pragma unroll
            for (int k = 0; k < R2; ++k) {
                int ky = (c.t % R1) + R1 * k;
                cf h = p.hfree[ky * N + kx];
                hf[k] = make_float2((float)((double)h.x / n2), (float)((double)h.y / n2));
            }
            convolve<N, R1, R2, true>(c, a, hf);
        }
    }
    // block reduction of the loss
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lsum += __shfl_down(lsum, off, 64);
    if (lane == 0) red[wave] = lsum;
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
        for (int w = 0; w < GE::NWAVES; ++w) s += red[w];
        p.loss_sum[b] = s;
    }
    if (!do_grad) return;

    // ================= reverse sweep =================
    const float sk1 = p.sigma * p.k1;
    cf psi[R1];
    if (c.act1) {
        load_db<R1, R2, BIN1>(db, tile_base, slice_stride, p.n_steps - 1, p.binning, p.Z);
#pragma unroll
        for (int k = 0; k < R1; ++k) psi[k] = stash[(size_t)((p.n_steps - 1) * R1 + k) * GE::NT];
    }
    for (int step = p.n_steps - 1; step >= 0; --step) {
        if (c.act1) {
#pragma unroll
            for (int k = 0; k < R1; ++k) {
                // z = conj(G) * psi'
                const float zr = a[k].x * psi[k].x + a[k].y * psi[k].y;
                const float zi = a[k].x * psi[k].y - a[k].y * psi[k].x;
                const float gd = sk1 * zi;
                const float gb = -p.k1 * zr;
                // tile gradient of this modulation step, thread-native layout (coalesced 8-B/lane store);
                // adm_tile_grad_accumulate overlap-adds the tiles afterwards (no atomics in this loop)
                gtile[(size_t)(step * R1 + k) * GE::NT] = make_float2(gd, gb);
#ifndef ADM_ABL_NOMOD
                a[k] = cmulc(a[k], modulator(db[k], p.k1, p.sigma));
#else
                a[k] = cmulc(a[k], db[k]);
#endif
            }
            if (step > 0) {
                load_db<R1, R2, BIN1>(db, tile_base, slice_stride, step - 1, p.binning, p.Z);
#pragma unroll
#ifndef ADM_ABL_NOSTASH
                for (int k = 0; k < R1; ++k) psi[k] = stash[(size_t)((step - 1) * R1 + k) * GE::NT];
#else
                for (int k = 0; k < R1; ++k) psi[k] = a[k];
#endif
            }
        }
#ifndef ADM_ABL_NOCONV
        if (step > 0) convolve<N, R1, R2, true>(c, a, c.hs);
#endif
    }
    if (p.grad_probe && c.act1) {
#pragma unroll
        for (int k = 0; k < R1; ++k) {
            float* gp = reinterpret_cast<float*>(p.grad_probe + c.line * N + k * R2 + c.t);
            atomicAdd(gp, a[k].x);
            atomicAdd(gp + 1, a[k].y);
        }
    }
}

template <int N, int R1, int R2> static hipError_t launch(const MsParams& p, int batch, hipStream_t st) {
    using GE = Geo<N, R1, R2>;
    if (p.binning == 1) hipLaunchKernelGGL((ms_fwd_adj_kernel<N, R1, R2, true>), dim3(batch), dim3(GE::NT), 0, st, p);
    else hipLaunchKernelGGL((ms_fwd_adj_kernel<N, R1, R2, false>), dim3(batch), dim3(GE::NT), 0, st, p);
    return hipGetLastError();
}

int ms_threads_for(int n) {
    switch (n) {
        case 12: return Geo<12, 3, 4>::NT;
        case 16: return Geo<16, 4, 4>::NT;
        case 32: return Geo<32, 4, 8>::NT;
        case 64: return Geo<64, 8, 8>::NT;
        case 72: return Geo<72, 8, 9>::NT;
        default: return 0;
    }
}
int ms_r2_for(int n) {
    switch (n) {
        case 12: return 4;
        case 16: return 4;
        case 32: return 8;
        case 64: return 8;
        case 72: return 9;
        default: return 0;
    }
}
int ms_r1_for(int n) {
    switch (n) {
        case 12: return 3;
        case 16: return 4;
        case 32: return 4;
        case 64: return 8;
        case 72: return 8;
        default: return 0;
    }
}

hipError_t ms_launch(int n, const MsParams& p, int batch, hipStream_t st) {
    switch (n) {
        case 12: return launch<12, 3, 4>(p, batch, st);
        case 16: return launch<16, 4, 4>(p, batch, st);
        case 32: return launch<32, 4, 8>(p, batch, st);
        case 64: return launch<64, 8, 8>(p, batch, st);
        case 72: return launch<72, 8, 9>(p, batch, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace adm
