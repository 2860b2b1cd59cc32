// Multislice forward + magnitude loss + hand-derived adjoint, one workgroup per probe position.
//
// Replaces (reference paths): adorym/propagate.py:195-280 (multislice_propagate_batch),
// adorym/forward_model.py:313-375 (tile gather, norm), :88-93 (LSQ loss) and the autograd backward
// (adorym/wrappers.py:322).  Math: SURVEY.md section 3.4.
//
// Design (gfx950):
//   * the complex Py x Px wavefield of one position lives in LDS, in the permuted image of Layout<> (adm_ms_math.h;
//     P = 72: pitch 73 with the nine residues of a row 65 elements apart, found by tools/lds/lds_layout_search.c);
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
#include "adm_ms_math.h"
#include <type_traits>
namespace adm {

#ifdef ADM_STAMPS
__device__ unsigned long long g_stamps[16 * 16];     // diagnostic build only, see ADM_STAMP in adm_ms_math.h
#endif


// pass 1 forward: a[n1] holds x[n1*R2 + t]; radix-R1 then twiddle W_N^(t*k1)
template <int R1> __device__ __forceinline__ void p1_fwd(cf (&a)[R1], const cf (&tw)[R1], FftK K) {
    Dft<R1, false>::run(a, K);
#pragma unroll
    for (int k = 1; k < R1; ++k) a[k] = cmul(a[k], tw[k]);
}
// pass 1 inverse: untwiddle, inverse radix-R1 -> x[n1*R2 + t]
template <int R1> __device__ __forceinline__ void p1_inv(cf (&a)[R1], const cf (&tw)[R1], FftK K) {
#pragma unroll
    for (int k = 1; k < R1; ++k) a[k] = cmulc(a[k], tw[k]);
    Dft<R1, true>::run(a, K);
}

template <int N, int R1, int R2> struct Ctx {
    using GE = Geo<N, R1, R2>;
    cf* fld;          // LDS field image
    int line, t;      // line owned in both roles (row for x passes, column for y passes), thread in line
    bool act1, act2;  // pass-1 role (t < R2) / pass-2 role (t < R1) active
    cf tw[R1];        // W_N^(t*k1)
    FftK K;           // butterfly constants of the transform in progress (wave-uniform; the sweeps dither them, adm_fft.h)
    // per-thread base offsets (complex elements) of the four access patterns
    int row_p1, row_p2, col_p1, col_p2;
};

// forward x passes starting from registers `a` (row role, pass 1); ends with the scrambled x spectrum in LDS
template <int N, int R1, int R2>
__device__ __forceinline__ void x_fwd(Ctx<N, R1, R2>& c, cf (&a)[R1]) {
    using GE = Geo<N, R1, R2>;
    if (c.act1) {
        p1_fwd<R1>(a, c.tw, c.K);
        st_line<R1, GE::ROW_P1_K>(a, c.fld + c.row_p1);
    }
    WAVE_SYNC();
    if (c.act2) {
        cf b[R2];
        ld_line<R2, GE::ROW_P2_K>(b, c.fld + c.row_p2);
        Dft<R2, false>::run(b, c.K);
        st_line<R2, GE::ROW_P2_K>(b, c.fld + c.row_p2);
    }
}
// inverse x passes; ends with the real-space row elements in registers `a`
template <int N, int R1, int R2>
__device__ __forceinline__ void x_inv(Ctx<N, R1, R2>& c, cf (&a)[R1]) {
    using GE = Geo<N, R1, R2>;
    if (c.act2) {
        cf b[R2];
        ld_line<R2, GE::ROW_P2_K>(b, c.fld + c.row_p2);
        Dft<R2, true>::run(b, c.K);
        st_line<R2, GE::ROW_P2_K>(b, c.fld + c.row_p2);
    }
    WAVE_SYNC();
    if (c.act1) {
        ld_line<R1, GE::ROW_P1_K>(a, c.fld + c.row_p1);
        p1_inv<R1>(a, c.tw, c.K);
    }
}
// forward y pass 1 (column role)
template <int N, int R1, int R2>
__device__ __forceinline__ void y_fwd_p1(Ctx<N, R1, R2>& c) {
    using GE = Geo<N, R1, R2>;
    if (c.act1) {
        cf a[R1];
        ld_line<R1, GE::COL_P1_K>(a, c.fld + c.col_p1);
        p1_fwd<R1>(a, c.tw, c.K);
        st_line<R1, GE::COL_P1_K>(a, c.fld + c.col_p1);
    }
    WAVE_SYNC();
}
template <int N, int R1, int R2>
__device__ __forceinline__ void y_inv_p1(Ctx<N, R1, R2>& c) {
    using GE = Geo<N, R1, R2>;
    WAVE_SYNC();
    if (c.act1) {
        cf a[R1];
        ld_line<R1, GE::COL_P1_K>(a, c.fld + c.col_p1);
        p1_inv<R1>(a, c.tw, c.K);
        st_line<R1, GE::COL_P1_K>(a, c.fld + c.col_p1);
    }
}

// psi <- IFFT2( Hmul * FFT2(psi) ), psi in registers `a` (row role) on entry and exit.
// CONJ: multiply by conj(H) (adjoint).  H already carries the 1/N^2 of the inverse.  The thread <-> (ky, kx)
// map is static, so H is read from registers `hs` (the slice kernel, or the one-off detector-plane Fresnel kernel).
//
// `hook(P<i>)`, i = 0..3, runs at four points spread over the convolution (after the x passes, after the first barrier and
// the first y pass, after the spectral multiply, after the second barrier).  The sweeps issue the NEXT step's global loads
// there (Prefetch below), a few per point.  Issued in one burst beside the slice modulation and its stores, the 16
// vector-memory instructions per thread of the 11 waves queue in the CU's single address / data path while the waves of a
// SIMD take their turn by age, and the whole workgroup then waits at the first barrier for the youngest wave's burst:
// 2.15 -> 1.81 ms at 32 positions, 2.95 -> 2.50 at 256, from the placement alone (same instructions, same registers).
// Measured placements (kernel ms at 32 positions): everything before the propagation 2.15; everything at P0 / P1 / P2 / P3
// 2.08 / 2.03 / 2.04 / 2.12; reverse loads over two points 1.91-1.93; over three (below) 1.81-1.83; stores moved too: no gain.
template <int I> using P = std::integral_constant<int, I>;
struct Prefetch {     // hook point at which each group of next-step loads is issued
    static constexpr int FWD_DB0 = 0, FWD_DB1 = 2;                  // forward sweep: the two halves of the tile slice
    static constexpr int REV_DB0 = 0, REV_DB1 = 1, REV_PSI = 2;     // reverse sweep: tile slice halves, stored wavefield
};
struct NoHook { template <class T> __device__ __forceinline__ void operator()(T) const {} };
template <int N, int R1, int R2, bool CONJ, class Hook = NoHook>
__device__ __forceinline__ void convolve(Ctx<N, R1, R2>& c, cf (&a)[R1], const cf (&hs)[R2], Hook hook = Hook()) {
    using GE = Geo<N, R1, R2>;
    x_fwd<N, R1, R2>(c, a);
    hook(P<0>());
    __syncthreads();
    y_fwd_p1<N, R1, R2>(c);
    hook(P<1>());
    if (c.act2) {
        cf b[R2];
        ld_line<R2, GE::COL_P2_K>(b, c.fld + c.col_p2);
        Dft<R2, false>::run(b, c.K);
#pragma unroll
        for (int k = 0; k < R2; ++k) b[k] = cmul_t<CONJ>(b[k], hs[k]);
        Dft<R2, true>::run(b, c.K);
        st_line<R2, GE::COL_P2_K>(b, c.fld + c.col_p2);
    }
    hook(P<2>());
    y_inv_p1<N, R1, R2>(c);
    __syncthreads();
    hook(P<3>());
    x_inv<N, R1, R2>(c, a);
}

// unnormalised 2-D transform of registers `a` up to (and including) the last y pass, result left in
// registers b[k2] of the pass-2 column role = spectrum at (ky = t + R1*k2, kx = freq_of_pos(line)).
// The inverse direction is obtained by the caller with IDFT(x) = conj(DFT(conj(x))).
template <int N, int R1, int R2>
__device__ __forceinline__ void fft2_to_regs(Ctx<N, R1, R2>& c, cf (&a)[R1], cf (&b)[R2]) {
    using GE = Geo<N, R1, R2>;
    x_fwd<N, R1, R2>(c, a);
    __syncthreads();
    y_fwd_p1<N, R1, R2>(c);
    if (c.act2) {
        ld_line<R2, GE::COL_P2_K>(b, c.fld + c.col_p2);
        Dft<R2, false>::run(b, c.K);
    }
}
// the matching unnormalised inverse, from registers b[k2] back to real-space registers a
template <int N, int R1, int R2>
__device__ __forceinline__ void ifft2_from_regs(Ctx<N, R1, R2>& c, cf (&b)[R2], cf (&a)[R1]) {
    using GE = Geo<N, R1, R2>;
    if (c.act2) {
        Dft<R2, true>::run(b, c.K);
        st_line<R2, GE::COL_P2_K>(b, c.fld + c.col_p2);
    }
    y_inv_p1<N, R1, R2>(c);
    __syncthreads();
    x_inv<N, R1, R2>(c, a);
}

// ---- building blocks of the kernel body -------------------------------------------------------------------------
template <int N, int R1, int R2>
__device__ __forceinline__ void load_probe(const Ctx<N, R1, R2>& c, cf (&a)[R1], const float2* __restrict__ probe) {
#pragma unroll
    for (int k = 0; k < R1; ++k) a[k] = c.act1 ? probe[c.line * N + k * R2 + c.t] : make_float2(0.f, 0.f);
}

template <int N, int R1, int R2>
__device__ __forceinline__ void add_probe_grad(const Ctx<N, R1, R2>& c, const cf (&a)[R1], float2* grad_probe, bool own_slot) {
    if (grad_probe && c.act1) {
        if (own_slot) {
            // this position's own [Py][Px] slot: plain stores; the slots are summed in a fixed order afterwards
            // (probe_grad_reduce_kernel), so the probe gradient is bit-reproducible like the object gradient
#pragma unroll
            for (int k = 0; k < R1; ++k) grad_probe[c.line * N + k * R2 + c.t] = a[k];
        } else {
#pragma unroll
            for (int k = 0; k < R1; ++k) {
                float* gp = reinterpret_cast<float*>(grad_probe + c.line * N + k * R2 + c.t);
                atomicAdd(gp, a[k].x);
                atomicAdd(gp + 1, a[k].y);
            }
        }
    }
}

template <int N, int R1, int R2>
__device__ __forceinline__ void block_loss(float lsum, float* red, float* out, int tid, int wave, int lane) {
    using GE = Geo<N, R1, R2>;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lsum += __shfl_down(lsum, off, 64);
    if (lane == 0) red[wave] = lsum;
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
        for (int w = 0; w < GE::NWAVES; ++w) s += red[w];
        *out = s;
    }
}

// sum of the (delta, beta) pairs of the slices of one modulation step for this thread's R1 pixels.
// BIN1 (binning == 1): pure loads, so the caller can issue them one step ahead and let the
// propagation hide their latency.
template <int R1, int R2, bool BIN1, int K0 = 0, int K1 = R1>
__device__ __forceinline__ void load_db(float2 (&db)[R1], const float2* __restrict__ base, size_t slice_stride, int step,
                                        int binning, int Z) {
    if (BIN1) {
        const float2* q = base + (size_t)step * slice_stride;
#pragma unroll
        for (int k = K0; k < K1; ++k) db[k] = q[k * R2];
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

// MODE 0: (delta, beta) slices, exp / sincos evaluated here; 1: real_imag (the slice IS the transmission, an unknown);
// 2: (delta, beta) unknowns, but `tile_base` points at the slice transmissions adm_rotate_fwd cached per voxel
//    (adm_plan_set_transmission_cache): the same numbers modulate<> would compute, evaluated once per voxel instead
//    of once per covering probe position and sweep -- the slice loop then has no transcendental at all.
template <int N, int R1, int R2, bool BIN1, int MODE>
__device__ __forceinline__ void fwd_sweep(Ctx<N, R1, R2>& c, cf (&a)[R1], const cf (&hs)[R2], const MsParams& p, float2* stash,
                                          const float2* tile_base, size_t slice_stride, bool do_grad) {
    using GE = Geo<N, R1, R2>;
    const int tid = threadIdx.x;
    ADM_STAMP_DECL;
    float2 db[R1];
    if (c.act1) load_db<R1, R2, BIN1>(db, tile_base, slice_stride, 0, p.binning, p.Z);
    for (int step = 0; step < p.n_steps; ++step) {
        ADM_STAMP_ON(blockIdx.x == 0 && step == 100);
        ADM_STAMP(0);
        if (c.act1) {
            if (MODE == 1) {
                // the slice IS the complex transmission; its gradient needs the PRE-modulation field
                if (do_grad) ws_store<R1>(stash + (size_t)step * R1 * GE::NT, GE::NT, tid, a);
#pragma unroll
                for (int k = 0; k < R1; ++k) a[k] = cmul(a[k], db[k]);
            } else {
                if (MODE == 2) {
#pragma unroll
                    for (int k = 0; k < R1; ++k) a[k] = cmul(a[k], db[k]);
                } else {
                    modulate<R1, false>(a, db, p.k1, p.sigma);
                }
                ADM_STAMP(1);
                if (do_grad) ws_store<R1>(stash + (size_t)step * R1 * GE::NT, GE::NT, tid, a);
            }
            ADM_STAMP(2);
        }
        ADM_STAMP(3);
        // The next step's tile slice is requested DURING the propagation, half at a time at two of its hook points: early
        // enough to land before the modulation, and not in one burst with the stores above (see Prefetch).
        auto hook = [&](auto pt) {
            constexpr int HP = decltype(pt)::value, H = BIN1 ? R1 / 2 : R1;
            if (c.act1 && step + 1 < p.n_steps) {
                if constexpr (HP == Prefetch::FWD_DB0) load_db<R1, R2, BIN1, 0, H>(db, tile_base, slice_stride, step + 1, p.binning, p.Z);
                if constexpr (HP == Prefetch::FWD_DB1 && BIN1) load_db<R1, R2, BIN1, H, R1>(db, tile_base, slice_stride, step + 1, p.binning, p.Z);
            }
        };
        if (step < p.n_steps - 1) {
            c.K = fft_k_dithered(step);          // propagation `step`: slice step -> step + 1
            convolve<N, R1, R2, false>(c, a, hs, hook);
        }
        ADM_STAMP(4);
    }
    c.K = fft_k_nominal();
    ADM_STAMP_ON(false);
}

// ACC: add to the tile gradient already stored by a previous probe mode instead of overwriting it
template <int N, int R1, int R2, bool BIN1, bool ACC, int MODE>
__device__ __forceinline__ void rev_sweep(Ctx<N, R1, R2>& c, cf (&a)[R1], const cf (&hs)[R2], const MsParams& p, const float2* stash,
                                          float2* gtile, const float2* tile_base, size_t slice_stride) {
    using GE = Geo<N, R1, R2>;
    constexpr bool RI = (MODE == 1);
    const v2f gw = {p.sigma * p.k1, -p.k1};
    const int tid = threadIdx.x;
    ADM_STAMP_DECL;
    float2 db[R1];
    cf psi[R1];
    if (c.act1) {
        load_db<R1, R2, BIN1>(db, tile_base, slice_stride, p.n_steps - 1, p.binning, p.Z);
        ws_load<R1>(stash + (size_t)(p.n_steps - 1) * R1 * GE::NT, GE::NT, tid, psi);
    }
    for (int step = p.n_steps - 1; step >= 0; --step) {
        ADM_STAMP_ON(blockIdx.x == 0 && step == 100);
        ADM_STAMP(8);
        if (c.act1) {
            // tile gradient of this modulation step, thread-native layout (coalesced 16-B/lane stores);
            // adm_tile_grad_accumulate overlap-adds the tiles afterwards (no atomics in this loop)
            float2 g[R1];
            float2* grow = gtile + (size_t)step * R1 * GE::NT;
            if (ACC) ws_load<R1>(grow, GE::NT, tid, g);
#pragma unroll
            for (int k = 0; k < R1; ++k) {
                // z = conj(G) * psi'  (delta_beta: psi' is the post-modulation field; real_imag: the stash holds the
                // pre-modulation field and (d/dre, d/dim) = G * conj(psi) = (Re z, -Im z))
                if (RI) {
                    const float zr = a[k].x * psi[k].x + a[k].y * psi[k].y;
                    const float zi = a[k].x * psi[k].y - a[k].y * psi[k].x;
                    g[k] = ACC ? make_float2(g[k].x + zr, g[k].y - zi) : make_float2(zr, -zi);
                    a[k] = cmulc(a[k], db[k]);
                } else {
                    // (d/ddelta, d/dbeta) = (sigma k1 Im z, -k1 Re z): three packed instructions per pixel
                    const cf z = cmulc(psi[k], a[k]);
                    g[k] = ACC ? fma_swapped(z, gw, g[k]) : mul_swapped(z, gw);
                }
            }
            ws_store<R1>(grow, GE::NT, tid, g);
            ADM_STAMP(9);
            if (MODE == 0) modulate<R1, true>(a, db, p.k1, p.sigma);
            if (MODE == 2) {
#pragma unroll
                for (int k = 0; k < R1; ++k) a[k] = cmulc(a[k], db[k]);
            }
            ADM_STAMP(10);
        }
        ADM_STAMP(11);
        auto hook = [&](auto pt) {
            constexpr int HP = decltype(pt)::value, H = BIN1 ? R1 / 2 : R1;
            if (c.act1 && step > 0) {
                if constexpr (HP == Prefetch::REV_DB0) load_db<R1, R2, BIN1, 0, H>(db, tile_base, slice_stride, step - 1, p.binning, p.Z);
                if constexpr (HP == Prefetch::REV_DB1 && BIN1) load_db<R1, R2, BIN1, H, R1>(db, tile_base, slice_stride, step - 1, p.binning, p.Z);
                if constexpr (HP == Prefetch::REV_PSI) ws_load<R1>(stash + (size_t)(step - 1) * R1 * GE::NT, GE::NT, tid, psi);
            }
        };
        if (step > 0) {
            c.K = fft_k_dithered(step - 1);      // the adjoint of propagation step - 1, with that propagation's constants
            convolve<N, R1, R2, true>(c, a, hs, hook);
        }
        ADM_STAMP(12);
    }
    c.K = fft_k_nominal();
    ADM_STAMP_ON(false);
}

// (a plan may hold several detector-plane kernels -- the distances of multi-distance data: position b takes kernel b % n_hfree.
// Only in the per-position-probe instantiations, which is where such launches come from: the default kernel's code is untouched.)
template <int N, int R1, int R2, bool PP>
__device__ __forceinline__ void load_hfree(cf (&hf)[R2], const MsParams& p, int kx, int tc2, int b) {
    const double n2 = (double)(N * N);
    const cf* hsel = PP ? p.hfree + (size_t)(b % p.n_hfree) * (N * N) : p.hfree;
#pragma unroll
    for (int k = 0; k < R2; ++k) {
        const cf h = hsel[(tc2 + R1 * k) * N + kx];
        hf[k] = make_float2((float)((double)h.x / n2), (float)((double)h.y / n2));
    }
}

// exit wave `a` -> detector plane.  Far field: unnormalised spectrum left in bb (pass-2 column role);
// near field / Fresnel: the detector field stays in `a` (real space).
template <int N, int R1, int R2, bool PP>
__device__ __forceinline__ void detector_forward(Ctx<N, R1, R2>& c, cf (&a)[R1], cf (&bb)[R2], const MsParams& p, int kx, int tc2, int b) {
    if (p.det_mode == ADM_DET_FRESNEL_) {
        cf hf[R2];
        load_hfree<N, R1, R2, PP>(hf, p, kx, tc2, b);
        convolve<N, R1, R2, false>(c, a, hf);
    } else if (p.det_mode == ADM_DET_FARFIELD_) {
        // Psi = scale * F(psi)  (F forward, or inverse via conjugation when det_inverse)
        if (p.det_inverse) {
#pragma unroll
            for (int k = 0; k < R1; ++k) a[k] = conjf2(a[k]);
        }
        fft2_to_regs<N, R1, R2>(c, a, bb);
    }
}
// dL/d(detector field) (in bb for the far field, in `a` otherwise) -> dL/d(exit wave) in `a`
template <int N, int R1, int R2, bool PP>
__device__ __forceinline__ void detector_adjoint(Ctx<N, R1, R2>& c, cf (&a)[R1], cf (&bb)[R2], const MsParams& p, int kx, int tc2, int b) {
    if (p.det_mode == ADM_DET_FARFIELD_) {
        ifft2_from_regs<N, R1, R2>(c, bb, a);
        if (p.det_inverse) {
#pragma unroll
            for (int k = 0; k < R1; ++k) a[k] = conjf2(a[k]);
        }
    } else if (p.det_mode == ADM_DET_FRESNEL_) {
        cf hf[R2];
        load_hfree<N, R1, R2, PP>(hf, p, kx, tc2, b);
        convolve<N, R1, R2, true>(c, a, hf);
    }
}

// position handled by workgroup `wg` of a launch of B: XCD k = wg % 8 owns positions [k*q + min(k, r), ...), q = B / 8, r = B % 8
__device__ __forceinline__ int xcd_position(int wg, int B) {
    const int k = wg & 7, slot = wg >> 3;
    const int q = B >> 3, r = B & 7;
    return k * q + min(k, r) + slot;
}

// PP: one probe set per position (sub-pixel probe positions); a template parameter because even the two extra address
// computations measurably perturb the schedule of the tuned default kernel (+3 %).
template <int N, int R1, int R2, bool BIN1, bool MULTI, int MODE, bool PP>
__global__ __launch_bounds__((Geo<N, R1, R2>::NT)) void ms_fwd_adj_kernel(MsParams p) {
    using GE = Geo<N, R1, R2>;
    __shared__ cf fld[GE::FLD];
    __shared__ float red[GE::NWAVES];

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int li = lane / GE::G;
    Ctx<N, R1, R2> c;
    c.fld = fld;
    c.K = fft_k_nominal();
    c.t = lane % GE::G;
    c.line = wave * GE::LPW + li;
    const bool line_ok = (li < GE::LPW) && (c.line < N);
    c.act1 = line_ok && (c.t < R2);
    c.act2 = line_ok && (c.t < R1);
    if (!line_ok) c.line = 0;   // keep addresses in range for inactive lanes
    const int tc2 = c.t % R1;   // clamp for lanes that are inactive in the pass-2 role
    c.row_p1 = c.line * GE::Q + c.t * GE::ROW_P1_T;
    c.row_p2 = c.line * GE::Q + tc2 * GE::ROW_P2_T;
    c.col_p1 = GE::posx(c.line) + c.t * GE::COL_P1_T;
    c.col_p2 = GE::posx(c.line) + tc2 * GE::COL_P2_T;
    // Workgroups are dealt round-robin over the 8 XCDs (observed, MI355X_MICROARCH.md: speed only, never correctness), and
    // consecutive positions of a scan overlap by ~90 % of a tile: XCD k takes the k-th CONTIGUOUS eighth of the batch, so
    // that the tile slices its workgroups read in near lock-step are largely the same lines of that XCD's L2.
    const int b = xcd_position(blockIdx.x, gridDim.x);

    // ---- static per-thread constants ----
#pragma unroll
    for (int k = 0; k < R1; ++k) c.tw[k] = p.twid[(c.t * k) % N];
    const int kx = freq_of_pos<R1, R2>(c.line);
    // 1/N^2 of the inverse transform is folded into H with ONE rounding per element (divide in
    // double): multiplying by fl(1/N^2) would scale every propagation by the same (1+eps) and the
    // bias would grow linearly with the number of slices.
    const double n2 = (double)(N * N);
    // The thread <-> (ky, kx) map is static, so each thread keeps its R2 values of H / N^2 in registers for the
    // whole kernel (ADM_H_IN_LDS: in an LDS image instead -- slower once the register file is not the limit).
    cf hs[R2];
#pragma unroll
    for (int k = 0; k < R2; ++k) {
        const int ky = tc2 + R1 * k;
        const cf h = p.h[ky * N + kx];
        hs[k] = make_float2((float)((double)h.x / n2), (float)((double)h.y / n2));
    }

    const int2 ps = p.pos[b];
    const int py = ps.x + p.pad_y0, px = ps.y + p.pad_x0;
    // element n1 of this thread: pixel (row = line, col = n1*R2 + t) of the tile
    const size_t slice_stride = (size_t)p.Yp * p.Xp;
    const size_t tile_off = (size_t)(py + c.line) * p.Xp + px + c.t;   // + n1*R2 + slice*slice_stride
    const bool do_grad = (p.want_grad != 0);
    float2* stash = p.stash + (size_t)b * p.n_modes * p.n_steps * R1 * GE::NT;
    float2* gtile = p.gtile + (size_t)b * p.n_steps * R1 * GE::NT;

    cf a[R1];
    const float2* tile_base = p.obj_rot + tile_off;
    cf bb[R2];
    float lsum = 0.f;
    if (!MULTI) {
        // ================= single probe mode: everything stays in registers =================
        load_probe<N, R1, R2>(c, a, PP ? p.probe + (size_t)b * p.probe_bstride : p.probe);
        fwd_sweep<N, R1, R2, BIN1, MODE>(c, a, hs, p, stash, tile_base, slice_stride, do_grad);
        detector_forward<N, R1, R2, PP>(c, a, bb, p, kx, tc2, b);
        if (p.det_mode == ADM_DET_FARFIELD_) {
            if (c.act2) {
                const int mx = (kx + N / 2) % N;
#pragma unroll
                for (int k = 0; k < R2; ++k) {
                    const int my = (c.t + R1 * k + N / 2) % N;
                    const size_t di = ((size_t)b * N + my) * N + mx;
                    const cf psi = cscale(bb[k], p.det_scale);      // (conjugated when det_inverse; |.| unaffected)
                    const float mag = sqrtf(psi.x * psi.x + psi.y * psi.y);
                    float g;
                    const float wq = p.det_weight ? p.det_weight[my * N + mx] : 1.f;     // beamstop mask (forward_model.py:128-136)
                    lsum += wq * loss_term(mag, p.target[di], p, g);
                    if (p.pred) p.pred[di] = mag;
                    bb[k] = cscale(psi, wq * g * p.det_scale);       // adjoint of (scale * F) is scale * F^H
                }
            }
        } else if (c.act1) {
#pragma unroll
            for (int k = 0; k < R1; ++k) {
                const size_t di = ((size_t)b * N + c.line) * N + k * R2 + c.t;
                const float mag = sqrtf(a[k].x * a[k].x + a[k].y * a[k].y);
                float g;
                const float wq = p.det_weight ? p.det_weight[c.line * N + k * R2 + c.t] : 1.f;
                lsum += wq * loss_term(mag, p.target[di], p, g);
                if (p.pred) p.pred[di] = mag;
                a[k] = cscale(a[k], wq * g);
            }
        }
        block_loss<N, R1, R2>(lsum, red, p.loss_sum + b, tid, wave, lane);
        if (!do_grad) return;
        detector_adjoint<N, R1, R2, PP>(c, a, bb, p, kx, tc2, b);
        rev_sweep<N, R1, R2, BIN1, false, MODE>(c, a, hs, p, stash, gtile, tile_base, slice_stride);
        add_probe_grad<N, R1, R2>(c, a, p.grad_probe ? p.grad_probe + (size_t)b * p.gprobe_bstride : nullptr, p.gprobe_bstride != 0);
    } else {
        // ================= several incoherent probe modes (adorym/forward_model.py:354-375) =================
        // pred = sqrt(sum_m |Psi_m|^2): the detector-plane fields of all modes are parked in HBM (thread-native
        // order), the intensity is summed in registers, then every mode is back-propagated with the common factor g.
        const int M = p.n_modes;
        const size_t per = (size_t)p.n_steps * R1 * GE::NT;
        const bool far = (p.det_mode == ADM_DET_FARFIELD_);
        float inten[GE::G];
#pragma unroll
        for (int k = 0; k < GE::G; ++k) inten[k] = 0.f;
        for (int m = 0; m < M; ++m) {
            load_probe<N, R1, R2>(c, a, p.probe + (PP ? (size_t)b * p.probe_bstride : 0) + (size_t)m * N * N);
            fwd_sweep<N, R1, R2, BIN1, MODE>(c, a, hs, p, stash + (size_t)m * per, tile_base, slice_stride, do_grad);
            detector_forward<N, R1, R2, PP>(c, a, bb, p, kx, tc2, b);
            float2* dq = p.det + ((size_t)b * M + m) * GE::G * GE::NT + tid;
            if (far) {
                if (c.act2) {
#pragma unroll
                    for (int k = 0; k < R2; ++k) {
                        const cf psi = cscale(bb[k], p.det_scale);
                        inten[k] += psi.x * psi.x + psi.y * psi.y;
                        dq[(size_t)k * GE::NT] = psi;
                    }
                }
            } else if (c.act1) {
#pragma unroll
                for (int k = 0; k < R1; ++k) {
                    inten[k] += a[k].x * a[k].x + a[k].y * a[k].y;
                    dq[(size_t)k * GE::NT] = a[k];
                }
            }
        }
        // loss and the common factor g (same for every mode): dL/dPsi_m = g * Psi_m
        float gf[GE::G];
        if (far) {
            if (c.act2) {
                const int mx = (kx + N / 2) % N;
#pragma unroll
                for (int k = 0; k < R2; ++k) {
                    const int my = (c.t + R1 * k + N / 2) % N;
                    const size_t di = ((size_t)b * N + my) * N + mx;
                    const float mag = sqrtf(inten[k]);
                    const float wq = p.det_weight ? p.det_weight[my * N + mx] : 1.f;
                    lsum += wq * loss_term_nz(mag, p.target[di], p, gf[k]);
                    gf[k] *= wq;
                    if (p.pred) p.pred[di] = mag;
                }
            }
        } else if (c.act1) {
#pragma unroll
            for (int k = 0; k < R1; ++k) {
                const size_t di = ((size_t)b * N + c.line) * N + k * R2 + c.t;
                const float mag = sqrtf(inten[k]);
                const float wq = p.det_weight ? p.det_weight[c.line * N + k * R2 + c.t] : 1.f;
                lsum += wq * loss_term_nz(mag, p.target[di], p, gf[k]);
                gf[k] *= wq;
                if (p.pred) p.pred[di] = mag;
            }
        }
        block_loss<N, R1, R2>(lsum, red, p.loss_sum + b, tid, wave, lane);
        if (!do_grad) return;
        for (int m = 0; m < M; ++m) {
            const float2* dq = p.det + ((size_t)b * M + m) * GE::G * GE::NT + tid;
            if (far) {
                if (c.act2) {
#pragma unroll
                    for (int k = 0; k < R2; ++k) bb[k] = cscale(dq[(size_t)k * GE::NT], gf[k] * p.det_scale);
                }
            } else if (c.act1) {
#pragma unroll
                for (int k = 0; k < R1; ++k) a[k] = cscale(dq[(size_t)k * GE::NT], gf[k]);
            }
            detector_adjoint<N, R1, R2, PP>(c, a, bb, p, kx, tc2, b);
            if (m == 0) rev_sweep<N, R1, R2, BIN1, false, MODE>(c, a, hs, p, stash, gtile, tile_base, slice_stride);
            else rev_sweep<N, R1, R2, BIN1, true, MODE>(c, a, hs, p, stash + (size_t)m * per, gtile, tile_base, slice_stride);
            add_probe_grad<N, R1, R2>(c, a, p.grad_probe ? p.grad_probe + (size_t)b * p.gprobe_bstride + (size_t)m * N * N : nullptr,
                                      p.gprobe_bstride != 0);
        }
    }
}

#ifdef ADM_STAMPS
}  // namespace adm
extern "C" int adm_debug_read_stamps(void* host, size_t bytes) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(adm::g_stamps), bytes < sizeof(adm::g_stamps) ? bytes : sizeof(adm::g_stamps));
}
namespace adm {
#endif
template <int N, int R1, int R2> static hipError_t launch(const MsParams& p, int batch, hipStream_t st) {
    using GE = Geo<N, R1, R2>;
    const dim3 g(batch), t(GE::NT);
#define ADM_LAUNCH(B1, MU, MODE_, PP_) hipLaunchKernelGGL((ms_fwd_adj_kernel<N, R1, R2, B1, MU, MODE_, PP_>), g, t, 0, st, p)
    const bool multi = p.n_modes > 1;
    // binning == 1 is enforced for real_imag at plan creation, for per-position probes and the cached transmissions
    // by the caller
    if (p.probe_bstride) {
        if (p.real_imag) { if (multi) ADM_LAUNCH(true, true, 1, true); else ADM_LAUNCH(true, false, 1, true); }
        else if (p.pre_t) { if (multi) ADM_LAUNCH(true, true, 2, true); else ADM_LAUNCH(true, false, 2, true); }
        else { if (multi) ADM_LAUNCH(true, true, 0, true); else ADM_LAUNCH(true, false, 0, true); }
    } else if (p.real_imag) {
        if (multi) ADM_LAUNCH(true, true, 1, false); else ADM_LAUNCH(true, false, 1, false);
    } else if (p.pre_t) {
        if (multi) ADM_LAUNCH(true, true, 2, false); else ADM_LAUNCH(true, false, 2, false);
    } else if (multi) {
        if (p.binning == 1) ADM_LAUNCH(true, true, 0, false); else ADM_LAUNCH(false, true, 0, false);
    } else {
        if (p.binning == 1) ADM_LAUNCH(true, false, 0, false); else ADM_LAUNCH(false, false, 0, false);
    }
#undef ADM_LAUNCH
    return hipGetLastError();
}

// =====================================================================================================================
// Sub-pixel probe positions: position b sees every probe mode Fourier-shifted by its correction (sy, sx)
//     p_b = IFFT2( Phi_b * FFT2(p) ),   Phi_b(ky,kx) = exp(-2 PI i (fx*sx + fy*sy)),  f = fftfreq      (adorym/util.py:380-397)
// and the adjoint:  dL/dp += IFFT2( conj(Phi_b) * FFT2(G_b) ),  dL/ds = 2 PI sum_k f_k Im( conj(Ghat_k) Phi_k F_k ),
// Ghat = FFT2(G_b) / N^2, F = FFT2(p).  Same LDS transform machinery and thread <-> frequency map as the multislice kernel.
// =====================================================================================================================
template <int N, int R1, int R2>
__device__ __forceinline__ void init_ctx(Ctx<N, R1, R2>& c, cf* fld, const float2* __restrict__ twid, int& tc2, int& kx) {
    using GE = Geo<N, R1, R2>;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int li = lane / GE::G;
    c.fld = fld;
    c.K = fft_k_nominal();
    c.t = lane % GE::G;
    c.line = wave * GE::LPW + li;
    const bool line_ok = (li < GE::LPW) && (c.line < N);
    c.act1 = line_ok && (c.t < R2);
    c.act2 = line_ok && (c.t < R1);
    if (!line_ok) c.line = 0;
    tc2 = c.t % R1;
    c.row_p1 = c.line * GE::Q + c.t * GE::ROW_P1_T;
    c.row_p2 = c.line * GE::Q + tc2 * GE::ROW_P2_T;
    c.col_p1 = GE::posx(c.line) + c.t * GE::COL_P1_T;
    c.col_p2 = GE::posx(c.line) + tc2 * GE::COL_P2_T;
#pragma unroll
    for (int k = 0; k < R1; ++k) c.tw[k] = twid[(c.t * k) % N];
    kx = freq_of_pos<R1, R2>(c.line);
}

template <int N> __device__ __forceinline__ float fftfreq_f(int k) {
    return (float)((double)(k < (N + 1) / 2 ? k : k - N) / (double)N);
}

// Phi for the R2 spectral elements this thread holds in the pass-2 column role
template <int N, int R1, int R2>
__device__ __forceinline__ void shift_phases(cf (&ph)[R2], float (&fy)[R2], float fx, float2 s, int tc2) {
#pragma clang fp contract(off)
    const float m2pi = (float)(-2.0 * 3.14159265359);
#pragma unroll
    for (int k = 0; k < R2; ++k) {
        fy[k] = fftfreq_f<N>(tc2 + R1 * k);
        const float arg = m2pi * (fx * s.y + fy[k] * s.x);
        float sn, cs;
        sincos_fast(arg, sn, cs);
        ph[k] = make_float2(cs, sn);
    }
}

template <int N, int R1, int R2>
__global__ __launch_bounds__((Geo<N, R1, R2>::NT)) void probe_shift_kernel(ShiftParams q) {
    using GE = Geo<N, R1, R2>;
    __shared__ cf fld[GE::FLD];
    Ctx<N, R1, R2> c;
    int tc2, kx;
    init_ctx<N, R1, R2>(c, fld, q.twid, tc2, kx);
    const int b = blockIdx.x / q.n_modes, m = blockIdx.x % q.n_modes;
    const float2 s = q.shifts[q.index ? q.index[b] : b];
    cf ph[R2];
    float fy[R2];
    shift_phases<N, R1, R2>(ph, fy, fftfreq_f<N>(kx), s, tc2);
    cf a[R1], bb[R2];
    load_probe<N, R1, R2>(c, a, q.probe + (size_t)m * N * N);
    fft2_to_regs<N, R1, R2>(c, a, bb);
    const float inv = (float)(1.0 / ((double)N * N));
    if (c.act2) {
#pragma unroll
        for (int k = 0; k < R2; ++k) bb[k] = cscale(cmul(bb[k], ph[k]), inv);
    }
    ifft2_from_regs<N, R1, R2>(c, bb, a);
    if (c.act1) {
        float2* o = q.probes_out + (size_t)blockIdx.x * N * N + c.line * N + c.t;
#pragma unroll
        for (int k = 0; k < R1; ++k) o[k * R2] = a[k];
    }
}

template <int N, int R1, int R2>
__global__ __launch_bounds__((Geo<N, R1, R2>::NT)) void probe_shift_adj_kernel(ShiftParams q) {
    using GE = Geo<N, R1, R2>;
    __shared__ cf fld[GE::FLD];
    __shared__ float red[2 * GE::NWAVES];
    Ctx<N, R1, R2> c;
    int tc2, kx;
    init_ctx<N, R1, R2>(c, fld, q.twid, tc2, kx);
    const int b = blockIdx.x / q.n_modes, m = blockIdx.x % q.n_modes;     // one workgroup per (position, mode)
    const int e = q.index ? q.index[b] : b;
    const float2 s = q.shifts[e];
    cf ph[R2];
    float fy[R2];
    const float fx = fftfreq_f<N>(kx);
    shift_phases<N, R1, R2>(ph, fy, fx, s, tc2);
    const float inv = (float)(1.0 / ((double)N * N));
    float gy = 0.f, gx = 0.f;
    {
        cf a[R1], bf[R2], bg[R2];
        load_probe<N, R1, R2>(c, a, q.probe + (size_t)m * N * N);
        fft2_to_regs<N, R1, R2>(c, a, bf);
        __syncthreads();
        load_probe<N, R1, R2>(c, a, q.grad_probes + ((size_t)b * q.n_modes + m) * N * N);
        fft2_to_regs<N, R1, R2>(c, a, bg);
        if (c.act2) {
#pragma unroll
            for (int k = 0; k < R2; ++k) {
                const cf gh = cscale(bg[k], inv);                 // Ghat
                const cf pf = cmul(ph[k], bf[k]);                 // Phi * F
                const float im = gh.x * pf.y - gh.y * pf.x;       // Im(conj(Ghat) * Phi F)
                gy += fy[k] * im;
                gx += fx * im;
                bg[k] = cmulc(gh, ph[k]);                         // conj(Phi) * Ghat  -> inverse transform below
            }
        }
        ifft2_from_regs<N, R1, R2>(c, bg, a);
        if (q.slots) add_probe_grad<N, R1, R2>(c, a, q.slots + ((size_t)b * q.n_modes + m) * N * N, true);
        else add_probe_grad<N, R1, R2>(c, a, q.grad_probe ? q.grad_probe + (size_t)m * N * N : nullptr, false);
    }
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        gy += __shfl_down(gy, off, 64);
        gx += __shfl_down(gx, off, 64);
    }
    if (lane == 0) { red[2 * wave] = gy; red[2 * wave + 1] = gx; }
    __syncthreads();
    if (tid == 0) {
        float sy = 0.f, sx = 0.f;
        for (int w = 0; w < GE::NWAVES; ++w) { sy += red[2 * w]; sx += red[2 * w + 1]; }
        const float twopi = (float)(2.0 * 3.14159265359);
        atomicAdd(q.grad_shifts + 2 * (size_t)e, twopi * sy);
        atomicAdd(q.grad_shifts + 2 * (size_t)e + 1, twopi * sx);
    }
}

template <int N, int R1, int R2> static hipError_t launch_shift(const ShiftParams& q, int batch, bool adjoint, hipStream_t st) {
    using GE = Geo<N, R1, R2>;
    if (adjoint) hipLaunchKernelGGL((probe_shift_adj_kernel<N, R1, R2>), dim3(batch * q.n_modes), dim3(GE::NT), 0, st, q);
    else hipLaunchKernelGGL((probe_shift_kernel<N, R1, R2>), dim3(batch * q.n_modes), dim3(GE::NT), 0, st, q);
    return hipGetLastError();
}

// compiled probe sizes N = R1 * R2 (radices from {2, 3, 4, 8, 9}; at most 16 waves per workgroup)
#define ADM_FOR_EACH_SIZE(X) X(8, 2, 4) X(12, 3, 4) X(16, 4, 4) X(18, 2, 9) X(24, 3, 8) X(27, 3, 9) X(32, 4, 8) X(36, 4, 9) X(64, 8, 8) X(72, 8, 9)

int ms_threads_for(int n) {
    switch (n) {
#define X(N_, A_, B_) case N_: return Geo<N_, A_, B_>::NT;
        ADM_FOR_EACH_SIZE(X)
#undef X
        default: return 0;
    }
}
int ms_r2_for(int n) {
    switch (n) {
#define X(N_, A_, B_) case N_: return B_;
        ADM_FOR_EACH_SIZE(X)
#undef X
        default: return 0;
    }
}
int ms_r1_for(int n) {
    switch (n) {
#define X(N_, A_, B_) case N_: return A_;
        ADM_FOR_EACH_SIZE(X)
#undef X
        default: return 0;
    }
}

hipError_t ms_launch(int n, const MsParams& p, int batch, hipStream_t st) {
    switch (n) {
#define X(N_, A_, B_) case N_: return launch<N_, A_, B_>(p, batch, st);
        ADM_FOR_EACH_SIZE(X)
#undef X
        default: return hipErrorInvalidValue;
    }
}

hipError_t shift_launch(int n, const ShiftParams& q, int batch, bool adjoint, hipStream_t st) {
    switch (n) {
#define X(N_, A_, B_) case N_: return launch_shift<N_, A_, B_>(q, batch, adjoint, st);
        ADM_FOR_EACH_SIZE(X)
#undef X
        default: return hipErrorInvalidValue;
    }
}

}  // namespace adm
