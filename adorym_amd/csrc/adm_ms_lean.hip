// Throughput variant of the multislice forward + loss + adjoint kernel: same mathematics, thread <-> pixel map and
// workspace layouts as ms_fwd_adj_kernel (adm_multislice.hip), but built so that TWO workgroups share a compute unit.
//
// Replaces (reference paths): adorym/propagate.py:195-280, adorym/forward_model.py:313-353, :88-93 and the autograd
// backward (adorym/wrappers.py:322) -- for the configuration the big batches run in: one probe mode, delta/beta
// unknowns, binning 1, one probe for all positions.
//
// Why a second kernel.  The latency-oriented kernel keeps its per-thread constants (transfer function, twiddles) and a
// one-step-ahead prefetch in registers: 148 VGPRs, one 11-wave workgroup per CU.  Per-wave stamps (tools/stamps.py) show
// that such a workgroup leaves the vector ALUs idle more than half of the time: the three waves of a SIMD serialise by
// age inside every barrier interval and the older ones then wait at the barrier.  With >= 512 positions in flight the
// cure is occupancy, not latency: here
//   * the transfer function comes from an LDS table that exploits H(ky, kx) = H(ky, N - kx) ((N/2+1) x N entries) and
//     the pass-1 twiddles from a G x R1 LDS table: no constant lives in a register across the slice loop;
//   * the field image is the dense permuted layout of adm_ms_math.h (45.7 KB at P = 72), so field + tables of two
//     workgroups fit the 160 KB of a CU;
//   * the slice modulation evaluates its wave-uniform small-phase test once for all R1 elements (one branch per slice
//     step instead of R1), and global addresses are a scalar base + one 32-bit lane offset (no 64-bit vector address
//     arithmetic in the loop);
//   * __launch_bounds__(NT, 6): at most 80 VGPRs => 22 waves = two workgroups per CU.
#include <hip/hip_runtime.h>
#include "adm_common.h"
#include "adm_fft.h"
#include "adm_ms_math.h"

namespace adm {

// the compiler must not prove an LDS table offset loop-invariant: it would hoist the table into registers
__device__ __forceinline__ int opaque(int x) {
    asm volatile("" : "+v"(x));
    return x;
}

template <int N, int R1, int R2> struct LeanGeo : Geo<N, R1, R2> {
    using GE = Geo<N, R1, R2>;
    static constexpr int HS = N / 2 + 1;       // classes of kx under kx -> N - kx
    static constexpr int HTAB = HS * N;        // complex entries of the transfer-function table [kxs][ky]
    static constexpr int TWTAB = GE::G * R1;   // complex entries of the twiddle table [t][k1]
};

template <int N, int R1, int R2> struct LCtx {
    cf* fld;
    const cf* htab;
    const cf* twtab;
    int row_p1, row_p2, col_p1, col_p2;   // complex-element offsets of the four access patterns
    int hb;                               // htab offset of this thread's first spectral element (pass-2 column role)
    int twb;                              // twtab offset of this thread's twiddles
    bool act1, act2;
};

template <int N, int R1, int R2> __device__ __forceinline__ void l_p1_fwd(const LCtx<N, R1, R2>& c, cf (&a)[R1]) {
    Dft<R1, false>::run(a);
    const cf* tw = c.twtab + opaque(c.twb);
#pragma unroll
    for (int k = 1; k < R1; ++k) a[k] = cmul(a[k], tw[k]);
}
template <int N, int R1, int R2> __device__ __forceinline__ void l_p1_inv(const LCtx<N, R1, R2>& c, cf (&a)[R1]) {
    const cf* tw = c.twtab + opaque(c.twb);
#pragma unroll
    for (int k = 1; k < R1; ++k) a[k] = cmulc(a[k], tw[k]);
    Dft<R1, true>::run(a);
}

template <int N, int R1, int R2> __device__ __forceinline__ void l_x_fwd(const LCtx<N, R1, R2>& c, cf (&a)[R1]) {
    using GE = Geo<N, R1, R2>;
    if (c.act1) {
        l_p1_fwd<N, R1, R2>(c, a);
        st_line<R1, GE::ROW_P1_K>(a, c.fld + c.row_p1);
    }
    WAVE_SYNC();
    if (c.act2) {
        cf b[R2];
        ld_line<R2, GE::ROW_P2_K>(b, c.fld + c.row_p2);
        Dft<R2, false>::run(b);
        st_line<R2, GE::ROW_P2_K>(b, c.fld + c.row_p2);
    }
}
template <int N, int R1, int R2> __device__ __forceinline__ void l_x_inv(const LCtx<N, R1, R2>& c, cf (&a)[R1]) {
    using GE = Geo<N, R1, R2>;
    if (c.act2) {
        cf b[R2];
        ld_line<R2, GE::ROW_P2_K>(b, c.fld + c.row_p2);
        Dft<R2, true>::run(b);
        st_line<R2, GE::ROW_P2_K>(b, c.fld + c.row_p2);
    }
    WAVE_SYNC();
    if (c.act1) {
        ld_line<R1, GE::ROW_P1_K>(a, c.fld + c.row_p1);
        l_p1_inv<N, R1, R2>(c, a);
    }
}
template <int N, int R1, int R2> __device__ __forceinline__ void l_y_fwd_p1(const LCtx<N, R1, R2>& c) {
    using GE = Geo<N, R1, R2>;
    if (c.act1) {
        cf a[R1];
        ld_line<R1, GE::COL_P1_K>(a, c.fld + c.col_p1);
        l_p1_fwd<N, R1, R2>(c, a);
        st_line<R1, GE::COL_P1_K>(a, c.fld + c.col_p1);
    }
    WAVE_SYNC();
}
template <int N, int R1, int R2> __device__ __forceinline__ void l_y_inv_p1(const LCtx<N, R1, R2>& c) {
    using GE = Geo<N, R1, R2>;
    WAVE_SYNC();
    if (c.act1) {
        cf a[R1];
        ld_line<R1, GE::COL_P1_K>(a, c.fld + c.col_p1);
        l_p1_inv<N, R1, R2>(c, a);
        st_line<R1, GE::COL_P1_K>(a, c.fld + c.col_p1);
    }
}

// psi <- IFFT2( H * FFT2(psi) ) (CONJ: conj(H)), psi in registers `a` (row role) on entry and exit; H (with the 1/N^2
// of the inverse folded in) from the LDS table, element k of the pass-2 column role at htab[hb + k*R1].
struct NoMid { __device__ __forceinline__ void operator()() const {} };
// `mid` runs between the spectral multiply and the inverse passes: the place where the caller issues global loads
// that the code after the convolution needs (late enough not to occupy registers across the whole convolution,
// early enough for their latency to hide behind the inverse passes).
template <int N, int R1, int R2, bool CONJ, class Mid = NoMid>
__device__ __forceinline__ void l_convolve(const LCtx<N, R1, R2>& c, cf (&a)[R1], Mid mid = Mid()) {
    using GE = Geo<N, R1, R2>;
    l_x_fwd<N, R1, R2>(c, a);
    __syncthreads();
    l_y_fwd_p1<N, R1, R2>(c);
    if (c.act2) {
        cf b[R2];
        ld_line<R2, GE::COL_P2_K>(b, c.fld + c.col_p2);
        Dft<R2, false>::run(b);
        __builtin_amdgcn_sched_barrier(0);     // keep the R2 table reads from being hoisted above the transform (registers)
        const cf* hp = c.htab + opaque(c.hb);
#pragma unroll
        for (int k = 0; k < R2; ++k) b[k] = cmul_t<CONJ>(b[k], hp[k * R1]);
        Dft<R2, true>::run(b);
        st_line<R2, GE::COL_P2_K>(b, c.fld + c.col_p2);
    }
    mid();
    l_y_inv_p1<N, R1, R2>(c);
    __syncthreads();
    l_x_inv<N, R1, R2>(c, a);
}
template <int N, int R1, int R2> __device__ __forceinline__ void l_fft2_to_regs(const LCtx<N, R1, R2>& c, cf (&a)[R1], cf (&b)[R2]) {
    using GE = Geo<N, R1, R2>;
    l_x_fwd<N, R1, R2>(c, a);
    __syncthreads();
    l_y_fwd_p1<N, R1, R2>(c);
    if (c.act2) {
        ld_line<R2, GE::COL_P2_K>(b, c.fld + c.col_p2);
        Dft<R2, false>::run(b);
    }
}
template <int N, int R1, int R2> __device__ __forceinline__ void l_ifft2_from_regs(const LCtx<N, R1, R2>& c, cf (&b)[R2], cf (&a)[R1]) {
    using GE = Geo<N, R1, R2>;
    if (c.act2) {
        Dft<R2, true>::run(b);
        st_line<R2, GE::COL_P2_K>(b, c.fld + c.col_p2);
    }
    l_y_inv_p1<N, R1, R2>(c);
    __syncthreads();
    l_x_inv<N, R1, R2>(c, a);
}

// global accesses as scalar base + 32-bit lane byte offset (saddr addressing: no 64-bit vector adds in the loop)
__device__ __forceinline__ float2 ldg2(const void* sbase, unsigned voff) {
    return *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(sbase) + voff);
}

template <int N, int R1, int R2>
__global__ __launch_bounds__((Geo<N, R1, R2>::NT), 6) void ms_lean_kernel(MsParams p) {
    using GE = Geo<N, R1, R2>;
    using LG = LeanGeo<N, R1, R2>;
    __shared__ cf fld[GE::FLD];
    __shared__ cf htab[LG::HTAB];
    __shared__ cf twtab[LG::TWTAB];
    __shared__ float red[GE::NWAVES];

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int li = lane / GE::G;
    const int t = lane % GE::G;
    int line = wave * GE::LPW + li;
    const bool line_ok = (li < GE::LPW) && (line < N);
    if (!line_ok) line = 0;
    const int tc2 = t % R1;
    LCtx<N, R1, R2> c;
    c.fld = fld;
    c.htab = htab;
    c.twtab = twtab;
    c.act1 = line_ok && (t < R2);
    c.act2 = line_ok && (t < R1);
    c.row_p1 = line * GE::Q + t * GE::ROW_P1_T;
    c.row_p2 = line * GE::Q + tc2 * GE::ROW_P2_T;
    c.col_p1 = GE::posx(line) + t * GE::COL_P1_T;
    c.col_p2 = GE::posx(line) + tc2 * GE::COL_P2_T;
    const int kx = freq_of_pos<R1, R2>(line);
    const int kxs = kx <= N / 2 ? kx : N - kx;
    c.hb = kxs * N + tc2;
    c.twb = t * R1;
    const int b = blockIdx.x;

    // ---- tables: H / N^2 (one rounding per element, division in double as in ms_fwd_adj_kernel) and W_N^(t*k1) ----
    {
        const double n2 = (double)(N * N);
        for (int i = tid; i < LG::HTAB; i += GE::NT) {
            const int ks = i / N, ky = i - ks * N;
            const cf h = p.h[ky * N + ks];
            htab[i] = make_float2((float)((double)h.x / n2), (float)((double)h.y / n2));
        }
        for (int i = tid; i < LG::TWTAB; i += GE::NT) twtab[i] = p.twid[((i / R1) * (i % R1)) % N];
    }
    __syncthreads();

    const int2 ps = p.pos[b];
    const int py = ps.x + p.pad_y0, px = ps.y + p.pad_x0;
    const size_t slice_stride = (size_t)p.Yp * p.Xp;                         // float2 elements
    const float2* tile0 = p.obj_rot + ((size_t)py * p.Xp + px);               // scalar: tile origin, slice 0
    const unsigned toff = (unsigned)((line * p.Xp + t) * (int)sizeof(float2)); // lane: pixel (line, t); + k*R2 elements
    const size_t per_pos = (size_t)p.n_steps * R1 * GE::NT;
    float2* stash0 = p.stash + (size_t)b * per_pos;                           // scalar
    float2* gtile0 = p.gtile + (size_t)b * per_pos;                           // scalar
    const bool do_grad = (p.want_grad != 0);
    const int n_steps = p.n_steps;
    const float k1 = p.k1, sigma = p.sigma;

    cf a[R1];
#pragma unroll
    for (int k = 0; k < R1; ++k) a[k] = c.act1 ? p.probe[line * N + k * R2 + t] : make_float2(0.f, 0.f);

    // =================================== forward sweep ===================================
    {
        float2 db[R1];
        if (c.act1) {
#pragma unroll
            for (int k = 0; k < R1; ++k) db[k] = ldg2(tile0 + k * R2, toff);
        }
        for (int step = 0; step < n_steps; ++step) {
            if (c.act1) {
                modulate<R1, false>(a, db, k1, sigma);
                if (do_grad) ws_store<R1>(stash0 + (size_t)step * R1 * GE::NT, GE::NT, tid, a);
                if (step + 1 < n_steps) {
                    const float2* tb = tile0 + (size_t)(step + 1) * slice_stride;
#pragma unroll
                    for (int k = 0; k < R1; ++k) db[k] = ldg2(tb + k * R2, toff);
                }
            }
            if (step < n_steps - 1) l_convolve<N, R1, R2, false>(c, a);
        }
    }

    // =================================== detector plane, loss ===================================
    cf bb[R2];
    float lsum = 0.f;
    if (p.det_mode == ADM_DET_FRESNEL_) {
        // one-off convolution with the detector-plane Fresnel kernel: rebuild the table, restore it afterwards
        __syncthreads();
        const double n2 = (double)(N * N);
        for (int i = tid; i < LG::HTAB; i += GE::NT) {
            const int ks = i / N, ky = i - ks * N;
            const cf h = p.hfree[ky * N + ks];
            htab[i] = make_float2((float)((double)h.x / n2), (float)((double)h.y / n2));
        }
        __syncthreads();
        l_convolve<N, R1, R2, false>(c, a);
    } else if (p.det_mode == ADM_DET_FARFIELD_) {
        if (p.det_inverse) {
#pragma unroll
            for (int k = 0; k < R1; ++k) a[k] = conjf2(a[k]);
        }
        l_fft2_to_regs<N, R1, R2>(c, a, bb);
    }
    if (p.det_mode == ADM_DET_FARFIELD_) {
        if (c.act2) {
            const int mx = (kx + N / 2) % N;
#pragma unroll
            for (int k = 0; k < R2; ++k) {
                const int my = (tc2 + R1 * k + N / 2) % N;
                const size_t di = ((size_t)b * N + my) * N + mx;
                const cf psi = cscale(bb[k], p.det_scale);
                const float mag = sqrtf(psi.x * psi.x + psi.y * psi.y);
                float g;
                const float wq = p.det_weight ? p.det_weight[my * N + mx] : 1.f;
                lsum += wq * loss_term(mag, p.target[di], p, g);
                if (p.pred) p.pred[di] = mag;
                bb[k] = cscale(psi, wq * g * p.det_scale);
            }
        }
    } else if (c.act1) {
#pragma unroll
        for (int k = 0; k < R1; ++k) {
            const size_t di = ((size_t)b * N + line) * N + k * R2 + t;
            const float mag = sqrtf(a[k].x * a[k].x + a[k].y * a[k].y);
            float g;
            const float wq = p.det_weight ? p.det_weight[line * N + k * R2 + t] : 1.f;
            lsum += wq * loss_term(mag, p.target[di], p, g);
            if (p.pred) p.pred[di] = mag;
            a[k] = cscale(a[k], wq * g);
        }
    }
    {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) lsum += __shfl_down(lsum, off, 64);
        if (lane == 0) red[wave] = lsum;
        __syncthreads();
        if (tid == 0) {
            float s = 0.f;
            for (int w = 0; w < GE::NWAVES; ++w) s += red[w];
            p.loss_sum[b] = s;
        }
    }
    if (!do_grad) return;

    // =================================== detector adjoint ===================================
    if (p.det_mode == ADM_DET_FARFIELD_) {
        l_ifft2_from_regs<N, R1, R2>(c, bb, a);
        if (p.det_inverse) {
#pragma unroll
            for (int k = 0; k < R1; ++k) a[k] = conjf2(a[k]);
        }
    } else if (p.det_mode == ADM_DET_FRESNEL_) {
        l_convolve<N, R1, R2, true>(c, a);
        __syncthreads();
        const double n2 = (double)(N * N);
        for (int i = tid; i < LG::HTAB; i += GE::NT) {
            const int ks = i / N, ky = i - ks * N;
            const cf h = p.h[ky * N + ks];
            htab[i] = make_float2((float)((double)h.x / n2), (float)((double)h.y / n2));
        }
        __syncthreads();
    }

    // =================================== reverse sweep ===================================
    {
        const float sk1 = sigma * k1;
        float2 db[R1];
        cf psi[R1];
        if (c.act1) {
            const float2* tb = tile0 + (size_t)(n_steps - 1) * slice_stride;
#pragma unroll
            for (int k = 0; k < R1; ++k) db[k] = ldg2(tb + k * R2, toff);
            ws_load<R1>(stash0 + (size_t)(n_steps - 1) * R1 * GE::NT, GE::NT, tid, psi);
        }
        for (int step = n_steps - 1; step >= 0; --step) {
            if (c.act1) {
                // z = conj(G) * psi'; d/ddelta = sigma k1 Im z, d/dbeta = -k1 Re z  
                {
                    float2 g[R1];
#pragma unroll
                    for (int k = 0; k < R1; ++k) {
                        const float zr = a[k].x * psi[k].x + a[k].y * psi[k].y;
                        const float zi = a[k].x * psi[k].y - a[k].y * psi[k].x;
                        g[k] = make_float2(sk1 * zi, -k1 * zr);
                    }
                    ws_store<R1>(gtile0 + (size_t)step * R1 * GE::NT, GE::NT, tid, g);
                }
                modulate<R1, true>(a, db, k1, sigma);
                if (step > 0) ws_load<R1>(stash0 + (size_t)(step - 1) * R1 * GE::NT, GE::NT, tid, psi);
            }
            if (step > 0) {
                // the stored field (HBM) is requested a whole convolution ahead, the slice data (L2 / Infinity Cache)
                // half a convolution ahead
                auto mid = [&]() {
                    if (c.act1) {
                        const float2* tb = tile0 + (size_t)(step - 1) * slice_stride;
#pragma unroll
                        for (int k = 0; k < R1; ++k) db[k] = ldg2(tb + k * R2, toff);
                    }
                };
                l_convolve<N, R1, R2, true>(c, a, mid);
            }
        }
    }
    if (p.grad_probe && c.act1) {
        // this position's own slot (plain stores); probe_grad_reduce_kernel sums the slots in a fixed order
        float2* gp = p.grad_probe + (size_t)b * p.gprobe_bstride + line * N + t;
#pragma unroll
        for (int k = 0; k < R1; ++k) gp[k * R2] = a[k];
    }
}

// The lean kernel serves: one probe mode, delta/beta unknowns, binning 1, one probe set for all positions, and a
// transfer function with H(ky, kx) == H(ky, N - kx) (every get_kernel() output; checked at plan creation).
bool ms_lean_supported(int n) { return n == 72 || n == 64; }

hipError_t ms_lean_launch(int n, const MsParams& p, int batch, hipStream_t st) {
    switch (n) {
        case 72: hipLaunchKernelGGL((ms_lean_kernel<72, 8, 9>), dim3(batch), dim3(Geo<72, 8, 9>::NT), 0, st, p); break;
        case 64: hipLaunchKernelGGL((ms_lean_kernel<64, 8, 8>), dim3(batch), dim3(Geo<64, 8, 8>::NT), 0, st, p); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace adm
