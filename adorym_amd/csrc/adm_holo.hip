// libadm -- multi-distance near-field holography (adorym/forward_model.py:809-1092).
//
// One undivided field of view (the reference's n_blocks == 1 case, config 5: 512 x 512), one object slice:
//     psi   = probe * c(obj)                                   c = re + i im (real_imag) or exp(-k1 beta) e^{-i sigma k1 delta}
//     Psi_d = IFFT2( FFT2(psi) * H_d ),  H_d = exp(-i sigma PI lambda d (u^2 + v^2))      (propagate.py:84-103, 556-568)
//     loss  = mean_{d,pixels} ( |Psi_d| - sqrt|A_d(data_d)| )^2                            (forward_model.py:88-93)
// with A_d the reference's affine registration of the measured hologram (wrappers.py:1158-1174: F.affine_grid +
// F.grid_sample, bilinear, border padding, align_corners=False), and the hand-derived gradients w.r.t. the object, the
// probe, every distance d and every affine matrix.
//
// The fields are far larger than the LDS-resident tiles of the multislice kernel (512^2 * 8 B = 2 MB), so the 2-D transforms
// run as line transforms (Stockham radix-8/4/2 in LDS, lines of N = 16 ... 2048) with transposed stores between the x and the y
// stage.  All traffic stays in L2 / Infinity Cache at these sizes; the path is bound by launch latency, so the element-wise
// stages are fused into the transforms: five kernels per minibatch (below).
#include <vector>
#include <cmath>
#include <cstring>
#include "adm_common.h"
#include "adm_optim.h"
#include "adm_fft.h"
#include "adm_ms_math.h"

struct adm_holo {
    adm_ctx* ctx;
    adm_holo_desc d;
    float2 *tw_y, *tw_x;      // exp(-2 pi i j / N) for N = ny, nx
    float* uv2t;              // [nx][ny]: u^2 + v^2 (nm^-2) of frequency (ky, kx) at [kx][ky], fp32 like the reference's tensors
    float2 *T14, *Ft;         // T14 [nx][ny] x spectra of the rows (K1 -> K2), later [ny][nx] (K4 -> K5); Ft [nx][ny] = FFT2(psi)
    float2 *Wq, *T3;          // [n_dists][ny][nx] / [n_dists][nx][ny] work fields
    float *part3, *part4;     // per-line partial sums: [n_dists][ny][8], [n_dists][nx]
    float* part_s;            // [n_dists][nx][2]: per-line sums of the shift gradient (adm_holo_shift_grad; allocated on first use)
    float* cot;               // adm_holo_set_registration: where K3 leaves dL/d(registered hologram sample), or nullptr
    int direct;               //   ... and whether `data` is taken pixel for pixel (already registered) instead of resampled
};

// transposed store index (row r of pitch p, column c); -DHOLO_ABL_STORE: timing-only variant that stores along the line instead
#ifdef HOLO_ABL_STORE
#define HOLO_TIDX(r, p, c, n) ((size_t)(c) * (n) + (r))
#else
#define HOLO_TIDX(r, p, c, n) ((size_t)(r) * (p) + (c))
#endif

namespace adm {

static inline int grid_for(size_t n) {
    size_t b = (n + 255) / 256;
    return (int)(b > 4096 ? 4096 : (b ? b : 1));
}

// ---------------------------------------------------------------------------------------------------------------
// batched row FFT with transposed store
// ---------------------------------------------------------------------------------------------------------------
template <int N, int NS, int R, bool INV, int TPR>
__device__ __forceinline__ void stockham_pass(const cf* __restrict__ src, cf* __restrict__ dst, const float2* __restrict__ tw, int t) {
    constexpr int NB = N / R;                 // butterflies per row
    constexpr int TWS = N / (NS * R);         // twiddle table stride
    for (int j = t; j < NB; j += TPR) {
        const int k = j % NS;
        cf v[R];
#pragma unroll
        for (int q = 0; q < R; ++q) {
            v[q] = src[j + q * NB];
            if (NS > 1 && q > 0) {
                const cf w = tw[q * k * TWS];
                v[q] = INV ? cmulc(v[q], w) : cmul(v[q], w);
            }
        }
        Dft<R, INV>::run(v);
        const int j0 = (j / NS) * NS * R + k;
#pragma unroll
        for (int q = 0; q < R; ++q) dst[j0 + q * NS] = v[q];
    }
}

// The TPR threads of a line are one wave or part of one when TPR <= 64 (lines of up to 512 points): LDS operations of a wave
// execute in order, so ordering the compiler is all a line-local exchange needs -- no workgroup barrier between the passes of a
// transform, the four lines of a block run at their own pace.  Longer lines span two or four waves and keep the barrier.
template <int TPR> __device__ __forceinline__ void line_sync() {
    if (TPR <= 64) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

template <int N, int NS, bool INV, int TPR> struct Passes {
    static __device__ __forceinline__ int run(cf* a, cf* b, const float2* tw, int t) {
        constexpr int REM = N / NS;
        constexpr int R = REM >= 8 ? 8 : REM;
        stockham_pass<N, NS, R, INV, TPR>(a, b, tw, t);
        line_sync<TPR>();
        return 1 + Passes<N, NS * R, INV, TPR>::run(b, a, tw, t);
    }
};
template <int N, bool INV, int TPR> struct Passes<N, N, INV, TPR> {
    static __device__ __forceinline__ int run(cf*, cf*, const float2*, int) { return 0; }
};

// ---------------------------------------------------------------------------------------------------------------
// Five kernels per minibatch.  Every stage is "lines of length N through LDS": a block of 256 threads works on LPB = 256 / (N / 8)
// lines at once (N / 8 threads per line, Stockham passes above), and everything element-wise -- slice modulation, transfer
// functions of all distances, loss and its derivative, the sums over distances, the object / probe gradient -- runs on the line
// while it is in LDS, between two transforms:
//   K1  rows y      : psi = probe * c(obj)                      -> FFT_x                       -> T1[kx][y]
//   K2  lines kx    : FFT_y -> F (kept: Ft[kx][ky])  ; per d: x H_d -> IFFT_y                  -> Wq[d][y][kx]
//   K3  rows (d, y) : IFFT_x / n -> Psi_d, loss, affine-gradient sums, dL/dPsi -> FFT_x        -> T3[d][kx][y]
//   K4  lines kx    : per d: FFT_y / n -> Gh_d, distance-gradient sums, GF += conj(H_d) Gh_d ; IFFT_y -> T4t[kx][y]
//   K5  rows y      : IFFT_x -> dL/dpsi -> dL/dobj (+=), dL/dprobe (=) ; block 0: the sums of K3 / K4 in fixed order
// (17 launches of 4.5-13 us each before round 4, back to back: the path is bound by the NUMBER of launches.)
// Partial sums: one slot per line (K3: per (d, y), 8 floats; K4: per (d, kx)), summed by K5's block 0 -- or by
// holo_sums_kernel when no gradient is wanted -- in a fixed order: bit-reproducible.
// ---------------------------------------------------------------------------------------------------------------
template <int N> struct LineGeo {
    static constexpr int TPR = N / 8;             // threads per line
    static constexpr int LPB = 256 / TPR;         // lines per block
    // pitch of a line's LDS buffer: N + 32 / min(LPB, 32) elements, so that the block-wide transposed read of
    // store_lines_transposed (a half wave = min(LPB, 32) lines x 32 / min(LPB, 32) consecutive points) touches 32 different
    // 8-byte bank pairs
    static constexpr int LP = N + (LPB > 1 ? 32 / (LPB < 32 ? LPB : 32) : 0);
};

// The x and y stages of a 2-D transform hand their results over transposed.  A line's own threads would store its points 8 bytes
// at a time, pitch * 8 bytes apart: one L2 write transaction per point (1 M of them per K2 / K3 launch at 512 x 512 x 4 -- the
// store-ablated build of round 6 is 15 us of 76 faster).  Instead the block stores its LPB adjacent lines together: thread e writes
// point e / LPB of line e % LPB, so LPB consecutive lanes fill LPB * 8 contiguous bytes (32 B at N = 512: a quarter of the
// transactions).  `base`: LDS address of line 0's result (all lines of a block finish in the same buffer); at(r, c): destination
// of point r of line c.
template <int N, typename At> __device__ __forceinline__ void store_lines_transposed(const cf* base, int n_lines, At at) {
    using LG = LineGeo<N>;
    __syncthreads();                               // every line of the block is complete
    for (int e = threadIdx.x; e < N * LG::LPB; e += 256) {
        const int c = e % LG::LPB, r = e / LG::LPB;
        if (c < n_lines) *at(r, c) = base[c * LG::LP + r];
    }
}
// The passes read 7 twiddles per butterfly: from an LDS copy of the table (N <= 1024; the caller's next barrier orders the copy
// before the first pass), from global memory otherwise (the line buffers of N = 2048 leave no room).
template <int N> struct TwLds {
    static constexpr bool USE = (N <= 1024);
    static constexpr int SIZE = USE ? N : 1;
};
template <int N> __device__ __forceinline__ const float2* stage_twiddles(float2* lds, const float2* __restrict__ tw) {
    if (!TwLds<N>::USE) return tw;
    for (int j = threadIdx.x; j < N; j += 256) lds[j] = tw[j];
    return lds;
}
// transform the line in `a` (filled, block synchronised); returns the buffer that holds the result (block synchronised)
template <int N, bool INV> __device__ __forceinline__ cf* line_fft(cf* a, cf* b, const float2* __restrict__ tw, int t) {
    const int np = Passes<N, 1, INV, LineGeo<N>::TPR>::run(a, b, tw, t);
    return (np & 1) ? b : a;
}

__device__ __forceinline__ cf holo_transmission(float2 o, int real_imag, float k1, float sigma) {
    if (real_imag) return o;
    const float e = expf(-k1 * o.y);
    float sn, cs;
    sincosf(-sigma * k1 * o.x, &sn, &cs);
    return make_float2(e * cs, e * sn);
}
__device__ __forceinline__ cf holo_h(float uv2, float dist_cm, float c1) {
    // -sigma*PI*lambda (c1, rounded once on the host) * dist_nm * (u^2+v^2), every product in fp32 like the reference
    const float arg = (c1 * (dist_cm * 1e7f)) * uv2;
    float sn, cs;
    sincos_fast(arg, sn, cs);       // branch-free, ~1 ulp (adm_ms_math.h); ocml's sincosf was a third of the y-stage kernels' instructions
    return make_float2(cs, sn);
}
// torch's affine_grid base coordinate (see oracle/adorym_oracle.py::affine_sample for the derivation)
__device__ __forceinline__ float base_coord(int i, int n) {
    const float step = 2.0f / (float)(n - 1);
    const float lin = (i < n / 2) ? fmaf(step, (float)i, -1.0f) : fmaf(-step, (float)(n - 1 - i), 1.0f);
    return (lin * (float)(n - 1)) / (float)n;
}

struct HoloArgs {
    const float2 *obj, *probe;
    const float *dists, *affine, *data, *uv2t;      // uv2t: [nx][ny] (u^2 + v^2 of (ky, kx) at [kx][ky])
    float2 *T1, *Ft, *Wq, *T3, *T4;                  // T1 [nx][ny], Ft [nx][ny], Wq [nd][ny][nx], T3 [nd][nx][ny], T4 [ny][nx]
    const float2 *tw_x, *tw_y;
    float *pred, *part3, *part4;                    // part3 [nd][ny][8], part4 [nd][nx]
    float* cot;                                     // [nd][ny][nx] dL/d(sample of the registered hologram), or nullptr
    int direct;                                     // data[d][y][x] IS the registered sample (no affine resampling)
    float2 *grad_obj, *grad_probe;
    float *loss_sum, *grad_affine, *grad_dists;
    int ny, nx, nd, real_imag, intensity, want_affine, want_dists, set_obj;
    float k1, sigma, c1, inv_n, gscale;
    // adm_holo_fwd_adj_adam: the Adam steps of the object, the distances and the affine matrices run where their gradients are
    // produced (K5), the gradients themselves are never stored.  NULL moments: that parameter is not updated.
    float2* obj_rw;
    float *dists_rw, *affine_rw;
    float *m_obj, *v_obj, *m_d, *v_d, *m_a, *v_a;
    const float* a_pin;
    unsigned long long a_pin_n;
    AdamScalars adam;                               // step of the object; the other two step sizes below
    float step_d, step_a;
};

// sum of v over the TPR threads of a line group (TPR a power of two <= 256); every thread of the block calls it.  A line of up to
// 64 threads is (part of) one wave: a fixed shuffle tree, no LDS, no workgroup barrier (the lines of a block stay independent);
// longer lines go through red ([256] floats of LDS).  Either way the order of the additions is fixed: bit-reproducible.
template <int TPR> __device__ __forceinline__ float line_sum(float v, float* red) {
    if (TPR <= 64) {
#pragma unroll
        for (int s = TPR / 2; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
        return v;
    }
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
#pragma unroll
    for (int s = TPR / 2; s > 0; s >>= 1) {
        if ((tid % TPR) < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    const float r = red[tid - tid % TPR];
    __syncthreads();
    return r;
}

// the same for Q quantities at once (one set of barriers); red: [Q][256]
template <int TPR, int Q> __device__ __forceinline__ void line_sums(float (&v)[Q], float (*red)[256]) {
    if (TPR <= 64) {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
#pragma unroll
            for (int s = TPR / 2; s > 0; s >>= 1) v[q] += __shfl_xor(v[q], s, 64);
        }
        return;
    }
    const int tid = threadIdx.x;
#pragma unroll
    for (int q = 0; q < Q; ++q) red[q][tid] = v[q];
    __syncthreads();
#pragma unroll
    for (int s = TPR / 2; s > 0; s >>= 1) {
        if ((tid % TPR) < s) {
#pragma unroll
            for (int q = 0; q < Q; ++q) red[q][tid] += red[q][tid + s];
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) v[q] = red[q][tid - tid % TPR];
    __syncthreads();
}

template <int NX> __global__ __launch_bounds__(256) void holo_k1(HoloArgs A) {
    using LG = LineGeo<NX>;
    __shared__ cf buf[2][LG::LPB * LG::LP];
    __shared__ float2 twl[TwLds<NX>::SIZE];
    const float2* tw = stage_twiddles<NX>(twl, A.tw_x);
    const int ll = threadIdx.x / LG::TPR, t = threadIdx.x % LG::TPR;
    const int y0 = blockIdx.x * LG::LPB, y = y0 + ll;
    const bool ok = y < A.ny;
    cf* a = buf[0] + ll * LG::LP;
    cf* b = buf[1] + ll * LG::LP;
    for (int x = t; x < NX; x += LG::TPR) {
        const size_t i = (size_t)y * NX + x;
        a[x] = ok ? cmul(A.probe[i], holo_transmission(A.obj[i], A.real_imag, A.k1, A.sigma)) : make_float2(0.f, 0.f);
    }
    __syncthreads();
    const cf* res = line_fft<NX, false>(a, b, tw, t);
    float2* T1 = A.T1;
    const int ny = A.ny;
    store_lines_transposed<NX>(res - ll * LG::LP, min(LG::LPB, ny - y0), [=](int k, int c) { return T1 + HOLO_TIDX(k, ny, y0 + c, NX); });
}

// LPB adjacent lines kx per block (so that the transposed stores are LPB x 8 bytes long) and one distance per blockIdx.y: the
// forward transform of a line is repeated for every distance (cheaper than a kernel boundary); blockIdx.y == 0 keeps F
template <int NY> __global__ __launch_bounds__(256) void holo_k2(HoloArgs A) {
    using LG = LineGeo<NY>;
    __shared__ cf buf[2][LG::LPB * LG::LP];
    __shared__ float2 twl[TwLds<NY>::SIZE];
    const float2* tw = stage_twiddles<NY>(twl, A.tw_y);
    const int ll = threadIdx.x / LG::TPR, t = threadIdx.x % LG::TPR;
    const int kx0 = blockIdx.x * LG::LPB, kx = kx0 + ll, d = blockIdx.y;
    const bool ok = kx < A.nx;
    cf* a = buf[0] + ll * LG::LP;
    cf* b = buf[1] + ll * LG::LP;
    for (int j = t; j < NY; j += LG::TPR) a[j] = ok ? A.T1[(size_t)kx * NY + j] : make_float2(0.f, 0.f);
    __syncthreads();
    cf* F = line_fft<NY, false>(a, b, tw, t);
    cf* p = (F == a) ? b : a;
    if (ok && d == 0)
        for (int k = t; k < NY; k += LG::TPR) A.Ft[(size_t)kx * NY + k] = F[k];
    const float dist = A.dists[d];
    for (int k = t; k < NY; k += LG::TPR) p[k] = ok ? cmul(F[k], holo_h(A.uv2t[(size_t)kx * NY + k], dist, A.c1)) : make_float2(0.f, 0.f);
    line_sync<LG::TPR>();
    const cf* res = line_fft<NY, true>(p, F, tw, t);      // (F's buffer is free once p is filled)
    float2* Wq = A.Wq + (size_t)d * NY * A.nx;
    const int nx = A.nx;
    store_lines_transposed<NY>(res - ll * LG::LP, min(LG::LPB, nx - kx0), [=](int y, int c) { return Wq + HOLO_TIDX(y, nx, kx0 + c, NY); });
}

template <int NX, bool GRAD> __global__ __launch_bounds__(256) void holo_k3(HoloArgs A) {
    using LG = LineGeo<NX>;
    __shared__ cf buf[2][LG::LPB * LG::LP];
    __shared__ float red[7][256];
    __shared__ float2 twl[TwLds<NX>::SIZE];
    const float2* tw = stage_twiddles<NX>(twl, A.tw_x);
    const int ll = threadIdx.x / LG::TPR, t = threadIdx.x % LG::TPR;
    const int job0 = blockIdx.x * LG::LPB, job = job0 + ll;        // job = d * ny + y
    const bool ok = job < A.nd * A.ny;
    const int d = ok ? job / A.ny : 0, y = ok ? job % A.ny : 0;
    cf* a = buf[0] + ll * LG::LP;
    cf* b = buf[1] + ll * LG::LP;
    float th[6] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f};
    if (A.affine) {
#pragma unroll
        for (int q = 0; q < 6; ++q) th[q] = A.affine[d * 6 + q];
    }
    const int ny = A.ny;
    const size_t n = (size_t)ny * NX;
    const float* img = A.data + (size_t)d * n;
    const float Y = base_coord(y, ny);
    // The four samples of the measured hologram every pixel of the line needs depend on the affine matrix only: their loads are
    // issued NOW, ahead of the line's fetch and inverse transform, and wait in registers (32 per thread) -- issued in the element-wise
    // stage they were a dependent trip to memory per pixel in the longest kernel of the minibatch.
    constexpr int PPT = NX / LG::TPR;
    float s00[PPT], s01[PPT], s10[PPT], s11[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const float X = base_coord(t + j * LG::TPR, NX);
        float ix = (((th[0] * X + th[1] * Y + th[2]) + 1.f) * (float)NX - 1.f) * 0.5f;
        float iy = (((th[3] * X + th[4] * Y + th[5]) + 1.f) * (float)ny - 1.f) * 0.5f;
        ix = fminf(fmaxf(ix, 0.f), (float)(NX - 1));
        iy = fminf(fmaxf(iy, 0.f), (float)(ny - 1));
        const int x0 = (int)floorf(ix), y0 = (int)floorf(iy);
        const int x1 = min(x0 + 1, NX - 1), y1 = min(y0 + 1, ny - 1);
        s00[j] = img[(size_t)y0 * NX + x0]; s01[j] = img[(size_t)y0 * NX + x1];
        s10[j] = img[(size_t)y1 * NX + x0]; s11[j] = img[(size_t)y1 * NX + x1];
    }
    for (int k = t; k < NX; k += LG::TPR) a[k] = ok ? A.Wq[((size_t)d * A.ny + y) * NX + k] : make_float2(0.f, 0.f);
    __syncthreads();
    cf* res = line_fft<NX, true>(a, b, tw, t);
    cf* oth = (res == a) ? b : a;
    float lsum = 0.f, ga[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const int x = t + j * LG::TPR;
        const float X = base_coord(x, NX);
        const float gx = th[0] * X + th[1] * Y + th[2];
        const float gy = th[3] * X + th[4] * Y + th[5];
        float ix = ((gx + 1.f) * (float)NX - 1.f) * 0.5f;
        float iy = ((gy + 1.f) * (float)ny - 1.f) * 0.5f;
        const float mx = (ix > 0.f && ix < (float)(NX - 1)) ? 1.f : 0.f;     // grid_sampler's clip_coordinates_set_grad
        const float my = (iy > 0.f && iy < (float)(ny - 1)) ? 1.f : 0.f;
        ix = fminf(fmaxf(ix, 0.f), (float)(NX - 1));
        iy = fminf(fmaxf(iy, 0.f), (float)(ny - 1));
        const int x0 = (int)floorf(ix), y0 = (int)floorf(iy);
        const float wx = ix - (float)x0, wy = iy - (float)y0;
        const float v00 = s00[j], v01 = s01[j], v10 = s10[j], v11 = s11[j];
        const float samp = A.direct ? img[(size_t)y * NX + x]
                                    : v00 * (1.f - wx) * (1.f - wy) + v01 * wx * (1.f - wy) + v10 * (1.f - wx) * wy + v11 * wx * wy;
        const float as = fabsf(samp);
        const float tgt = A.intensity ? sqrtf(as) : as;
        const float2 ps = cscale(res[x], A.inv_n);                  // Psi_d(y, x): normalised inverse
        const float mag = sqrtf(ps.x * ps.x + ps.y * ps.y);
        const float diff = mag - tgt;
        if (ok) {
            lsum += diff * diff;
            if (A.pred) A.pred[(size_t)d * n + (size_t)y * NX + x] = mag;
        }
        if (GRAD) {
            const float g = (mag > 0.f) ? A.gscale * diff / mag : 0.f;
            oth[x] = cscale(ps, g);                                  // dL/dPsi_d
            if ((A.want_affine || A.cot) && ok) {
                const float sg = (float)((samp > 0.f) - (samp < 0.f));
                float cot = -A.gscale * diff * (A.intensity ? sg / (2.f * sqrtf(as)) : sg);
                if (!(fabsf(cot) <= 3.0e38f)) cot = 0.f;             // 0/0 at an exactly zero sample
                if (A.cot) A.cot[(size_t)d * n + (size_t)y * NX + x] = cot;
                const float dix = ((v01 - v00) * (1.f - wy) + (v11 - v10) * wy) * mx * (0.5f * (float)NX) * cot;
                const float diy = ((v10 - v00) * (1.f - wx) + (v11 - v01) * wx) * my * (0.5f * (float)ny) * cot;
                ga[0] += dix * X; ga[1] += dix * Y; ga[2] += dix;
                ga[3] += diy * X; ga[4] += diy * Y; ga[5] += diy;
            }
        }
    }
    // one slot of 8 floats per line (slot 0: loss, 1..6: affine gradient)
    if (GRAD && A.want_affine) {
        float v[7] = {lsum, ga[0], ga[1], ga[2], ga[3], ga[4], ga[5]};
        line_sums<LG::TPR, 7>(v, red);
        if (ok && t == 0) {
#pragma unroll
            for (int q = 0; q < 7; ++q) A.part3[(size_t)job * 8 + q] = v[q];
        }
    } else {
        const float ls = line_sum<LG::TPR>(lsum, red[0]);
        if (ok && t == 0) A.part3[(size_t)job * 8] = ls;
    }
    if (!GRAD) return;
    line_sync<LG::TPR>();
    const cf* G = line_fft<NX, false>(oth, res, tw, t);
    float2* T3 = A.T3;
    store_lines_transposed<NX>(G - ll * LG::LP, min(LG::LPB, A.nd * ny - job0), [=](int k, int c) {
        const int j = job0 + c, dd = j / ny, yy = j % ny;          // (the lines of a block may belong to different distances when ny < LPB)
        return T3 + (size_t)dd * NX * ny + HOLO_TIDX(k, ny, yy, NX);
    });
}

// Gh_d = FFT2(dL/dPsi_d) / n.  GF = sum_d conj(H_d) Gh_d;  dL/dd_cm = 1e7 * Re sum_k conj(Gh_d) (-i sigma PI lambda uv2) H_d F
template <int NY> __global__ __launch_bounds__(256) void holo_k4(HoloArgs A) {
    using LG = LineGeo<NY>;
    static_assert(3 * LG::LPB * LG::LP * sizeof(cf) + 256 * sizeof(float) + TwLds<NY>::SIZE * sizeof(float2) <= 64 * 1024,
                  "holo_k4: three line buffers + reduction scratch + twiddle copy must fit the 64 KB a workgroup may allocate");
    __shared__ cf buf[3][LG::LPB * LG::LP];
    __shared__ float red[256];
    __shared__ float2 twl[TwLds<NY>::SIZE];
    const float2* tw = stage_twiddles<NY>(twl, A.tw_y);
    const int ll = threadIdx.x / LG::TPR, t = threadIdx.x % LG::TPR;
    const int kx = blockIdx.x;                        // one line per block, slot s works on distances s, s + LPB, ...
    cf* a = buf[0] + ll * LG::LP;
    cf* b = buf[1] + ll * LG::LP;
    cf* GF = buf[2] + ll * LG::LP;
    for (int k = t; k < NY; k += LG::TPR) GF[k] = make_float2(0.f, 0.f);
    for (int d0 = 0; d0 < A.nd; d0 += LG::LPB) {
        const int d = d0 + ll;
        const bool ok = d < A.nd;
        for (int j = t; j < NY; j += LG::TPR) a[j] = ok ? A.T3[((size_t)d * A.nx + kx) * NY + j] : make_float2(0.f, 0.f);
        if (d0 == 0) __syncthreads();                 // (also orders the twiddle copy of stage_twiddles before the first pass)
        else line_sync<LG::TPR>();
        const cf* res = line_fft<NY, false>(a, b, tw, t);
        const float dist = ok ? A.dists[d] : 0.f;
        float acc = 0.f;
        for (int k = t; k < NY; k += LG::TPR) {
            const size_t e = (size_t)kx * NY + k;
            const float u2 = A.uv2t[e];
            const cf h = holo_h(u2, dist, A.c1);
            const cf g = cscale(res[k], A.inv_n);
            GF[k] = cadd(GF[k], cmulc(g, h));
            if (A.want_dists && ok) {
                const cf hf = cmul(h, A.Ft[e]);
                acc += -A.c1 * u2 * (g.x * hf.y - g.y * hf.x);      // Re( conj(g) (i c1 u2) hf ) = -c1 u2 Im(conj(g) hf)
            }
        }
        if (A.want_dists) {
            const float sd = line_sum<LG::TPR>(acc, red);
            if (ok && t == 0) A.part4[(size_t)d * A.nx + kx] = sd;
        }
        line_sync<LG::TPR>();                         // a / b are refilled by the next round of distances
    }
    // the slots' sums over their distances, added in slot order into slot 0, which transforms back
    if (LG::LPB > 1) {
        __syncthreads();                              // every slot's GF is complete
        for (int k = threadIdx.x; k < NY; k += 256) {
            cf sum = buf[2][k];
#pragma unroll
            for (int s_ = 1; s_ < LG::LPB; ++s_) sum = cadd(sum, buf[2][s_ * LG::LP + k]);
            buf[2][k] = sum;
        }
        __syncthreads();
    }
    const cf* gy = line_fft<NY, true>(GF, a, tw, t);      // (every slot runs it on its own buffers; slot 0's counts)
    // one line kx per block: nothing to store side by side, so the line goes out as it is (T4t[kx][y], contiguous) and K5 -- whose
    // blocks own LPB adjacent rows y -- does the transposition on its way in, LPB * 8 bytes per column
    if (ll == 0)
        for (int y = t; y < NY; y += LG::TPR) A.T4[(size_t)kx * NY + y] = gy[y];
}

// the per-line sums of K3 / K4 in a fixed order: loss_sum[d] =, grad_affine[d][q] +=, grad_dists[d] += 1e7 * ...
// One wave per output: lanes stride over the lines, then a fixed shuffle tree.
template <bool FUSE = false>
__device__ __forceinline__ void holo_sum_one(const HoloArgs& A, bool grad, int o) {      // called by one whole wave
    const int lane = threadIdx.x & 63;
    const int d = o >> 3, q = o & 7;
    const bool want = (q == 0) || (q < 7 ? (grad && A.want_affine) : (grad && A.want_dists));
    if (!want) return;
    float s = 0.f;
    if (q < 7) { for (int y = lane; y < A.ny; y += 64) s += A.part3[((size_t)d * A.ny + y) * 8 + q]; }
    else { for (int x = lane; x < A.nx; x += 64) s += A.part4[(size_t)d * A.nx + x]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) {
        if (q == 0) A.loss_sum[d] = s;
        else if (FUSE) {
            // the parameter's Adam step right here (adm_adam_step_small's arithmetic on the gradient an overwriting launch would store)
            AdamScalars a = A.adam;
            float mo, vo;
            if (q < 7) {
                const int i = d * 6 + q - 1;
                a.step = A.step_a;
                float xn = adam_value(A.affine_rw[i], 0.f + s, A.m_a[i], A.v_a[i], a, (size_t)i, mo, vo);
                A.m_a[i] = mo; A.v_a[i] = vo;
                if (A.a_pin && (unsigned long long)i < A.a_pin_n) xn = A.a_pin[i];
                A.affine_rw[i] = xn;
            } else {
                a.step = A.step_d;
                const float xn = adam_value(A.dists_rw[d], 0.f + 1e7f * s, A.m_d[d], A.v_d[d], a, (size_t)d, mo, vo);
                A.m_d[d] = mo; A.v_d[d] = vo;
                A.dists_rw[d] = xn;
            }
        }
        else if (q < 7) A.grad_affine[d * 6 + q - 1] = (A.set_obj ? 0.f : A.grad_affine[d * 6 + q - 1]) + s;
        else A.grad_dists[d] = (A.set_obj ? 0.f : A.grad_dists[d]) + 1e7f * s;
    }
}
__global__ __launch_bounds__(64) void holo_sums_kernel(HoloArgs A) { holo_sum_one<false>(A, false, blockIdx.x); }

// dL/dpsi -> dL/dobj (accumulated) and dL/dprobe (written)
template <int NX, bool FUSE> __global__ __launch_bounds__(256) void holo_k5(HoloArgs A) {
    using LG = LineGeo<NX>;
    __shared__ cf buf[2][LG::LPB * LG::LP];
    __shared__ float2 twl[TwLds<NX>::SIZE];
    const float2* tw = stage_twiddles<NX>(twl, A.tw_x);
    const int ll = threadIdx.x / LG::TPR, t = threadIdx.x % LG::TPR;
    const int y = blockIdx.x * LG::LPB + ll;
    const bool ok = y < A.ny;
    cf* a = buf[0] + ll * LG::LP;
    cf* b = buf[1] + ll * LG::LP;
    // what the epilogue needs of the thread's 8 pixels -- object, probe and (fused update) the moments -- is requested NOW, before
    // the line is fetched and transformed: with 128 blocks on 256 CUs nothing else hides these loads (they cost a dependent trip to
    // memory per pixel when issued in the epilogue: 12 -> 18 us with the moments added)
    constexpr int PPT = NX / LG::TPR;
    float2 ob[PPT], pb[PPT], mob[FUSE ? PPT : 1], vob[FUSE ? PPT : 1];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        const size_t i = (size_t)(ok ? y : 0) * NX + t + j * LG::TPR;
        ob[j] = A.obj[i];
        pb[j] = A.probe[i];
        if (FUSE) {
            mob[j] = reinterpret_cast<const float2*>(A.m_obj)[i];
            vob[j] = reinterpret_cast<const float2*>(A.v_obj)[i];
        }
    }
    // T4t[kx][y]: thread e fetches column e / LPB of row e % LPB, so LPB consecutive lanes read LPB * 8 contiguous bytes
    {
        const int y0 = blockIdx.x * LG::LPB, ny = A.ny;
        for (int e = threadIdx.x; e < NX * LG::LPB; e += 256) {
            const int c = e % LG::LPB, k = e / LG::LPB;
            buf[0][c * LG::LP + k] = (y0 + c < ny) ? A.T4[(size_t)k * ny + y0 + c] : make_float2(0.f, 0.f);
        }
    }
    __syncthreads();
    const cf* res = line_fft<NX, true>(a, b, tw, t);
    if (ok)
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const int x = t + j * LG::TPR;
            const size_t i = (size_t)y * NX + x;
            const cf g = res[x];
            const cf c = holo_transmission(ob[j], A.real_imag, A.k1, A.sigma);
            const cf p = pb[j];
            if (A.grad_probe) A.grad_probe[i] = cmulc(g, c);
            float2 go = (FUSE || A.set_obj) ? make_float2(0.f, 0.f) : A.grad_obj[i];
            if (A.real_imag) {
                const cf z = cmulc(g, p);                  // G conj(probe)
                go.x += z.x;
                go.y += z.y;
            } else {
                const cf pm = cmul(p, c);                  // post-modulation field psi'
                const float wre = g.x * pm.x + g.y * pm.y, wim = g.x * pm.y - g.y * pm.x;     // w = conj(G) psi'
                go.y += -A.k1 * wre;                       // d/dbeta
                go.x += A.sigma * A.k1 * wim;              // d/ddelta
            }
            if (FUSE) {
                // the object's Adam step on the gradient just formed (channels 2 i and 2 i + 1 of the flat object array)
                const float2 xo = ob[j];
                const float2 mo = mob[j], vo = vob[j];
                float2 xn, mn, vn;
                xn.x = adam_value(xo.x, go.x, mo.x, vo.x, A.adam, 2 * i, mn.x, vn.x);
                xn.y = adam_value(xo.y, go.y, mo.y, vo.y, A.adam, 2 * i + 1, mn.y, vn.y);
                reinterpret_cast<float2*>(A.m_obj)[i] = mn;
                reinterpret_cast<float2*>(A.v_obj)[i] = vn;
                A.obj_rw[i] = xn;
            } else {
                A.grad_obj[i] = go;
            }
        }
    // the sums of K3 / K4: output o by wave 0 of block o % gridDim.x (a handful of blocks carry one or two each)
    if (threadIdx.x < 64)
        for (int o = blockIdx.x; o < A.nd * 8; o += gridDim.x) holo_sum_one<FUSE>(A, true, o);
}

#define ADM_HOLO_SIZES(X) X(16) X(32) X(64) X(128) X(256) X(512) X(1024) X(2048)
// ---------------------------------------------------------------------------------------------------------------
// Per-distance shift refinement of the measured holograms (optimize_all_probe_pos with multi-distance data,
// adorym/forward_model.py:1075-1085; demos/2d_multidist_holography_w_position_correction.py): the loss compares with
//     T_d = Re IFFT2( FFT2(|data_d|) * Phi_d ),   Phi_d(ky, kx) = exp(-2 PI i (fx(kx) s_d[1] + fy(ky) s_d[0]))     (util.py:380-397)
// instead of data_d.  D^_d = FFT2(|data_d|) does not depend on anything that is optimised: computed once per dataset (SR + SLF<0>),
// kept transposed [d][kx][ky].  Per minibatch: SLI (lines kx: D^ Phi -> IFFT_y) + SRI (rows: IFFT_x, real part) in front of K1..K5,
// which then take T_d pixel for pixel (HoloArgs::direct) and leave c = dL/dT_d (HoloArgs::cot); behind them SR + SLF<1>:
//     dL/ds_d[q] = sum_pixels c dT_d/ds_d[q] = (1/n) Re sum_k conj(FFT2(c))_k D^_k Phi_k (-2 PI i f_q(k)) = (2 PI / n) sum_k f_q(k) Im(conj(C^_k) D^_k Phi_k)
// per-line sums in fixed order, then one wave per output.  Same line transforms, LDS layout and transposed stores as K1..K5.
// ---------------------------------------------------------------------------------------------------------------
struct ShiftArgs {
    const float* img;          // SR: real rows [nd][ny][nx]
    int absval;                // SR: take |img| (raw data: forward_model.py:1055) or img as it is (the cotangent)
    float2* T;                 // SR out / SLF in: [nd][nx][ny]
    float2* spec_out;          // SLF<0>: [nd][nx][ny]
    const float2* spec;        // SLI / SLF<1>: D^
    const float* shifts;       // [nd][2] (sy, sx)
    float2* W;                 // SLI out / SRI in: [nd][ny][nx]
    float* out;                // SRI: [nd][ny][nx]
    float* part;               // SLF<1>: [nd][nx][2]
    float* grad_shifts;        // [nd][2], accumulated
    const float2 *tw_x, *tw_y;
    int ny, nx, nd;
    float inv_n;
};
__device__ __forceinline__ float fft_freq(int k, int n) { return (float)(k < (n + 1) / 2 ? k : k - n) / (float)n; }   // np.fft.fftfreq(n, 1)

// rows (d, y) of a real image -> FFT_x -> T[d][kx][y]
template <int NX> __global__ __launch_bounds__(256) void holo_sr(ShiftArgs A) {
    using LG = LineGeo<NX>;
    __shared__ cf buf[2][LG::LPB * LG::LP];
    __shared__ float2 twl[TwLds<NX>::SIZE];
    const float2* tw = stage_twiddles<NX>(twl, A.tw_x);
    const int ll = threadIdx.x / LG::TPR, t = threadIdx.x % LG::TPR;
    const int y0 = blockIdx.x * LG::LPB, y = y0 + ll, d = blockIdx.y;
    const bool ok = y < A.ny;
    cf* a = buf[0] + ll * LG::LP;
    cf* b = buf[1] + ll * LG::LP;
    const float* row = A.img + ((size_t)d * A.ny + (ok ? y : 0)) * NX;
    for (int x = t; x < NX; x += LG::TPR) {
        const float v = ok ? row[x] : 0.f;
        a[x] = make_float2(A.absval ? fabsf(v) : v, 0.f);
    }
    __syncthreads();
    const cf* res = line_fft<NX, false>(a, b, tw, t);
    float2* T = A.T + (size_t)d * NX * A.ny;
    const int ny = A.ny;
    store_lines_transposed<NX>(res - ll * LG::LP, min(LG::LPB, ny - y0), [=](int k, int c) { return T + HOLO_TIDX(k, ny, y0 + c, NX); });
}

// lines (d, kx): FFT_y of T.  MODE 0: the spectrum is stored ([d][kx][ky]); MODE 1: the two sums of the shift gradient
template <int NY, int MODE> __global__ __launch_bounds__(256) void holo_slf(ShiftArgs A) {
    using LG = LineGeo<NY>;
    __shared__ cf buf[2][LG::LPB * LG::LP];
    __shared__ float red[2][256];
    __shared__ float2 twl[TwLds<NY>::SIZE];
    const float2* tw = stage_twiddles<NY>(twl, A.tw_y);
    const int ll = threadIdx.x / LG::TPR, t = threadIdx.x % LG::TPR;
    const int kx0 = blockIdx.x * LG::LPB, kx = kx0 + ll, d = blockIdx.y;
    const bool ok = kx < A.nx;
    cf* a = buf[0] + ll * LG::LP;
    cf* b = buf[1] + ll * LG::LP;
    const size_t line = ((size_t)d * A.nx + (ok ? kx : 0)) * NY;
    for (int j = t; j < NY; j += LG::TPR) a[j] = ok ? A.T[line + j] : make_float2(0.f, 0.f);
    __syncthreads();
    const cf* G = line_fft<NY, false>(a, b, tw, t);
    if (MODE == 0) {
        if (ok)
            for (int k = t; k < NY; k += LG::TPR) A.spec_out[line + k] = G[k];
        return;
    }
    const float sy = A.shifts[2 * d], sx = A.shifts[2 * d + 1];
    const float fx = fft_freq(kx, A.nx);
    // (the terms cancel almost completely -- for uncorrelated data the sum is ~1e-3 of the sum of their magnitudes: the eight terms of
    // a thread and, below, the nx line sums are added in double; the tree over a line's threads stays in float)
    double v0 = 0.0, v1 = 0.0;
    if (ok) {
        for (int k = t; k < NY; k += LG::TPR) {
            const float fy = fft_freq(k, NY);
            float sn, cs;
            sincosf(-2.f * 3.14159265359f * (fx * sx + fy * sy), &sn, &cs);
            const cf c = cmul(cmul(conjf2(G[k]), A.spec[line + k]), make_float2(cs, sn));
            v0 += (double)(fy * c.y);
            v1 += (double)(fx * c.y);
        }
    }
    float v[2] = {(float)v0, (float)v1};
    line_sums<LG::TPR, 2>(v, red);
    if (ok && t == 0) {
        const float s = 2.f * 3.14159265359f * A.inv_n;
        A.part[((size_t)d * A.nx + kx) * 2] = s * v[0];
        A.part[((size_t)d * A.nx + kx) * 2 + 1] = s * v[1];
    }
}
// one wave per output (d, q): the per-line sums in a fixed order
__global__ __launch_bounds__(64) void holo_shift_sum_kernel(ShiftArgs A) {
    const int o = blockIdx.x, d = o >> 1, q = o & 1, lane = threadIdx.x;
    double acc = 0.0;
    for (int kx = lane; kx < A.nx; kx += 64) acc += (double)A.part[((size_t)d * A.nx + kx) * 2 + q];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) A.grad_shifts[o] += (float)acc;
}

// lines (d, kx): D^ Phi -> IFFT_y -> W[d][y][kx]
template <int NY> __global__ __launch_bounds__(256) void holo_sli(ShiftArgs A) {
    using LG = LineGeo<NY>;
    __shared__ cf buf[2][LG::LPB * LG::LP];
    __shared__ float2 twl[TwLds<NY>::SIZE];
    const float2* tw = stage_twiddles<NY>(twl, A.tw_y);
    const int ll = threadIdx.x / LG::TPR, t = threadIdx.x % LG::TPR;
    const int kx0 = blockIdx.x * LG::LPB, kx = kx0 + ll, d = blockIdx.y;
    const bool ok = kx < A.nx;
    cf* a = buf[0] + ll * LG::LP;
    cf* b = buf[1] + ll * LG::LP;
    const size_t line = ((size_t)d * A.nx + (ok ? kx : 0)) * NY;
    const float sy = A.shifts[2 * d], sx = A.shifts[2 * d + 1];
    const float fx = fft_freq(kx, A.nx);
    for (int k = t; k < NY; k += LG::TPR) {
        float sn, cs;
        sincosf(-2.f * 3.14159265359f * (fx * sx + fft_freq(k, NY) * sy), &sn, &cs);
        a[k] = ok ? cmul(A.spec[line + k], make_float2(cs, sn)) : make_float2(0.f, 0.f);
    }
    __syncthreads();
    const cf* res = line_fft<NY, true>(a, b, tw, t);
    float2* W = A.W + (size_t)d * NY * A.nx;
    const int nx = A.nx;
    store_lines_transposed<NY>(res - ll * LG::LP, min(LG::LPB, nx - kx0), [=](int y, int c) { return W + HOLO_TIDX(y, nx, kx0 + c, NY); });
}
// rows (d, y): IFFT_x / n, real part -> out[d][y][x]
template <int NX> __global__ __launch_bounds__(256) void holo_sri(ShiftArgs A) {
    using LG = LineGeo<NX>;
    __shared__ cf buf[2][LG::LPB * LG::LP];
    __shared__ float2 twl[TwLds<NX>::SIZE];
    const float2* tw = stage_twiddles<NX>(twl, A.tw_x);
    const int ll = threadIdx.x / LG::TPR, t = threadIdx.x % LG::TPR;
    const int y = blockIdx.x * LG::LPB + ll, d = blockIdx.y;
    const bool ok = y < A.ny;
    cf* a = buf[0] + ll * LG::LP;
    cf* b = buf[1] + ll * LG::LP;
    const size_t row = ((size_t)d * A.ny + (ok ? y : 0)) * NX;
    for (int k = t; k < NX; k += LG::TPR) a[k] = ok ? A.W[row + k] : make_float2(0.f, 0.f);
    __syncthreads();
    const cf* res = line_fft<NX, true>(a, b, tw, t);
    if (ok)
        for (int x = t; x < NX; x += LG::TPR) A.out[row + x] = res[x].x * A.inv_n;
}

template <int N> static int blocks_for(int lines) { return (lines + LineGeo<N>::LPB - 1) / LineGeo<N>::LPB; }

static hipError_t holo_run(const HoloArgs& A, bool want_grad, hipStream_t st, bool fuse = false) {
#define K_X(KERNEL, N_, LINES) case N_: hipLaunchKernelGGL((KERNEL<N_>), dim3(blocks_for<N_>(LINES)), dim3(256), 0, st, A); break;
#define K1(N_) K_X(holo_k1, N_, A.ny)
#define K2(N_) case N_: hipLaunchKernelGGL((holo_k2<N_>), dim3(blocks_for<N_>(A.nx), A.nd), dim3(256), 0, st, A); break;
#define K4(N_) case N_: hipLaunchKernelGGL((holo_k4<N_>), dim3(A.nx), dim3(256), 0, st, A); break;
#define K5(N_) case N_: if (fuse) hipLaunchKernelGGL((holo_k5<N_, true>), dim3(blocks_for<N_>(A.ny)), dim3(256), 0, st, A); \
                         else hipLaunchKernelGGL((holo_k5<N_, false>), dim3(blocks_for<N_>(A.ny)), dim3(256), 0, st, A); break;
#define K3G(N_) case N_: hipLaunchKernelGGL((holo_k3<N_, true>), dim3(blocks_for<N_>(A.nd * A.ny)), dim3(256), 0, st, A); break;
#define K3N(N_) case N_: hipLaunchKernelGGL((holo_k3<N_, false>), dim3(blocks_for<N_>(A.nd * A.ny)), dim3(256), 0, st, A); break;
    switch (A.nx) { ADM_HOLO_SIZES(K1) default: return hipErrorInvalidValue; }
    switch (A.ny) { ADM_HOLO_SIZES(K2) default: return hipErrorInvalidValue; }
    if (want_grad) {
        switch (A.nx) { ADM_HOLO_SIZES(K3G) default: return hipErrorInvalidValue; }
        switch (A.ny) { ADM_HOLO_SIZES(K4) default: return hipErrorInvalidValue; }
        switch (A.nx) { ADM_HOLO_SIZES(K5) default: return hipErrorInvalidValue; }
    } else {
        switch (A.nx) { ADM_HOLO_SIZES(K3N) default: return hipErrorInvalidValue; }
        hipLaunchKernelGGL(holo_sums_kernel, dim3(A.nd * 8), dim3(64), 0, st, A);
    }
#undef K_X
#undef K1
#undef K2
#undef K4
#undef K5
#undef K3G
#undef K3N
    return hipGetLastError();
}

}  // namespace adm

using namespace adm;

static int upload_twiddles(adm_ctx* ctx, int N, float2** out) {
    std::vector<float2> t(N);
    for (int j = 0; j < N; ++j) {
        const double a = -2.0 * M_PI * (double)j / (double)N;
        t[j] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    int rc = adm_malloc(ctx, N * sizeof(float2), (void**)out);
    if (rc) return rc;
    return adm_h2d(ctx, *out, t.data(), N * sizeof(float2));
}

static bool pow2_in_range(int n) { return n >= 16 && n <= 2048 && (n & (n - 1)) == 0; }

extern "C" int adm_holo_create(adm_ctx* ctx, const adm_holo_desc* desc, adm_holo** out) {
    if (!ctx || !desc || !out) return fail(ADM_ERR_INVALID, "adm_holo_create: null argument");
    const adm_holo_desc& d = *desc;
    if (!pow2_in_range(d.ny) || !pow2_in_range(d.nx))
        return fail(ADM_ERR_UNSUPPORTED, "adm_holo_create: field sizes must be powers of two in [16, 2048]");
    if (d.n_dists < 1 || d.n_dists > 64) return fail(ADM_ERR_INVALID, "adm_holo_create: n_dists must be in [1, 64]");
    if (d.sign_convention != 1 && d.sign_convention != -1) return fail(ADM_ERR_INVALID, "adm_holo_create: sign_convention must be +-1");
    adm_holo* h = new adm_holo();
    std::memset(h, 0, sizeof(*h));
    h->ctx = ctx;
    h->d = d;
    const size_t n = (size_t)d.ny * d.nx;
    int rc = upload_twiddles(ctx, d.ny, &h->tw_y);
    if (!rc) rc = upload_twiddles(ctx, d.nx, &h->tw_x);
    if (!rc) {
        // gen_freq_mesh (adorym/propagate.py:54-60): u along y, v along x; cast to fp32 before squaring like the tensors
        std::vector<float> uv(n);
        for (int y = 0; y < d.ny; ++y) {
            const int ky = y < (d.ny + 1) / 2 ? y : y - d.ny;
            const float u = (float)(((double)ky / d.ny) / d.voxel_nm_y);
            for (int x = 0; x < d.nx; ++x) {
                const int kx = x < (d.nx + 1) / 2 ? x : x - d.nx;
                const float v = (float)(((double)kx / d.nx) / d.voxel_nm_x);
                uv[(size_t)x * d.ny + y] = u * u + v * v;
            }
        }
        rc = adm_malloc(ctx, n * sizeof(float), (void**)&h->uv2t);
        if (!rc) rc = adm_h2d(ctx, h->uv2t, uv.data(), n * sizeof(float));
    }
    if (!rc) rc = adm_malloc(ctx, n * sizeof(float2), (void**)&h->T14);
    if (!rc) rc = adm_malloc(ctx, n * sizeof(float2), (void**)&h->Ft);
    if (!rc) rc = adm_malloc(ctx, n * d.n_dists * sizeof(float2), (void**)&h->Wq);
    if (!rc) rc = adm_malloc(ctx, n * d.n_dists * sizeof(float2), (void**)&h->T3);
    if (!rc) rc = adm_malloc(ctx, (size_t)d.n_dists * d.ny * 8 * sizeof(float), (void**)&h->part3);
    if (!rc) rc = adm_malloc(ctx, (size_t)d.n_dists * d.nx * sizeof(float), (void**)&h->part4);
    // (the affine-gradient slots of part3 are summed only when they were written: zero once for the others)
    if (!rc) rc = adm_memset(ctx, h->part3, 0, (size_t)d.n_dists * d.ny * 8 * sizeof(float));
    if (rc) {
        adm_holo_destroy(h);
        return rc;
    }
    *out = h;
    return ADM_OK;
}

extern "C" int adm_holo_destroy(adm_holo* h) {
    if (!h) return ADM_OK;
    void* bufs[] = {h->tw_y, h->tw_x, h->uv2t, h->T14, h->Ft, h->Wq, h->T3, h->part3, h->part4, h->part_s};
    for (void* b : bufs)
        if (b) adm_free(h->ctx, b);
    delete h;
    return ADM_OK;
}

extern "C" int adm_holo_fwd_adj(adm_holo* h, const float* obj, const float* probe, const float* dists_cm, const float* affine,
                                const float* data, int want_grad, float* grad_obj, float* grad_probe, float* grad_dists,
                                float* grad_affine, float* pred, float* loss_sum) {
    if (!h || !obj || !probe || !dists_cm || !data || !loss_sum) return fail(ADM_ERR_INVALID, "adm_holo_fwd_adj: null argument");
    if (want_grad && !grad_obj) return fail(ADM_ERR_INVALID, "adm_holo_fwd_adj: want_grad needs grad_obj");
    const adm_holo_desc& d = h->d;
    const size_t n = (size_t)d.ny * d.nx;
    HoloArgs A;
    std::memset(&A, 0, sizeof(A));
    A.obj = (const float2*)obj; A.probe = (const float2*)probe;
    A.dists = dists_cm; A.affine = affine; A.data = data; A.uv2t = h->uv2t;
    A.T1 = h->T14; A.Ft = h->Ft; A.Wq = h->Wq; A.T3 = h->T3; A.T4 = h->T14;
    A.tw_x = h->tw_x; A.tw_y = h->tw_y;
    A.pred = pred; A.part3 = h->part3; A.part4 = h->part4;
    A.cot = want_grad ? h->cot : nullptr; A.direct = h->direct;
    if (h->direct && affine) return fail(ADM_ERR_INVALID, "adm_holo_fwd_adj: data registered elsewhere (adm_holo_set_registration, direct) excludes affine matrices");
    A.grad_obj = (float2*)grad_obj; A.grad_probe = (float2*)grad_probe;
    A.loss_sum = loss_sum; A.grad_affine = grad_affine; A.grad_dists = grad_dists;
    A.ny = d.ny; A.nx = d.nx; A.nd = d.n_dists; A.real_imag = d.unknown_type; A.intensity = d.raw_intensity;
    A.want_affine = (want_grad && grad_affine) ? 1 : 0;
    A.want_dists = (want_grad && grad_dists) ? 1 : 0;
    A.set_obj = (want_grad == 2) ? 1 : 0;
    A.k1 = d.k1; A.sigma = (float)d.sign_convention;
    A.c1 = (float)(-(double)d.sign_convention * 3.14159265359 * d.lambda_nm);
    A.inv_n = (float)(1.0 / (double)n);
    A.gscale = want_grad ? (float)(2.0 / ((double)n * d.n_dists)) : 0.f;
    ADM_HIP(holo_run(A, want_grad != 0, h->ctx->stream));
    return ADM_OK;
}

extern "C" int adm_holo_fwd_adj_adam(adm_holo* h, float* obj, const float* probe, float* dists_cm, float* affine, const float* data,
                                     const adm_holo_adam* opt, float* pred, float* loss_sum) {
    if (!h || !obj || !probe || !dists_cm || !data || !loss_sum || !opt) return fail(ADM_ERR_INVALID, "adm_holo_fwd_adj_adam: null argument");
    if (!opt->m_obj || !opt->v_obj) return fail(ADM_ERR_INVALID, "adm_holo_fwd_adj_adam: the object's moments are required");
    if ((opt->m_dists == nullptr) != (opt->v_dists == nullptr) || (opt->m_affine == nullptr) != (opt->v_affine == nullptr))
        return fail(ADM_ERR_INVALID, "adm_holo_fwd_adj_adam: m and v of a parameter come together");
    if (opt->m_affine && !affine) return fail(ADM_ERR_INVALID, "adm_holo_fwd_adj_adam: affine moments without affine matrices");
    const adm_holo_desc& d = h->d;
    const size_t n = (size_t)d.ny * d.nx;
    HoloArgs A;
    std::memset(&A, 0, sizeof(A));
    A.obj = (const float2*)obj; A.probe = (const float2*)probe;
    A.dists = dists_cm; A.affine = affine; A.data = data; A.uv2t = h->uv2t;
    A.T1 = h->T14; A.Ft = h->Ft; A.Wq = h->Wq; A.T3 = h->T3; A.T4 = h->T14;
    A.tw_x = h->tw_x; A.tw_y = h->tw_y;
    A.pred = pred; A.part3 = h->part3; A.part4 = h->part4;
    if (h->cot || h->direct) return fail(ADM_ERR_INVALID, "adm_holo_fwd_adj_adam: not with adm_holo_set_registration (the shift refinement takes the general path)");
    A.loss_sum = loss_sum;
    A.ny = d.ny; A.nx = d.nx; A.nd = d.n_dists; A.real_imag = d.unknown_type; A.intensity = d.raw_intensity;
    A.want_affine = opt->m_affine ? 1 : 0;
    A.want_dists = opt->m_dists ? 1 : 0;
    A.set_obj = 1;
    A.k1 = d.k1; A.sigma = (float)d.sign_convention;
    A.c1 = (float)(-(double)d.sign_convention * 3.14159265359 * d.lambda_nm);
    A.inv_n = (float)(1.0 / (double)n);
    A.gscale = (float)(2.0 / ((double)n * d.n_dists));
    A.obj_rw = (float2*)obj; A.dists_rw = dists_cm; A.affine_rw = affine;
    A.m_obj = opt->m_obj; A.v_obj = opt->v_obj; A.m_d = opt->m_dists; A.v_d = opt->v_dists; A.m_a = opt->m_affine; A.v_a = opt->v_affine;
    A.a_pin = opt->affine_pin; A.a_pin_n = opt->affine_pin_n;
    A.adam = adam_scalars(opt->i_batch, opt->step_obj, opt->b1, opt->b2, opt->eps, 0, nullptr);
    A.step_d = (float)opt->step_dists; A.step_a = (float)opt->step_affine;
    ADM_HIP(holo_run(A, true, h->ctx->stream, true));
    return ADM_OK;
}

// ---- per-distance shift refinement (kernels holo_sr / holo_slf / holo_sli / holo_sri above) ----
static void shift_args(const adm_holo* h, ShiftArgs& S) {
    std::memset(&S, 0, sizeof(S));
    S.tw_x = h->tw_x; S.tw_y = h->tw_y;
    S.ny = h->d.ny; S.nx = h->d.nx; S.nd = h->d.n_dists;
    S.inv_n = (float)(1.0 / ((double)h->d.ny * h->d.nx));
}
static hipError_t launch_sr(const ShiftArgs& S, hipStream_t st) {
#define X(N_) case N_: hipLaunchKernelGGL((holo_sr<N_>), dim3(blocks_for<N_>(S.ny), S.nd), dim3(256), 0, st, S); break;
    switch (S.nx) { ADM_HOLO_SIZES(X) default: return hipErrorInvalidValue; }
#undef X
    return hipGetLastError();
}
template <int MODE> static hipError_t launch_slf(const ShiftArgs& S, hipStream_t st) {
#define X(N_) case N_: hipLaunchKernelGGL((holo_slf<N_, MODE>), dim3(blocks_for<N_>(S.nx), S.nd), dim3(256), 0, st, S); break;
    switch (S.ny) { ADM_HOLO_SIZES(X) default: return hipErrorInvalidValue; }
#undef X
    return hipGetLastError();
}
static hipError_t launch_sli(const ShiftArgs& S, hipStream_t st) {
#define X(N_) case N_: hipLaunchKernelGGL((holo_sli<N_>), dim3(blocks_for<N_>(S.nx), S.nd), dim3(256), 0, st, S); break;
    switch (S.ny) { ADM_HOLO_SIZES(X) default: return hipErrorInvalidValue; }
#undef X
    return hipGetLastError();
}
static hipError_t launch_sri(const ShiftArgs& S, hipStream_t st) {
#define X(N_) case N_: hipLaunchKernelGGL((holo_sri<N_>), dim3(blocks_for<N_>(S.ny), S.nd), dim3(256), 0, st, S); break;
    switch (S.nx) { ADM_HOLO_SIZES(X) default: return hipErrorInvalidValue; }
#undef X
    return hipGetLastError();
}

extern "C" int adm_holo_set_registration(adm_holo* h, float* cot_out, int direct) {
    if (!h) return fail(ADM_ERR_INVALID, "adm_holo_set_registration: null handle");
    h->cot = cot_out;
    h->direct = direct ? 1 : 0;
    return ADM_OK;
}

extern "C" int adm_holo_data_spectrum(adm_holo* h, const float* data, float* spectrum) {
    if (!h || !data || !spectrum) return fail(ADM_ERR_INVALID, "adm_holo_data_spectrum: null argument");
    ShiftArgs S;
    shift_args(h, S);
    S.img = data; S.absval = 1; S.T = h->T3; S.spec_out = (float2*)spectrum;
    ADM_HIP(launch_sr(S, h->ctx->stream));
    ADM_HIP(launch_slf<0>(S, h->ctx->stream));
    return ADM_OK;
}

extern "C" int adm_holo_shift_targets(adm_holo* h, const float* spectrum, const float* shifts, float* targets) {
    if (!h || !spectrum || !shifts || !targets) return fail(ADM_ERR_INVALID, "adm_holo_shift_targets: null argument");
    ShiftArgs S;
    shift_args(h, S);
    S.spec = (const float2*)spectrum; S.shifts = shifts; S.W = h->Wq; S.out = targets;
    ADM_HIP(launch_sli(S, h->ctx->stream));
    ADM_HIP(launch_sri(S, h->ctx->stream));
    return ADM_OK;
}

extern "C" int adm_holo_shift_grad(adm_holo* h, const float* cot, const float* spectrum, const float* shifts, float* grad_shifts) {
    if (!h || !cot || !spectrum || !shifts || !grad_shifts) return fail(ADM_ERR_INVALID, "adm_holo_shift_grad: null argument");
    if (!h->part_s) {
        int rc = adm_malloc(h->ctx, (size_t)h->d.n_dists * h->d.nx * 2 * sizeof(float), (void**)&h->part_s);
        if (rc) return rc;
    }
    ShiftArgs S;
    shift_args(h, S);
    S.img = cot; S.absval = 0; S.T = h->T3; S.spec = (const float2*)spectrum; S.shifts = shifts; S.part = h->part_s;
    S.grad_shifts = grad_shifts;
    ADM_HIP(launch_sr(S, h->ctx->stream));
    ADM_HIP(launch_slf<1>(S, h->ctx->stream));
    hipLaunchKernelGGL(holo_shift_sum_kernel, dim3(S.nd * 2), dim3(64), 0, h->ctx->stream, S);
    ADM_HIP(hipGetLastError());
    return ADM_OK;
}
