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
// The fields are far larger than the LDS-resident tiles of the multislice kernel (512^2 * 8 B = 2 MB), so the
// 2-D transforms run as two passes of a batched row FFT (Stockham radix-8/4/2 in LDS, rows of N = 16 ... 2048) whose
// store is transposed: rows -> [kx][y], then "rows" of that -> [ky][kx].  All traffic stays in L2 / Infinity Cache
// at these sizes; the path is launch-latency bound, not bandwidth bound.
#include <vector>
#include <cmath>
#include <cstring>
#include "adm_common.h"
#include "adm_fft.h"

struct adm_holo {
    adm_ctx* ctx;
    adm_holo_desc d;
    float2 *tw_y, *tw_x;      // exp(-2 pi i j / N) for N = ny, nx
    float* uv2;               // [ny][nx] u^2 + v^2 (nm^-2), fp32 like the reference's tensors
    float2 *psi, *F, *W, *T;  // psi [ny][nx]; F = FFT2(psi); W, T: [n_dists][ny][nx] work fields
    float* partial;           // [n_dists][256 blocks][8] per-block partial sums
};

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

template <int N, int NS, bool INV, int TPR> struct Passes {
    static __device__ __forceinline__ int run(cf* a, cf* b, const float2* tw, int t) {
        constexpr int REM = N / NS;
        constexpr int R = REM >= 8 ? 8 : REM;
        stockham_pass<N, NS, R, INV, TPR>(a, b, tw, t);
        __syncthreads();
        return 1 + Passes<N, NS * R, INV, TPR>::run(b, a, tw, t);
    }
};
template <int N, bool INV, int TPR> struct Passes<N, N, INV, TPR> {
    static __device__ __forceinline__ int run(cf*, cf*, const float2*, int) { return 0; }
};

// in [nrows_total][N] -> out [b][N][rows] (b = row / rows), values multiplied by `scale`
template <int N, bool INV>
__global__ __launch_bounds__(256) void fft_rows_t_kernel(const float2* __restrict__ in, float2* __restrict__ out,
                                                         const float2* __restrict__ tw, int rows, int nrows_total, float scale) {
    constexpr int TPR = N / 8;
    constexpr int RPW = 256 / TPR;
    __shared__ cf buf[2][RPW * N];
    const int rl = threadIdx.x / TPR, t = threadIdx.x % TPR;
    const int gr = blockIdx.x * RPW + rl;
    const bool ok = gr < nrows_total;
    cf* a = buf[0] + rl * N;
    cf* b = buf[1] + rl * N;
    for (int n = t; n < N; n += TPR) a[n] = ok ? in[(size_t)gr * N + n] : make_float2(0.f, 0.f);
    __syncthreads();
    const int np = Passes<N, 1, INV, TPR>::run(a, b, tw, t);
    const cf* res = (np & 1) ? b : a;
    if (ok) {
        const int bi = gr / rows, r = gr % rows;
        for (int k = t; k < N; k += TPR) out[((size_t)bi * N + k) * rows + r] = cscale(res[k], scale);
    }
}

template <bool INV> static hipError_t fft_rows_t(int n, const float2* in, float2* out, const float2* tw, int rows, int nb, float scale,
                                                 hipStream_t st) {
    const int total = rows * nb;
#define ADM_CASE(N_)                                                                                                        \
    case N_: {                                                                                                              \
        constexpr int RPW = 256 / (N_ / 8);                                                                                 \
        hipLaunchKernelGGL((fft_rows_t_kernel<N_, INV>), dim3((total + RPW - 1) / RPW), dim3(256), 0, st, in, out, tw, rows, total, \
                           scale);                                                                                          \
        break;                                                                                                              \
    }
    switch (n) {
        ADM_CASE(16) ADM_CASE(32) ADM_CASE(64) ADM_CASE(128) ADM_CASE(256) ADM_CASE(512) ADM_CASE(1024) ADM_CASE(2048)
        default: return hipErrorInvalidValue;
    }
#undef ADM_CASE
    return hipGetLastError();
}

// [nb][ny][nx] -> [nb][ny][nx] spectrum (natural order); `tmp` is scratch of the same size
template <bool INV> static hipError_t fft2(const adm_holo* h, const float2* in, float2* out, float2* tmp, int nb, float scale) {
    hipStream_t st = h->ctx->stream;
    hipError_t e = fft_rows_t<INV>(h->d.nx, in, tmp, h->tw_x, h->d.ny, nb, 1.0f, st);
    if (e != hipSuccess) return e;
    return fft_rows_t<INV>(h->d.ny, tmp, out, h->tw_y, h->d.nx, nb, scale, st);
}

// ---------------------------------------------------------------------------------------------------------------
// elementwise stages
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ cf holo_transmission(float2 o, int real_imag, float k1, float sigma) {
    if (real_imag) return o;
    const float e = expf(-k1 * o.y);
    float sn, cs;
    sincosf(-sigma * k1 * o.x, &sn, &cs);
    return make_float2(e * cs, e * sn);
}

__global__ __launch_bounds__(256) void holo_modulate_kernel(const float2* __restrict__ obj, const float2* __restrict__ probe,
                                                            float2* __restrict__ psi, size_t n, int real_imag, float k1, float sigma) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        psi[i] = cmul(probe[i], holo_transmission(obj[i], real_imag, k1, sigma));
}

__device__ __forceinline__ cf holo_h(float uv2, float dist_cm, float c1) {
    // -sigma*PI*lambda (c1, rounded once on the host) * dist_nm * (u^2+v^2), every product in fp32 like the reference
    const float arg = (c1 * (dist_cm * 1e7f)) * uv2;
    float sn, cs;
    sincosf(arg, &sn, &cs);
    return make_float2(cs, sn);
}

// W[d] = F * H_d
__global__ __launch_bounds__(256) void holo_apply_h_kernel(const float2* __restrict__ F, const float* __restrict__ uv2,
                                                           const float* __restrict__ dists, float2* __restrict__ W, size_t n, int nd,
                                                           float c1) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n * nd; i += (size_t)gridDim.x * blockDim.x) {
        const int d = (int)(i / n);
        const size_t k = i % n;
        W[i] = cmul(F[k], holo_h(uv2[k], dists[d], c1));
    }
}

// torch's affine_grid base coordinate (see oracle/adorym_oracle.py::affine_sample for the derivation)
__device__ __forceinline__ float base_coord(int i, int n) {
    const float step = 2.0f / (float)(n - 1);
    const float lin = (i < n / 2) ? fmaf(step, (float)i, -1.0f) : fmaf(-step, (float)(n - 1 - i), 1.0f);
    return (lin * (float)(n - 1)) / (float)n;
}

__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// per (distance, pixel): registered target, loss term, dL/dPsi (in place of Psi), affine-matrix gradient partial sums.
// grid = (blocks, n_dists): one distance per blockIdx.y so that the reductions are per distance.
__global__ __launch_bounds__(256) void holo_loss_kernel(float2* __restrict__ Psi, const float* __restrict__ data,
                                                        const float* __restrict__ affine, int ny, int nx, int intensity, float gscale,
                                                        float* __restrict__ pred_out, float* __restrict__ partial,
                                                        int want_affine) {
    __shared__ float red[4];
    const int d = blockIdx.y;
    const size_t n = (size_t)ny * nx;
    float th[6] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f};
    if (affine) {
#pragma unroll
        for (int q = 0; q < 6; ++q) th[q] = affine[d * 6 + q];
    }
    const float* img = data + (size_t)d * n;
    float lsum = 0.f, ga[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / nx), x = (int)(i % nx);
        const float X = base_coord(x, nx), Y = base_coord(y, ny);
        const float gx = th[0] * X + th[1] * Y + th[2];
        const float gy = th[3] * X + th[4] * Y + th[5];
        float ix = ((gx + 1.f) * (float)nx - 1.f) * 0.5f;
        float iy = ((gy + 1.f) * (float)ny - 1.f) * 0.5f;
        const float mx = (ix > 0.f && ix < (float)(nx - 1)) ? 1.f : 0.f;     // grid_sampler's clip_coordinates_set_grad
        const float my = (iy > 0.f && iy < (float)(ny - 1)) ? 1.f : 0.f;
        ix = fminf(fmaxf(ix, 0.f), (float)(nx - 1));
        iy = fminf(fmaxf(iy, 0.f), (float)(ny - 1));
        const int x0 = (int)floorf(ix), y0 = (int)floorf(iy);
        const float wx = ix - (float)x0, wy = iy - (float)y0;
        const int x1 = min(x0 + 1, nx - 1), y1 = min(y0 + 1, ny - 1);
        const float v00 = img[(size_t)y0 * nx + x0], v01 = img[(size_t)y0 * nx + x1];
        const float v10 = img[(size_t)y1 * nx + x0], v11 = img[(size_t)y1 * nx + x1];
        const float samp = v00 * (1.f - wx) * (1.f - wy) + v01 * wx * (1.f - wy) + v10 * (1.f - wx) * wy + v11 * wx * wy;
        const float as = fabsf(samp);
        const float tgt = intensity ? sqrtf(as) : as;
        const float2 ps = Psi[(size_t)d * n + i];
        const float mag = sqrtf(ps.x * ps.x + ps.y * ps.y);
        const float diff = mag - tgt;
        lsum += diff * diff;
        if (pred_out) pred_out[(size_t)d * n + i] = mag;
        const float g = (mag > 0.f) ? gscale * diff / mag : 0.f;
        Psi[(size_t)d * n + i] = cscale(ps, g);
        if (want_affine) {
            const float sg = (float)((samp > 0.f) - (samp < 0.f));
            float cot = -gscale * diff * (intensity ? sg / (2.f * sqrtf(as)) : sg);
            if (!(fabsf(cot) <= 3.0e38f)) cot = 0.f;               // 0/0 at an exactly zero sample
            const float dix = ((v01 - v00) * (1.f - wy) + (v11 - v10) * wy) * mx * (0.5f * (float)nx) * cot;
            const float diy = ((v10 - v00) * (1.f - wx) + (v11 - v01) * wx) * my * (0.5f * (float)ny) * cot;
            ga[0] += dix * X; ga[1] += dix * Y; ga[2] += dix;
            ga[3] += diy * X; ga[4] += diy * Y; ga[5] += diy;
        }
    }
    // per-block partial sums (slot 0: loss, 1..6: affine gradient); summed by holo_reduce_kernel -- hundreds of float
    // atomics on one address serialise in L2 and made this kernel 10x slower than its arithmetic
    float* out = partial + ((size_t)d * gridDim.x + blockIdx.x) * 8;
    const float ls = block_sum(lsum, red);
    if (threadIdx.x == 0) out[0] = ls;
    if (want_affine) {
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const float s = block_sum(ga[q], red);
            if (threadIdx.x == 0) out[1 + q] = s;
        }
    }
}

// dst[d*stride + q] (=|+=) scale * sum_b partial[(d*nb + b)*8 + off + q]   for q < nq;  one wave per distance
__global__ __launch_bounds__(64) void holo_reduce_kernel(const float* __restrict__ partial, int nb, int off, int nq, float scale,
                                                         float* __restrict__ dst, int stride, int accumulate) {
    const int d = blockIdx.x;
    for (int q = 0; q < nq; ++q) {
        float acc = 0.f;
        for (int b = threadIdx.x; b < nb; b += 64) acc += partial[((size_t)d * nb + b) * 8 + off + q];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
        if (threadIdx.x == 0) {
            if (accumulate) dst[d * stride + q] += scale * acc;
            else dst[d * stride + q] = scale * acc;
        }
    }
}

// Gh[d] = FFT2(dL/dPsi_d) / (ny nx) (already scaled).  GF = sum_d conj(H_d) Gh_d;
// dL/dd_cm = 1e7 * Re sum_k conj(Gh_d) (-i sigma PI lambda uv2) H_d F     (one distance per blockIdx.y)
__global__ __launch_bounds__(256) void holo_adjoint_kernel(const float2* __restrict__ Gh, const float2* __restrict__ F,
                                                           const float* __restrict__ uv2, const float* __restrict__ dists, size_t n, int nd,
                                                           float c1, float* __restrict__ partial) {
    __shared__ float red[4];
    const int d = blockIdx.y;
    float acc = 0.f;
    const float dist = dists[d];
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) {
        const float u2 = uv2[k];
        const cf h = holo_h(u2, dist, c1);
        const cf g = Gh[(size_t)d * n + k];
        const cf hf = cmul(h, F[k]);
        // Re( conj(g) * (i c1 u2) * hf ) = -c1 u2 * Im(conj(g) hf)
        acc += -c1 * u2 * (g.x * hf.y - g.y * hf.x);
    }
    const float s = block_sum(acc, red);
    if (threadIdx.x == 0) partial[((size_t)d * gridDim.x + blockIdx.x) * 8] = s;
}

__global__ __launch_bounds__(256) void holo_sum_conj_h_kernel(const float2* __restrict__ Gh, const float* __restrict__ uv2,
                                                              const float* __restrict__ dists, float2* __restrict__ GF, size_t n, int nd,
                                                              float c1) {
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) {
        cf acc = make_float2(0.f, 0.f);
        const float u2 = uv2[k];
        for (int d = 0; d < nd; ++d) acc = cadd(acc, cmulc(Gh[(size_t)d * n + k], holo_h(u2, dists[d], c1)));
        GF[k] = acc;
    }
}

// dL/dpsi -> dL/dobj (accumulated) and dL/dprobe (written)
__global__ __launch_bounds__(256) void holo_obj_grad_kernel(const float2* __restrict__ gpsi, const float2* __restrict__ obj,
                                                            const float2* __restrict__ probe, float2* __restrict__ grad_obj,
                                                            float2* __restrict__ grad_probe, size_t n, int real_imag, float k1,
                                                            float sigma) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const cf g = gpsi[i];
        const cf c = holo_transmission(obj[i], real_imag, k1, sigma);
        const cf p = probe[i];
        if (grad_probe) grad_probe[i] = cmulc(g, c);
        float2 go = grad_obj[i];
        if (real_imag) {
            const cf z = cmulc(g, p);                  // G conj(probe)
            go.x += z.x;
            go.y += z.y;
        } else {
            const cf pm = cmul(p, c);                  // post-modulation field psi'
            const float wre = g.x * pm.x + g.y * pm.y, wim = g.x * pm.y - g.y * pm.x;     // w = conj(G) psi'
            go.y += -k1 * wre;                         // d/dbeta
            go.x += sigma * k1 * wim;                  // d/ddelta
        }
        grad_obj[i] = go;
    }
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
                uv[(size_t)y * d.nx + x] = u * u + v * v;
            }
        }
        rc = adm_malloc(ctx, n * sizeof(float), (void**)&h->uv2);
        if (!rc) rc = adm_h2d(ctx, h->uv2, uv.data(), n * sizeof(float));
    }
    if (!rc) rc = adm_malloc(ctx, n * sizeof(float2), (void**)&h->psi);
    if (!rc) rc = adm_malloc(ctx, n * sizeof(float2), (void**)&h->F);
    if (!rc) rc = adm_malloc(ctx, n * d.n_dists * sizeof(float2), (void**)&h->W);
    if (!rc) rc = adm_malloc(ctx, n * d.n_dists * sizeof(float2), (void**)&h->T);
    if (!rc) rc = adm_malloc(ctx, (size_t)d.n_dists * 256 * 8 * sizeof(float), (void**)&h->partial);
    if (rc) {
        adm_holo_destroy(h);
        return rc;
    }
    *out = h;
    return ADM_OK;
}

extern "C" int adm_holo_destroy(adm_holo* h) {
    if (!h) return ADM_OK;
    void* bufs[] = {h->tw_y, h->tw_x, h->uv2, h->psi, h->F, h->W, h->T, h->partial};
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
    hipStream_t st = h->ctx->stream;
    const size_t n = (size_t)d.ny * d.nx;
    const int nd = d.n_dists;
    const float sigma = (float)d.sign_convention;
    const float c1 = (float)(-(double)d.sign_convention * 3.14159265359 * d.lambda_nm);
    const float inv_n = (float)(1.0 / (double)n);
    hipLaunchKernelGGL(holo_modulate_kernel, dim3(grid_for(n)), dim3(256), 0, st, (const float2*)obj, (const float2*)probe, h->psi, n,
                       d.unknown_type, d.k1, sigma);
    ADM_HIP(fft2<false>(h, h->psi, h->F, h->T, 1, 1.0f));
    hipLaunchKernelGGL(holo_apply_h_kernel, dim3(grid_for(n * nd)), dim3(256), 0, st, (const float2*)h->F, (const float*)h->uv2, dists_cm,
                       h->W, n, nd, c1);
    ADM_HIP(fft2<true>(h, h->W, h->W, h->T, nd, inv_n));            // Psi_d (normalised inverse), in place via T
    const float gscale = want_grad ? (float)(2.0 / ((double)n * nd)) : 0.f;
    int nb = grid_for(n);
    if (nb > 256) nb = 256;
    const int want_affine = (want_grad && grad_affine) ? 1 : 0;
    hipLaunchKernelGGL(holo_loss_kernel, dim3(nb, nd), dim3(256), 0, st, h->W, data, affine, d.ny, d.nx, d.raw_intensity, gscale, pred,
                       h->partial, want_affine);
    hipLaunchKernelGGL(holo_reduce_kernel, dim3(nd), dim3(64), 0, st, (const float*)h->partial, nb, 0, 1, 1.0f, loss_sum, 1, 0);
    if (want_affine)
        hipLaunchKernelGGL(holo_reduce_kernel, dim3(nd), dim3(64), 0, st, (const float*)h->partial, nb, 1, 6, 1.0f, grad_affine, 6, 1);
    ADM_HIP(hipGetLastError());
    if (!want_grad) return ADM_OK;
    ADM_HIP(fft2<false>(h, h->W, h->W, h->T, nd, inv_n));           // Gh_d = FFT2(dL/dPsi_d) / N
    if (grad_dists) {
        hipLaunchKernelGGL(holo_adjoint_kernel, dim3(nb, nd), dim3(256), 0, st, (const float2*)h->W, (const float2*)h->F,
                           (const float*)h->uv2, dists_cm, n, nd, c1, h->partial);
        hipLaunchKernelGGL(holo_reduce_kernel, dim3(nd), dim3(64), 0, st, (const float*)h->partial, nb, 0, 1, 1e7f, grad_dists, 1, 1);
    }
    hipLaunchKernelGGL(holo_sum_conj_h_kernel, dim3(grid_for(n)), dim3(256), 0, st, (const float2*)h->W, (const float*)h->uv2, dists_cm,
                       h->psi, n, nd, c1);
    ADM_HIP(fft2<true>(h, h->psi, h->psi, h->T, 1, 1.0f));          // dL/dpsi = unnormalised inverse of GF
    hipLaunchKernelGGL(holo_obj_grad_kernel, dim3(grid_for(n)), dim3(256), 0, st, (const float2*)h->psi, (const float2*)obj,
                       (const float2*)probe, (float2*)grad_obj, (float2*)grad_probe, n, d.unknown_type, d.k1, sigma);
    ADM_HIP(hipGetLastError());
    return ADM_OK;
}
