// Multislice forward + loss + adjoint for ANY probe size Py x Px (non-square, prime factors, 128 x 128 ...): the fallback
// behind the tuned kernels of adm_multislice.hip, which are compiled for a fixed set of square sizes.  Same mathematics
// (SURVEY.md section 3.4), same workspace contract, one workgroup per probe position, the field in LDS.
//
// Replaces (reference paths): adorym/propagate.py:195-280 for whatever `probe_size = prj.shape[-2:]` the data brings
// (adorym/ptychography.py:313-317), adorym/forward_model.py:313-375, :88-103, and the autograd backward.
//
// Transforms: autosort (Stockham) passes with a run-time radix list per axis (8, 4, 2, 9, 3, 5, 7, then any remaining
// prime as one radix).  Every pass is evaluated output-centrically -- output o of a line is the R-term sum
// sum_u in[j + u*N/R] * W_N^(u*(k*N/(Ns*R) + t*N/R)) -- so one code path serves every radix, and a pass is
// "all threads compute their outputs into registers, barrier, write, barrier": in place, natural order out.
// Cost per point and pass = R complex multiply-adds instead of the ~log R of a butterfly: 2-3x the arithmetic of the tuned
// kernels at P = 72, which is the price of generality here, not a design target.
//
// Rows of the workspace (stored fields, tile gradients) are pixel-major [Py][Px]; tile_accumulate follows through
// TileGeom::pixel_major.  Supported: delta_beta and real_imag, any binning, several probe modes, all detector / loss
// variants, beamstop weights, one probe set per position (adm_multislice_fwd_adj_pp).  Not supported here: the Fourier
// shift of the probes (adm_probe_shift: sub-pixel position refinement).
#include <hip/hip_runtime.h>
#include "adm_common.h"
#include "adm_fft.h"
#include "adm_ms_math.h"

namespace adm {

constexpr int GEN_E = 16;         // field elements per thread (Py*Px <= GEN_E * 1024)

struct GenCtx {
    cf* fld;            // LDS [Py*Px]
    const cf* twx;      // LDS W_Px^j
    const cf* twy;      // LDS W_Py^j
    int Py, Px, n, ne, tid, nt;
};

// one Stockham pass of radix R over every line of the field along x (ALONG_Y: along y); v = this thread's outputs
template <bool ALONG_Y, bool INV>
__device__ __forceinline__ void gen_pass(const GenCtx& g, int R, int Ns, cf (&v)[GEN_E]) {
    const int N = ALONG_Y ? g.Py : g.Px;
    const cf* tw = ALONG_Y ? g.twy : g.twx;
    const int m = N / R;
    const int NsR = Ns * R;
    const int tstep = N / NsR;
#pragma unroll
    for (int j = 0; j < GEN_E; ++j) {
        const int i = g.tid + j * g.nt;
        if (j < g.ne && i < g.n) {
            const int y = i / g.Px, x = i - y * g.Px;
            const int o = ALONG_Y ? y : x;
            const int k = o % Ns, t = (o / Ns) % R, bq = o / NsR;
            const int jb = bq * Ns + k;
            const int base = (k * tstep + t * m) % N;
            const cf* src = ALONG_Y ? g.fld + (size_t)jb * g.Px + x : g.fld + (size_t)y * g.Px + jb;
            const int sstride = ALONG_Y ? m * g.Px : m;
            cf acc = make_float2(0.f, 0.f);
            int e = 0;
            for (int u = 0; u < R; ++u) {
                const cf a = src[(size_t)u * sstride];
                const cf w = tw[e];
                acc = INV ? cadd(acc, cmulc(a, w)) : cadd(acc, cmul(a, w));
                e += base;
                if (e >= N) e -= N;
            }
            v[j] = acc;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < GEN_E; ++j) {
        const int i = g.tid + j * g.nt;
        if (j < g.ne && i < g.n) g.fld[i] = v[j];
    }
    __syncthreads();
}

// unnormalised 2-D transform of the LDS field, in place, natural order (INV: conjugate twiddles)
template <bool INV> __device__ __forceinline__ void gen_fft2(const GenCtx& g, const MsParams& p, cf (&v)[GEN_E]) {
    int Ns = 1;
    for (int s = 0; s < p.gen_nrx; ++s) { gen_pass<false, INV>(g, p.gen_rx[s], Ns, v); Ns *= p.gen_rx[s]; }
    Ns = 1;
    for (int s = 0; s < p.gen_nry; ++s) { gen_pass<true, INV>(g, p.gen_ry[s], Ns, v); Ns *= p.gen_ry[s]; }
}

// field <- IFFT2( H * FFT2(field) ) (CONJ: conj(H)); hs = H / (Py*Px), one rounding per element, natural order, global memory
template <bool CONJ> __device__ __forceinline__ void gen_convolve(const GenCtx& g, const MsParams& p, const float2* __restrict__ hs, cf (&v)[GEN_E]) {
    gen_fft2<false>(g, p, v);
#pragma unroll
    for (int j = 0; j < GEN_E; ++j) {
        const int i = g.tid + j * g.nt;
        if (j < g.ne && i < g.n) g.fld[i] = cmul_t<CONJ>(g.fld[i], hs[i]);
    }
    __syncthreads();
    gen_fft2<true>(g, p, v);
}

__device__ __forceinline__ cf gen_modulator(float2 db, float k1, float sigma) {
    const float e = exp_fast(-k1 * db.y);
    float sn, cs;
    sincos_fast(-sigma * k1 * db.x, sn, cs);
    return make_float2(e * cs, e * sn);
}

// (delta, beta) -- or (re, im) -- of modulation step `step` at tile pixel i (sum over the bin's slices)
__device__ __forceinline__ float2 gen_slice(const MsParams& p, const float2* __restrict__ tile, size_t slice_stride, int step, size_t pix_off) {
    const int s_lo = step * p.binning, s_hi = min(s_lo + p.binning, p.Z);
    float2 acc = make_float2(0.f, 0.f);
    for (int s = s_lo; s < s_hi; ++s) {
        const float2 q = tile[(size_t)s * slice_stride + pix_off];
        acc.x += q.x;
        acc.y += q.y;
    }
    return acc;
}

__device__ __forceinline__ float gen_block_sum(float val, float* red, int tid, int nt) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) val += __shfl_down(val, off, 64);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = val;
    __syncthreads();
    float s = 0.f;
    if (tid == 0) for (int w = 0; w < (nt >> 6); ++w) s += red[w];
    return s;          // valid on thread 0
}

__global__ __launch_bounds__(1024) void ms_generic_kernel(MsParams p) {
    extern __shared__ cf gen_lds[];
    __shared__ float red[16];
    GenCtx g;
    g.Py = p.gen_py; g.Px = p.gen_px; g.n = g.Py * g.Px;
    g.tid = threadIdx.x; g.nt = blockDim.x;
    g.ne = (g.n + g.nt - 1) / g.nt;
    g.fld = gen_lds;
    cf* twx = gen_lds + g.n;
    cf* twy = twx + g.Px;
    for (int i = g.tid; i < g.Px; i += g.nt) twx[i] = p.twid[i];
    for (int i = g.tid; i < g.Py; i += g.nt) twy[i] = p.gen_twid_y[i];
    g.twx = twx; g.twy = twy;
    const int b = blockIdx.x;
    const int M = p.n_modes;
    const bool RI = p.real_imag != 0;
    const bool do_grad = p.want_grad != 0;
    const bool far = p.det_mode == ADM_DET_FARFIELD_;
    const int2 ps = p.pos[b];
    const size_t slice_stride = (size_t)p.Yp * p.Xp;
    const float2* tile = p.obj_rot + (size_t)(ps.x + p.pad_y0) * p.Xp + (ps.y + p.pad_x0);
    const size_t row = (size_t)g.n;                                 // elements of one workspace row
    const size_t per = (size_t)p.n_steps * row;
    float2* gtile = p.gtile + (size_t)b * per;
    cf v[GEN_E];
    float inten[GEN_E], gf[GEN_E];
#pragma unroll
    for (int j = 0; j < GEN_E; ++j) inten[j] = 0.f;
    float lsum = 0.f;

    // pixel i of the tile -> offset inside one slice of the rotated object
    auto pix = [&](int i) { const int y = i / g.Px; return (size_t)y * p.Xp + (i - y * g.Px); };

    for (int m = 0; m < M; ++m) {
        float2* stash = p.stash + ((size_t)b * M + m) * per;
        const float2* probe = p.probe + (size_t)b * p.probe_bstride + (size_t)m * row;     // (stride 0: one probe set for all positions)
        for (int i = g.tid; i < g.n; i += g.nt) g.fld[i] = probe[i];
        __syncthreads();
        // ---------------- forward sweep ----------------
        for (int step = 0; step < p.n_steps; ++step) {
            for (int i = g.tid; i < g.n; i += g.nt) {
                const float2 db = gen_slice(p, tile, slice_stride, step, pix(i));
                cf a = g.fld[i];
                if (RI) {
                    if (do_grad) stash[(size_t)step * row + i] = a;      // pre-modulation field
                    a = cmul(a, db);
                } else {
                    a = cmul(a, p.pre_t ? db : gen_modulator(db, p.k1, p.sigma));      // pre_t: cached slice transmissions
                    if (do_grad) stash[(size_t)step * row + i] = a;      // post-modulation field
                }
                g.fld[i] = a;
            }
            __syncthreads();
            if (step < p.n_steps - 1) gen_convolve<false>(g, p, p.gen_hs, v);
        }
        // ---------------- detector plane ----------------
        if (p.det_mode == ADM_DET_FRESNEL_) gen_convolve<false>(g, p, p.gen_hfree_s + (p.n_hfree > 1 ? (size_t)(b % p.n_hfree) * g.n : 0), v);
        else if (far) { if (p.det_inverse) gen_fft2<true>(g, p, v); else gen_fft2<false>(g, p, v); }
        if (M > 1) {
            float2* dq = p.det + ((size_t)b * M + m) * row;
#pragma unroll
            for (int j = 0; j < GEN_E; ++j) {
                const int i = g.tid + j * g.nt;
                if (j < g.ne && i < g.n) {
                    const cf psi = far ? cscale(g.fld[i], p.det_scale) : g.fld[i];
                    inten[j] += psi.x * psi.x + psi.y * psi.y;
                    dq[i] = psi;
                }
            }
            __syncthreads();
        }
    }
    // ---------------- loss, dL/dPsi = g * Psi (one factor for all modes) ----------------
#pragma unroll
    for (int j = 0; j < GEN_E; ++j) {
        const int i = g.tid + j * g.nt;
        gf[j] = 0.f;
        if (j < g.ne && i < g.n) {
            const int ky = i / g.Px, kx = i - ky * g.Px;
            // far field: natural-order spectrum element (ky, kx) is the fftshifted detector pixel ((ky + Py/2) % Py, ...)
            const int my = far ? (ky + g.Py / 2) % g.Py : ky, mx = far ? (kx + g.Px / 2) % g.Px : kx;
            const size_t di = ((size_t)b * g.Py + my) * g.Px + mx;
            const float wq = p.det_weight ? p.det_weight[my * g.Px + mx] : 1.f;
            float mag;
            if (M > 1) mag = sqrtf(inten[j]);
            else {
                const cf psi = far ? cscale(g.fld[i], p.det_scale) : g.fld[i];
                mag = sqrtf(psi.x * psi.x + psi.y * psi.y);
            }
            float gg;
            lsum += wq * (M > 1 ? loss_term_nz(mag, p.target[di], p, gg) : loss_term(mag, p.target[di], p, gg));
            if (p.pred) p.pred[di] = mag;
            gf[j] = wq * gg;
        }
    }
    {
        const float s = gen_block_sum(lsum, red, g.tid, g.nt);
        if (g.tid == 0) p.loss_sum[b] = s;
    }
    if (!do_grad) return;

    // ---------------- adjoint, mode by mode ----------------
    for (int m = 0; m < M; ++m) {
        const float2* stash = p.stash + ((size_t)b * M + m) * per;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < GEN_E; ++j) {
            const int i = g.tid + j * g.nt;
            if (j < g.ne && i < g.n) {
                // Psi = scale * F(psi): the adjoint of (scale * F) applied to g * Psi is scale * F^H (g * scale * F psi)
                const cf psi = (M > 1) ? p.det[((size_t)b * M + m) * row + i] : (far ? cscale(g.fld[i], p.det_scale) : g.fld[i]);
                g.fld[i] = cscale(psi, gf[j] * (far ? p.det_scale : 1.f));
            }
        }
        __syncthreads();
        if (far) { if (p.det_inverse) gen_fft2<false>(g, p, v); else gen_fft2<true>(g, p, v); }
        else if (p.det_mode == ADM_DET_FRESNEL_) gen_convolve<true>(g, p, p.gen_hfree_s + (p.n_hfree > 1 ? (size_t)(b % p.n_hfree) * g.n : 0), v);
        // ---------------- reverse sweep ----------------
        const float sk1 = p.sigma * p.k1;
        for (int step = p.n_steps - 1; step >= 0; --step) {
            for (int i = g.tid; i < g.n; i += g.nt) {
                const cf a = g.fld[i];
                const cf psi = stash[(size_t)step * row + i];
                const float2 db = gen_slice(p, tile, slice_stride, step, pix(i));
                const float zr = a.x * psi.x + a.y * psi.y;
                const float zi = a.x * psi.y - a.y * psi.x;
                float2 gr = make_float2(RI ? zr : sk1 * zi, RI ? -zi : -p.k1 * zr);
                float2* gq = gtile + (size_t)step * row + i;
                if (m > 0) { const float2 o = *gq; gr.x += o.x; gr.y += o.y; }
                *gq = gr;
                g.fld[i] = cmulc(a, (RI || p.pre_t) ? db : gen_modulator(db, p.k1, p.sigma));
            }
            __syncthreads();
            if (step > 0) gen_convolve<true>(g, p, p.gen_hs, v);
        }
        if (p.grad_probe) {
            float2* gp = p.grad_probe + (size_t)b * p.gprobe_bstride + (size_t)m * row;     // this position's own slot
            for (int i = g.tid; i < g.n; i += g.nt) gp[i] = g.fld[i];
        }
    }
}

// threads per workgroup: at most GEN_E field elements per thread
int ms_generic_threads(int py, int px) {
    const int n = py * px;
    int nt = ((n + GEN_E - 1) / GEN_E + 63) / 64 * 64;
    if (nt < 256) nt = 256;
    return nt;
}
bool ms_generic_supported(int py, int px) {
    const size_t lds = ((size_t)py * px + px + py) * sizeof(float2) + 64;
    return ms_generic_threads(py, px) <= 1024 && lds <= 160 * 1024 - 256;
}

hipError_t ms_generic_launch(const MsParams& p, int batch, hipStream_t st) {
    const int nt = ms_generic_threads(p.gen_py, p.gen_px);
    const size_t lds = ((size_t)p.gen_py * p.gen_px + p.gen_px + p.gen_py) * sizeof(float2);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ms_generic_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(ms_generic_kernel, dim3(batch), dim3(nt), lds, st, p);
    return hipGetLastError();
}

}  // namespace adm
