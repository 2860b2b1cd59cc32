// Whole-object passes: rotation gather / scatter (R2), regulariser gradient (R9),
// fused Adam / GD + constraints (R13-R15), axpy.  All HBM-bound streaming kernels.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include <hip/hip_fp16.h>
#include "adm_common.h"
#include "adm_optim.h"
#include "adm_ms_math.h"

namespace adm {

// --------------------------------------------------------------------------------------------
// Rotation about axis 0.  Reference: apply_rotation -> w.grid_sample (adorym/util.py:536-552,
// adorym/wrappers.py:1105-1147), torch grid_sampler bilinear / border / align_corners=False.
// One thread owns one rotated-frame (x', z') and loops over the y planes, so the coordinate
// pipeline (fp16 table -> fp64 normalise -> fp32 un-normalise -> clamp -> weights) runs once per
// thread.  A 16x16 (x', z') patch per block keeps the source footprint compact for any angle;
// writes are 128-B row segments of the slice-major [Z][Yp][Xp][2] layout.
// --------------------------------------------------------------------------------------------
struct RotGeom {
    int Y, X, Z, Yp, Xp, pad_y0, pad_x0;
};

struct Bilin {
    int i00, i01, i10, i11;   // float2 offsets inside one y plane of obj ([X][Z])
    float w00, w01, w10, w11;
};

__device__ __forceinline__ Bilin make_bilin(const uint16_t* coords, int xr, int zr, int X, int Z) {
#pragma clang fp contract(off)      // torch / NumPy evaluate this pipeline without fused multiply-adds: match them bit for bit
    Bilin b;
    if (coords == nullptr) {
        b.i00 = b.i01 = b.i10 = b.i11 = xr * Z + zr;
        b.w00 = 1.f; b.w01 = b.w10 = b.w11 = 0.f;
        return b;
    }
    const __half* ch = reinterpret_cast<const __half*>(coords) + 2 * ((size_t)xr * Z + zr);
    const double x_old = (double)__half2float(ch[0]);
    const double z_old = (double)__half2float(ch[1]);
    // wrappers.py:1137: grid = -1 + 2*grid/arr_shape + 1/arr_shape on the flipped (z, x) pair, arr_shape = (X, Z)
    const float gz = (float)(-1.0 + 2.0 * z_old / (double)X + 1.0 / (double)X);
    const float gx = (float)(-1.0 + 2.0 * x_old / (double)Z + 1.0 / (double)Z);
    float iz = ((gz + 1.f) * (float)Z - 1.f) / 2.f;
    float ix = ((gx + 1.f) * (float)X - 1.f) / 2.f;
    iz = fminf((float)(Z - 1), fmaxf(iz, 0.f));
    ix = fminf((float)(X - 1), fmaxf(ix, 0.f));
    const float fz = floorf(iz), fx = floorf(ix);
    const float tz = iz - fz, tx = ix - fx;
    const int z0 = (int)fz, x0 = (int)fx;
    const bool vz = (z0 + 1 <= Z - 1), vx = (x0 + 1 <= X - 1);
    const int z1 = vz ? z0 + 1 : z0, x1 = vx ? x0 + 1 : x0;
    b.i00 = x0 * Z + z0; b.i01 = x0 * Z + z1; b.i10 = x1 * Z + z0; b.i11 = x1 * Z + z1;
    b.w00 = (1.f - tx) * (1.f - tz);
    b.w01 = vz ? (1.f - tx) * tz : 0.f;
    b.w10 = vx ? tx * (1.f - tz) : 0.f;
    b.w11 = (vx && vz) ? tx * tz : 0.f;
    return b;
}

// trans != nullptr: the slice transmission of every gathered voxel is stored beside it (same layout), so that the
// multislice kernel multiplies with a loaded number instead of evaluating exp / sincos per covering position and sweep.
__global__ __launch_bounds__(256) void rotate_fwd_kernel(const float2* __restrict__ obj, const uint16_t* __restrict__ coords,
                                                         float2* __restrict__ rot, float2* __restrict__ trans, float k1, float sigma,
                                                         RotGeom g, int y_lo, int y_hi, int y_chunk) {
    const int xr = blockIdx.x * 16 + (threadIdx.x & 15);
    const int zr = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (xr >= g.X || zr >= g.Z) return;
    const Bilin b = make_bilin(coords, xr, zr, g.X, g.Z);
    const int ya = y_lo + blockIdx.z * y_chunk;
    const int yb = min(ya + y_chunk, y_hi);
    const size_t plane = (size_t)g.X * g.Z;
    for (int y = ya; y < yb; ++y) {
        const float2* o = obj + (size_t)y * plane;
        const float2 v00 = o[b.i00], v01 = o[b.i01], v10 = o[b.i10], v11 = o[b.i11];
        float2 r;
        r.x = v00.x * b.w00 + v01.x * b.w01 + v10.x * b.w10 + v11.x * b.w11;
        r.y = v00.y * b.w00 + v01.y * b.w01 + v10.y * b.w10 + v11.y * b.w11;
        const size_t o_rot = ((size_t)zr * g.Yp + g.pad_y0 + y) * g.Xp + g.pad_x0 + xr;
        if (rot) rot[o_rot] = r;
        if (trans) trans[o_rot] = slice_transmission(r, k1, sigma);
    }
}

// R objects stacked along y, block r (planes [r * Yb, (r + 1) * Yb) of the stacked rotated frame) gathered with ITS angle's table
// from the ONE real object (adorym_amd.AngleBatch: the 16 angles of a config-2 update): one launch instead of R launches of a
// few blocks each.  Same arithmetic per voxel as rotate_fwd_kernel.
__global__ __launch_bounds__(256) void rotate_fwd_stack_kernel(const float2* __restrict__ obj, const uint16_t* const* __restrict__ tables,
                                                               int Yb, float2* __restrict__ rot, float2* __restrict__ trans, float k1,
                                                               float sigma, RotGeom g, int y_chunk) {
    const int xr = blockIdx.x * 16 + (threadIdx.x & 15);
    const int zr = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (xr >= g.X || zr >= g.Z) return;
    const int ya = blockIdx.z * y_chunk;                 // (y_chunk divides Yb: a block never straddles two angles)
    const int r = ya / Yb;
    const Bilin b = make_bilin(tables[r], xr, zr, g.X, g.Z);
    const int yb = min(ya + y_chunk, g.Y);
    const size_t plane = (size_t)g.X * g.Z;
    for (int y = ya; y < yb; ++y) {
        const float2* o = obj + (size_t)(y - r * Yb) * plane;
        const float2 v00 = o[b.i00], v01 = o[b.i01], v10 = o[b.i10], v11 = o[b.i11];
        float2 q;
        q.x = v00.x * b.w00 + v01.x * b.w01 + v10.x * b.w10 + v11.x * b.w11;
        q.y = v00.y * b.w00 + v01.y * b.w01 + v10.y * b.w10 + v11.y * b.w11;
        const size_t o_rot = ((size_t)zr * g.Yp + g.pad_y0 + y) * g.Xp + g.pad_x0 + xr;
        if (rot) rot[o_rot] = q;
        if (trans) trans[o_rot] = slice_transmission(q, k1, sigma);
    }
}

// No rotation (coords == nullptr) on a thin object -- the 2-D modes, Z = 1: the general kernels above would keep one thread in
// sixteen busy (their 16 x 16 patches span (x', z')).  One thread per voxel, z fastest like the object: the same numbers (weights
// 1, 0, 0, 0), 13-14 us -> a plain copy's time on the config-1 shape.
__global__ __launch_bounds__(256) void identity_fwd_kernel(const float2* __restrict__ obj, float2* __restrict__ rot, float2* __restrict__ trans,
                                                           float k1, float sigma, RotGeom g, int y_lo, int y_hi) {
    const size_t n = (size_t)(y_hi - y_lo) * g.X * g.Z;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int z = (int)(i % g.Z);
        const size_t yx = i / g.Z;
        const int x = (int)(yx % g.X), y = y_lo + (int)(yx / g.X);
        const float2 r = obj[((size_t)y * g.X + x) * g.Z + z];
        const size_t o_rot = ((size_t)z * g.Yp + g.pad_y0 + y) * g.Xp + g.pad_x0 + x;
        if (rot) rot[o_rot] = r;
        if (trans) trans[o_rot] = slice_transmission(r, k1, sigma);
    }
}
__global__ __launch_bounds__(256) void identity_adj_kernel(const float2* __restrict__ grot, float2* __restrict__ gobj, RotGeom g, int y_lo,
                                                           int y_hi) {
    const size_t n = (size_t)(y_hi - y_lo) * g.X * g.Z;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int z = (int)(i % g.Z);
        const size_t yx = i / g.Z;
        const int x = (int)(yx % g.X), y = y_lo + (int)(yx / g.X);
        const float2 v = grot[((size_t)z * g.Yp + g.pad_y0 + y) * g.Xp + g.pad_x0 + x];
        float2* o = gobj + ((size_t)y * g.X + x) * g.Z + z;
        float2 c = *o;
        c.x += v.x;
        c.y += v.y;
        *o = c;
    }
}

// slice transmissions of rows [row_lo, row_hi) of every slice of a rotated-frame buffer (pads included): the cache's
// initial fill (obj_rot == nullptr: vacuum, 1 + 0i) and adm_transmission_refresh
__global__ __launch_bounds__(256) void transmission_kernel(const float2* __restrict__ rot, float2* __restrict__ trans, float k1,
                                                           float sigma, int Z, int Yp, int Xp, int row_lo, int row_hi) {
    const size_t per = (size_t)(row_hi - row_lo) * Xp;
    const size_t n = per * Z;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t z = i / per, r = i - z * per;
        const size_t o = (z * Yp + row_lo) * Xp + r;
        trans[o] = rot ? slice_transmission(rot[o], k1, sigma) : make_float2(1.f, 0.f);
    }
}

__global__ __launch_bounds__(256) void rotate_adj_kernel(const float2* __restrict__ grot, const uint16_t* __restrict__ coords,
                                                         float* __restrict__ gobj, RotGeom g, int y_lo, int y_hi, int y_chunk) {
    const int xr = blockIdx.x * 16 + (threadIdx.x & 15);
    const int zr = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (xr >= g.X || zr >= g.Z) return;
    const Bilin b = make_bilin(coords, xr, zr, g.X, g.Z);
    const int ya = y_lo + blockIdx.z * y_chunk;
    const int yb = min(ya + y_chunk, y_hi);
    const size_t plane = (size_t)g.X * g.Z;
    for (int y = ya; y < yb; ++y) {
        const float2 v = grot[((size_t)zr * g.Yp + g.pad_y0 + y) * g.Xp + g.pad_x0 + xr];
        float* o = gobj + 2 * (size_t)y * plane;
        if (b.w00 != 0.f) { atomicAdd(o + 2 * b.i00, v.x * b.w00); atomicAdd(o + 2 * b.i00 + 1, v.y * b.w00); }
        if (b.w01 != 0.f) { atomicAdd(o + 2 * b.i01, v.x * b.w01); atomicAdd(o + 2 * b.i01 + 1, v.y * b.w01); }
        if (b.w10 != 0.f) { atomicAdd(o + 2 * b.i10, v.x * b.w10); atomicAdd(o + 2 * b.i10 + 1, v.y * b.w10); }
        if (b.w11 != 0.f) { atomicAdd(o + 2 * b.i11, v.x * b.w11); atomicAdd(o + 2 * b.i11 + 1, v.y * b.w11); }
    }
}

// Rotation adjoint as a GATHER (deterministic, no atomics): the transpose of the bilinear sampling
// operator is prebuilt per angle as a CSR matrix over object-plane voxels (host: adorym_amd/util.py
// build_rotation_adjoint_csr, same fp32 coordinate pipeline as make_bilin).  One thread owns one object
// voxel column (x, z) -- consecutive threads are consecutive z, the fastest object axis, so the
// read-modify-write of grad_obj is coalesced -- and walks the y planes four at a time.
// A block owns a 16 x 16 patch of object-plane voxels and four y planes.  ALONG_X = false: lanes run along z (the
// fastest object axis): gobj accesses are coalesced, and so are the gathers from grad_rot when |sin(theta)| is large
// (a step in z is then a step in x').  ALONG_X = true (|cos| > |sin|): lanes run along x so that the gathered
// rotated-frame voxels are consecutive in x' (the fastest axis of [Z][Yp][Xp]); the patch is then transposed through
// LDS so that the read-modify-write of gobj is still issued along z.
template <bool ALONG_X>
__global__ __launch_bounds__(256) void rotate_adj_csr_kernel(const float2* __restrict__ grot, const int* __restrict__ ptr,
                                                             const int* __restrict__ src, const float* __restrict__ wgt,
                                                             float2* __restrict__ gobj, RotGeom g, int y_lo, int y_hi) {
    __shared__ float2 tile[4][16][17];
    const int lx = ALONG_X ? (threadIdx.x & 15) : (threadIdx.x >> 4);
    const int lz = ALONG_X ? (threadIdx.x >> 4) : (threadIdx.x & 15);
    const int x = blockIdx.x * 16 + lx, z = blockIdx.y * 16 + lz;
    const bool ok = (x < g.X) && (z < g.Z);
    const int t = ok ? x * g.Z + z : 0;
    const int beg = ok ? ptr[t] : 0, end = ok ? ptr[t + 1] : 0;
    const int y0 = y_lo + blockIdx.z * 4;
    const int ny = min(4, y_hi - y0);
    const size_t plane = (size_t)g.X * g.Z;
    float2 a[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = make_float2(0.f, 0.f);
    const size_t row = (size_t)(g.pad_y0 + y0) * g.Xp;
    if (ny == 4) {
        for (int j = beg; j < end; ++j) {
            const float w = wgt[j];
            const float2* q = grot + (size_t)src[j] + row;
            const float2 v0 = q[0], v1 = q[g.Xp], v2 = q[2 * (size_t)g.Xp], v3 = q[3 * (size_t)g.Xp];
            a[0].x += w * v0.x; a[0].y += w * v0.y;
            a[1].x += w * v1.x; a[1].y += w * v1.y;
            a[2].x += w * v2.x; a[2].y += w * v2.y;
            a[3].x += w * v3.x; a[3].y += w * v3.y;
        }
    } else {
        for (int j = beg; j < end; ++j) {
            const float w = wgt[j];
            const float2* q = grot + (size_t)src[j] + row;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < ny) { const float2 v = q[(size_t)i * g.Xp]; a[i].x += w * v.x; a[i].y += w * v.y; }
        }
    }
    int wx = lx, wz = lz;
    if (ALONG_X) {
#pragma unroll
        for (int i = 0; i < 4; ++i) tile[i][lx][lz] = a[i];
        __syncthreads();
        wx = threadIdx.x >> 4; wz = threadIdx.x & 15;          // now lanes run along z
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = tile[i][wx][wz];
    }
    const int ox = blockIdx.x * 16 + wx, oz = blockIdx.y * 16 + wz;
    if (ox < g.X && oz < g.Z) {
        float2* o = gobj + (size_t)y0 * plane + (size_t)ox * g.Z + oz;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < ny) { float2 c = o[(size_t)i * plane]; c.x += a[i].x; c.y += a[i].y; o[(size_t)i * plane] = c; }
    }
}

// LDS-staged variant of the CSR rotation adjoint: the sources of a 16 x 16 patch of object-plane voxels lie in a
// (x', z') bounding box of at most ~27 x 27 rotated-frame voxels (host-computed per patch and angle).  The box rows are
// loaded contiguously along x' into LDS for four y planes, and the bilinear-transpose gather then runs out of LDS -- the
// memory access pattern no longer depends on the angle (the direct gather is 10x slower at 45 degrees than at 0).
// lsrc[j] = (z' - box.z0) * box.w + (x' - box.x0) of CSR entry j.
#define ADM_STAGE_MAX 1024          // float2 elements per plane in LDS when four planes are staged at once
__global__ __launch_bounds__(256) void rotate_adj_staged_kernel(const float2* __restrict__ grot, const int* __restrict__ ptr,
                                                                const int* __restrict__ src, const unsigned short* __restrict__ lsrc,
                                                                const float* __restrict__ wgt,
                                                                const int4* __restrict__ boxes, float2* __restrict__ gobj, RotGeom g,
                                                                int y_lo, int y_hi) {
    __shared__ float2 stage[4 * ADM_STAGE_MAX];
    const int4 box = boxes[blockIdx.y * gridDim.x + blockIdx.x];      // (x0, z0, w, h)
    const int y0 = y_lo + blockIdx.z * 4;
    const int ny = min(4, y_hi - y0);
    const int bw = box.z, bh = box.w, per = bw * bh;
    const size_t slice = (size_t)g.Yp * g.Xp;
    const size_t plane = (size_t)g.X * g.Z;
    const int lx = threadIdx.x >> 4, lz = threadIdx.x & 15;          // lanes along z, the fastest object axis
    const int x = blockIdx.x * 16 + lx, z = blockIdx.y * 16 + lz;
    const bool ok = (x < g.X) && (z < g.Z);
    const int t = ok ? x * g.Z + z : 0;
    const int beg = ok ? ptr[t] : 0, end = ok ? ptr[t + 1] : 0;
    float2* o = gobj + (size_t)y0 * plane + (size_t)x * g.Z + z;
    if (bw == 0) {
        // no usable box (only for objects much larger than 256^3): gather from global memory
        if (!ok) return;
        const size_t row = (size_t)(g.pad_y0 + y0) * g.Xp;
        float2 a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = make_float2(0.f, 0.f);
        for (int j = beg; j < end; ++j) {
            const float w = wgt[j];
            const float2* q = grot + (size_t)src[j] + row;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < ny) { const float2 v = q[(size_t)i * g.Xp]; a[i].x += w * v.x; a[i].y += w * v.y; }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < ny) { float2 c = o[(size_t)i * plane]; c.x += a[i].x; c.y += a[i].y; o[(size_t)i * plane] = c; }
        return;
    }
    if (per <= ADM_STAGE_MAX) {
        // interior patch: the boxes of four y planes fit at once
        for (int idx = threadIdx.x; idx < 4 * per; idx += 256) {
            const int p = idx / per, rem = idx - p * per;
            const int zz = rem / bw, xx = rem - zz * bw;
            float2 v = make_float2(0.f, 0.f);
            if (p < ny) v = grot[(size_t)(box.y + zz) * slice + (size_t)(g.pad_y0 + y0 + p) * g.Xp + g.pad_x0 + box.x + xx];
            stage[p * ADM_STAGE_MAX + rem] = v;
        }
        __syncthreads();
        if (!ok) return;
        float2 a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = make_float2(0.f, 0.f);
        for (int j = beg; j < end; ++j) {
            const float w = wgt[j];
            const int q = lsrc[j];
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float2 v = stage[i * ADM_STAGE_MAX + q]; a[i].x += w * v.x; a[i].y += w * v.y; }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < ny) { float2 c = o[(size_t)i * plane]; c.x += a[i].x; c.y += a[i].y; o[(size_t)i * plane] = c; }
        return;
    }
    // rim patch (border clamping folds a corner of the rotated frame onto it: box of up to 4096 voxels): one or two planes
    // per pass over the CSR entries, whatever fits the 32 KB stage
    const int npp = (2 * per <= 4 * ADM_STAGE_MAX) ? 2 : 1;
    for (int p0 = 0; p0 < ny; p0 += npp) {
        for (int idx = threadIdx.x; idx < npp * per; idx += 256) {
            const int pp = idx / per, rem = idx - pp * per;
            const int zz = rem / bw, xx = rem - zz * bw;
            float2 v = make_float2(0.f, 0.f);
            if (p0 + pp < ny) v = grot[(size_t)(box.y + zz) * slice + (size_t)(g.pad_y0 + y0 + p0 + pp) * g.Xp + g.pad_x0 + box.x + xx];
            stage[idx] = v;
        }
        __syncthreads();
        if (ok) {
            float2 acc0 = make_float2(0.f, 0.f), acc1 = make_float2(0.f, 0.f);
            const int second = (npp == 2) ? per : 0;
            for (int j = beg; j < end; j += 8) {       // eight entries' weight/offset loads in flight, then the LDS gathers
                float w[8];
                int q[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int jj = min(j + u, end - 1);
                    w[u] = (j + u < end) ? wgt[jj] : 0.f;
                    q[u] = lsrc[jj];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float2 v0 = stage[q[u]];
                    const float2 v1 = stage[q[u] + second];
                    acc0.x += w[u] * v0.x; acc0.y += w[u] * v0.y;
                    acc1.x += w[u] * v1.x; acc1.y += w[u] * v1.y;
                }
            }
            float2 c = o[(size_t)p0 * plane];
            c.x += acc0.x;
            c.y += acc0.y;
            o[(size_t)p0 * plane] = c;
            if (npp == 2 && p0 + 1 < ny) {
                float2 d = o[(size_t)(p0 + 1) * plane];
                d.x += acc1.x;
                d.y += acc1.y;
                o[(size_t)(p0 + 1) * plane] = d;
            }
        }
        __syncthreads();
    }
}

// The staged adjoint for R objects stacked along y (adorym_amd.AngleBatch): block r of the stacked gradient image is back-rotated with
// ITS angle's CSR and all R contributions are added, r ascending, to the ONE real gradient -- c = g; c += a_0; c += a_1; ... ; g = c:
// the additions R sequential launches of rotate_adj_staged_kernel would make, in the same order, in one launch (16 launches of
// 6 - 12 us each per config-2 update).  Same three cases per (patch, angle) as above, written without early exits so that every
// thread reaches the barriers that separate one angle's use of the LDS stage from the next.
struct AdjTables { const int* ptr; const int* src; const unsigned short* lsrc; const float* wgt; const int4* boxes; };
// `part` != nullptr: the angles run in PARALLEL -- blockIdx.z = (angle, plane group), the block leaves its angle's term a_r in
// part[r][y][x][z] and stack_sum_kernel forms c = g; c += a_0; c += a_1; ... afterwards (the same additions in the same order: same
// bits); a block walking all R angles one after the other is a chain of R stage-fill / barrier / gather rounds (86 us for 16
// angles of a 64^3 object against 20 + 8).
__global__ __launch_bounds__(256) void rotate_adj_staged_stack_kernel(const float2* __restrict__ grot, const AdjTables* __restrict__ tabs,
                                                                      int R, int Yb, float2* __restrict__ gobj, RotGeom g, int npl,
                                                                      float2* __restrict__ part) {
    __shared__ float2 stage[4 * ADM_STAGE_MAX];
    const int nzg = (Yb + npl - 1) / npl;                // plane groups
    const int zg = part ? (int)blockIdx.z % nzg : (int)blockIdx.z;
    const int r_lo = part ? (int)blockIdx.z / nzg : 0, r_hi = part ? r_lo + 1 : R;
    const int y0 = zg * npl;                             // first plane (of the REAL object, Yb planes) of this block: npl = 1, 2 or 4 planes
    const int ny = min(npl, Yb - y0);
    const size_t slice = (size_t)g.Yp * g.Xp;
    const size_t plane = (size_t)g.X * g.Z;
    const int lx = threadIdx.x >> 4, lz = threadIdx.x & 15;
    const int x = blockIdx.x * 16 + lx, z = blockIdx.y * 16 + lz;
    const bool ok = (x < g.X) && (z < g.Z);
    const int t = ok ? x * g.Z + z : 0;
    float2* o = (part ? part + (size_t)r_lo * Yb * plane : gobj) + (size_t)y0 * plane + (size_t)x * g.Z + z;
    float2 c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = (ok && i < ny && !part) ? o[(size_t)i * plane] : make_float2(0.f, 0.f);
    for (int r = r_lo; r < r_hi; ++r) {
        const AdjTables T = tabs[r];
        const int4 box = T.boxes[blockIdx.y * gridDim.x + blockIdx.x];      // (x0, z0, w, h)
        const int ys = r * Yb + y0;                     // the same planes in block r of the stacked image
        const int bw = box.z, bh = box.w, per = bw * bh;
        const int beg = ok ? T.ptr[t] : 0, end = ok ? T.ptr[t + 1] : 0;
        if (bw == 0) {
            const size_t row = (size_t)(g.pad_y0 + ys) * g.Xp;
            float2 a[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = make_float2(0.f, 0.f);
            for (int j = beg; j < end; ++j) {
                const float w = T.wgt[j];
                const float2* q = grot + (size_t)T.src[j] + row;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < ny) { const float2 v = q[(size_t)i * g.Xp]; a[i].x += w * v.x; a[i].y += w * v.y; }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) { c[i].x += a[i].x; c[i].y += a[i].y; }
        } else if (per <= ADM_STAGE_MAX) {
            for (int idx = threadIdx.x; idx < npl * per; idx += 256) {
                const int p = idx / per, rem = idx - p * per;
                const int zz = rem / bw, xx = rem - zz * bw;
                float2 v = make_float2(0.f, 0.f);
                if (p < ny) v = grot[(size_t)(box.y + zz) * slice + (size_t)(g.pad_y0 + ys + p) * g.Xp + g.pad_x0 + box.x + xx];
                stage[p * ADM_STAGE_MAX + rem] = v;
            }
            __syncthreads();
            float2 a[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = make_float2(0.f, 0.f);
            for (int j = beg; j < end; ++j) {
                const float w = T.wgt[j];
                const int q = T.lsrc[j];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < ny) { const float2 v = stage[i * ADM_STAGE_MAX + q]; a[i].x += w * v.x; a[i].y += w * v.y; }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) { c[i].x += a[i].x; c[i].y += a[i].y; }
        } else {
            const int npp = (npl >= 2 && 2 * per <= 4 * ADM_STAGE_MAX) ? 2 : 1;
            for (int p0 = 0; p0 < ny; p0 += npp) {
                for (int idx = threadIdx.x; idx < npp * per; idx += 256) {
                    const int pp = idx / per, rem = idx - pp * per;
                    const int zz = rem / bw, xx = rem - zz * bw;
                    float2 v = make_float2(0.f, 0.f);
                    if (p0 + pp < ny) v = grot[(size_t)(box.y + zz) * slice + (size_t)(g.pad_y0 + ys + p0 + pp) * g.Xp + g.pad_x0 + box.x + xx];
                    stage[idx] = v;
                }
                __syncthreads();
                float2 acc0 = make_float2(0.f, 0.f), acc1 = make_float2(0.f, 0.f);
                const int second = (npp == 2) ? per : 0;
                for (int j = beg; j < end; j += 8) {
                    float w[8];
                    int q[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int jj = min(j + u, end - 1);
                        w[u] = (j + u < end) ? T.wgt[jj] : 0.f;
                        q[u] = T.lsrc[jj];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const float2 v0 = stage[q[u]];
                        const float2 v1 = stage[q[u] + second];
                        acc0.x += w[u] * v0.x; acc0.y += w[u] * v0.y;
                        acc1.x += w[u] * v1.x; acc1.y += w[u] * v1.y;
                    }
                }
                // (p0 is uniform: the static indices below keep c[] in registers)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i == p0) { c[i].x += acc0.x; c[i].y += acc0.y; }
                    if (npp == 2 && i == p0 + 1) { c[i].x += acc1.x; c[i].y += acc1.y; }
                }
                __syncthreads();
            }
        }
        __syncthreads();                                 // the next angle refills the stage
    }
    if (ok) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < ny) o[(size_t)i * plane] = c[i];
    }
}

// --------------------------------------------------------------------------------------------
// Overlap-add of the per-position tile gradients written by the multislice kernel (adjoint of the
// tile gather, adorym/forward_model.py:313-331).  Tiles overlap, so this is a gather per rotated-frame
// pixel over the positions that cover it: deterministic, no atomics.
//   cover_build_kernel : per padded pixel (y, x) the list of (position, element) sources
//   tile_accumulate_kernel : grad_rot[s][y][x] = sum of the sources, for every slice
// gtile layout per position: [step] rows of R1 x NT elements, element (k, tid) at ws_elem_offset() (adm_ms_math.h), with
// pixel (row, col) -> k = col / R2, tid = (row / LPW) * 64 + (row % LPW) * G + col % R2 (the multislice kernel's
// thread-native order).
// --------------------------------------------------------------------------------------------
#define ADM_MAXCOVER 64

struct TileGeom {
    int Yp, Xp, pad_y0, pad_x0, Py, Px, R1, R2, G, LPW, NT, n_steps, binning, Z;
    int pixel_major;      // rows written by the generic kernel: [Py][Px]; else the tuned kernels' thread-native order
    int row_elems;        // float2 elements of one row (one modulation step of one position)
    int row0, nrows;      // padded-row window touched by the batch
    int add_lo, add_hi;   // padded rows [add_lo, add_hi) already hold an earlier part of the same batch: accumulate there
};

// positions [b0, B) of the batch enter the lists (b0 > 0: one pass of a batch whose coverage exceeds ADM_MAXCOVER, see
// adm_tile_grad_accumulate_range)
__global__ __launch_bounds__(256) void cover_build_kernel(const int2* __restrict__ pos, int b0, int B, TileGeom g,
                                                          unsigned* __restrict__ cover, int* __restrict__ overflow) {
    // The positions are read 256 at a time, and only those whose tile reaches this block's 32 x 8 pixels go on -- in position
    // order (ballot + prefix count per wave) -- to the per-pixel loop: a pixel used to walk ALL positions of the batch, one
    // dependent load each (15 us per 64 positions whatever the number of pixels; 29 us for the 200 entries of a tiled
    // multi-distance launch, longer than the multislice kernel beside it).
    __shared__ int2 sp[256];
    __shared__ int sb[256];
    __shared__ int wcnt[4];
    const int x = blockIdx.x * 32 + (threadIdx.x & 31);
    const int r = blockIdx.y * 8 + (threadIdx.x >> 5);
    const bool live = x < g.Xp && r < g.nrows;
    const int y = g.row0 + r;
    const int bx_lo = blockIdx.x * 32, by_lo = g.row0 + blockIdx.y * 8;         // the block's pixels: [bx_lo, +32) x [by_lo, +8)
    // structure-of-arrays: entry c of pixel (r, x) at cover[(c * nrows + r) * Xp + x], so that the lanes of a wave
    // (consecutive x) read consecutive words
    const size_t cplane = (size_t)g.nrows * g.Xp;
    unsigned* out = cover + (size_t)(live ? r : 0) * g.Xp + (live ? x : 0);
    int cnt = 0;
    const unsigned per_pos = (unsigned)g.n_steps * g.row_elems;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int c0 = b0; c0 < B; c0 += 256) {
        const int b = c0 + (int)threadIdx.x;
        int2 p = make_int2(0, 0);
        bool hit = false;
        if (b < B) {
            p = pos[b];
            const int ty = p.x + g.pad_y0, tx = p.y + g.pad_x0;                   // the tile's first padded row / column
            hit = ty < by_lo + 8 && ty + g.Py > by_lo && tx < bx_lo + 32 && tx + g.Px > bx_lo;
        }
        const unsigned long long m = __ballot(hit);
        __syncthreads();                                                          // (the previous chunk's candidates are consumed)
        if (lane == 0) wcnt[wave] = __popcll(m);
        __syncthreads();
        int off = 0, n = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { if (w < wave) off += wcnt[w]; n += wcnt[w]; }
        if (hit) {
            const int k = off + __popcll(m & ((1ull << lane) - 1ull));
            sp[k] = p;
            sb[k] = b;
        }
        __syncthreads();
        if (!live) continue;
        for (int j = 0; j < n; ++j) {
            const int2 q = sp[j];
            const int row = y - (q.x + g.pad_y0), col = x - (q.y + g.pad_x0);
            if (row >= 0 && row < g.Py && col >= 0 && col < g.Px) {
                if (cnt < ADM_MAXCOVER) {
                    unsigned off_e;
                    if (g.pixel_major) off_e = (unsigned)(row * g.Px + col);
                    else {
                        const int tid = (row / g.LPW) * 64 + (row % g.LPW) * g.G + col % g.R2;
                        off_e = adm::ws_elem_offset(g.R1, g.NT, col / g.R2, tid);
                    }
                    out[(size_t)(1 + cnt) * cplane] = (unsigned)sb[j] * per_pos + off_e;
                }
                ++cnt;
            }
        }
    }
    if (!live) return;
    if (cnt > ADM_MAXCOVER) { atomicExch(overflow, 1); cnt = ADM_MAXCOVER; }
    out[0] = (unsigned)cnt;
}

// One thread owns one padded pixel and TA_STEPS consecutive modulation steps: for every covering tile it issues TA_STEPS
// independent 8-B loads (stride = one step of that tile) before touching the accumulators (TA_CU > 1: of TA_CU tiles at a
// time).  With the XCD-aware grid below the kernel is bound by how much of a fetched line its neighbours still find in
// their L2, so FEWER steps per block are better: 256 positions take 0.76 / 0.87 / 1.01 ms at TA_STEPS 2 / 4 / 8 (round 1,
// blockIdx = (x, y, z): 0.97 at 4, the best of 2/4/8/16 then); TA_CU 2-8 and an x loop inside the block change nothing.
#ifndef TA_STEPS
#define TA_STEPS 2
#endif
#ifndef TA_CU
#define TA_CU 1
#endif
// the work of one block of tile_accumulate_kernel: step chunk zc, pixel block `rem` (x fastest)
__device__ __forceinline__ void ta_block(const float2* __restrict__ gtile, const unsigned* __restrict__ cover, float2* __restrict__ grad_rot,
                                         const TileGeom& g, int zc, int rem) {
    const int nbx = (g.Xp + 31) / 32;
    const int x = (rem % nbx) * 32 + (threadIdx.x & 31);
    const int r = (rem / nbx) * 8 + (threadIdx.x >> 5);
    if (x >= g.Xp || r >= g.nrows) return;
    const size_t cplane = (size_t)g.nrows * g.Xp;
    const unsigned* cv = cover + (size_t)r * g.Xp + x;
    const int cnt = (int)cv[0];
    const size_t step_stride = (size_t)g.row_elems;
    const size_t slice_stride = (size_t)g.Yp * g.Xp;
    float2* out = grad_rot + (size_t)(g.row0 + r) * g.Xp + x;
    const bool add = (g.row0 + r >= g.add_lo) && (g.row0 + r < g.add_hi);
    const int st0 = zc * TA_STEPS;
    const int nst = min(TA_STEPS, g.n_steps - st0);
    float2 acc[TA_STEPS];
#pragma unroll
    for (int i = 0; i < TA_STEPS; ++i) acc[i] = make_float2(0.f, 0.f);
    if (nst == TA_STEPS) {
        int c = 0;
        for (; c + TA_CU <= cnt; c += TA_CU) {         // TA_CU tiles x TA_STEPS steps in flight; added in list order
            float2 v[TA_CU][TA_STEPS];
#pragma unroll
            for (int u = 0; u < TA_CU; ++u) {
                const float2* src = gtile + (size_t)cv[(size_t)(1 + c + u) * cplane] + (size_t)st0 * step_stride;
#pragma unroll
                for (int i = 0; i < TA_STEPS; ++i) v[u][i] = src[(size_t)i * step_stride];
            }
#pragma unroll
            for (int u = 0; u < TA_CU; ++u)
#pragma unroll
                for (int i = 0; i < TA_STEPS; ++i) { acc[i].x += v[u][i].x; acc[i].y += v[u][i].y; }
        }
        for (; c < cnt; ++c) {
            const float2* src = gtile + (size_t)cv[(size_t)(1 + c) * cplane] + (size_t)st0 * step_stride;
            float2 v[TA_STEPS];
#pragma unroll
            for (int i = 0; i < TA_STEPS; ++i) v[i] = src[(size_t)i * step_stride];
#pragma unroll
            for (int i = 0; i < TA_STEPS; ++i) { acc[i].x += v[i].x; acc[i].y += v[i].y; }
        }
    } else if (nst == 1) {
        // one step left (every thin object: 2-D ptychography, sub-tiles of holograms): the loop over the covering tiles is all
        // there is -- four list entries and their four values in flight at a time, added in list order (same bits as one by one)
        int c = 0;
        for (; c + 4 <= cnt; c += 4) {
            unsigned e[4];
            float2 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) e[u] = cv[(size_t)(1 + c + u) * cplane];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = gtile[(size_t)e[u] + (size_t)st0 * step_stride];
#pragma unroll
            for (int u = 0; u < 4; ++u) { acc[0].x += v[u].x; acc[0].y += v[u].y; }
        }
        for (; c < cnt; ++c) {
            const float2 v = gtile[(size_t)cv[(size_t)(1 + c) * cplane] + (size_t)st0 * step_stride];
            acc[0].x += v.x; acc[0].y += v.y;
        }
    } else {
        for (int c = 0; c < cnt; ++c) {
            const float2* src = gtile + (size_t)cv[(size_t)(1 + c) * cplane] + (size_t)st0 * step_stride;
#pragma unroll
            for (int i = 0; i < TA_STEPS; ++i)
                if (i < nst) { const float2 v = src[(size_t)i * step_stride]; acc[i].x += v.x; acc[i].y += v.y; }
        }
    }
#pragma unroll
    for (int i = 0; i < TA_STEPS; ++i) {
        if (i < nst) {
            const int st = st0 + i;
            const int s_lo = st * g.binning, s_hi = min(s_lo + g.binning, g.Z);
            for (int sl = s_lo; sl < s_hi; ++sl) {
                float2 v = acc[i];
                if (add) { const float2 o = out[(size_t)sl * slice_stride]; v.x += o.x; v.y += o.y; }
                out[(size_t)sl * slice_stride] = v;
            }
        }
    }
}

__global__ __launch_bounds__(256) void tile_accumulate_kernel(const float2* __restrict__ gtile, const unsigned* __restrict__ cover,
                                                              float2* __restrict__ grad_rot, TileGeom g) {
    // 1-D grid, XCD-aware: blocks are dealt round-robin over the 8 XCDs, and the 72-144-byte runs a block reads from a
    // tile row straddle 128-byte lines that its x / y neighbours read too.  XCD k takes the step chunks k, k + 8, ... and
    // ALL pixel blocks of each, x fastest, so that neighbours share an L2 (with blockIdx = (x, y, z) every straddled
    // line was fetched from HBM by two or three XCDs; FETCH_SIZE -12 %, time -10 % from the mapping alone).
    const int nbx = (g.Xp + 31) / 32, nby = (g.nrows + 7) / 8;
    const int idx = blockIdx.x >> 3;
    const int zc = (blockIdx.x & 7) + 8 * (idx / (nbx * nby));
    if (zc * TA_STEPS >= g.n_steps) return;
    ta_block(gtile, cover, grad_rot, g, zc, idx % (nbx * nby));
}

// --------------------------------------------------------------------------------------------
// Regulariser gradient.  L1Regularizer (adorym/regularizers.py:30-46): alpha_c * mean|x_c|;
// TVRegularizer (regularizers.py:95-110 -> util.py:1427-1440): gamma * sum_axes sum|roll(a,1)-a| / V.
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ float sgn(float v) { return (float)((v > 0.f) - (v < 0.f)); }

// SET: g = regulariser gradient (replaces a zero fill + accumulate when the gradient buffer is being initialised).
// One thread per voxel (both channels, float2), threads along z (the fastest axis), blockIdx = (x, y): no integer divisions,
// every load an 8-byte coalesced access (the y / x neighbours are whole rows one plane / one row away).
// MODE 0: g += gradient; 1: g = gradient; 2: value only (g untouched)
template <int SET>
__global__ __launch_bounds__(256) void reg_grad_kernel(const float2* __restrict__ x, float2* __restrict__ g, int Y, int X, int Z,
                                                       float a_d, float a_b, float gamma, float* reg_partial) {
    const float invV = 1.0f / (float)((size_t)Y * X * Z);
    float val = 0.f;
    const int xx = blockIdx.x, y = blockIdx.y;
    const size_t sx = (size_t)Z, sy = (size_t)Z * X;
    const size_t row = (size_t)y * sy + (size_t)xx * sx;
    const size_t row_xm = (size_t)y * sy + (size_t)((xx + X - 1) % X) * sx, row_xp = (size_t)y * sy + (size_t)((xx + 1) % X) * sx;
    const size_t row_ym = (size_t)((y + Y - 1) % Y) * sy + (size_t)xx * sx, row_yp = (size_t)((y + 1) % Y) * sy + (size_t)xx * sx;
    for (int z = threadIdx.x; z < Z; z += blockDim.x) {
        const float2 v = x[row + z];
        float2 gr = make_float2(0.f, 0.f);
        if (a_d != 0.f) { gr.x += a_d * sgn(v.x) * invV; val += a_d * fabsf(v.x) * invV; }
        if (a_b != 0.f) { gr.y += a_b * sgn(v.y) * invV; val += a_b * fabsf(v.y) * invV; }
        if (gamma != 0.f) {
            const float2 zm = x[row + (z + Z - 1) % Z], zp = x[row + (z + 1) % Z];
            const float2 xm = x[row_xm + z], xp = x[row_xp + z];
            const float2 ym = x[row_ym + z], yp = x[row_yp + z];
            gr.x += gamma * invV * ((sgn(v.x - zp.x) - sgn(zm.x - v.x)) + (sgn(v.x - xp.x) - sgn(xm.x - v.x)) + (sgn(v.x - yp.x) - sgn(ym.x - v.x)));
            gr.y += gamma * invV * ((sgn(v.y - zp.y) - sgn(zm.y - v.y)) + (sgn(v.y - xp.y) - sgn(xm.y - v.y)) + (sgn(v.y - yp.y) - sgn(ym.y - v.y)));
            val += gamma * invV * (fabsf(zm.x - v.x) + fabsf(xm.x - v.x) + fabsf(ym.x - v.x));
            val += gamma * invV * (fabsf(zm.y - v.y) + fabsf(xm.y - v.y) + fabsf(ym.y - v.y));
        }
        if (SET == 1) g[row + z] = gr;
        else if (SET == 0) { float2 o = g[row + z]; o.x += gr.x; o.y += gr.y; g[row + z] = o; }
    }
    if (reg_partial) {
        // one partial per (y, x) row, summed in a fixed order by reg_value_reduce_kernel: 65 536 atomics on ONE address
        // serialise at the memory side and made this kernel 6x slower (0.84 instead of 0.13 ms at 256^3)
        __shared__ float red[4];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) val += __shfl_down(val, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = val;
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.f;
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
            reg_partial[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = t;
        }
    }
}

// The same gradient for the flat fp32 elements [lo, hi) of the object only (a rank's shard of a sharded update), ADDED where the
// element lies in [add_lo, add_hi) and WRITTEN elsewhere: the footprint-restricted exchange reduces the data term over the planes
// the global batch touched and leaves everything else of the gradient buffer undefined -- the regulariser term, the same on every
// rank, is put in afterwards by the owner with R-fold weights.  blockIdx.y counts planes from y0.
__global__ __launch_bounds__(256) void reg_grad_range_kernel(const float2* __restrict__ x, float* __restrict__ g, int Y, int X, int Z,
                                                             float a_d, float a_b, float gamma, int y0, size_t lo, size_t hi, size_t add_lo,
                                                             size_t add_hi) {
    const float invV = 1.0f / (float)((size_t)Y * X * Z);
    const int xx = blockIdx.x, y = y0 + blockIdx.y;
    const size_t sx = (size_t)Z, sy = (size_t)Z * X;
    const size_t row = (size_t)y * sy + (size_t)xx * sx;
    const size_t row_xm = (size_t)y * sy + (size_t)((xx + X - 1) % X) * sx, row_xp = (size_t)y * sy + (size_t)((xx + 1) % X) * sx;
    const size_t row_ym = (size_t)((y + Y - 1) % Y) * sy + (size_t)xx * sx, row_yp = (size_t)((y + 1) % Y) * sy + (size_t)xx * sx;
    for (int z = threadIdx.x; z < Z; z += blockDim.x) {
        const size_t e = 2 * (row + z);
        if (e + 1 < lo || e >= hi) continue;
        const float2 v = x[row + z];
        float2 gr = make_float2(0.f, 0.f);
        if (a_d != 0.f) gr.x += a_d * sgn(v.x) * invV;
        if (a_b != 0.f) gr.y += a_b * sgn(v.y) * invV;
        if (gamma != 0.f) {
            const float2 zm = x[row + (z + Z - 1) % Z], zp = x[row + (z + 1) % Z];
            const float2 xm = x[row_xm + z], xp = x[row_xp + z];
            const float2 ym = x[row_ym + z], yp = x[row_yp + z];
            gr.x += gamma * invV * ((sgn(v.x - zp.x) - sgn(zm.x - v.x)) + (sgn(v.x - xp.x) - sgn(xm.x - v.x)) + (sgn(v.x - yp.x) - sgn(ym.x - v.x)));
            gr.y += gamma * invV * ((sgn(v.y - zp.y) - sgn(zm.y - v.y)) + (sgn(v.y - xp.y) - sgn(xm.y - v.y)) + (sgn(v.y - yp.y) - sgn(ym.y - v.y)));
        }
        if (e >= lo && e < hi) g[e] = (e >= add_lo && e < add_hi) ? g[e] + gr.x : gr.x;
        if (e + 1 >= lo && e + 1 < hi) g[e + 1] = (e + 1 >= add_lo && e + 1 < add_hi) ? g[e + 1] + gr.y : gr.y;
    }
}

// out[i] += sum_b part[b][i], b ascending: the per-position probe gradients of a launch, summed deterministically
__global__ __launch_bounds__(256) void probe_grad_reduce_kernel(const float2* __restrict__ part, int batch, size_t stride, size_t n,
                                                                float2* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float2 acc = out[i];
    for (int b = 0; b < batch; ++b) {
        const float2 v = part[(size_t)b * stride + i];
        acc.x += v.x;
        acc.y += v.y;
    }
    out[i] = acc;
}
// Large batches (a dense 2-D scan taken as one minibatch: 2704 positions): one thread per element walking the whole batch is a
// chain of `batch` dependent adds on 21 workgroups.  Two levels instead: chunk c of PGR_CHUNK slots is summed in slot order into
// its first slot (every thread touches element i of every slot only: in place), then the chunk sums are added in chunk order.
// A fixed order either way: bit-reproducible.
#define PGR_CHUNK 32
__global__ __launch_bounds__(256) void probe_grad_reduce_chunks_kernel(float2* __restrict__ part, int batch, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int b0 = blockIdx.y * PGR_CHUNK, b1 = min(b0 + PGR_CHUNK, batch);
    float2 acc = part[(size_t)b0 * n + i];
    for (int b = b0 + 1; b < b1; ++b) {
        const float2 v = part[(size_t)b * n + i];
        acc.x += v.x;
        acc.y += v.y;
    }
    part[(size_t)b0 * n + i] = acc;
}
hipError_t probe_grad_reduce_large(float2* part, int batch, size_t n, float2* out, hipStream_t st) {
    const int chunks = (batch + PGR_CHUNK - 1) / PGR_CHUNK;
    hipLaunchKernelGGL(probe_grad_reduce_chunks_kernel, dim3((unsigned)((n + 255) / 256), chunks), dim3(256), 0, st, part, batch, n);
    // the chunk sums sit PGR_CHUNK slots apart
    hipLaunchKernelGGL(probe_grad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float2*)part, chunks, (size_t)PGR_CHUNK * n, n, out);
    return hipGetLastError();
}
hipError_t probe_grad_reduce(const float2* part, int batch, size_t n, float2* out, hipStream_t st) {
    hipLaunchKernelGGL(probe_grad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, part, batch, n, n, out);
    return hipGetLastError();
}

// *out += sum(partial[0..n)) in a fixed order (one block)
__global__ __launch_bounds__(1024) void reg_value_reduce_kernel(const float* __restrict__ partial, int n, float* out) {
    __shared__ float red[16];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) acc += partial[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += red[w];
        *out += t;
    }
}

// real_imag branches (adorym/regularizers.py:38-45, 105-110): the regularised quantities are u = re^2 + im^2 (TV),
// |o| = sqrt(u) (L1, about its mean) and phi = atan2(im, re) (TV and L1); chain rule back to (re, im).
// stats[0] = sum |o|, stats[1] = sum sgn(|o| - mean|o|) (filled by the two pre-passes when alpha_d != 0).
// wgt (reweighted L1, adorym/regularizers.py:73-82): the sign sum of pass 1 is weighted with wm = w_re^2 + w_im^2 per voxel.
__global__ __launch_bounds__(256) void ri_stats_kernel(const float2* __restrict__ x, size_t V, float* stats, int pass,
                                                       const float2* __restrict__ wgt = nullptr) {
    float acc = 0.f;
    const float mean = pass ? stats[0] / (float)V : 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += (size_t)gridDim.x * blockDim.x) {
        const float2 o = x[i];
        const float om = sqrtf(o.x * o.x + o.y * o.y);
        float wm = 1.f;
        if (wgt && pass) { const float2 w = wgt[i]; wm = w.x * w.x + w.y * w.y; }
        acc += pass ? wm * sgn(om - mean) : om;
    }
    __shared__ float red[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(stats + pass, red[0] + red[1] + red[2] + red[3]);
}

__device__ __forceinline__ void ri_up(const float2* __restrict__ x, size_t i, float& u, float& ph) {
    const float2 o = x[i];
    u = o.x * o.x + o.y * o.y;
    ph = atan2f(o.y, o.x);
}

__global__ __launch_bounds__(256) void reg_grad_ri_kernel(const float2* __restrict__ x, float2* __restrict__ g, int Y, int X, int Z,
                                                          float a_d, float a_b, float gamma, const float* __restrict__ stats,
                                                          float* reg_value, int set) {
    const size_t V = (size_t)Y * X * Z;
    const float invV = 1.0f / (float)V;
    float val = 0.f;
    const float mean_om = (a_d != 0.f) ? stats[0] * invV : 0.f;
    const float mean_sg = (a_d != 0.f) ? stats[1] * invV : 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += (size_t)gridDim.x * blockDim.x) {
        size_t vox = i;
        const int z = (int)(vox % Z);
        vox /= Z;
        const int xx = (int)(vox % X);
        const int y = (int)(vox / X);
        const float2 o = x[i];
        const float u = o.x * o.x + o.y * o.y;
        const float ph = atan2f(o.y, o.x);
        float gr = 0.f, gi = 0.f;
        if (a_d != 0.f) {
            const float om = sqrtf(u);
            const float dev = om - mean_om;
            const float gom = a_d * (sgn(dev) - mean_sg) * invV;
            gr += gom * o.x / om;
            gi += gom * o.y / om;
            val += a_d * fabsf(dev) * invV;
        }
        if (a_b != 0.f) {
            const float gph = a_b * sgn(ph) * invV;
            gr += -gph * o.y / u;
            gi += gph * o.x / u;
            val += a_b * fabsf(ph) * invV;
        }
        if (gamma != 0.f) {
            const size_t sz = 1, sx = (size_t)Z, sy = (size_t)Z * X;
            float um, pm, up, pp, gu = 0.f, gp = 0.f;
            ri_up(x, i - (size_t)z * sz + (size_t)((z + Z - 1) % Z) * sz, um, pm);
            ri_up(x, i - (size_t)z * sz + (size_t)((z + 1) % Z) * sz, up, pp);
            gu += sgn(u - up) - sgn(um - u); gp += sgn(ph - pp) - sgn(pm - ph);
            val += gamma * invV * (fabsf(um - u) + fabsf(pm - ph));
            ri_up(x, i - (size_t)xx * sx + (size_t)((xx + X - 1) % X) * sx, um, pm);
            ri_up(x, i - (size_t)xx * sx + (size_t)((xx + 1) % X) * sx, up, pp);
            gu += sgn(u - up) - sgn(um - u); gp += sgn(ph - pp) - sgn(pm - ph);
            val += gamma * invV * (fabsf(um - u) + fabsf(pm - ph));
            ri_up(x, i - (size_t)y * sy + (size_t)((y + Y - 1) % Y) * sy, um, pm);
            ri_up(x, i - (size_t)y * sy + (size_t)((y + 1) % Y) * sy, up, pp);
            gu += sgn(u - up) - sgn(um - u); gp += sgn(ph - pp) - sgn(pm - ph);
            val += gamma * invV * (fabsf(um - u) + fabsf(pm - ph));
            gr += gamma * invV * (gu * 2.f * o.x - gp * o.y / u);
            gi += gamma * invV * (gu * 2.f * o.y + gp * o.x / u);
        }
        float2 gv = set ? make_float2(0.f, 0.f) : g[i];
        gv.x += gr;
        gv.y += gi;
        g[i] = gv;
    }
    if (reg_value) {
        __shared__ float red[4];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) val += __shfl_down(val, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = val;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(reg_value, red[0] + red[1] + red[2] + red[3]);
    }
}

// Reweighted L1 for complex-transmission unknowns (adorym/regularizers.py:73-82): with wm = w_re^2 + w_im^2 (constants),
//   alpha_d * mean(wm * | |o| - mean|o| |) + alpha_b * mean(wm * |atan2(im, re)|);   stats as in reg_grad_ri_kernel, the
// sign sum weighted.  g += gradient w.r.t. (re, im); value added to *reg_value.
__global__ __launch_bounds__(256) void reg_grad_ri_weighted_kernel(const float2* __restrict__ x, const float2* __restrict__ wgt,
                                                                   float2* __restrict__ g, size_t V, float a_d, float a_b,
                                                                   const float* __restrict__ stats, float* reg_value) {
    const float invV = 1.0f / (float)V;
    float val = 0.f;
    const float mean_om = (a_d != 0.f) ? stats[0] * invV : 0.f;
    const float mean_sg = (a_d != 0.f) ? stats[1] * invV : 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += (size_t)gridDim.x * blockDim.x) {
        const float2 o = x[i];
        const float2 w = wgt[i];
        const float wm = w.x * w.x + w.y * w.y;
        const float u = o.x * o.x + o.y * o.y;
        float gr = 0.f, gi = 0.f;
        if (a_d != 0.f) {
            const float om = sqrtf(u);
            const float dev = om - mean_om;
            const float gom = a_d * (wm * sgn(dev) - mean_sg) * invV;
            gr += gom * o.x / om;
            gi += gom * o.y / om;
            val += a_d * wm * fabsf(dev) * invV;
        }
        if (a_b != 0.f) {
            const float ph = atan2f(o.y, o.x);
            const float gph = a_b * wm * sgn(ph) * invV;
            gr += -gph * o.y / u;
            gi += gph * o.x / u;
            val += a_b * wm * fabsf(ph) * invV;
        }
        float2 gv = g[i];
        gv.x += gr;
        gv.y += gi;
        g[i] = gv;
    }
    if (reg_value) {
        __shared__ float red[4];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) val += __shfl_down(val, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = val;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(reg_value, red[0] + red[1] + red[2] + red[3]);
    }
}

// x[r][c] -= mean_r x[r][c]   (adorym/optimizers.py:1046-1048), one workgroup
// body of center_rows_kernel for one workgroup of 256 threads (also the tail of small_adam_kernel: same sums, same bits)
__device__ __forceinline__ void center_rows_block(float* __restrict__ x, size_t n_rows, int n_cols, float* red, float* mean) {
    for (int c = 0; c < n_cols; ++c) {
        float acc = 0.f;
        for (size_t r = threadIdx.x; r < n_rows; r += blockDim.x) acc += x[r * n_cols + c];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) *mean = (red[0] + red[1] + red[2] + red[3]) / (float)n_rows;
        __syncthreads();
        const float mu = *mean;
        for (size_t r = threadIdx.x; r < n_rows; r += blockDim.x) x[r * n_cols + c] -= mu;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void center_rows_kernel(float* __restrict__ x, size_t n_rows, int n_cols) {
    __shared__ float red[4];
    __shared__ float mean;
    center_rows_block(x, n_rows, n_cols, red, &mean);
}

// --------------------------------------------------------------------------------------------
// Fused optimiser steps.  AdamOptimizer.apply_gradient (adorym/optimizers.py:309-318):
//   m = b1*m; m = m + (1-b1)*g; v = b2*v; v = v + (1-b2)*g^2;
//   x = x - step*(m/q1) / (sqrt(v/q2) + eps),  q = 1 - b^(i_batch+1)
// followed by the constraints of adorym/ptychography.py:1135-1158 and the support mask
// (adorym/array_ops.py:239-251).  Same operation order as the reference, fp32.
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ x, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, size_t lo, size_t hi, AdamScalars a) {
    for (size_t i = lo + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += (size_t)gridDim.x * blockDim.x) {
        float mv, vv;
        const float xn = adam_value(x[i], g[i], m[i], v[i], a, i, mv, vv);
        m[i] = mv;
        v[i] = vv;
        x[i] = xn;
    }
}

// The small optimisable parameters of a minibatch (probe modes, sub-pixel position corrections, propagation distances, affine
// matrices: KBs each) updated in ONE launch, one workgroup per array: Adam with adam_kernel's arithmetic, then -- per array, as
// asked -- the drift guard of the position corrections (center_rows_kernel's sums), the pin of the first entries to fixed values,
// and the zero fill of the gradient accumulator for the next minibatch.  These paths are launch-bound (4-5 us per launch whatever
// its size): config-5 shape 24 -> 17 launches per minibatch, config-1 shape 17 -> 10.
// An array without a drift guard is element-wise, so it is spread over ceil(n / chunk) workgroups (five probe modes of
// 64 x 64 took 77 us in ONE workgroup, 40 % of a config-1-shape minibatch); an array with one stays in a single workgroup
// (its column means need every element).  first_block[i]: the first workgroup of array i; first_block[count] = grid size.
// Elements per workgroup of an element-wise array: 2048 for the genuinely small ones (a few KB); anything larger -- five 64 x 64
// probe modes, the 2-D object itself, which the driver adds to this launch on one rank -- gets 256: one element per thread, like adam_kernel
// (8 dependent iterations per thread made the launch 8.4 us for a 512 x 512 x 2 object against adam_kernel's 5.6 us).
static inline int small_chunk(uint64_t n) { return n > 4096 ? 256 : 2048; }
struct SmallParams { adm_small_param p[ADM_SMALL_PARAMS_MAX]; int first_block[ADM_SMALL_PARAMS_MAX + 1]; int chunk[ADM_SMALL_PARAMS_MAX]; int count; };

// An array with a drift guard, held in registers: thread t owns the rows t, t + 256, ... (all NC columns of each), so that the
// Adam step, the column sums (in center_rows_block's order: rows ascending per thread, the same shuffle tree, the same four partial
// sums -- the same bits), the subtraction and the pin need ONE round of loads and ONE of stores.  The generic path below makes
// ~35 dependent trips to memory for the 1 400 x 2 position corrections of a config-1 minibatch (17 us, the longest block of the launch).
#define SMALL_WHOLE_RPT 8
template <int NC>
__device__ __forceinline__ void small_adam_whole(const adm_small_param& q, const AdamScalars& a, float* red, float* mean) {
    const size_t n_rows = q.n / NC;
    float xv[SMALL_WHOLE_RPT][NC];
#pragma unroll
    for (int j = 0; j < SMALL_WHOLE_RPT; ++j) {
        const size_t r = threadIdx.x + (size_t)256 * j;
        if (r < n_rows) {
            float gv[NC], mv[NC], vv[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) { const size_t i = r * NC + c; xv[j][c] = q.x[i]; gv[c] = q.g[i]; mv[c] = q.m[i]; vv[c] = q.v[i]; }
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const size_t i = r * NC + c;
                float mo, vo;
                xv[j][c] = adam_value(xv[j][c], gv[c], mv[c], vv[c], a, i, mo, vo);
                q.m[i] = mo;
                q.v[i] = vo;
                if (q.zero_grad) q.g[i] = 0.f;
            }
        } else {
#pragma unroll
            for (int c = 0; c < NC; ++c) xv[j][c] = 0.f;
        }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < SMALL_WHOLE_RPT; ++j)
            if (threadIdx.x + (size_t)256 * j < n_rows) acc += xv[j][c];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) *mean = (red[0] + red[1] + red[2] + red[3]) / (float)n_rows;
        __syncthreads();
        const float mu = *mean;
#pragma unroll
        for (int j = 0; j < SMALL_WHOLE_RPT; ++j) xv[j][c] -= mu;
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < SMALL_WHOLE_RPT; ++j) {
        const size_t r = threadIdx.x + (size_t)256 * j;
        if (r < n_rows) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const size_t i = r * NC + c;
                q.x[i] = (q.pin && i < q.pin_n) ? q.pin[i] : xv[j][c];
            }
        }
    }
}

__global__ __launch_bounds__(256) void small_adam_kernel(SmallParams sp, AdamScalars a) {
    __shared__ float red[4];
    __shared__ float mean;
    int k = 0;
    while (k + 1 < sp.count && (int)blockIdx.x >= sp.first_block[k + 1]) ++k;
    const adm_small_param q = sp.p[k];
    const bool whole = q.center_cols > 0;
    if (whole && q.center_cols <= 2 && q.n / (size_t)q.center_cols <= (size_t)256 * SMALL_WHOLE_RPT) {
        a.step = (float)q.step_size;
        if (q.center_cols == 1) small_adam_whole<1>(q, a, red, &mean);
        else small_adam_whole<2>(q, a, red, &mean);
        return;
    }
    const size_t chunk = (size_t)sp.chunk[k];
    const size_t lo = whole ? 0 : (size_t)(blockIdx.x - sp.first_block[k]) * chunk;
    const size_t hi = whole ? q.n : (lo + chunk < q.n ? lo + chunk : q.n);
    a.step = (float)q.step_size;
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        float mv, vv;
        float xn = adam_value(q.x[i], q.g[i], q.m[i], q.v[i], a, i, mv, vv);
        q.m[i] = mv;
        q.v[i] = vv;
        if (!whole && q.pin && i < q.pin_n) xn = q.pin[i];       // (with a drift guard the pin follows the re-centring, below)
        q.x[i] = xn;
        if (q.zero_grad) q.g[i] = 0.f;
    }
    if (!whole) return;
    __syncthreads();
    center_rows_block(q.x, q.n / (size_t)q.center_cols, q.center_cols, red, &mean);
    if (q.pin) {
        for (size_t i = threadIdx.x; i < q.pin_n; i += blockDim.x) q.x[i] = q.pin[i];
    }
}

__global__ __launch_bounds__(256) void gd_kernel(float* __restrict__ x, const float* __restrict__ g, size_t lo, size_t hi,
                                                 float step, int flags, const float* __restrict__ mask) {
    for (size_t i = lo + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += (size_t)gridDim.x * blockDim.x)
        x[i] = gd_value(x[i], g[i], step, i, flags, mask);
}

__global__ __launch_bounds__(256) void momentum_kernel(float* __restrict__ x, const float* __restrict__ g, float* __restrict__ v,
                                                       size_t lo, size_t hi, float step, float gamma, int flags,
                                                       const float* __restrict__ mask) {
    for (size_t i = lo + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += (size_t)gridDim.x * blockDim.x) {
        float vv;
        x[i] = momentum_value(x[i], g[i], v[i], step, gamma, i, flags, mask, vv);
        v[i] = vv;
    }
}

// reweighted L1: block partials of (max, sum) -> final scalars -> weights
__global__ __launch_bounds__(256) void rwl1_partial_kernel(const float* __restrict__ x, size_t n, float* __restrict__ part) {
    float mx = -3.4e38f;
    double sm = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        mx = fmaxf(mx, v);
        sm += (double)v;
    }
    __shared__ float smx[256];
    __shared__ double ssm[256];
    smx[threadIdx.x] = mx; ssm[threadIdx.x] = sm;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { smx[threadIdx.x] = fmaxf(smx[threadIdx.x], smx[threadIdx.x + o]); ssm[threadIdx.x] += ssm[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = smx[0]; part[2 * blockIdx.x + 1] = (float)(ssm[0] / (double)n); }
}
__global__ void rwl1_final_kernel(float* part, int nblocks) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float mx = -3.4e38f; double mean = 0.0;
        for (int b = 0; b < nblocks; ++b) { mx = fmaxf(mx, part[2 * b]); mean += (double)part[2 * b + 1]; }
        part[2 * nblocks] = mx; part[2 * nblocks + 1] = (float)mean;
    }
}
__global__ __launch_bounds__(256) void rwl1_weight_kernel(const float* __restrict__ x, float* __restrict__ w, size_t n,
                                                          const float* __restrict__ scal) {
    const float mx = scal[0], off = 1e-4f * scal[1];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        w[i] = mx / (fabsf(x[i]) + off);
}
__global__ __launch_bounds__(256) void reg_grad_weighted_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                float* __restrict__ g, size_t n, float a_d, float a_b,
                                                                float invV, float* reg_value) {
    float val = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float al = (i & 1) ? a_b : a_d;
        const float v = x[i], wi = w[i];
        g[i] += al * wi * sgn(v) * invV;
        val += al * wi * fabsf(v) * invV;
    }
    if (reg_value) {
        __shared__ float red[4];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) val += __shfl_down(val, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = val;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(reg_value, red[0] + red[1] + red[2] + red[3]);
    }
}

__global__ __launch_bounds__(256) void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float a, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] += a * x[i];
}

static inline int stream_grid(size_t n) {
    size_t b = (n + 255) / 256;
    return (int)(b > 4096 ? 4096 : (b ? b : 1));
}

}  // namespace adm

using namespace adm;

extern "C" int adm_rotate_fwd(adm_plan* plan, const float* obj, const uint16_t* coords, float* obj_rot, int y_lo, int y_hi) {
    if (!plan || !obj || !obj_rot) return fail(ADM_ERR_INVALID, "adm_rotate_fwd: null argument");
    const adm_plan_desc& d = plan->d;
    if (y_lo < 0 || y_hi > d.obj_y || y_lo > y_hi) return fail(ADM_ERR_INVALID, "adm_rotate_fwd: bad y range");
    if (y_lo == y_hi) return ADM_OK;
    RotGeom g{d.obj_y, d.obj_x, d.obj_z, plan->Yp, plan->Xp, d.pad_y0, d.pad_x0};
    // y planes per block: 32 (the (x', z') sampling weights are decoded once per block and reused) -- unless that leaves the chip
    // empty: a 64^3 object has 16 patches x 2 chunks = 32 blocks that each walk 32 planes one dependent gather after the other
    // (17 us per launch, 16 launches per config-2 update); then fewer planes per block, down to one
    int y_chunk = 32;
    const int n_patch = ((d.obj_x + 15) / 16) * ((d.obj_z + 15) / 16);
    while (y_chunk > 1 && n_patch * ((y_hi - y_lo + y_chunk - 1) / y_chunk) < 512) y_chunk >>= 1;
    dim3 grid((d.obj_x + 15) / 16, (d.obj_z + 15) / 16, (y_hi - y_lo + y_chunk - 1) / y_chunk);
    // cache mode 2: only the transmissions are written (half the stores); obj_rot then merely names the image the cache holds
    float2* rot_out = (plan->trans_dev && plan->trans_only) ? (float2*)nullptr : (float2*)obj_rot;
    if (!coords && d.obj_z < 16)
        hipLaunchKernelGGL(identity_fwd_kernel, dim3(stream_grid((size_t)(y_hi - y_lo) * d.obj_x * d.obj_z)), dim3(256), 0, plan->ctx->stream,
                           (const float2*)obj, rot_out, plan->trans_dev, d.k1, (float)d.sign_convention, g, y_lo, y_hi);
    else
        hipLaunchKernelGGL(rotate_fwd_kernel, grid, dim3(256), 0, plan->ctx->stream, (const float2*)obj, coords, rot_out, plan->trans_dev,
                           d.k1, (float)d.sign_convention, g, y_lo, y_hi, y_chunk);
    ADM_HIP(hipGetLastError());
    if (plan->trans_dev) plan->trans_src = obj_rot;
    return ADM_OK;
}

extern "C" int adm_rotate_fwd_stack(adm_plan* plan, const float* obj, const void* tables_dev, int n_tables, float* obj_rot) {
    if (!plan || !obj || !tables_dev || !obj_rot) return fail(ADM_ERR_INVALID, "adm_rotate_fwd_stack: null argument");
    const adm_plan_desc& d = plan->d;
    if (n_tables < 1 || d.obj_y % n_tables) return fail(ADM_ERR_INVALID, "adm_rotate_fwd_stack: the plan's y extent is not n_tables blocks");
    const int Yb = d.obj_y / n_tables;
    RotGeom g{d.obj_y, d.obj_x, d.obj_z, plan->Yp, plan->Xp, d.pad_y0, d.pad_x0};
    int y_chunk = 32;
    while (y_chunk > 1 && Yb % y_chunk) y_chunk >>= 1;
    const int n_patch = ((d.obj_x + 15) / 16) * ((d.obj_z + 15) / 16);
    while (y_chunk > 1 && n_patch * (d.obj_y / y_chunk) < 512) y_chunk >>= 1;
    dim3 grid((d.obj_x + 15) / 16, (d.obj_z + 15) / 16, d.obj_y / y_chunk);
    float2* rot_out = (plan->trans_dev && plan->trans_only) ? (float2*)nullptr : (float2*)obj_rot;
    hipLaunchKernelGGL(rotate_fwd_stack_kernel, grid, dim3(256), 0, plan->ctx->stream, (const float2*)obj,
                       (const uint16_t* const*)tables_dev, Yb, rot_out, plan->trans_dev, d.k1, (float)d.sign_convention, g, y_chunk);
    ADM_HIP(hipGetLastError());
    if (plan->trans_dev) plan->trans_src = obj_rot;
    return ADM_OK;
}

static int transmission_fill(adm_plan* plan, const float* obj_rot, int row_lo, int row_hi) {
    const adm_plan_desc& d = plan->d;
    const size_t n = (size_t)(row_hi - row_lo) * plan->Xp * d.obj_z;
    hipLaunchKernelGGL(transmission_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 8192)), dim3(256), 0, plan->ctx->stream,
                       (const float2*)obj_rot, plan->trans_dev, d.k1, (float)d.sign_convention, d.obj_z, plan->Yp, plan->Xp, row_lo,
                       row_hi);
    ADM_HIP(hipGetLastError());
    return ADM_OK;
}

extern "C" int adm_plan_set_transmission_cache(adm_plan* plan, int on) {
    if (!plan) return fail(ADM_ERR_INVALID, "adm_plan_set_transmission_cache: null plan");
    const adm_plan_desc& d = plan->d;
    if (!on) {
        if (plan->trans_dev) {
            ADM_HIP(hipStreamSynchronize(plan->ctx->main_stream));
            ADM_HIP(hipFree(plan->trans_dev));
        }
        plan->trans_dev = nullptr;
        plan->trans_src = nullptr;
        plan->trans_only = false;
        return ADM_OK;
    }
    if (d.unknown_type != 0 || d.binning != 1)      // (validated first: a refused call leaves the plan as it was)
        return fail(ADM_ERR_UNSUPPORTED, "adm_plan_set_transmission_cache: needs unknown_type delta_beta and binning 1");
    if (!plan->trans_dev) {
        float2* t = nullptr;
        ADM_HIP(hipMalloc((void**)&t, (size_t)d.obj_z * plan->Yp * plan->Xp * sizeof(float2)));
        plan->trans_dev = t;
        plan->trans_src = nullptr;
        const int rc = transmission_fill(plan, nullptr, 0, plan->Yp);      // vacuum everywhere: the pads keep it forever
        if (rc) {
            (void)hipFree(t);
            plan->trans_dev = nullptr;
            plan->trans_only = false;
            return rc;
        }
    }
    plan->trans_only = (on == 2);
    return ADM_OK;
}

extern "C" int adm_transmission_refresh(adm_plan* plan, const float* obj_rot, int y_lo, int y_hi) {
    if (!plan || !obj_rot) return fail(ADM_ERR_INVALID, "adm_transmission_refresh: null argument");
    if (!plan->trans_dev) return fail(ADM_ERR_INVALID, "adm_transmission_refresh: the plan has no transmission cache");
    const adm_plan_desc& d = plan->d;
    if (y_lo < 0 || y_hi > d.obj_y || y_lo > y_hi) return fail(ADM_ERR_INVALID, "adm_transmission_refresh: bad y range");
    plan->trans_src = obj_rot;
    if (y_lo == y_hi) return ADM_OK;
    return transmission_fill(plan, obj_rot, d.pad_y0 + y_lo, d.pad_y0 + y_hi);
}

extern "C" int adm_rotate_adj(adm_plan* plan, const float* grad_rot, const uint16_t* coords, float* grad_obj, int y_lo, int y_hi) {
    if (!plan || !grad_rot || !grad_obj) return fail(ADM_ERR_INVALID, "adm_rotate_adj: null argument");
    const adm_plan_desc& d = plan->d;
    if (y_lo < 0 || y_hi > d.obj_y || y_lo > y_hi) return fail(ADM_ERR_INVALID, "adm_rotate_adj: bad y range");
    if (y_lo == y_hi) return ADM_OK;
    RotGeom g{d.obj_y, d.obj_x, d.obj_z, plan->Yp, plan->Xp, d.pad_y0, d.pad_x0};
    const int y_chunk = 32;
    dim3 grid((d.obj_x + 15) / 16, (d.obj_z + 15) / 16, (y_hi - y_lo + y_chunk - 1) / y_chunk);
    if (!coords && d.obj_z < 16)
        hipLaunchKernelGGL(identity_adj_kernel, dim3(stream_grid((size_t)(y_hi - y_lo) * d.obj_x * d.obj_z)), dim3(256), 0, plan->ctx->stream,
                           (const float2*)grad_rot, (float2*)grad_obj, g, y_lo, y_hi);
    else
        hipLaunchKernelGGL(rotate_adj_kernel, grid, dim3(256), 0, plan->ctx->stream, (const float2*)grad_rot, coords, grad_obj, g, y_lo,
                           y_hi, y_chunk);
    ADM_HIP(hipGetLastError());
    return ADM_OK;
}

extern "C" int adm_rotate_adj_csr(adm_plan* plan, const float* grad_rot, const int32_t* csr_ptr, const int32_t* csr_src,
                                  const float* csr_w, float* grad_obj, int y_lo, int y_hi, int lanes_along_x) {
    if (!plan || !grad_rot || !csr_ptr || !csr_src || !csr_w || !grad_obj) return fail(ADM_ERR_INVALID, "adm_rotate_adj_csr: null argument");
    const adm_plan_desc& d = plan->d;
    if (y_lo < 0 || y_hi > d.obj_y || y_lo > y_hi) return fail(ADM_ERR_INVALID, "adm_rotate_adj_csr: bad y range");
    if (y_lo == y_hi) return ADM_OK;
    RotGeom g{d.obj_y, d.obj_x, d.obj_z, plan->Yp, plan->Xp, d.pad_y0, d.pad_x0};
    dim3 grid((d.obj_x + 15) / 16, (d.obj_z + 15) / 16, (y_hi - y_lo + 3) / 4);
    if (lanes_along_x)
        hipLaunchKernelGGL(rotate_adj_csr_kernel<true>, grid, dim3(256), 0, plan->ctx->stream, (const float2*)grad_rot, csr_ptr, csr_src,
                           csr_w, (float2*)grad_obj, g, y_lo, y_hi);
    else
        hipLaunchKernelGGL(rotate_adj_csr_kernel<false>, grid, dim3(256), 0, plan->ctx->stream, (const float2*)grad_rot, csr_ptr, csr_src,
                           csr_w, (float2*)grad_obj, g, y_lo, y_hi);
    ADM_HIP(hipGetLastError());
    return ADM_OK;
}

extern "C" int adm_rotate_adj_staged(adm_plan* plan, const float* grad_rot, const int32_t* csr_ptr, const int32_t* csr_src,
                                     const uint16_t* csr_lsrc, const float* csr_w, const int32_t* boxes, float* grad_obj, int y_lo,
                                     int y_hi) {
    if (!plan || !grad_rot || !csr_ptr || !csr_src || !csr_lsrc || !csr_w || !boxes || !grad_obj)
        return fail(ADM_ERR_INVALID, "adm_rotate_adj_staged: null argument");
    const adm_plan_desc& d = plan->d;
    if (y_lo < 0 || y_hi > d.obj_y || y_lo > y_hi) return fail(ADM_ERR_INVALID, "adm_rotate_adj_staged: bad y range");
    if (y_lo == y_hi) return ADM_OK;
    RotGeom g{d.obj_y, d.obj_x, d.obj_z, plan->Yp, plan->Xp, d.pad_y0, d.pad_x0};
    dim3 grid((d.obj_x + 15) / 16, (d.obj_z + 15) / 16, (y_hi - y_lo + 3) / 4);
    hipLaunchKernelGGL(rotate_adj_staged_kernel, grid, dim3(256), 0, plan->ctx->stream, (const float2*)grad_rot, csr_ptr, csr_src,
                       (const unsigned short*)csr_lsrc, csr_w, (const int4*)boxes, (float2*)grad_obj, g, y_lo, y_hi);
    ADM_HIP(hipGetLastError());
    return ADM_OK;
}

#ifndef ADM_STACK_MIN_BLOCKS
#define ADM_STACK_MIN_BLOCKS 2048
#endif
// g[i] = ((g[i] + a_0[i]) + a_1[i]) + ... : the terms of the R angles in angle order
__global__ __launch_bounds__(256) void stack_sum_kernel(const float2* __restrict__ part, int R, size_t n, float2* __restrict__ gobj) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float2 c = gobj[i];
    for (int r = 0; r < R; ++r) {
        const float2 a = part[(size_t)r * n + i];
        c.x += a.x;
        c.y += a.y;
    }
    gobj[i] = c;
}

extern "C" int adm_rotate_adj_staged_stack(adm_plan* plan, const float* grad_rot, const void* tables_dev, int n_tables, float* grad_obj,
                                           float* scratch, size_t scratch_bytes) {
    if (!plan || !grad_rot || !tables_dev || !grad_obj) return fail(ADM_ERR_INVALID, "adm_rotate_adj_staged_stack: null argument");
    const adm_plan_desc& d = plan->d;
    if (n_tables < 1 || d.obj_y % n_tables) return fail(ADM_ERR_INVALID, "adm_rotate_adj_staged_stack: the plan's y extent is not n_tables blocks");
    const int Yb = d.obj_y / n_tables;
    RotGeom g{d.obj_y, d.obj_x, d.obj_z, plan->Yp, plan->Xp, d.pad_y0, d.pad_x0};
    // planes per block: four (one pass over a patch's CSR entries serves four planes) unless that leaves the chip short of blocks --
    // every block walks the n_tables angles one after the other, so a 64^3 object wants all 1024 (patch, plane) pairs in flight
    int npl = 4;
    const int n_patch = ((d.obj_x + 15) / 16) * ((d.obj_z + 15) / 16);
    while (npl > 1 && n_patch * ((Yb + npl - 1) / npl) < 1024) npl >>= 1;
    const size_t n_real = (size_t)Yb * d.obj_x * d.obj_z;
    if (scratch) {
        // the angles side by side (scratch: n_tables copies of the real gradient's size), then their sum in angle order
        if (scratch_bytes < (size_t)n_tables * n_real * sizeof(float2))
            return fail(ADM_ERR_INVALID, "adm_rotate_adj_staged_stack: scratch smaller than n_tables x the real object");
        npl = 4;
        while (npl > 1 && (size_t)n_patch * ((Yb + npl - 1) / npl) * n_tables < (size_t)ADM_STACK_MIN_BLOCKS) npl >>= 1;
        dim3 gridp((d.obj_x + 15) / 16, (d.obj_z + 15) / 16, ((Yb + npl - 1) / npl) * n_tables);
        hipLaunchKernelGGL(rotate_adj_staged_stack_kernel, gridp, dim3(256), 0, plan->ctx->stream, (const float2*)grad_rot,
                           (const AdjTables*)tables_dev, n_tables, Yb, (float2*)grad_obj, g, npl, (float2*)scratch);
        hipLaunchKernelGGL(stack_sum_kernel, dim3((unsigned)((n_real + 255) / 256)), dim3(256), 0, plan->ctx->stream,
                           (const float2*)scratch, n_tables, n_real, (float2*)grad_obj);
        ADM_HIP(hipGetLastError());
        return ADM_OK;
    }
    dim3 grid((d.obj_x + 15) / 16, (d.obj_z + 15) / 16, (Yb + npl - 1) / npl);
    hipLaunchKernelGGL(rotate_adj_staged_stack_kernel, grid, dim3(256), 0, plan->ctx->stream, (const float2*)grad_rot,
                       (const AdjTables*)tables_dev, n_tables, Yb, (float2*)grad_obj, g, npl, (float2*)nullptr);
    ADM_HIP(hipGetLastError());
    return ADM_OK;
}

extern "C" int adm_tile_grad_accumulate(adm_plan* plan, void* workspace, size_t workspace_bytes, const int32_t* pos, int batch,
                                       const int32_t* pos_host, float* grad_rot) {
    return adm_tile_grad_accumulate_part(plan, workspace, workspace_bytes, pos, batch, pos_host, grad_rot, 0, 0, 0);
}

// Geometry of the overlap-add of one (part of a) batch; shared by the cover build and the accumulate so that both see the
// same window.  Returns 0 or a negative status.
static int tile_geom(adm_plan* plan, int batch, const int32_t* pos_host, int win_y_lo, int win_y_hi, int add, TileGeom& g) {
    const adm_plan_desc& d = plan->d;
    const int N = d.probe_x;
    g.Yp = plan->Yp; g.Xp = plan->Xp; g.pad_y0 = d.pad_y0; g.pad_x0 = d.pad_x0; g.Py = d.probe_y; g.Px = d.probe_x;
    g.pixel_major = plan->generic ? 1 : 0;
    g.row_elems = (int)ms_row_elems(plan);
    g.R1 = g.R2 = g.G = g.LPW = 1; g.NT = 64;
    if (!plan->generic) { g.R1 = ms_r1_for(N); g.R2 = ms_r2_for(N); g.G = g.R1 > g.R2 ? g.R1 : g.R2; g.LPW = 64 / g.G; g.NT = ms_threads_for(N); }
    g.n_steps = plan->n_steps; g.binning = d.binning; g.Z = d.obj_z;
    int ymin = pos_host[0], ymax = pos_host[0];
    for (int b = 1; b < batch; ++b) { ymin = pos_host[2 * b] < ymin ? pos_host[2 * b] : ymin; ymax = pos_host[2 * b] > ymax ? pos_host[2 * b] : ymax; }
    g.row0 = ymin + d.pad_y0;
    g.nrows = ymax - ymin + d.probe_y;
    g.add_lo = g.add_hi = 0;
    if (add) {                      // accumulate into this part's own rows (an earlier part wrote the whole batch window)
        g.add_lo = g.row0;
        g.add_hi = g.row0 + g.nrows;
    } else if (win_y_hi > win_y_lo) {   // first part: write the whole batch window (zeros where no tile of this part reaches)
        if (win_y_lo + d.pad_y0 > g.row0 || win_y_hi + d.pad_y0 < g.row0 + g.nrows)
            return fail(ADM_ERR_INVALID, "adm_tile_grad_accumulate_part: the window must contain the part's own rows");
        g.row0 = win_y_lo + d.pad_y0;
        g.nrows = win_y_hi - win_y_lo;
    }
    if (g.row0 < 0 || g.row0 + g.nrows > g.Yp) return fail(ADM_ERR_INVALID, "adm_tile_grad_accumulate: a position lies outside the padded frame");
    const size_t per = (size_t)plan->n_steps * g.row_elems;
    if ((size_t)batch * per >= 0xFFFFFFFFull) return fail(ADM_ERR_UNSUPPORTED, "adm_tile_grad_accumulate: batch too large for 32-bit tile offsets");
    return ADM_OK;
}

// what adm_tile_cover_build has built and nobody has used yet (a workspace holds one set of lists: a new build for the same
// workspace replaces its entry)
// (the key includes a fingerprint of the positions THEMSELVES: the device position buffer is reused from minibatch to minibatch, so
// its address says nothing about its contents)
static unsigned long long pos_fingerprint(const int32_t* pos_host, int batch) {
    unsigned long long h = 1469598103934665603ull;                     // FNV-1a over the (y, x) pairs
    for (int i = 0; i < 2 * batch; ++i) { h ^= (unsigned)pos_host[i]; h *= 1099511628211ull; }
    return h;
}
static void cover_key_put(adm_plan* plan, const void* ws, const void* pos, const int32_t* pos_host, int batch, const TileGeom& g) {
    int slot = 0;
    for (int i = 0; i < 4; ++i) {
        if (plan->cover_keys[i].ws == ws) { slot = i; break; }
        if (!plan->cover_keys[i].ws) slot = i;
    }
    plan->cover_keys[slot] = {ws, pos, batch, g.row0, g.nrows, pos_fingerprint(pos_host, batch)};
}
static bool cover_key_take(adm_plan* plan, const void* ws, const void* pos, const int32_t* pos_host, int batch, const TileGeom& g) {
    for (int i = 0; i < 4; ++i) {
        adm_plan::CoverKey& k = plan->cover_keys[i];
        if (k.ws == ws) {
            const bool ok = k.pos == pos && k.batch == batch && k.row0 == g.row0 && k.nrows == g.nrows && k.fp == pos_fingerprint(pos_host, batch);
            k.ws = nullptr;
            return ok;
        }
    }
    return false;
}

static int cover_build(adm_plan* plan, void* workspace, const int32_t* pos, int batch, const TileGeom& g, int b_lo = 0, int b_hi = -1) {
    if (b_hi < 0) b_hi = batch;
    char* ws = (char*)workspace;
    unsigned* cover = (unsigned*)(ws + ws_off_cover(plan, batch));
    int* overflow = (int*)(cover + (size_t)g.Yp * g.Xp * (ADM_MAXCOVER + 1));
    hipStream_t st = plan->ctx->stream;
    // (a batch of at most ADM_MAXCOVER positions cannot overflow a cover list: no flag to reset, one launch less per minibatch)
    if (b_hi - b_lo > ADM_MAXCOVER) ADM_HIP(hipMemsetAsync(overflow, 0, sizeof(int), st));
    dim3 grid((g.Xp + 31) / 32, (g.nrows + 7) / 8, 1);
    hipLaunchKernelGGL(cover_build_kernel, grid, dim3(256), 0, st, (const int2*)pos, b_lo, b_hi, g, cover, overflow);
    ADM_HIP(hipGetLastError());
    return ADM_OK;
}

// The cover lists depend on the positions only, not on the tile gradients: building them EARLY -- on the side stream, beside the
// multislice launch -- takes a launch and its dependency gap off the chain that follows the kernel.  The plan remembers what
// was built (workspace, positions, batch, window); the next adm_tile_grad_accumulate[_part] with the same arguments skips its
// own build.  The caller orders the two (adm_ctx_join before the accumulate when the build was queued on the side stream).
extern "C" int adm_tile_cover_build(adm_plan* plan, void* workspace, size_t workspace_bytes, const int32_t* pos, int batch,
                                    const int32_t* pos_host, int win_y_lo, int win_y_hi, int add) {
    if (!plan || !workspace || !pos || !pos_host) return fail(ADM_ERR_INVALID, "adm_tile_cover_build: null argument");
    if (batch <= 0) return fail(ADM_ERR_INVALID, "adm_tile_cover_build: batch must be positive");
    if (workspace_bytes < adm_plan_workspace_bytes(plan, batch)) return fail(ADM_ERR_INVALID, "adm_tile_cover_build: workspace too small");
    TileGeom g;
    int rc = tile_geom(plan, batch, pos_host, win_y_lo, win_y_hi, add, g);
    if (rc) return rc;
    rc = cover_build(plan, workspace, pos, batch, g);
    if (rc) return rc;
    cover_key_put(plan, workspace, pos, pos_host, batch, g);
    return ADM_OK;
}

extern "C" int adm_tile_grad_accumulate_part(adm_plan* plan, void* workspace, size_t workspace_bytes, const int32_t* pos, int batch,
                                            const int32_t* pos_host, float* grad_rot, int win_y_lo, int win_y_hi, int add) {
    if (!plan || !workspace || !pos || !pos_host || !grad_rot) return fail(ADM_ERR_INVALID, "adm_tile_grad_accumulate: null argument");
    if (batch <= 0) return fail(ADM_ERR_INVALID, "adm_tile_grad_accumulate: batch must be positive");
    if (workspace_bytes < adm_plan_workspace_bytes(plan, batch)) return fail(ADM_ERR_INVALID, "adm_tile_grad_accumulate: workspace too small");
    TileGeom g;
    int rc = tile_geom(plan, batch, pos_host, win_y_lo, win_y_hi, add, g);
    if (rc) return rc;
    const bool prebuilt = cover_key_take(plan, workspace, pos, pos_host, batch, g);     // one use: the position buffer may be rewritten later
    if (!prebuilt) {
        rc = cover_build(plan, workspace, pos, batch, g);
        if (rc) return rc;
    }
    char* ws = (char*)workspace;
    const float2* gtile = (const float2*)(ws + ws_off_gtile(plan, batch));
    unsigned* cover = (unsigned*)(ws + ws_off_cover(plan, batch));
    hipStream_t st = plan->ctx->stream;
    dim3 grid((g.Xp + 31) / 32, (g.nrows + 7) / 8, 1);
    const unsigned nz8 = ((plan->n_steps + TA_STEPS - 1) / TA_STEPS + 7) / 8;      // step chunks per XCD
    hipLaunchKernelGGL(tile_accumulate_kernel, dim3(8u * nz8 * grid.x * grid.y), dim3(256), 0, st, gtile, (const unsigned*)cover,
                       (float2*)grad_rot, g);
    ADM_HIP(hipGetLastError());
    return ADM_OK;
}

// A batch in which a pixel is covered by more than ADM_MAXCOVER tiles (dense 2-D scans taken as ONE minibatch,
// demos/2d_ptychography_w_probe_optimization.py: 2704 positions 5 pixels apart under a 72 x 72 probe): the overlap-add runs in
// passes over position ranges [b_lo, b_hi) of at most ADM_MAXCOVER positions each -- a range cannot overflow a list --, the first
// writing the batch's rows, the others adding to them.  Sums in position order, deterministic like the one-pass form.
extern "C" int adm_tile_grad_accumulate_range(adm_plan* plan, void* workspace, size_t workspace_bytes, const int32_t* pos, int batch,
                                             const int32_t* pos_host, float* grad_rot, int b_lo, int b_hi, int add) {
    if (!plan || !workspace || !pos || !pos_host || !grad_rot) return fail(ADM_ERR_INVALID, "adm_tile_grad_accumulate_range: null argument");
    if (batch <= 0 || b_lo < 0 || b_hi <= b_lo || b_hi > batch) return fail(ADM_ERR_INVALID, "adm_tile_grad_accumulate_range: bad range");
    if (b_hi - b_lo > ADM_MAXCOVER) return fail(ADM_ERR_INVALID, "adm_tile_grad_accumulate_range: at most 64 positions per pass");
    if (workspace_bytes < adm_plan_workspace_bytes(plan, batch)) return fail(ADM_ERR_INVALID, "adm_tile_grad_accumulate_range: workspace too small");
    TileGeom g, gb;
    int rc = tile_geom(plan, batch, pos_host, 0, 0, 0, gb);        // the window of the WHOLE batch
    if (rc) return rc;
    (void)cover_key_take(plan, workspace, pos, pos_host, batch, gb); // lists built ahead for the one-pass form are void now
    if (!add) {
        // first pass: the batch's rows start from zero, every pass then ADDS inside the rows its own positions reach (a pass over
        // the whole batch window cost the same whatever the range covered: 43 passes x 276 rows for the 2704-position demo)
        hipStream_t st0 = plan->ctx->stream;
        const size_t slice = (size_t)gb.Yp * gb.Xp;
        ADM_HIP(hipMemset2DAsync(grad_rot + 2 * (size_t)gb.row0 * gb.Xp, slice * sizeof(float2), 0, (size_t)gb.nrows * gb.Xp * sizeof(float2),
                                 (size_t)plan->d.obj_z, st0));
    }
    rc = tile_geom(plan, b_hi - b_lo, pos_host + 2 * (size_t)b_lo, 0, 0, 1, g);      // this range's own rows, accumulated
    if (rc) return rc;
    if ((size_t)batch * plan->n_steps * g.row_elems >= 0xFFFFFFFFull)
        return fail(ADM_ERR_UNSUPPORTED, "adm_tile_grad_accumulate_range: batch too large for 32-bit tile offsets");
    rc = cover_build(plan, workspace, pos, batch, g, b_lo, b_hi);
    if (rc) return rc;
    char* ws = (char*)workspace;
    const float2* gtile = (const float2*)(ws + ws_off_gtile(plan, batch));
    unsigned* cover = (unsigned*)(ws + ws_off_cover(plan, batch));
    hipStream_t st = plan->ctx->stream;
    dim3 grid((g.Xp + 31) / 32, (g.nrows + 7) / 8, 1);
    const unsigned nz8 = ((plan->n_steps + TA_STEPS - 1) / TA_STEPS + 7) / 8;
    hipLaunchKernelGGL(tile_accumulate_kernel, dim3(8u * nz8 * grid.x * grid.y), dim3(256), 0, st, gtile, (const unsigned*)cover,
                       (float2*)grad_rot, g);
    ADM_HIP(hipGetLastError());
    return ADM_OK;
}

extern "C" int adm_tile_grad_status(adm_plan* plan, void* workspace, size_t workspace_bytes, int batch, int* overflow_host) {
    if (!plan || !workspace || !overflow_host) return fail(ADM_ERR_INVALID, "adm_tile_grad_status: null argument");
    if (batch <= 0 || workspace_bytes < adm_plan_workspace_bytes(plan, batch)) return fail(ADM_ERR_INVALID, "adm_tile_grad_status: bad workspace");
    if (batch <= ADM_MAXCOVER) {        // cannot overflow; the flag is not maintained for such batches
        *overflow_host = 0;
        return ADM_OK;
    }
    char* ws = (char*)workspace;
    unsigned* cover = (unsigned*)(ws + ws_off_cover(plan, batch));
    int* overflow = (int*)(cover + (size_t)plan->Yp * plan->Xp * (ADM_MAXCOVER + 1));
    return adm_d2h(plan->ctx, overflow_host, overflow, sizeof(int));
}

static int reg_grad_impl(adm_plan* plan, const float* obj, float alpha_d, float alpha_b, float gamma, float* grad_obj,
                         float* reg_value, bool set) {
    if (!plan || !obj || (!grad_obj && !reg_value)) return fail(ADM_ERR_INVALID, "adm_reg_grad: null argument");
    if (!grad_obj && plan->d.unknown_type == 1) return fail(ADM_ERR_UNSUPPORTED, "adm_reg_grad: value-only evaluation is built for delta_beta unknowns");
    const adm_plan_desc& d = plan->d;
    const size_t n = (size_t)d.obj_y * d.obj_x * d.obj_z * 2;
    hipStream_t st = plan->ctx->stream;
    if (d.unknown_type == 1) {
        const size_t V = n / 2;
        if (alpha_d != 0.f) {
            if (!plan->reg_stats) ADM_HIP(hipMalloc((void**)&plan->reg_stats, 2 * sizeof(float)));
            ADM_HIP(hipMemsetAsync(plan->reg_stats, 0, 2 * sizeof(float), st));
            int nb = stream_grid(V);
            if (nb > 1024) nb = 1024;
            hipLaunchKernelGGL(ri_stats_kernel, dim3(nb), dim3(256), 0, st, (const float2*)obj, V, plan->reg_stats, 0, (const float2*)nullptr);
            hipLaunchKernelGGL(ri_stats_kernel, dim3(nb), dim3(256), 0, st, (const float2*)obj, V, plan->reg_stats, 1, (const float2*)nullptr);
        }
        hipLaunchKernelGGL(reg_grad_ri_kernel, dim3(stream_grid(V)), dim3(256), 0, st, (const float2*)obj, (float2*)grad_obj, d.obj_y,
                           d.obj_x, d.obj_z, alpha_d, alpha_b, gamma, (const float*)plan->reg_stats, reg_value, set ? 1 : 0);
        ADM_HIP(hipGetLastError());
        return ADM_OK;
    }
    const int nt = d.obj_z >= 192 ? 256 : (d.obj_z >= 96 ? 128 : 64);
    const dim3 grid(d.obj_x, d.obj_y);
    float* partial = nullptr;
    if (reg_value) {
        if (!plan->reg_partial) ADM_HIP(hipMalloc((void**)&plan->reg_partial, (size_t)d.obj_x * d.obj_y * sizeof(float)));
        partial = plan->reg_partial;
    }
    if (!grad_obj)
        hipLaunchKernelGGL(reg_grad_kernel<2>, grid, dim3(nt), 0, st, (const float2*)obj, (float2*)nullptr, d.obj_y, d.obj_x, d.obj_z,
                           alpha_d, alpha_b, gamma, partial);
    else if (set)
        hipLaunchKernelGGL(reg_grad_kernel<1>, grid, dim3(nt), 0, st, (const float2*)obj, (float2*)grad_obj, d.obj_y, d.obj_x, d.obj_z,
                           alpha_d, alpha_b, gamma, partial);
    else
        hipLaunchKernelGGL(reg_grad_kernel<0>, grid, dim3(nt), 0, st, (const float2*)obj, (float2*)grad_obj, d.obj_y, d.obj_x, d.obj_z,
                           alpha_d, alpha_b, gamma, partial);
    if (reg_value)
        hipLaunchKernelGGL(reg_value_reduce_kernel, dim3(1), dim3(1024), 0, st, (const float*)partial, d.obj_x * d.obj_y, reg_value);
    ADM_HIP(hipGetLastError());
    return ADM_OK;
}

extern "C" int adm_reg_grad(adm_plan* plan, const float* obj, float alpha_d, float alpha_b, float gamma, float* grad_obj,
                            float* reg_value) {
    return reg_grad_impl(plan, obj, alpha_d, alpha_b, gamma, grad_obj, reg_value, false);
}

extern "C" int adm_reg_grad_set(adm_plan* plan, const float* obj, float alpha_d, float alpha_b, float gamma, float* grad_obj,
                                float* reg_value) {
    return reg_grad_impl(plan, obj, alpha_d, alpha_b, gamma, grad_obj, reg_value, true);
}

extern "C" int adm_reg_grad_range(adm_plan* plan, const float* obj, float alpha_d, float alpha_b, float gamma, float* grad_obj,
                                  size_t lo, size_t hi, size_t add_lo, size_t add_hi) {
    if (!plan || !obj || !grad_obj) return fail(ADM_ERR_INVALID, "adm_reg_grad_range: null argument");
    const adm_plan_desc& d = plan->d;
    if (d.unknown_type != 0) return fail(ADM_ERR_UNSUPPORTED, "adm_reg_grad_range: built for delta_beta unknowns");
    const size_t n = (size_t)d.obj_y * d.obj_x * d.obj_z * 2, plane = (size_t)d.obj_x * d.obj_z * 2;
    if (hi > n) hi = n;
    if (lo >= hi) return ADM_OK;
    const int y0 = (int)(lo / plane), y1 = (int)((hi + plane - 1) / plane);
    const int nt = d.obj_z >= 192 ? 256 : (d.obj_z >= 96 ? 128 : 64);
    hipLaunchKernelGGL(reg_grad_range_kernel, dim3(d.obj_x, y1 - y0), dim3(nt), 0, plan->ctx->stream, (const float2*)obj, grad_obj, d.obj_y,
                       d.obj_x, d.obj_z, alpha_d, alpha_b, gamma, y0, lo, hi, add_lo, add_hi);
    ADM_HIP(hipGetLastError());
    return ADM_OK;
}

extern "C" int adm_center_rows(adm_ctx* ctx, float* x, size_t n_rows, int n_cols) {
    if (!ctx || !x) return fail(ADM_ERR_INVALID, "adm_center_rows: null argument");
    if (n_rows == 0 || n_cols <= 0) return ADM_OK;
    hipLaunchKernelGGL(center_rows_kernel, dim3(1), dim3(256), 0, ctx->stream, x, n_rows, n_cols);
    ADM_HIP(hipGetLastError());
    return ADM_OK;
}

extern "C" int adm_adam_step(adm_ctx* ctx, float* x, const float* g, float* m, float* v, size_t lo, size_t hi, int i_batch,
                             double step_size, double b1, double b2, double eps, int flags, const float* mask) {
    if (!ctx || !x || !g || !m || !v) return fail(ADM_ERR_INVALID, "adm_adam_step: null argument");
    if (hi <= lo) return ADM_OK;
    hipLaunchKernelGGL(adam_kernel, dim3(stream_grid(hi - lo)), dim3(256), 0, ctx->stream, x, g, m, v, lo, hi,
                       adam_scalars(i_batch, step_size, b1, b2, eps, flags, mask));
    ADM_HIP(hipGetLastError());
    return ADM_OK;
}

extern "C" int adm_adam_step_small(adm_ctx* ctx, const adm_small_param* params, int count, int i_batch, double b1, double b2, double eps) {
    if (!ctx || !params) return fail(ADM_ERR_INVALID, "adm_adam_step_small: null argument");
    if (count <= 0) return ADM_OK;
    if (count > ADM_SMALL_PARAMS_MAX) return fail(ADM_ERR_INVALID, "adm_adam_step_small: too many arrays in one call");
    SmallParams sp;
    std::memset(&sp, 0, sizeof(sp));
    int nblocks = 0;
    for (int k = 0; k < count; ++k) {
        const adm_small_param& q = params[k];
        if (!q.x || !q.g || !q.m || !q.v) return fail(ADM_ERR_INVALID, "adm_adam_step_small: null array");
        if (q.center_cols > 0 && q.n % (uint64_t)q.center_cols) return fail(ADM_ERR_INVALID, "adm_adam_step_small: n is not a multiple of center_cols");
        if (q.pin && q.pin_n > q.n) return fail(ADM_ERR_INVALID, "adm_adam_step_small: pin_n exceeds n");
        sp.p[k] = q;
        sp.first_block[k] = nblocks;
        sp.chunk[k] = small_chunk(q.n);
        nblocks += q.center_cols > 0 ? 1 : (int)((q.n + sp.chunk[k] - 1) / sp.chunk[k]);
    }
    sp.first_block[count] = nblocks;
    sp.count = count;
    if (nblocks == 0) return ADM_OK;
    hipLaunchKernelGGL(small_adam_kernel, dim3(nblocks), dim3(256), 0, ctx->stream, sp, adam_scalars(i_batch, 0.0, b1, b2, eps, 0, nullptr));
    ADM_HIP(hipGetLastError());
    return ADM_OK;
}

extern "C" int adm_gd_step(adm_ctx* ctx, float* x, const float* g, size_t lo, size_t hi, double step_size, int flags,
                           const float* mask) {
    if (!ctx || !x || !g) return fail(ADM_ERR_INVALID, "adm_gd_step: null argument");
    if (hi <= lo) return ADM_OK;
    hipLaunchKernelGGL(gd_kernel, dim3(stream_grid(hi - lo)), dim3(256), 0, ctx->stream, x, g, lo, hi, (float)step_size, flags, mask);
    ADM_HIP(hipGetLastError());
    return ADM_OK;
}

extern "C" int adm_momentum_step(adm_ctx* ctx, float* x, const float* g, float* v, size_t lo, size_t hi, double step_size,
                                 double gamma, int flags, const float* mask) {
    if (!ctx || !x || !g || !v) return fail(ADM_ERR_INVALID, "adm_momentum_step: null argument");
    if (hi <= lo) return ADM_OK;
    hipLaunchKernelGGL(momentum_kernel, dim3(stream_grid(hi - lo)), dim3(256), 0, ctx->stream, x, g, v, lo, hi, (float)step_size,
                       (float)gamma, flags, mask);
    ADM_HIP(hipGetLastError());
    return ADM_OK;
}

extern "C" int adm_rwl1_update(adm_plan* plan, const float* obj, float* weight, float* scratch) {
    if (!plan || !obj || !weight || !scratch) return fail(ADM_ERR_INVALID, "adm_rwl1_update: null argument");
    const adm_plan_desc& d = plan->d;
    const size_t n = (size_t)d.obj_y * d.obj_x * d.obj_z * 2;
    int nb = stream_grid(n);
    if (nb > 1024) nb = 1024;
    hipStream_t st = plan->ctx->stream;
    hipLaunchKernelGGL(rwl1_partial_kernel, dim3(nb), dim3(256), 0, st, obj, n, scratch);
    hipLaunchKernelGGL(rwl1_final_kernel, dim3(1), dim3(64), 0, st, scratch, nb);
    hipLaunchKernelGGL(rwl1_weight_kernel, dim3(stream_grid(n)), dim3(256), 0, st, obj, weight, n, (const float*)(scratch + 2 * nb));
    ADM_HIP(hipGetLastError());
    return ADM_OK;
}

extern "C" int adm_reg_grad_weighted(adm_plan* plan, const float* obj, const float* weight, float alpha_d, float alpha_b,
                                     float* grad_obj, float* reg_value) {
    if (!plan || !obj || !weight || !grad_obj) return fail(ADM_ERR_INVALID, "adm_reg_grad_weighted: null argument");
    const adm_plan_desc& d = plan->d;
    const size_t n = (size_t)d.obj_y * d.obj_x * d.obj_z * 2;
    if (d.unknown_type == 1) {          // real_imag (adorym/regularizers.py:73-82)
        const size_t V = n / 2;
        hipStream_t st = plan->ctx->stream;
        if (alpha_d != 0.f) {
            if (!plan->reg_stats) ADM_HIP(hipMalloc((void**)&plan->reg_stats, 2 * sizeof(float)));
            ADM_HIP(hipMemsetAsync(plan->reg_stats, 0, 2 * sizeof(float), st));
            int nb = stream_grid(V);
            if (nb > 1024) nb = 1024;
            hipLaunchKernelGGL(ri_stats_kernel, dim3(nb), dim3(256), 0, st, (const float2*)obj, V, plan->reg_stats, 0, (const float2*)weight);
            hipLaunchKernelGGL(ri_stats_kernel, dim3(nb), dim3(256), 0, st, (const float2*)obj, V, plan->reg_stats, 1, (const float2*)weight);
        }
        hipLaunchKernelGGL(reg_grad_ri_weighted_kernel, dim3(stream_grid(V)), dim3(256), 0, st, (const float2*)obj, (const float2*)weight,
                           (float2*)grad_obj, V, alpha_d, alpha_b, (const float*)plan->reg_stats, reg_value);
        ADM_HIP(hipGetLastError());
        return ADM_OK;
    }
    hipLaunchKernelGGL(reg_grad_weighted_kernel, dim3(stream_grid(n)), dim3(256), 0, plan->ctx->stream, obj, weight, grad_obj, n,
                       alpha_d, alpha_b, 1.0f / (float)(n / 2), reg_value);
    ADM_HIP(hipGetLastError());
    return ADM_OK;
}

extern "C" int adm_axpy(adm_ctx* ctx, float* y, const float* x, float a, size_t n) {
    if (!ctx || !y || !x) return fail(ADM_ERR_INVALID, "adm_axpy: null argument");
    if (!n) return ADM_OK;
    hipLaunchKernelGGL(axpy_kernel, dim3(stream_grid(n)), dim3(256), 0, ctx->stream, y, x, a, n);
    ADM_HIP(hipGetLastError());
    return ADM_OK;
}
