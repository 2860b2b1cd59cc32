// Internal definitions shared by the libadm translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <cstdio>
#include "../../include/adm.h"

#define ADM_DET_NONE_ 0
#define ADM_DET_FARFIELD_ 1
#define ADM_DET_FRESNEL_ 2

struct adm_ctx {
    int device;
    hipStream_t stream;      // stream new work is enqueued on (main, or aux between adm_ctx_fork / adm_ctx_end_fork)
    hipStream_t main_stream;
    hipStream_t aux_stream;  // side stream for work that is independent of the multislice chain
    hipEvent_t ev_fork, ev_join;
    bool owns_stream;
    bool join_pending;
    void* comm;              // ncclComm_t of adm_comm_init (adm_comm.hip) or nullptr
    void* comm_aux;          // ncclComm_t of adm_comm_init_aux: collectives queued on the side stream, or nullptr
    int comm_rank, comm_size;
    void* p2p;               // peer-to-peer group of adm_p2p_create (adm_p2p.hip) or nullptr
};

struct adm_plan {
    adm_ctx* ctx;
    adm_plan_desc d;       // host pointers inside are NOT kept valid; see device copies below
    int Yp, Xp;            // padded rotated-frame extents
    int n_steps;           // ceil(obj_z / binning)
    float2* h_dev;         // [Py*Px] slice transfer function
    float2* hfree_dev;     // [n_hfree][Py*Px] or nullptr
    int n_hfree;           // detector-plane Fresnel kernels held (1; adm_plan_set_detector_kernels: the distances of multi-distance data)
    float2* twid_dev;      // [Px] exp(-2 pi i j / N)
    float* det_weight_dev; // [Py*Px] beamstop weights or nullptr (adm_plan_set_detector_mask)
    float* reg_stats;      // 2 floats of scratch for the real_imag L1 regulariser (lazily allocated)
    float* reg_partial;    // [obj_y*obj_x] per-row partial sums of the regulariser value (lazily allocated)
    bool generic;          // probe size outside the tuned kernels' set (or forced): adm_ms_generic.hip, pixel-major workspace rows
    int gen_nrx, gen_nry, gen_rx[8], gen_ry[8];   // radix lists of the x / y transforms of the generic kernel
    float2* hs_dev;        // [Py*Px] H / (Py*Px), one rounding per element (generic kernel)
    float2* hfree_s_dev;   // same for the detector-plane Fresnel kernel, or nullptr
    float2* twid_y_dev;    // [Py] exp(-2 pi i j / Py)
    float2* trans_dev;     // [Z][Yp][Xp] slice transmissions of the voxels of trans_src, or nullptr (adm_plan_set_transmission_cache)
    const void* trans_src; // the obj_rot buffer trans_dev was last filled from (adm_rotate_fwd / adm_transmission_refresh)
    bool trans_only;       // adm_plan_set_transmission_cache(plan, 2): adm_rotate_fwd writes ONLY the transmissions (obj_rot is an identity, not data)
    // adm_tile_cover_build: the cover lists in workspace `ws` are current for (pos, batch, window); a few entries, so that every
    // round of a batch launched in parts can have its lists built ahead
    struct CoverKey { const void* ws; const void* pos; int batch, row0, nrows; unsigned long long fp; } cover_keys[4];
};

namespace adm {
void set_error(const std::string& msg);
int fail(int code, const std::string& msg);
int hip_fail(hipError_t e, const char* what);

struct MsParams {
    const float2* obj_rot;     // [Z][Yp][Xp] (delta, beta); pre_t: the cached slice transmissions instead
    int want_grad;             // 0 = forward only
    const float2* probe;       // [P][P]
    float2* grad_probe;        // [P][P] or nullptr
    const int2* pos;           // [B] (y, x) in object coordinates
    const float* target;       // [B][P][P]
    float* pred;               // [B][P][P] or nullptr
    float* loss_sum;           // [B]
    float2* stash;             // [B][n_steps][R1][NT] post-modulation wavefields
    float2* gtile;             // [B][n_steps][R1][NT] per-position tile gradients (d/ddelta, d/dbeta)
    const float2* h;           // [P][P] natural order, unscaled
    const float2* hfree;       // [n_hfree][P][P] or nullptr
    const float2* twid;        // [N] exp(-2 pi i j / N)
    int Z, Yp, Xp, pad_y0, pad_x0;
    int binning, n_steps;
    int det_mode, det_inverse;     // det_inverse: far field uses the inverse transform (sign_convention -1)
    float det_scale;               // far-field scale: 1, 1/N^2, or 1/N (ortho)
    float k1, sigma;
    float grad_scale;
    int n_modes;               // incoherent probe modes
    float2* det;               // [B][n_modes][G][NT] detector-plane fields of the modes (n_modes > 1 only)
    int loss_type;             // 0 LSQ on magnitudes, 1 Poisson
    float poisson_mult;
    int real_imag;             // unknown_type == 'real_imag'
    int pre_t;                 // obj_rot holds exp(-k1 beta) (cos, sin)(-sigma k1 delta) per voxel (adm_plan_set_transmission_cache)
    const float* det_weight;   // [P][P] 0/1 weights of the detector pixels in the loss (beamstop), reference layout; or nullptr
    size_t probe_bstride;      // float2 elements between the probes of consecutive positions (0 = one shared probe set)
    size_t gprobe_bstride;     // same for grad_probe (per-position gradients when the probes are per position)
    // generic kernel (adm_ms_generic.hip) only
    int gen_py, gen_px, gen_nrx, gen_nry, gen_rx[8], gen_ry[8];
    const float2* gen_twid_y;  // [Py]; twid is [Px]
    const float2* gen_hs;      // [Py][Px] H / (Py*Px)
    const float2* gen_hfree_s; // [n_hfree][Py][Px] or nullptr
    // (last, so that the fields above keep their kernel-argument offsets)
    int n_hfree;               // detector-plane kernels in hfree / gen_hfree_s: position b is propagated with kernel b % n_hfree
};
// per-position sub-pixel probe shifts (adorym/util.py:380-397, forward_model.py:296-311)
struct ShiftParams {
    const float2* probe;       // [M][P][P]
    const float2* shifts;      // (sy, sx) entries; entry of position b = shifts[index ? index[b] : b]
    const int* index;          // [B] or nullptr
    float2* probes_out;        // forward: [B][M][P][P]
    const float2* grad_probes; // adjoint: [B][M][P][P]
    float2* slots;             // adjoint, large batches: every (position, mode) leaves its probe-gradient term in its own [P][P] slot
                               // (= grad_probes, in place) for a fixed-order sum afterwards instead of atomics on grad_probe
    float2* grad_probe;        // adjoint: [M][P][P] accumulated (atomics) or nullptr
    float* grad_shifts;        // adjoint: accumulated (atomics) at [index ? index[b] : b][2]
    const float2* twid;
    int n_modes;
};
int ms_threads_for(int n);
int ms_r2_for(int n);
size_t ms_ws_per_pos(const adm_plan* plan);
size_t ms_row_elems(const adm_plan* plan);
size_t ws_det_bytes(const adm_plan* plan, int batch);   // float2 elements of one position's stash of ONE mode
size_t ws_off_gtile(const adm_plan* plan, int batch);
size_t ws_off_cover(const adm_plan* plan, int batch);
size_t ws_off_det(const adm_plan* plan, int batch);
size_t ws_off_gprobe(const adm_plan* plan, int batch);
hipError_t probe_grad_reduce(const float2* part, int batch, size_t n, float2* out, hipStream_t st);
hipError_t probe_grad_reduce_large(float2* part, int batch, size_t n, float2* out, hipStream_t st);   // two levels, `part` is scratch
int ms_r1_for(int n);
hipError_t ms_launch(int n, const MsParams& p, int batch, hipStream_t st);
// adm_multislice_fwd_adj's body (per_position: one probe set per position)
int multislice_impl(adm_plan* plan, const float* obj_rot, const float* probe, const int32_t* pos, int batch, const float* target,
                    int want_grad, float* grad_probe, float* pred, float* loss_sum, float grad_scale, void* workspace,
                    size_t workspace_bytes, bool per_position);
bool ms_generic_supported(int py, int px);
int ms_generic_threads(int py, int px);
hipError_t ms_generic_launch(const MsParams& p, int batch, hipStream_t st);
hipError_t shift_launch(int n, const ShiftParams& q, int batch, bool adjoint, hipStream_t st);
}  // namespace adm

#define ADM_HIP(call)                                          \
    do {                                                       \
        hipError_t e__ = (call);                               \
        if (e__ != hipSuccess) return adm::hip_fail(e__, #call); \
    } while (0)
