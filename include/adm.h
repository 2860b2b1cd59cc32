/*
 * adm.h -- C ABI of libadm.so: the MI355X (gfx950) implementation of Adorym's multislice
 * forward model + hand-derived adjoint + object update.
 *
 * The reference (mdw771/adorym) is 100 % Python and has no FFI; this header defines the
 * boundary a reference maintainer would bind with ctypes.  Each entry
 * point names the reference code it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - plain C, every function returns 0 on success and a negative adm_status otherwise;
 *     adm_last_error() returns a thread-local, library-owned message for the last failure.
 *   - "device pointer" = memory of the context's GPU (hipMalloc'ed by adm_malloc or by anybody
 *     else in the process, e.g. torch); "host pointer" = borrowed for the duration of the call.
 *   - all work is enqueued on the context's HIP stream; calls are asynchronous unless noted.
 *   - fp32 arithmetic throughout (the reference's default dtype).
 *   - a context is not thread-safe.
 */
#ifndef ADM_H
#define ADM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADM_VERSION 100

typedef enum {
    ADM_OK = 0,
    ADM_ERR_INVALID = -1,     /* bad argument                                   */
    ADM_ERR_HIP = -2,         /* HIP runtime error (message has the hipError)    */
    ADM_ERR_UNSUPPORTED = -3, /* valid in the reference, not implemented here    */
    ADM_ERR_NOMEM = -4
} adm_status;

typedef struct adm_ctx adm_ctx;
typedef struct adm_plan adm_plan;

/* ---- library / context ------------------------------------------------------------ */
int adm_version(void);
const char* adm_last_error(void);
/* (no reference counterpart) A line to leave on stdout should the process be killed by SIGABRT / SIGSEGV / SIGBUS -- the ROCm
 * runtime abort()s on a GPU memory fault -- followed by _exit(exit_code).  NULL disarms.  bench.py arms it around the secondary
 * legs of a multi-rank run so that a fault inside one of them still leaves the measured headline line behind. */
int adm_crash_line_set(const char* line, int exit_code);
/* Number of GPUs visible to the process (0 if none / on error); creates no context.  Replaces the device enumeration
 * behind `gpu_index` (adorym/ptychography.py:203-205) for launchers that map local ranks to devices. */
int adm_device_count(void);
/* Free and total bytes of the context's GPU (hipMemGetInfo): what a driver logs as device memory in use; the reference has
 * no counterpart beyond torch's allocator statistics. */
int adm_mem_info(adm_ctx* ctx, size_t* free_bytes, size_t* total_bytes);

/* One context = one GPU + one stream.  `stream` may be an existing hipStream_t (e.g.
 * torch.cuda.current_stream().cuda_stream) or NULL to let the context own a new one.
 * Replaces: the device selection of reconstruct_ptychography (adorym/ptychography.py:203-205). */
int adm_ctx_create(int device, void* stream, adm_ctx** out);
int adm_ctx_destroy(adm_ctx* ctx);
int adm_ctx_sync(adm_ctx* ctx);                 /* blocks until the stream is idle */
void* adm_ctx_stream(adm_ctx* ctx);             /* the hipStream_t in use */
int adm_ctx_device(adm_ctx* ctx);
/* Side stream for work that does not depend on the multislice chain (e.g. zeroing the gradient buffer and the
 * regulariser gradient, which only read the object): calls made between adm_ctx_fork and adm_ctx_end_fork are
 * enqueued on an auxiliary stream that first waits for everything enqueued so far; adm_ctx_join makes the main
 * stream wait for that side work (call it before the first consumer of its results). */
int adm_ctx_fork(adm_ctx* ctx);
int adm_ctx_end_fork(adm_ctx* ctx);
int adm_ctx_join(adm_ctx* ctx);

/* ---- device memory (replaces w.create_variable / w.zeros / w.to_numpy on device tensors,
 *      adorym/wrappers.py:121-147, 186-205) -------------------------------------------- */
int adm_malloc(adm_ctx* ctx, size_t bytes, void** dptr);
int adm_free(adm_ctx* ctx, void* dptr);
int adm_memset(adm_ctx* ctx, void* dptr, int byte_value, size_t bytes);       /* async */
int adm_h2d(adm_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes); /* blocking */
int adm_d2h(adm_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes); /* blocking */
int adm_d2d(adm_ctx* ctx, void* dst_dev, const void* src_dev, size_t bytes);  /* async */
/* Pinned host memory + asynchronous read-back: lets the caller queue the next minibatch before it looks at the loss of
 * the previous one (the reference blocks on w.to_numpy(loss) every minibatch, adorym/forward_model.py:140).
 * adm_d2h_async: dst must come from adm_host_alloc; complete once an event recorded after it has been synchronised. */
int adm_host_alloc(adm_ctx* ctx, size_t bytes, void** hptr);
int adm_host_free(adm_ctx* ctx, void* hptr);
int adm_d2h_async(adm_ctx* ctx, void* dst_pinned, const void* src_dev, size_t bytes);
/* adm_h2d_async: src must come from adm_host_alloc and must not be rewritten before an event recorded after the call has
 * happened; the copy is ordered on the context's stream like a kernel (no host synchronisation).  Replaces the
 * per-minibatch host->device hand-over of the measured data (adorym/forward_model.py:113-119). */
int adm_h2d_async(adm_ctx* ctx, void* dst_dev, const void* src_pinned, size_t bytes);

/* ---- events: kernel timing on the context's stream --------------------------------- */
int adm_event_create(adm_ctx* ctx, void** ev);
int adm_event_destroy(adm_ctx* ctx, void* ev);
int adm_event_record(adm_ctx* ctx, void* ev);
int adm_event_elapsed_ms(adm_ctx* ctx, void* ev_start, void* ev_stop, float* ms); /* blocks on ev_stop */
int adm_event_sync(adm_ctx* ctx, void* ev);                                        /* blocks until ev has happened */

/* ---------------------------------------------------------------------------------------------------------------------
 * Collectives of the data-parallel mode (one process per GPU): RCCL over xGMI, loaded with dlopen on first use.
 * They replace the reference's mpi4py object collectives (pickle through host memory):
 *   gradient.arr = comm.allreduce(gradient.arr)      adorym/ptychography.py:1113-1114  -> adm_reduce_scatter of the object
 *       gradient, fused optimiser step on the owned shard, adm_all_gather of the updated shards
 *   comm.allreduce(w.to_numpy(opt.grads))            adorym/optimizers.py:1025,1041,1053,1064,1079 -> adm_all_reduce
 *   comm.bcast / comm.Bcast                          adorym/ptychography.py:217,411,485-487,664-665,796 -> adm_broadcast
 * adm_comm_unique_id: 128 bytes, created by one rank and handed to all ranks by the host side's rendezvous.
 * adm_comm_init: collective over the `nranks` contexts.  All transfers use device pointers, are enqueued on the
 * context's stream and are asynchronous; counts are in floats (bytes for adm_broadcast).
 * adm_reduce_scatter: recv[0..recv_count) = sum over ranks of send[rank*recv_count ...]; send holds nranks*recv_count.
 * adm_all_gather:     recv[r*send_count ...] = rank r's send[0..send_count).  adm_all_reduce: in place, sum (op_max = 0) or max. */
int adm_comm_unique_id(void* out128);
/* 0 if librccl and all entry points libadm needs resolve in this process; no communicator, no GPU work.  Every rank
 * checks its own installation with it BEFORE any rank enters the rendezvous of adm_comm_init (a rank that fails there
 * would leave the others waiting).  No reference counterpart (mpi4py either imports or the run is serial,
 * adorym/ptychography.py:45-50). */
int adm_comm_available(void);
int adm_comm_init(adm_ctx* ctx, int rank, int nranks, const void* unique_id128);
/* Optional second communicator over the same ranks (own unique id, collective like adm_comm_init) for collectives issued
 * between adm_ctx_fork and adm_ctx_end_fork, i.e. on the side stream: the deferred part of the object all-gather that
 * runs beside the next multislice kernel.  Each communicator then only ever sees one stream. */
int adm_comm_init_aux(adm_ctx* ctx, const void* unique_id128);
int adm_comm_destroy(adm_ctx* ctx);
int adm_comm_rank(adm_ctx* ctx);
int adm_comm_size(adm_ctx* ctx);
int adm_reduce_scatter(adm_ctx* ctx, const float* send, float* recv, size_t recv_count);
int adm_all_gather(adm_ctx* ctx, const float* send, float* recv, size_t send_count);
int adm_all_reduce(adm_ctx* ctx, float* buf, size_t count, int op_max);
int adm_broadcast(adm_ctx* ctx, void* buf, size_t bytes, int root);
/* buf[0 .. count) of rank `root` = the sum over ranks of their buf[0 .. count) (in place; other ranks' buffers unchanged): the
 * data term of `comm.allreduce(gradient.arr)` (adorym/ptychography.py:1113-1114) delivered to the rank that owns those elements
 * of a sharded update, for the part of the gradient the global batch touched (adm_reg_grad_range supplies the rest). */
int adm_reduce(adm_ctx* ctx, float* buf, size_t count, int root);
/* Collectives issued between the two calls are launched together (ncclGroupStart / ncclGroupEnd). */
int adm_comm_group_start(adm_ctx* ctx);
int adm_comm_group_end(adm_ctx* ctx);

/* ---------------------------------------------------------------------------------------------------------------------
 * Peer-to-peer transport (adm_p2p.hip): the same exchange as above -- `gradient.arr = comm.allreduce(gradient.arr)`
 * (adorym/ptychography.py:1113-1114) followed by the identical optimiser step on every rank (:1120-1129, optimizers.py:309-318)
 * and the constraints / mask (:1135-1158, array_ops.py:239-251) -- as a DIRECT all-pairs exchange without RCCL: every rank maps
 * the object and gradient buffers of its peers (IPC handles; on one node every peer is one xGMI hop away) and ONE kernel per
 * update sums the ranks' gradients of the owned shard in rank order, applies the optimiser in registers and writes the new
 * values into every replica: reduce-scatter + optimiser + all-gather in one pass.  Ordering between the ranks uses device-side
 * flags in uncached memory (signal / wait kernels on the context's stream: no host synchronisation).  Several ranks may share
 * one GPU (the mapped pointers then name local memory), which is how the path is exercised on a single-GPU machine.
 *   adm_p2p_create       per rank: flag block + mailbox (for small all-reduces; bytes, 0 = 8 MB).  nranks <= ADM_P2P_MAX_RANKS.
 *   adm_p2p_local        which = 0: the rank's flag block, 1: its mailbox -- to be exported to the peers.
 *   adm_p2p_export/open/close   ADM_P2P_HANDLE_BYTES-byte IPC handle of a device allocation (its base address) / mapping of a
 *                        peer's allocation in this process / unmapping.  A handle cannot be opened by the process that made it.
 *   adm_p2p_connect      the flag blocks and mailboxes of all ranks (array of nranks device pointers; own entries ignored).
 *   adm_p2p_bind_object  the object and gradient buffers of all ranks, n floats each (own entries = own buffers).
 *   adm_p2p_update       kind = ADM_OPT_*.  For the flat elements [lo, hi) -- this rank's shard: g = sum over ranks q of g_q[i]
 *                        (rank order) where i is in [sum_lo, sum_hi), g = own g[i] elsewhere (footprint-restricted exchange: the
 *                        owner's buffer is complete there); then adm_adam_step / adm_gd_step / adm_momentum_step's arithmetic
 *                        with the moments m, v indexed from the shard's start (m[i - lo]; momentum: m is the velocity, b1 is
 *                        gamma); the result is written to x_q[i] of every rank.  A barrier of the ranks' streams on entry
 *                        (every gradient buffer complete) and on exit (every replica updated, every gradient buffer free).
 *   adm_p2p_all_reduce   in-place rank-order sum of a small device array through the mailboxes (comm.allreduce of the small
 *                        parameter gradients, adorym/optimizers.py:1025,1041,1053,1064,1079).
 *   adm_p2p_barrier      barrier of the ranks' streams (device side).
 *   adm_p2p_status       ADM_OK, or an error if a wait timed out (ADM_P2P_TIMEOUT_S, default 30 s): which rank did not arrive.
 *   adm_p2p_destroy      frees flag block and mailbox; the peers must have closed their mappings. */
#define ADM_P2P_HANDLE_BYTES 64
#define ADM_P2P_MAX_RANKS 16
#define ADM_OPT_ADAM 0
#define ADM_OPT_GD 1
#define ADM_OPT_MOMENTUM 2
int adm_p2p_create(adm_ctx* ctx, int rank, int nranks, size_t mailbox_bytes);
int adm_p2p_local(adm_ctx* ctx, int which, void** dptr);
int adm_p2p_export(adm_ctx* ctx, const void* dptr, void* handle64);
int adm_p2p_open(adm_ctx* ctx, const void* handle64, void** dptr);
int adm_p2p_close(adm_ctx* ctx, void* dptr);
int adm_p2p_connect(adm_ctx* ctx, void* const* peer_flags, void* const* peer_mailbox);
int adm_p2p_bind_object(adm_ctx* ctx, void* const* peer_x, void* const* peer_g, size_t n);
int adm_p2p_rank(adm_ctx* ctx);
int adm_p2p_size(adm_ctx* ctx);
int adm_p2p_update(adm_ctx* ctx, int kind, float* m, float* v, size_t lo, size_t hi, size_t sum_lo, size_t sum_hi, int i_batch,
                   double step_size, double b1, double b2, double eps, int flags, const float* mask);
int adm_p2p_all_reduce(adm_ctx* ctx, float* buf, size_t count);
int adm_p2p_barrier(adm_ctx* ctx);
int adm_p2p_status(adm_ctx* ctx);
int adm_p2p_destroy(adm_ctx* ctx);

/* ---- plan: static geometry + physics of one reconstruction ------------------------- */
typedef enum { ADM_DET_NONE = 0, ADM_DET_FARFIELD = 1, ADM_DET_FRESNEL = 2 } adm_det_mode;
typedef enum { ADM_LOSS_LSQ = 0, ADM_LOSS_POISSON = 1 } adm_loss_type;

typedef struct {
    int32_t obj_y, obj_x, obj_z;      /* object [Y, X, Z, 2] (delta, beta interleaved, z fastest)    */
    int32_t probe_y, probe_x;         /* tile / probe / detector size                                 */
    int32_t pad_y0, pad_y1;           /* zero padding of the rotated-frame buffers so every tile fits */
    int32_t pad_x0, pad_x1;           /*   (adorym/util.py:1374-1406 applied to ALL probe positions)  */
    int32_t binning;                  /* slices summed per modulation (adorym/propagate.py:207-241)   */
    int32_t n_modes;                  /* probe modes (adorym/forward_model.py:354-375)                */
    int32_t sign_convention;          /* +1 / -1 (adorym/propagate.py:241)                            */
    int32_t det_mode;                 /* adm_det_mode: free_prop_cm 0/None, 'inf', finite             */
    int32_t normalize_fft;            /* far field with norm='ortho' (adorym/wrappers.py:725-770)     */
    float   k1;                       /* 2*PI*delta_nm/lmbda_nm (adorym/propagate.py:215)             */
    const float* h_re;                /* host [probe_y*probe_x] slice-to-slice transfer function,     */
    const float* h_im;                /*   unshifted, = get_kernel() cast to fp32 (propagate.py:62-81, 202-204) */
    const float* hfree_re;            /* host, detector-plane Fresnel kernel for ADM_DET_FRESNEL      */
    const float* hfree_im;            /*   (adorym/propagate.py:537-553), else NULL                   */
    int32_t loss_type;                /* adm_loss_type: LSQ on magnitudes, or Poisson (adorym/forward_model.py:88-103) */
    float   poisson_multiplier;       /* forward_model.py:94-102; ignored for LSQ                     */
    int32_t unknown_type;             /* 0 'delta_beta': slices hold (delta, beta), c = exp(-k1 beta) e^{-i sigma k1 delta};  */
                                      /* 1 'real_imag': slices hold c = re + i im directly (adorym/propagate.py:236-249);     */
                                      /*   the caller pre-fills the pads of obj_rot with (1, 0) (adorym/util.py:1338-1350)    */
} adm_plan_desc;

int adm_plan_create(adm_ctx* ctx, const adm_plan_desc* desc, adm_plan** out);
int adm_plan_destroy(adm_plan* plan);
/* Beamstop (adorym/forward_model.py:128-136): host float [Py][Px] in the detector layout of the data; pixels with
 * mask >= 1e-5 take part in the loss, the others are dropped.  The per-position loss sums then run over the kept pixels only
 * and the caller's grad_scale / mean use their count.  NULL removes the mask. */
int adm_plan_set_detector_mask(adm_plan* plan, const float* mask_host);
/* Several detector-plane Fresnel kernels in one plan (det_mode ADM_DET_FRESNEL): multi-distance data divided into sub-tiles
 * propagates the SAME tiles to every distance (adorym/forward_model.py:999-1018, one multislice_propagate_batch per distance).
 * hfree_re / hfree_im: host [n][Py*Px], kernel i as adm_plan_desc.hfree_* would hold it for distance i.  Afterwards position b
 * of a launch is propagated with kernel b % n (a launch of n * B positions, tile-major: tile j at every distance, then tile
 * j + 1); such launches go through adm_multislice_fwd_adj_pp (every tile brings its own probe window) and their batch must
 * be a multiple of n.  n = 1 restores the one-kernel behaviour. */
int adm_plan_set_detector_kernels(adm_plan* plan, int n, const float* hfree_re, const float* hfree_im);
/* number of floats of one rotated-frame buffer: obj_z * (obj_y+pads) * (obj_x+pads) * 2.
 * Internal layout is slice-major [Z][Yp][Xp][2] so that a tile slice is Py contiguous rows. */
size_t adm_plan_rot_elems(const adm_plan* plan);
/* bytes of scratch adm_multislice_fwd_adj needs for `batch` positions (stored post-modulation wavefields) */
size_t adm_plan_workspace_bytes(const adm_plan* plan, int batch);

/* ---- R1/R2  rotation ----------------------------------------------------------------
 * adm_rotate_fwd replaces apply_rotation -> w.grid_sample (adorym/util.py:536-552,
 * adorym/wrappers.py:1105-1147) and pad_object (adorym/util.py:1327-1351):
 *   obj [Y,X,Z,2] --bilinear gather, border clamp--> interior of obj_rot [Z][Yp][Xp][2]
 * for y-planes y_lo <= y < y_hi (planes are independent under a rotation about axis 0).
 * coords: device uint16 (IEEE fp16 bits) [X*Z, 2] = the reference's lookup table
 * (adorym/util.py:492-516); NULL = no rotation (two_d_mode / theta-independent copy).
 * adm_rotate_adj is its transpose (autograd of grid_sampler_2d): grad_obj += R^T grad_rot. */
int adm_rotate_fwd(adm_plan* plan, const float* obj, const uint16_t* coords, float* obj_rot, int y_lo, int y_hi);
/* The same gather for a plan whose y extent is n_tables blocks of equal height -- R objects stacked along y, one rotation angle
 * each (adorym_amd.AngleBatch: BASELINE config 2's 16 angles per update, adorym/ptychography.py:342-346 forces minibatch 1 per rank
 * there) -- all from the ONE object `obj` [obj_y / n_tables, X, Z, 2]: block r of obj_rot = obj rotated with tables_dev[r]
 * (device array of n_tables pointers to fp16 lookup tables).  One launch instead of n_tables. */
int adm_rotate_fwd_stack(adm_plan* plan, const float* obj, const void* tables_dev, int n_tables, float* obj_rot);
int adm_rotate_adj(adm_plan* plan, const float* grad_rot, const uint16_t* coords, float* grad_obj, int y_lo, int y_hi);
/* Same operator as adm_rotate_adj, evaluated as a deterministic gather: the transpose of the bilinear sampling
 * matrix of one angle in CSR form over object-plane voxels t = x*Z + z:
 *   csr_ptr [X*Z+1], csr_src [nnz] = z'*(Yp*Xp) + pad_x0 + x' (float2 offset of the rotated-frame voxel inside
 *   grad_rot, without the y row), csr_w [nnz] bilinear weights.   grad_obj[y][t] += sum_j w_j * grad_rot[src_j + row(y)].
 * The host builds the CSR once per angle from the same fp16 lookup table (adorym_amd/util.py).
 * lanes_along_x: performance hint only (same result): 1 when |cos(theta)| > |sin(theta)| so that neighbouring lanes
 * gather neighbouring x' of grad_rot. */
int adm_rotate_adj_csr(adm_plan* plan, const float* grad_rot, const int32_t* csr_ptr, const int32_t* csr_src,
                       const float* csr_w, float* grad_obj, int y_lo, int y_hi, int lanes_along_x);
/* The same gather staged through LDS: boxes int32 [ceil(Z/16)][ceil(X/16)][4] = (x'0, z'0, width, height) is the
 * rotated-frame bounding box (width*height <= 4096; four planes are staged at once up to 1024, else one at a time: rim patches
 * collect the border-clamped samples of a whole corner of the rotated frame) of the sources of each 16 x 16 patch of
 * object-plane voxels (x-patch index fastest) and csr_lsrc uint16 [nnz] = (z' - z'0) * width + (x' - x'0); width = 0 marks a
 * patch without a usable box, which gathers through csr_src as adm_rotate_adj_csr does. */
int adm_rotate_adj_staged(adm_plan* plan, const float* grad_rot, const int32_t* csr_ptr, const int32_t* csr_src,
                          const uint16_t* csr_lsrc, const float* csr_w, const int32_t* boxes, float* grad_obj, int y_lo, int y_hi);
/* adm_rotate_adj_staged for a plan whose y extent is n_tables stacked blocks (see adm_rotate_fwd_stack): block r of grad_rot is
 * back-rotated with the r-th set of CSR tables and the n_tables contributions are ADDED, r ascending, into the ONE gradient
 * grad_obj [obj_y / n_tables, X, Z, 2] -- the additions n_tables sequential calls would make (`gradient.arr = comm.allreduce(...)`
 * of 16 single-angle ranks, adorym/ptychography.py:1113-1114), in one launch.  tables_dev: device array of n_tables records of five
 * device pointers (csr_ptr, csr_src, csr_lsrc, csr_w, boxes: the outputs of adm_rotation_csr_build, in that order).
 * scratch (device, n_tables x the size of grad_obj; or NULL): with it the angles' terms are formed side by side and then added in
 * angle order by a second launch -- the same additions, the same bits, without a chain of n_tables rounds inside every block. */
int adm_rotate_adj_staged_stack(adm_plan* plan, const float* grad_rot, const void* tables_dev, int n_tables, float* grad_obj,
                                float* scratch, size_t scratch_bytes);
/* The fp16 lookup table of one angle formed on the device: what save_rotation_lookup / read_origin_coords hand to apply_rotation
 * (adorym/util.py:446-477, 492-525) -- source coordinates (x_old, z_old) of every rotated-frame cell, flat index x * Z + z, float32
 * arithmetic with one rounding per operation, stored as float16; bit-identical to the reference's table.  cos_theta / sin_theta:
 * np.cos / np.sin of float32(theta) evaluated in float32 on the host (the reference's torch scalars).  coords: device uint16 [X*Z][2]. */
int adm_rotation_table_build(adm_ctx* ctx, int X, int Z, float cos_theta, float sin_theta, uint16_t* coords);
/* Builds, on the device, everything adm_rotate_adj_staged needs for one angle from the fp16 lookup table `coords`
 * (device, [X*Z][2]): csr_ptr [X*Z+1], csr_src / csr_lsrc / csr_w [4*X*Z] (only the first csr_ptr[X*Z] entries are
 * meaningful), boxes [ceil(Z/16)*ceil(X/16)][4].  Rows ordered by target voxel, entries by ascending source offset: the same
 * table the host builder (adorym_amd/util.py:build_rotation_adjoint_csr) produces, so the gather stays deterministic.
 * scratch: device memory of adm_rotation_csr_scratch_bytes(plan) bytes.  Asynchronous on the context's stream.
 * Transposes adorym/util.py:536-552 + adorym/wrappers.py:1105-1147 (what torch's grid_sampler_2d_backward scatters). */
size_t adm_rotation_csr_scratch_bytes(const adm_plan* plan);
int adm_rotation_csr_build(adm_plan* plan, const uint16_t* coords, int32_t* csr_ptr, int32_t* csr_src, uint16_t* csr_lsrc,
                           float* csr_w, int32_t* boxes, void* scratch, size_t scratch_bytes);

/* ---- R3,R5-R8,R10  multislice forward + loss + adjoint ------------------------------
 * Replaces, for one minibatch of `batch` probe positions of one rotation angle:
 *   tile extraction            adorym/forward_model.py:313-331
 *   multislice_propagate_batch adorym/propagate.py:195-280 (delta_beta, non-projection branch)
 *   w.norm / mode sum          adorym/forward_model.py:337-375
 *   LSQ magnitude loss         adorym/forward_model.py:88-93
 *   torch.autograd.grad        adorym/wrappers.py:322   (hand-derived adjoint)
 * obj_rot   device [Z][Yp][Xp][2]
 * probe     device [n_modes][Py][Px][2] (real, imag interleaved)
 * pos       device int32 [batch][2] = (y, x) top-left corner of each tile in OBJECT coordinates
 *           (may be negative / overhang: the pads cover it)
 * target    device [batch][Py][Px], in the reference's fftshift-ed detector layout.  LSQ: target magnitude abs(prj)
 *           (sqrt(abs(prj)) for intensity data).  Poisson: abs(prj)^2 for magnitude data, abs(prj) for intensity data.
 * want_grad 0 = forward / loss only (predict); 1 = also run the adjoint sweep, leaving one tile gradient per
 *           position in `workspace` for adm_tile_grad_accumulate
 * grad_probe device [n_modes][Py][Px][2], accumulated into; may be NULL
 * pred      device [batch][Py][Px] predicted magnitude (reference layout); may be NULL
 * loss_sum  device [batch] : per-position sum over pixels of (pred-target)^2, or of the Poisson terms
 *           pred^2*pm - target*pm*log(pred^2*pm) (overwritten)
 * grad_scale multiplies d loss/d pred: 2/(batch*Py*Px) for the reference's mean()
 * workspace device scratch of adm_plan_workspace_bytes(plan, batch) bytes (unused when want_grad==0 and n_modes==1) */
int adm_multislice_fwd_adj(adm_plan* plan, const float* obj_rot, const float* probe, const int32_t* pos, int batch,
                           const float* target, int want_grad, float* grad_probe, float* pred, float* loss_sum,
                           float grad_scale, void* workspace, size_t workspace_bytes);

/* Slice-transmission cache (delta_beta unknowns, binning 1).  The reference evaluates exp(-k1*beta) * (cos, sin)(-sigma*k1*delta)
 * for every voxel of every tile it extracts (adorym/propagate.py:241 through wrappers.py:600-608), i.e. once per covering
 * probe position -- up to ~40 times per voxel and angle in config 3 -- and autograd once more in the backward pass.  With the
 * cache on, adm_rotate_fwd stores that factor per rotated-frame voxel (a plan-owned [Z][Yp][Xp] complex buffer, pads = 1+0i)
 * beside obj_rot, and adm_multislice_fwd_adj[_pp] multiplies with the loaded number: bit-identical results (the same fp32
 * expression, evaluated once), no transcendental in the slice loop.  The cache is used only for the obj_rot buffer it was
 * last filled from; a caller that writes obj_rot by any other route than adm_rotate_fwd must call adm_transmission_refresh
 * for the rows [y_lo, y_hi) (object coordinates) it changed.  One rotated-object buffer per plan while the cache is on: the
 * cache has a single image, so alternating PARTIAL-row rotations between two obj_rot buffers would leave rows of one beside
 * rows of the other.  Off by default at this level; adorym_amd's engine (one obj_rot per plan) turns it on.
 * on = 2: the cache REPLACES the rotated object -- adm_rotate_fwd writes the transmissions only (half the stores of the rotation:
 * nobody reads the rotated (delta, beta) once the slice loop multiplies with cached numbers) and obj_rot is just the name of the
 * image the cache holds; adm_multislice_fwd_adj then refuses an obj_rot the cache was not filled from. */
int adm_plan_set_transmission_cache(adm_plan* plan, int on);
int adm_transmission_refresh(adm_plan* plan, const float* obj_rot, int y_lo, int y_hi);
/* Probe sizes.  Any Py x Px with Py*Px <= 16384 whose field fits the LDS is accepted (the reference takes whatever
 * prj.shape[-2:] is, adorym/ptychography.py:313-317).  Square sizes in {8,12,16,18,24,27,32,36,64,72} run the tuned
 * register-resident kernels; every other size -- and every size after adm_plan_set_generic(plan, 1) -- runs the generic
 * kernel (adm_ms_generic.hip: run-time radix lists, any prime factors, non-square).  Call it before the first workspace is
 * sized: the two kernels lay their workspace rows out differently.  Per-position probes (adm_multislice_fwd_adj_pp,
 * adm_probe_shift*) exist for the tuned sizes only. */
int adm_plan_set_generic(adm_plan* plan, int on);

/* Same as adm_multislice_fwd_adj with ONE PROBE SET PER POSITION (sub-pixel probe positions, adorym/forward_model.py:
 * 296-311 + 337-375): probes device [batch][n_modes][Py][Px][2]; grad_probes device [batch][n_modes][Py][Px][2] or NULL,
 * OVERWRITTEN with the per-position probe gradients (input of adm_probe_shift_adj). */
int adm_multislice_fwd_adj_pp(adm_plan* plan, const float* obj_rot, const float* probes, const int32_t* pos, int batch,
                              const float* target, int want_grad, float* grad_probes, float* pred, float* loss_sum,
                              float grad_scale, void* workspace, size_t workspace_bytes);

/* ---- f2  sub-pixel probe positions -----------------------------------------------------
 * realign_image_fourier (adorym/util.py:380-397) applied to every probe mode for every position of a minibatch:
 *   probes_out[b][m] = IFFT2( exp(-2 PI i (fx*sx_b + fy*sy_b)) * FFT2(probe[m]) ),  PI = 3.14159265359, f = fftfreq.
 * shifts  device float [n_entries][2] = (sy, sx), the reference's probe_pos_correction flattened over (theta, position)
 * index   device int32 [batch]: entry used by position b; NULL = entry b
 * adm_probe_shift_adj is the adjoint (what torch.autograd.grad returns through that op for probe_real/imag and
 * probe_pos_correction): grad_probe [n_modes][Py][Px][2] += sum_b shift_{-s_b}(grad_probes[b]) (may be NULL);
 * grad_shifts[index[b]][0..1] += dL/d(sy, sx) (float [n_entries][2], NOT zeroed by the call).
 * grad_probes is CONSUMED: with more than 256 (position, mode) pairs its slots are reused for the terms of the sum, which is then
 * formed in a fixed order (no atomics); its contents are undefined afterwards. */
int adm_probe_shift(adm_plan* plan, const float* probe, const float* shifts, const int32_t* index, int batch, float* probes_out);
int adm_probe_shift_adj(adm_plan* plan, const float* probe, const float* shifts, const int32_t* index, int batch,
                        float* grad_probes, float* grad_probe, float* grad_shifts);
/* x[r][c] -= mean_r x[r][c]: the drift guard applied to probe_pos_correction after its update
 * (adorym/optimizers.py:1046-1048).  Single workgroup; meant for small parameter arrays. */
int adm_center_rows(adm_ctx* ctx, float* x, size_t n_rows, int n_cols);

/* Adjoint of the tile gather (adorym/forward_model.py:313-331 under autograd): overlap-adds the tile
 * gradients left in `workspace` by adm_multislice_fwd_adj(want_grad=1) for the same pos/batch and WRITES
 * (not accumulates) them into every padded row of grad_rot [Z][Yp][Xp][2] touched by the batch; pixels of
 * those rows not covered by any tile are set to zero, other rows are left untouched.  Deterministic (no
 * atomics).  pos_host = the same positions on the host (used for the row window). */
int adm_tile_grad_accumulate(adm_plan* plan, void* workspace, size_t workspace_bytes, const int32_t* pos, int batch,
                             const int32_t* pos_host, float* grad_rot);
/* The same for a batch that was launched in several parts (each with its own workspace), so that the overlap-add of one
 * part can run beside the multislice launch of the next.  First part: add = 0 and [win_y_lo, win_y_hi) = the y-window
 * (object coordinates; may reach into the pads) of the WHOLE batch -- all of it is written (zeros where no tile of this part
 * reaches).  Later parts: add = 1, their tiles are accumulated into what is there (win_* ignored). */
int adm_tile_grad_accumulate_part(adm_plan* plan, void* workspace, size_t workspace_bytes, const int32_t* pos, int batch,
                                  const int32_t* pos_host, float* grad_rot, int win_y_lo, int win_y_hi, int add);
/* The cover lists of the overlap-add (which tiles reach which padded pixel) depend on the positions only.  Building them early --
 * e.g. between adm_ctx_fork and adm_ctx_end_fork, beside the multislice launch -- takes one launch and its dependency gap off the
 * chain behind the kernel: the next adm_tile_grad_accumulate[_part] with the same workspace / pos / batch / window finds them built
 * and skips its own build (one use).  Same arguments as adm_tile_grad_accumulate_part without grad_rot.  Ordering between the
 * two calls is the caller's (adm_ctx_join).  No reference counterpart: autograd's index bookkeeping (adorym/forward_model.py:313-331). */
int adm_tile_cover_build(adm_plan* plan, void* workspace, size_t workspace_bytes, const int32_t* pos, int batch,
                         const int32_t* pos_host, int win_y_lo, int win_y_hi, int add);
/* Blocking: *overflow_host = 1 if some pixel of the last adm_tile_grad_accumulate was covered by more than 64 tiles
 * (the overlap-add then dropped contributions; use smaller batches). */
/* The overlap-add of a batch in which a pixel is covered by more than 64 tiles (dense 2-D scans taken as one minibatch,
 * demos/2d_ptychography_w_probe_optimization.py): one pass per range [b_lo, b_hi) of at most 64 positions of the batch -- the
 * first with add = 0 (writes the batch's rows), the others with add = 1.  Same arguments as adm_tile_grad_accumulate otherwise;
 * replaces autograd's index_add over the tile stack (adorym/forward_model.py:313-331) for such batches. */
int adm_tile_grad_accumulate_range(adm_plan* plan, void* workspace, size_t workspace_bytes, const int32_t* pos, int batch,
                                   const int32_t* pos_host, float* grad_rot, int b_lo, int b_hi, int add);
int adm_tile_grad_status(adm_plan* plan, void* workspace, size_t workspace_bytes, int batch, int* overflow_host);

/* ---- R9  regulariser gradients --------------------------------------------------------
 * L1Regularizer / TVRegularizer (adorym/regularizers.py:30-46, 95-110; adorym/util.py:1427-1440):
 * grad_obj += d/dobj [ alpha_d*mean|delta| + alpha_b*mean|beta| + gamma*(TV(delta)+TV(beta)) ];
 * reg_value (device float[1], may be NULL) += the regulariser value.
 * Plans with unknown_type 'real_imag' evaluate the reference's real_imag branches instead (regularizers.py:38-45,
 * 105-110): alpha_d*mean| |o| - mean|o| | + alpha_b*mean|arg o| + gamma*(TV(re^2+im^2) + TV(atan2(im, re))). */
/* adm_reg_grad_set: grad_obj = (instead of +=) the regulariser gradient -- initialises the gradient buffer of a new
 * minibatch in one pass instead of a zero fill followed by an accumulate. */
int adm_reg_grad_set(adm_plan* plan, const float* obj, float alpha_d, float alpha_b, float gamma, float* grad_obj,
                     float* reg_value);
int adm_reg_grad(adm_plan* plan, const float* obj, float alpha_d, float alpha_b, float gamma, float* grad_obj,
                 float* reg_value);
/* (grad_obj == NULL with reg_value != NULL: the value only; delta_beta unknowns.)
 * adm_reg_grad_range: the same gradient for the flat fp32 elements [lo, hi) of the object only -- a rank's shard of a sharded
 * update -- ADDED where the element lies in [add_lo, add_hi) and WRITTEN elsewhere.  Every rank of the reference adds its own,
 * identical, regulariser term before the all-reduce (adorym/forward_model.py:138-139); with the exchange restricted to the planes
 * the global batch touched, the owner adds it here instead, with R-fold weights (delta_beta unknowns, plain L1 / TV). */
int adm_reg_grad_range(adm_plan* plan, const float* obj, float alpha_d, float alpha_b, float gamma, float* grad_obj, size_t lo,
                       size_t hi, size_t add_lo, size_t add_hi);

/* ---- R13-R15  fused optimiser step + constraints on elements [lo, hi) of flat arrays -----
 * AdamOptimizer.apply_gradient math (adorym/optimizers.py:309-318), then non-negativity clip,
 * phase/absorption-only (adorym/ptychography.py:1135-1158) and the finite-support mask
 * (adorym/array_ops.py:239-251).  Channel = flat index & 1 (delta even, beta odd).
 * flags: bit0 non_negativity, bit1 zero channel 0 (absorption_only), bit2 zero channel 1 (phase_only).
 * mask: device float [n/2] per-voxel multiplier or NULL.  Hyper-parameters are doubles because the
 * reference forms 1-b1, 1-b1^(t+1) ... in Python doubles before torch casts them to fp32. */
#define ADM_FLAG_NONNEG 1
#define ADM_FLAG_ZERO_CH0 2
#define ADM_FLAG_ZERO_CH1 4
int adm_adam_step(adm_ctx* ctx, float* x, const float* g, float* m, float* v, size_t lo, size_t hi, int i_batch,
                  double step_size, double b1, double b2, double eps, int flags, const float* mask);
/* The SMALL optimisable parameters of a minibatch -- probe modes, sub-pixel position corrections, propagation distances, affine
 * matrices (adorym/optimizers.py:1022-1083) -- updated in one launch, one workgroup per array, with adm_adam_step's arithmetic:
 *   x, g, m, v   device arrays of n floats (g is the accumulated gradient; flags = 0, no mask)
 *   step_size    per array (the reference gives every parameter its own optimiser and learning rate)
 *   center_cols  > 0: afterwards subtract from every column of x viewed as [n / center_cols][center_cols] its mean over the rows
 *                (the drift guard of probe_pos_correction, optimizers.py:1046-1048; same sums as adm_center_rows)
 *   pin, pin_n   != NULL: afterwards x[0 .. pin_n) = pin[0 .. pin_n) ("regularize transformation of image 0", optimizers.py:1067-1073)
 *   zero_grad    != 0: g is zero-filled once used (the accumulator of the next minibatch, ptychography.py:1017-1031)
 * i_batch, b1, b2, eps are common to the arrays of one call (at most ADM_SMALL_PARAMS_MAX).  These updates are launch-bound. */
#define ADM_SMALL_PARAMS_MAX 6
typedef struct adm_small_param {
    float* x; float* g; float* m; float* v;
    uint64_t n;
    double step_size;
    int32_t center_cols;
    int32_t zero_grad;
    const float* pin;
    uint64_t pin_n;
} adm_small_param;
int adm_adam_step_small(adm_ctx* ctx, const adm_small_param* params, int count, int i_batch, double b1, double b2, double eps);
/* GDOptimizer.apply_gradient (adorym/optimizers.py:440-464); step_size already scheduled by the host */
int adm_gd_step(adm_ctx* ctx, float* x, const float* g, size_t lo, size_t hi, double step_size, int flags,
                const float* mask);
/* MomentumOptimizer.apply_gradient (adorym/optimizers.py:376-411): v = gamma*v + step*g; x = x - v; + constraints */
int adm_momentum_step(adm_ctx* ctx, float* x, const float* g, float* v, size_t lo, size_t hi, double step_size, double gamma,
                      int flags, const float* mask);
/* Reweighted L1 (adorym/regularizers.py:49-84).  adm_rwl1_update: weight = max(obj) / (|obj| + 1e-4*mean(obj)) over both
 * channels jointly (adorym/ptychography.py:995-1000); scratch = device float[2*1024+2].  adm_reg_grad_weighted:
 * grad_obj += alpha_c * weight * sign(obj) / V, reg_value += alpha_d*mean(w_d|delta|) + alpha_b*mean(w_b|beta|).
 * Plans with unknown_type 'real_imag' evaluate the reference's real_imag branch instead (regularizers.py:73-82): with
 * wm = w_re^2 + w_im^2, alpha_d*mean(wm*| |o| - mean|o| |) + alpha_b*mean(wm*|atan2(im, re)|), gradient w.r.t. (re, im). */
int adm_rwl1_update(adm_plan* plan, const float* obj, float* weight, float* scratch);
int adm_reg_grad_weighted(adm_plan* plan, const float* obj, const float* weight, float alpha_d, float alpha_b, float* grad_obj,
                          float* reg_value);
/* ---- f1  multi-distance near-field holography ------------------------------------------
 * MultiDistModel (adorym/forward_model.py:809-1092) for one undivided field of view (n_blocks == 1, config 5) and one
 * object slice:  psi = probe * c(obj);  Psi_d = IFFT2(FFT2(psi) * exp(-i sigma PI lambda d (u^2+v^2)))
 * (fresnel_propagate_wrapped, adorym/propagate.py:84-103, 556-568);  loss = mean_{d,pixels}(|Psi_d| - t_d)^2 with
 * t_d = sqrt|A_d(data_d)| for intensity data (|A_d(data_d)| for magnitudes) and A_d = w.affine_transform
 * (adorym/wrappers.py:1158-1174: F.affine_grid + F.grid_sample, bilinear, border padding, align_corners=False). */
typedef struct adm_holo adm_holo;
typedef struct adm_holo_desc {
    int32_t ny, nx;                   /* field size = object size (powers of two, 16 ... 2048)         */
    int32_t n_dists;                  /* number of holograms / propagation distances                   */
    double  lambda_nm;                /* 1240 / energy_ev                                              */
    double  voxel_nm_y, voxel_nm_x;   /* psize_cm * 1e7                                                */
    int32_t sign_convention;          /* +1 / -1                                                       */
    int32_t unknown_type;             /* 0 delta_beta (uses k1), 1 real_imag                           */
    int32_t raw_intensity;            /* raw_data_type: 0 'magnitude', 1 'intensity'                   */
    float   k1;                       /* 2 PI delta_nm / lambda_nm (delta_beta only)                   */
} adm_holo_desc;
int adm_holo_create(adm_ctx* ctx, const adm_holo_desc* desc, adm_holo** out);
int adm_holo_destroy(adm_holo* holo);
/* obj [ny][nx][2], probe [ny][nx][2], dists_cm [n_dists] (the reference's free_prop_cm), affine [n_dists][2][3]
 * (prj_affine_ls; NULL = identity), data [n_dists][ny][nx] raw measurements: all device pointers.
 * loss_sum [n_dists] (overwritten) = per-distance sum of squared residuals; loss = sum / (n_dists*ny*nx).  May be page-locked host
 * memory (adm_host_alloc): the last kernel then writes the sums where the host reads them, no copy is queued.
 * want_grad = 1: grad_obj [ny][nx][2] += dL/dobj; grad_probe [ny][nx][2] = dL/dprobe (NULL ok);
 * grad_dists [n_dists] += dL/dfree_prop_cm (NULL ok); grad_affine [n_dists][2][3] += dL/dprj_affine_ls (NULL ok);
 * want_grad = 2: the same with '=' instead of '+=' (the buffers need no zero fill: three launches less per minibatch on a
 * path that is bound by the number of launches);
 * pred [n_dists][ny][nx] = |Psi_d| (NULL ok). */
int adm_holo_fwd_adj(adm_holo* holo, const float* obj, const float* probe, const float* dists_cm, const float* affine,
                     const float* data, int want_grad, float* grad_obj, float* grad_probe, float* grad_dists,
                     float* grad_affine, float* pred, float* loss_sum);
/* adm_holo_fwd_adj (want_grad = 2: every gradient overwritten) FUSED with the Adam steps that consume those gradients --
 * `opt.apply_gradient` of the object (adorym/ptychography.py:1120-1129, optimizers.py:309-318) and the updates of `free_prop_cm`
 * and `prj_affine_ls` with the identity pin of matrix 0 (optimizers.py:1062-1083) -- in the launch group's last kernel: the object
 * gradient never reaches memory and no optimiser launch follows (a config-5 minibatch is five dependent kernels of 8 - 19 us; the
 * sixth launch was 7 % of it).  obj, dists_cm and affine are updated IN PLACE.  NULL moments of the distances / of the affine
 * matrices: that parameter is left alone (its gradient is not formed).  Same arithmetic, same bits as adm_holo_fwd_adj followed by
 * adm_adam_step_small on the three arrays.  Valid where the reference's update is exactly that: one rank, no regulariser on the
 * object, no constraint or mask, plain Adam with common (b1, b2, eps), a minibatch per update. */
typedef struct adm_holo_adam {
    float* m_obj; float* v_obj; double step_obj;                 /* [ny*nx*2] each */
    float* m_dists; float* v_dists; double step_dists;           /* [n_dists] each, or NULL */
    float* m_affine; float* v_affine; double step_affine;        /* [n_dists*6] each, or NULL */
    const float* affine_pin; uint64_t affine_pin_n;              /* the first affine_pin_n entries of affine are set to these afterwards */
    int32_t i_batch; double b1, b2, eps;
} adm_holo_adam;
int adm_holo_fwd_adj_adam(adm_holo* h, float* obj, const float* probe, float* dists_cm, float* affine, const float* data,
                          const adm_holo_adam* opt, float* pred, float* loss_sum);
/* Per-distance shift refinement of the measured holograms: `optimize_all_probe_pos` with multi-distance data,
 * adorym/forward_model.py:1075-1085 (demos/2d_multidist_holography_w_position_correction.py) -- the loss compares with
 *     T_d = Re IFFT2( FFT2(|data_d|) * exp(-2 PI i (fx s_d[1] + fy s_d[0])) )        (realign_image_fourier, util.py:380-397)
 * and torch.autograd.grad returns dL/ds_d.  Device pointers throughout.
 *   adm_holo_data_spectrum   spectrum [n_dists][nx][ny][2] <- FFT2(|data_d|), transposed ([kx][ky]); once per dataset.
 *   adm_holo_shift_targets   targets [n_dists][ny][nx] <- T_d for shifts [n_dists][2] = (sy, sx).
 *   adm_holo_set_registration(h, cot_out, direct): from now on adm_holo_fwd_adj takes `data` pixel for pixel when direct != 0
 *                            (it holds T_d; no affine matrices then) and, with want_grad, leaves c = dL/dT_d in cot_out
 *                            [n_dists][ny][nx] (NULL: not wanted).  (NULL, 0) restores the default.
 *   adm_holo_shift_grad      grad_shifts [n_dists][2] += sum_pixels c dT_d/ds_d.
 * The work fields of the handle are shared with adm_holo_fwd_adj: the calls of one minibatch follow each other on the stream. */
int adm_holo_data_spectrum(adm_holo* h, const float* data, float* spectrum);
int adm_holo_shift_targets(adm_holo* h, const float* spectrum, const float* shifts, float* targets);
int adm_holo_set_registration(adm_holo* h, float* cot_out, int direct);
int adm_holo_shift_grad(adm_holo* h, const float* cot, const float* spectrum, const float* shifts, float* grad_shifts);

/* y[i] += a * x[i]  (gradient accumulation, adorym/ptychography.py:1063-1066) */
int adm_axpy(adm_ctx* ctx, float* y, const float* x, float a, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* ADM_H */
