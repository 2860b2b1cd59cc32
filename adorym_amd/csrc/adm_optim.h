// Element-wise optimiser arithmetic shared by the kernels that apply it: adam_kernel / small_adam_kernel / gd_kernel /
// momentum_kernel (adm_object.hip: one rank, or the owned shard after an RCCL reduce-scatter) and the peer-to-peer fused
// exchange (adm_p2p.hip: sum over the ranks' gradient buffers + the same step + write to every replica).  One definition,
// contraction off, so that every caller produces the same bits.
//   AdamOptimizer.apply_gradient      adorym/optimizers.py:309-318
//   GDOptimizer.apply_gradient        adorym/optimizers.py:440-464
//   MomentumOptimizer.apply_gradient  adorym/optimizers.py:376-411
//   constraints / support mask        adorym/ptychography.py:1135-1158, adorym/array_ops.py:239-251
#pragma once
#include <hip/hip_runtime.h>
#include "adm_common.h"

namespace adm {

__device__ __forceinline__ float constrain(float xv, size_t i, int flags, const float* mask) {
    if ((flags & ADM_FLAG_NONNEG) && xv < 0.f) xv = 0.f;
    if ((flags & ADM_FLAG_ZERO_CH0) && !(i & 1)) xv *= 0.f;
    if ((flags & ADM_FLAG_ZERO_CH1) && (i & 1)) xv *= 0.f;
    if (mask) xv *= mask[i >> 1];
    return xv;
}

struct AdamScalars {
    float step, b1, b2, omb1, omb2, q1, q2, eps;
    int flags;
    const float* mask;
};

// one element of AdamOptimizer.apply_gradient + constraints
__device__ __forceinline__ float adam_value(float xv, float gv, float m_in, float v_in, const AdamScalars& a, size_t i, float& m_out,
                                            float& v_out) {
#pragma clang fp contract(off)
    float mv = a.b1 * m_in;
    mv = mv + a.omb1 * gv;
    float vv = a.b2 * v_in;
    vv = vv + a.omb2 * (gv * gv);
    const float mhat = mv / a.q1;
    const float vhat = vv / a.q2;
    const float d = a.step * mhat / (sqrtf(vhat) + a.eps);
    m_out = mv;
    v_out = vv;
    return constrain(xv - d, i, a.flags, a.mask);
}

__device__ __forceinline__ float gd_value(float xv, float gv, float step, size_t i, int flags, const float* mask) {
#pragma clang fp contract(off)
    return constrain(xv - step * gv, i, flags, mask);
}

__device__ __forceinline__ float momentum_value(float xv, float gv, float v_in, float step, float gamma, size_t i, int flags,
                                                const float* mask, float& v_out) {
#pragma clang fp contract(off)
    const float vv = gamma * v_in + step * gv;
    v_out = vv;
    return constrain(xv - vv, i, flags, mask);
}

static inline AdamScalars adam_scalars(int i_batch, double step_size, double b1, double b2, double eps, int flags, const float* mask) {
    // the reference evaluates the scalars in Python doubles and torch casts them to fp32 at the op
    AdamScalars a;
    double p1 = 1.0, p2 = 1.0;
    for (int k = 0; k < i_batch + 1; ++k) { p1 *= b1; p2 *= b2; }
    a.step = (float)step_size; a.b1 = (float)b1; a.b2 = (float)b2;
    a.omb1 = (float)(1.0 - b1); a.omb2 = (float)(1.0 - b2);
    a.q1 = (float)(1.0 - p1); a.q2 = (float)(1.0 - p2);
    a.eps = (float)eps; a.flags = flags; a.mask = mask;
    return a;
}

}  // namespace adm
