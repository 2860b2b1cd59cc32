#!/usr/bin/env python3
"""
Development-container check (needs /root/reference; never runs on the GPU box): times the real reference's
fwd + autograd backward on a C3-shape minibatch and the from-scratch restatement oracle/torch_structured.py on
the same inputs, so that the latter can stand in as the "reference PyTorch-CPU path" timing in bench.py
(SURVEY.md 8d asks for +-15 %).  Also prints the numerical agreement of loss and gradient.

    python oracle/time_vs_reference.py [B] [S]         # defaults: B=4 positions, S=256 slices, P=72
"""
import os
import sys
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
sys.path.insert(0, ROOT)
import gen_goldens as G          # installs the h5py/dxchange shims and imports the reference  # noqa: E402
import torch                     # noqa: E402
from oracle import torch_structured as T   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
P, Y = 72, 256
torch.set_num_threads(os.cpu_count())
r = np.random.default_rng(0)
obj = np.stack([r.normal(8.7e-7, 1e-7, (Y, Y, S)), r.normal(5.1e-8, 1e-8, (Y, Y, S))], -1).astype(np.float32)
pos = np.array([[-36 + 12 * i, 60 + 24 * i] for i in range(B)])
probe = (r.standard_normal((P, P)) + 1j * r.standard_normal((P, P))).astype(np.complex64)
meas = np.abs(r.standard_normal((B, P, P))).astype(np.float32)
lm = 1240. / 5000.
vox = np.array([1e-7] * 3) * 1e7
h = G.get_kernel(1., lm, vox, (P, P), fresnel_approx=True, sign_convention=1)
k1 = 2 * 3.14159265359 * 1. / lm


def run_reference():
    o = torch.tensor(obj, requires_grad=True)
    t0 = time.perf_counter()
    op, pad = G.U.pad_object(o, [Y, Y, S], pos, [P, P], unknown_type='delta_beta')
    tiles = torch.stack([op[y + pad[0, 0]:y + pad[0, 0] + P, x + pad[1, 0]:x + pad[1, 0] + P] for y, x in pos])
    er, ei = G.multislice_propagate_batch(tiles, torch.tensor(probe.real.copy()), torch.tensor(probe.imag.copy()), 5000., 1e-7,
                                          kernel=h, free_prop_cm='inf', obj_batch_shape=[B, P, P, S])
    loss = torch.mean((G.w.norm(er, ei) - torch.tensor(meas)) ** 2)
    t1 = time.perf_counter()
    g, = torch.autograd.grad(loss, [o])
    t2 = time.perf_counter()
    return float(loss.detach()), g.numpy(), t1 - t0, t2 - t1


def run_structured():
    t0 = time.perf_counter()
    l, g = T.loss_and_grad(obj, pos, probe, h, k1, meas)
    return l, g, time.perf_counter() - t0


run_reference(); run_structured()            # warm-up (allocator, FFT plans)
REPS = int(os.environ.get('REPS', '5'))
tr_, ts_ = [], []
for _ in range(REPS):                        # interleaved, so that drifts of the shared 8-core container hit both alike
    lr, gr, tf, tb = run_reference(); tr_.append((tf + tb, tf, tb))
    ls, gs_, ts = run_structured(); ts_.append(ts)
med_r, med_s = float(np.median([a[0] for a in tr_])), float(np.median(ts_))
print('all runs: reference', ['%.2f' % a[0] for a in tr_], 'structured', ['%.2f' % a for a in ts_])
print('median: reference %.2f s, structured %.2f s, ratio %.3f   (SURVEY 8d asks for 1 +- 0.15)' % (med_r, med_s, med_s / med_r))
(_, tf, tb), ts = min(tr_), min(ts_)
print('best:   reference fwd %.2f s + bwd %.2f s = %.2f s, structured %.2f s, ratio %.3f  (%d threads, B = %d, S = %d)'
      % (tf, tb, tf + tb, ts, ts / (tf + tb), torch.get_num_threads(), B, S))
print('loss rel diff %.2e, grad rel-L2 %.2e' % (abs(lr - ls) / abs(lr), np.linalg.norm(gr - gs_) / np.linalg.norm(gr)))
