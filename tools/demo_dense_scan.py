#!/usr/bin/env python3
"""demos/2d_ptychography_w_probe_optimization.py of the reference AT ITS OWN SIZE on synthetic data: 256 x 256 x 1 phase-only object,
72 x 72 probe estimated from the data (probe_type='ifft'), 52 x 52 = 2704 positions 5 pixels apart taken as ONE minibatch (up to
~210 tiles on a pixel: the multi-pass overlap-add), Adam on object + probe + sub-pixel position corrections.  Prints one JSON line:
ms per update (= per epoch), positions/s, the loss at the first and the last update.

    python tools/demo_dense_scan.py [n_epochs]
"""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import adorym_amd as A                      # noqa: E402


def main():
    n_epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    r = np.random.default_rng(5)
    N, P = 256, 72
    pos = np.array([(y, x) for y in np.arange(-10, 246, 5) for x in np.arange(-10, 246, 5)], dtype=float)
    # synthetic data: the product's own forward model on a smooth phase object (predict), so that the run has something to fit
    yy, xx = np.mgrid[0:N, 0:N] / N
    delta = 2e-5 * (np.sin(6 * yy) * np.cos(5 * xx) + 1)[..., None]
    g = np.exp(-(((np.arange(P) - P / 2 + 0.5) / 12.) ** 2))
    probe = (g[:, None] * g[None, :]) * np.exp(0.3j * (g[:, None] + g[None, :]))
    ctx = A.Context(0)
    eng = A.MultisliceEngine(ctx, (N, N, 1), (P, P), np.round(pos).astype(int), 5000., 1e-7, max_batch=len(pos))
    obj = ctx.array(np.stack([delta, np.zeros_like(delta)], -1).astype(np.float32))
    eng.set_batch(np.round(pos).astype(int), np.zeros((len(pos), P, P), np.float32))
    eng.rotate(obj, None, None)
    eng.multislice(ctx.array(np.stack([probe.real, probe.imag], -1)[None].astype(np.float32)), want_grad=False, want_pred=True)
    prj = eng.pred()[None].astype(np.float32)
    del eng
    ctx.close()
    with tempfile.TemporaryDirectory() as td:
        t0 = time.perf_counter()
        st = A.reconstruct_ptychography(
            fname=prj, theta_st=0, theta_end=0, n_epochs=n_epochs, obj_size=(N, N, 1), alpha_d=0, alpha_b=0, gamma=0, learning_rate=2e-7,
            energy_ev=5000, psize_cm=1.e-7, minibatch_size=len(pos), output_folder='recon', save_path=td, save_intermediate=False,
            initial_guess=None, n_dp_batch=20, probe_type='ifft', optimize_probe=True, object_type='phase_only', probe_pos=pos,
            free_prop_cm='inf', optimizer='adam', two_d_mode=True, use_checkpoint=False, store_checkpoint=False, optimize_all_probe_pos=True,
            raw_data_type='magnitude', return_state=True)
        wall = time.perf_counter() - t0
        with open(os.path.join(td, 'recon', 'convergence', 'loss_rank_0.txt')) as f:
            ts = np.array([float(l.split(',')[3]) for l in f.read().strip().split('\n')[1:]])
    dt = float(np.median(np.diff(ts))) if len(ts) > 2 else wall / n_epochs
    print(json.dumps({'workload': 'dense 2-D scan as one minibatch: 256x256x1 phase-only, probe 72x72 from the data, 2704 positions, object+probe+position Adam',
                      'ms_per_update': 1e3 * dt, 'positions_per_s': len(pos) / dt, 'updates': n_epochs, 'wall_s_incl_setup': wall,
                      'loss_first': st['losses'][0], 'loss_last': st['losses'][-1]}))


if __name__ == '__main__':
    main()
