#!/usr/bin/env python3
"""
TEST / BASELINE INFRASTRUCTURE ONLY -- never imported by the product (adorym_amd/).

Second CPU-baseline flavour of bench.py (SURVEY.md 8d ii): the reference's own op structure -- PyTorch-CPU tensors, separate
re/im, strided slice selects, torch.autograd.grad -- restated in oracle/torch_structured.py (timed within 15 % of the imported
reference in the build container: oracle/time_vs_reference.py).  A minibatch of B = 4 positions of config 3 (full 256^3
object: tile gather + forward + autograd backward; the size SURVEY / BASELINE.md quote the reference at) is timed for torch
thread counts 8 ... host cores, the best is reported with its thread count, then -- time permitting -- a larger minibatch at
that count.  Runs as a CHILD process of bench.py: the benchmark's GPU process never imports torch (a PyTorch-ROCm wheel
brings its own copy of the HIP runtime, and two of them in one process do not shut down cleanly).  Prints one JSON line.

    python oracle/torch_structured_bench.py [seconds_budget]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    budget_s = float(sys.argv[1]) if len(sys.argv) > 1 else 75.0
    import torch
    from oracle import adorym_oracle as O
    from oracle import torch_structured as T
    from adorym_amd.workloads import c3_config, probe_array          # host-side NumPy generators only (no device layer)
    cfg = c3_config()
    phys = O.Physics(cfg['probe_size'], cfg['energy_ev'], cfg['psize_cm'], free_prop_cm=cfg['free_prop_cm'])
    pa = probe_array(cfg)
    probe = pa[..., 0] + 1j * pa[..., 1]
    Y, X, Z = cfg['obj_size']
    r = np.random.default_rng(1)
    obj = np.stack([r.normal(8.7e-7, 1e-7, (Y, X, Z)), r.normal(5.1e-8, 1e-8, (Y, X, Z))], -1).astype(np.float32)
    n_cpu = os.cpu_count() or 1
    t_start = time.perf_counter()

    def run(nb, threads):
        torch.set_num_threads(threads)
        pos = cfg['probe_pos'][200:200 + nb].astype(int)
        meas = np.abs(np.random.default_rng(2).standard_normal((nb,) + tuple(cfg['probe_size']))).astype(np.float32)
        t0 = time.perf_counter()
        T.loss_and_grad(obj, pos, probe, phys.h, phys.k1, meas)
        return time.perf_counter() - t0

    T.loss_and_grad(obj[:, :, :4], cfg['probe_pos'][200:202].astype(int), probe, phys.h, phys.k1,
                    np.ones((2,) + tuple(cfg['probe_size']), np.float32))          # warm-up (thread pool, FFT plans)
    sweep = []
    for th in [t for t in (8, 16, 32, 64, 128) if t <= max(8, n_cpu)]:
        if sweep and time.perf_counter() - t_start > 0.55 * budget_s:
            break
        dt = run(4, min(th, n_cpu))
        sweep.append({'threads': min(th, n_cpu), 'positions': 4, 'seconds': dt, 'positions_per_s': 4 / dt})
    best = max(sweep, key=lambda q: q['positions_per_s'])
    out = {'value': best['positions_per_s'], 'unit': 'probe-positions/s', 'cores': best['threads'], 'kind': 'port', 'thread_sweep': sweep,
           'sample': '4 positions of config 3 (256^3 object, P=72, 256 slices): tile gather + fwd + torch.autograd backward, fp32, '
                     'reference op structure (oracle/torch_structured.py), %.1f s at the best of the thread counts tried, torch %s'
                     % (best['seconds'], torch.__version__)}
    left = budget_s - (time.perf_counter() - t_start)
    nb = int(min(32, 0.8 * left * best['positions_per_s']))
    if nb >= 8:
        dt = run(nb, best['threads'])
        out['larger_minibatch'] = {'threads': best['threads'], 'positions': nb, 'seconds': dt, 'positions_per_s': nb / dt}
        if nb / dt > out['value']:
            out['value'] = nb / dt
    print(json.dumps(out))


if __name__ == '__main__':
    main()
