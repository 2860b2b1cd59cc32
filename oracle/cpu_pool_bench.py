#!/usr/bin/env python3
"""
TEST / BASELINE INFRASTRUCTURE ONLY -- never imported by the product (adorym_amd/).

The NumPy oracle's forward + hand adjoint of the multislice chain (oracle/adorym_oracle.py, fp32: the reference's dtype)
timed on ALL host cores: a pool of worker processes, each running whole probe positions of config 3's shape (P = 72,
256 slices) -- positions are independent, so this is how the port scales over cores.  Started by bench.py as a CHILD process
(a fresh interpreter that never touches the GPU, so forking a pool is safe) and prints one JSON line.

    python oracle/cpu_pool_bench.py [n_procs] [seconds_budget]
"""
import json
import os
import sys
import time

os.environ.setdefault('OMP_NUM_THREADS', '1')          # one core per worker: the pool is the parallelism
os.environ.setdefault('OPENBLAS_NUM_THREADS', '1')
os.environ.setdefault('MKL_NUM_THREADS', '1')
import numpy as np                                       # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import adorym_oracle as O                    # noqa: E402

P, S = 72, 256
_state = {}


def _init():
    r = np.random.default_rng(os.getpid())
    _state['phys'] = O.Physics((P, P), 5000., 1e-7, free_prop_cm='inf')
    yy, xx = np.mgrid[:P, :P] - (P - 1) / 2.
    mag = np.exp(-(xx ** 2 + yy ** 2) / (2 * 6. ** 2))
    _state['probe'] = mag * np.exp(0.5j * mag)
    _state['tiles'] = np.stack([r.normal(8.7e-7, 1e-7, (1, P, P, S)), r.normal(5.1e-8, 1e-8, (1, P, P, S))], -1).astype(np.float32)
    _state['meas'] = np.abs(r.standard_normal((1, P, P))).astype(np.float32)


def _one(_):
    t0 = time.perf_counter()
    O.forward_adjoint_tiles(_state['tiles'], _state['probe'], _state['meas'], _state['phys'], 'float32')
    return time.perf_counter() - t0


def main():
    import multiprocessing as mp
    n_cpu = os.cpu_count() or 1
    n_procs = int(sys.argv[1]) if len(sys.argv) > 1 else n_cpu
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
    with mp.get_context('fork').Pool(n_procs, initializer=_init) as pool:
        pool.map(_one, range(n_procs))                                   # warm-up: FFT plans, page faults
        t1 = sum(pool.map(_one, range(n_procs))) / n_procs               # seconds per position per worker, pool loaded
        n = max(n_procs, int(budget / t1) * n_procs)
        n = min(n, 64 * n_procs)
        t0 = time.perf_counter()
        pool.map(_one, range(n), chunksize=1)
        dt = time.perf_counter() - t0
    print(json.dumps({'value': n / dt, 'unit': 'probe-positions/s', 'cores': n_procs, 'kind': 'port', 'positions': n, 'seconds': dt,
                      'seconds_per_position_per_worker': t1, 'host_cores': n_cpu, 'numpy': np.__version__}))


if __name__ == '__main__':
    main()
