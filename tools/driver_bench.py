#!/usr/bin/env python3
"""Wall-clock per minibatch of the DRIVER (reconstruct_ptychography) on config 3's shape, to compare with bench.py's
engine-level loop: python tools/driver_bench.py [n_theta]"""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adorym_amd as A
from adorym_amd import workloads as W

n_theta = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfg = W.c3_config()
r = np.random.default_rng(0)
n_pos = len(cfg['probe_pos'])
prj = (np.abs(r.standard_normal((n_theta, n_pos, 72, 72))) * 30).astype(np.float32)
g = W.random_guess(cfg['obj_size'], seed=1)
with tempfile.TemporaryDirectory() as td:
    for rep in range(2):        # first run pays the one-off set-up (rotation tables, CSR build)
        t0 = time.time()
        st = A.reconstruct_ptychography(fname=prj, obj_size=cfg['obj_size'], probe_pos=cfg['probe_pos'], theta_st=0, theta_end=2 * np.pi,
                                        n_theta=n_theta, energy_ev=cfg['energy_ev'], psize_cm=cfg['psize_cm'], free_prop_cm='inf',
                                        minibatch_size=32, n_epochs=1, alpha_d=cfg['alpha_d'], alpha_b=cfg['alpha_b'], gamma=cfg['gamma'],
                                        learning_rate=cfg['learning_rate'], optimizer='adam', probe_type='gaussian', probe_mag_sigma=6,
                                        probe_phase_sigma=6, probe_phase_max=0.5, initial_guess=[g[..., 0], g[..., 1]], save_path=td,
                                        output_folder='d%d' % rep, store_checkpoint=False, use_checkpoint=False, return_state=True)
        dt = time.time() - t0
        nb = len(st['losses'])
        print('run %d: %d minibatches in %.2f s total (%.2f ms per minibatch incl. set-up and output)' % (rep, nb, dt, 1e3 * dt / nb))
        ts = np.array([float(l.split(',')[3]) for l in open(os.path.join(st['output_folder'], 'convergence', 'loss_rank_0.txt')).read().strip().split('\n')[1:]])
        d = np.diff(ts) * 1e3
        print('   steady state from the convergence log: median %.2f ms, min %.2f ms per minibatch' % (np.median(d), d.min()))
# per-minibatch timestamps from the convergence log would include set-up; report the steady-state from the log spacing
