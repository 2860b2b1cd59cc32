#!/usr/bin/env python3
"""cProfile of the driver's host side on config 3's shape: python tools/driver_profile.py [n_theta] [update_scheme]"""
import os, sys, tempfile, cProfile, pstats, io, contextlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adorym_amd as A
from adorym_amd import workloads as W
n_theta = int(sys.argv[1]) if len(sys.argv) > 1 else 4
scheme = sys.argv[2] if len(sys.argv) > 2 else 'immediate'
cfg = W.c3_config()
r = np.random.default_rng(0)
prj = (np.abs(r.standard_normal((n_theta, len(cfg['probe_pos']), 72, 72), dtype=np.float32)) * 30)
g = W.random_guess(cfg['obj_size'], seed=1)
pr = cProfile.Profile()
with tempfile.TemporaryDirectory() as td, open(os.devnull, 'w') as sink, contextlib.redirect_stdout(sink):
    pr.enable()
    A.reconstruct_ptychography(fname=prj, obj_size=cfg['obj_size'], probe_pos=cfg['probe_pos'], theta_st=0, theta_end=2 * np.pi,
                               n_theta=n_theta, energy_ev=cfg['energy_ev'], psize_cm=cfg['psize_cm'], free_prop_cm='inf',
                               minibatch_size=32, n_epochs=1, alpha_d=cfg['alpha_d'], alpha_b=cfg['alpha_b'], gamma=cfg['gamma'],
                               learning_rate=cfg['learning_rate'], optimizer='adam', initial_guess=[g[..., 0], g[..., 1]], save_path=td,
                               output_folder='d', store_checkpoint=False, use_checkpoint=False, update_scheme=scheme, **cfg['probe'])
    pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22)
print(s.getvalue()[:6000])
