#!/usr/bin/env python3
"""The DRIVER on the config-1 shape (2-D ptychography 618 x 606 x 1 real_imag, P = 64, 5 probe modes, minibatch 35, intensity data,
probe + sub-pixel position refinement, TV): wall time per minibatch of adorym_amd.reconstruct_ptychography itself (mean spacing of the
convergence log) and, with `profile` as first argument, a cProfile of its host side.   python tools/driver_c1.py [profile]"""
import os, sys, tempfile, cProfile, pstats, io, contextlib, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adorym_amd as A

Y, X, P, M, B = 618, 606, 64, 5, 35
r = np.random.default_rng(0)
energy, psize = 8801.121930115722, 1.32789376566526e-06
pos = np.array([(y, x) for y in range(-20, Y - 40, 16) for x in range(-20, X - 40, 16)], dtype=float)
pos += r.uniform(-0.4, 0.4, pos.shape)
prj = (np.abs(r.standard_normal((1, len(pos), P, P), dtype=np.float32)) * 50) ** 2
probe = r.standard_normal((M, P, P)) + 1j * r.standard_normal((M, P, P))
prof = len(sys.argv) > 1 and sys.argv[1] == 'profile'
pr = cProfile.Profile()
with tempfile.TemporaryDirectory() as td, open(os.devnull, 'w') as sink, contextlib.redirect_stdout(sink):
    if prof:
        pr.enable()
    t0 = time.perf_counter()
    st = A.reconstruct_ptychography(fname=prj, obj_size=(Y, X, 1), probe_pos=pos, energy_ev=energy, psize_cm=psize, free_prop_cm='inf',
                                    raw_data_type='intensity', unknown_type='real_imag', minibatch_size=B, n_epochs=3, optimizer='adam',
                                    learning_rate=1e-3, initial_guess=[np.ones((Y, X, 1)), np.zeros((Y, X, 1))], probe_type='supplied',
                                    probe_initial=[np.abs(probe), np.angle(probe)], n_probe_modes=M, optimize_probe=True,
                                    probe_learning_rate=1e-3, optimize_all_probe_pos=True, all_probe_pos_learning_rate=1e-2, gamma=1e-6,
                                    alpha_d=0, alpha_b=0, save_path=td, output_folder='c1', store_checkpoint=False, use_checkpoint=False,
                                    return_state=True)
    wall = time.perf_counter() - t0
    if prof:
        pr.disable()
    lines = open(os.path.join(st['output_folder'], 'convergence', 'loss_rank_0.txt')).read().strip().split('\n')[1:]
ts = np.array([float(l.split(',')[3]) for l in lines])
n = len(ts)
ms = 1e3 * (ts[-1] - ts[n // 3]) / (n - 1 - n // 3)          # epochs 2 and 3 (the first one meets every position set for the first time)
print('driver, config-1 shape: %d minibatches of %d, %.3f ms per minibatch = %.0f positions/s (wall %.2f s incl. setup)' % (n, B, ms, B / (ms * 1e-3), wall))
if prof:
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(25)
    print(s.getvalue()[:7000])
