"""GPU diagnostic script (not collected by pytest; lives under tests/ because it uses the oracle): coherent energy drift of
the propagation chain (object = 0 => c = 1 exactly).  python tests/diag_energy.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import adorym_amd as A
from oracle import adorym_oracle as O

ctx = A.Context(0)
r = np.random.default_rng(0)
for P in (16, 32, 64, 72, 12):
    for S in (2, 64, 256):
        pos = np.array([(0, 0)])
        eng = A.MultisliceEngine(ctx, (P, P, S), (P, P), pos, 5000., 1e-7)
        obj = np.zeros((P, P, S, 2), np.float32)
        probe = (0.5 + r.uniform(0, 1, (P, P))) * np.exp(1j * r.uniform(-np.pi, np.pi, (P, P)))
        d_probe = ctx.array(np.stack([probe.real, probe.imag], -1), np.float32)
        eng.set_batch(pos, np.zeros((1, P, P), np.float32))
        eng.rotate(ctx.array(obj), None)
        eng.multislice(d_probe, want_grad=False, want_pred=True)
        pred = eng.pred().astype(np.float64)
        phys = O.Physics((P, P), 5000., 1e-7)
        tiles = obj[None].astype(np.float64)
        p64 = np.abs(O.multislice_forward(tiles, probe, phys, 'float64'))
        p32 = np.abs(O.multislice_forward(tiles.astype(np.float32), probe, phys, 'float32'))
        rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
        e_in = P * P * np.sum(np.abs(probe.astype(np.complex64)) ** 2)
        print('P=%3d S=%3d  energy drift gpu %+.2e  cpu32 %+.2e | pred err gpu %.2e cpu32 %.2e' % (
            P, S, (pred ** 2).sum() / e_in - 1, (p32.astype(np.float64) ** 2).sum() / e_in - 1, rel(pred[0], p64[0]), rel(p32[0], p64[0])))
