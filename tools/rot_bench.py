"""Time the rotation adjoint alone at config 3's shape: python tools/rot_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, adorym_amd as A
from adorym_amd import workloads as W
cfg = W.c3_config(); ctx = A.Context(0)
eng = A.MultisliceEngine(ctx, cfg['obj_size'], cfg['probe_size'], cfg['probe_pos'], cfg['energy_ev'], cfg['psize_cm'], max_batch=32)
g = ctx.zeros((256, 256, 256, 2))
eng.grad_rot.set(np.random.default_rng(0).standard_normal(eng.grad_rot.shape).astype(np.float32))
thetas = np.linspace(0, 2 * np.pi, 500, dtype='float32')
e0, e1 = ctx.event(), ctx.event()
out = []
for it in (0, 20, 41, 62, 83, 104, 125):
    tab = A.RotationTable(ctx, cfg['obj_size'], thetas[it]); tab.csr(eng.plan)
    ts = []
    for r in range(4):
        e0.record(); eng.rotate_adjoint(g, tab, (60, 168)); e1.record(); ts.append(e0.elapsed_ms(e1))
    out.append('%.2f rad: %.0f us' % (thetas[it], 1e3 * min(ts[1:])))
print(os.environ.get('ADM_LIB_PATH', 'default'), ' | '.join(out))
