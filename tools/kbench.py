"""Time the multislice kernel alone at the C3 shape (B positions, one theta).  Experiment helper:
ADM_LIB_PATH=adorym_amd/libadm_X.so python tools/kbench.py [B] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import adorym_amd as A
from adorym_amd import workloads as W

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
fwd_only = len(sys.argv) > 3 and sys.argv[3] == 'fwd'
cfg = W.c3_config()
ctx = A.Context(0)
eng = A.MultisliceEngine(ctx, cfg['obj_size'], cfg['probe_size'], cfg['probe_pos'], cfg['energy_ev'], cfg['psize_cm'], max_batch=B,
                         transmission_cache=os.environ.get('ADM_TCACHE', '1') == '1')      # ADM_TCACHE=0: exp / sincos in the slice loop
Y, X, Z = cfg['obj_size']
obj = ctx.array(W.random_guess((Y, X, Z), seed=1))
probe = ctx.array(W.probe_array(cfg))
pos = cfg['probe_pos'][np.arange(B) % len(cfg['probe_pos'])] if B > 200 else cfg['probe_pos'][200:200 + B]
eng.set_batch(pos, np.abs(np.random.default_rng(0).standard_normal((B, 72, 72))).astype(np.float32) * 30)
eng.rotate(obj, None)
e0, e1 = ctx.event(), ctx.event()
ts = []
for r in range(reps + 1):
    e0.record()
    eng.multislice(probe, accumulate=False, want_grad=not fwd_only)
    e1.record()
    ts.append(e0.elapsed_ms(e1))
print('%s B=%d %s: kernel ms min %.3f median %.3f  -> %.0f pos/s' % (os.environ.get('ADM_LIB_PATH', 'libadm.so'), B,
      'fwd' if fwd_only else 'fwd+adj', min(ts[1:]), float(np.median(ts[1:])), B / (min(ts[1:]) * 1e-3)))
