"""Diagnostic (ADM_STAMPS build): per-wave shader-clock stamps of slice step 100 of workgroup 0.
ADM_LIB_PATH=adorym_amd/libadm_stamps.so python tools/stamps.py [B]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import adorym_amd as A
from adorym_amd import workloads as W
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = W.c3_config()
ctx = A.Context(0)
eng = A.MultisliceEngine(ctx, cfg['obj_size'], cfg['probe_size'], cfg['probe_pos'], cfg['energy_ev'], cfg['psize_cm'], max_batch=B)
obj = ctx.array(W.random_guess(tuple(cfg['obj_size']), seed=1))
probe = ctx.array(W.probe_array(cfg))
pos = cfg['probe_pos'][np.arange(B) % len(cfg['probe_pos'])]
eng.set_batch(pos, np.abs(np.random.default_rng(0).standard_normal((B, 72, 72))).astype(np.float32) * 30)
eng.rotate(obj, None)
for r in range(3):
    eng.multislice(probe, accumulate=False, want_grad=True)
ctx.sync()
buf = np.zeros(16 * 16, np.uint64)
lib = ctx.lib
lib.adm_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
lib.adm_debug_read_stamps(buf.ctypes.data, buf.nbytes)
s = buf.reshape(16, 16).astype(np.int64)
def show(names, idx, title):
    t0 = s[:11, idx[0]].min()
    print(title)
    print('wave ' + ' '.join('%9s' % n for n in names))
    for w in range(11):
        print('%4d ' % w + ' '.join('%9d' % (s[w, i] - t0) for i in idx))
print('B=%d  stamps relative to the earliest wave, shader cycles' % B)
show(['start', 'mod', 'stores', 'loads', 'conv'], [0, 1, 2, 3, 4], 'forward step 100')
show(['start', 'grad+st', 'mod', 'loads', 'conv'], [8, 9, 10, 11, 12], 'reverse step 100')
