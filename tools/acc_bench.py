import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, adorym_amd as A
from adorym_amd import workloads as W
cfg = W.c3_config(); ctx = A.Context(0)
B = int(sys.argv[1])
eng = A.MultisliceEngine(ctx, cfg['obj_size'], cfg['probe_size'], cfg['probe_pos'], cfg['energy_ev'], cfg['psize_cm'], max_batch=B)
obj = ctx.array(W.random_guess(cfg['obj_size'], seed=1)); probe = ctx.array(W.probe_array(cfg))
pos = cfg['probe_pos'][np.arange(B) % 529]
eng.set_batch(pos, np.ones((B, 72, 72), np.float32)); eng.rotate(obj, None)
e = [ctx.event() for _ in range(3)]
for r in range(3):
    e[0].record(); eng.multislice(probe, accumulate=False); e[1].record(); eng.accumulate_tiles(); e[2].record()
    t1, t2 = e[0].elapsed_ms(e[1]), e[1].elapsed_ms(e[2])
print(os.environ.get('ADM_LIB_PATH', 'default'), 'B', B, 'ms kernel %.3f  accumulate %.3f' % (t1, t2))
