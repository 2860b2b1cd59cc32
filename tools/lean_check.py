"""Throughput kernel vs latency kernel on the same inputs (config-3 shape): loss sums, predictions, stored fields,
per-position tile gradients, overlap-added gradient.  python tools/lean_check.py [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import adorym_amd as A
from adorym_amd import workloads as W
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = W.c3_config()
ctx = A.Context(0)
eng = A.MultisliceEngine(ctx, cfg['obj_size'], cfg['probe_size'], cfg['probe_pos'], cfg['energy_ev'], cfg['psize_cm'], max_batch=B)
# well-conditioned problem: data = forward of a structured truth, guess = a scaled copy of it
truth = W.foam_object(tuple(cfg['obj_size']), seed=0)
probe = ctx.array(W.probe_array(cfg))
pos = cfg['probe_pos'][(np.arange(B) * 7 + 200) % len(cfg['probe_pos'])]
eng.set_batch(pos, np.zeros((B, 72, 72), np.float32))
eng.plan.set_lean_min_batch(0)
eng.rotate(ctx.array(truth), None)
eng.multislice(probe, want_grad=False, want_pred=True)
target = eng.pred().copy()
eng.set_batch(pos, target)
obj = ctx.array((0.7 * truth).astype(np.float32))
eng.rotate(obj, None)
S, R1, NT = 256, 8, 704
per = S * R1 * NT
res = []
for lean in (0, 1):
    eng.plan.set_lean_min_batch(lean)
    eng.multislice(probe, want_grad=True, want_pred=True)
    ctx.sync()
    ws = eng._ws.get().view(np.float32)
    stash = ws[:B * per * 2].reshape(B, S, R1, NT, 2).copy()
    gt = ws[B * per * 2:2 * B * per * 2].reshape(B, S, R1, NT, 2).copy()
    res.append((eng._loss.view(0, (B,)).get().copy(), eng.pred().copy(), eng.grad_rot.get().copy(), stash, gt))
(l0, p0, g0, s0, t0), (l1, p1, g1, s1, t1) = res
rel = lambda a, b: np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
print('loss rel diff max', np.abs(l1 - l0).max() / np.abs(l0).max())
print('pred rel-L2', rel(p1, p0))
print('stash rel-L2', rel(s1, s0), ' per step (0,1,128,254,255):', [float(rel(s1[:, s], s0[:, s])) for s in (0, 1, 128, 254, 255)])
print('gtile rel-L2', rel(t1, t0), ' per step (255,254,128,1,0):', [float(rel(t1[:, s], t0[:, s])) for s in (255, 254, 128, 1, 0)])
print('gtile per k at step 200:', [float(rel(t1[:, 200, k], t0[:, 200, k])) for k in range(R1)])
print('grad_rot rel-L2', rel(g1, g0))
ok = np.abs(l1 - l0).max() / np.abs(l0).max() < 1e-5 and rel(g1, g0) < 1e-4
print('LEAN_CHECK', 'OK' if ok else 'FAIL')
