"""fp32 drift at config 3's depth (P = 72, 256 slices) against the reference's fp64 results and its OWN fp32 errors (golden F17):
the body of tests/test_gpu_variants.py::test_depth_256_against_the_references_own_fp32_error as a tool, for A/B runs of kernel
variants.  ADM_LIB_PATH=adorym_amd/libadm_X.so python tools/depth_err.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import numpy as np
import adorym_amd as A
import cases


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - b) / np.linalg.norm(b))


g = np.load(os.path.join(ROOT, 'tests', 'golden', 'F17_depth256.npz'))
d = cases.depth256_inputs()
P = d['P']
Y, X, S = d['obj'].shape[:3]
ctx = A.Context(0)
eng = A.MultisliceEngine(ctx, (Y, X, S), (P, P), d['pos'], cases.ENERGY_EV, cases.PSIZE_CM)
d_obj = ctx.array(d['obj'], np.float32)
d_grad = ctx.zeros(d['obj'].shape)
d_probe = ctx.array(np.stack([d['probe'].real, d['probe'].imag], -1)[None], np.float32)
eng.set_batch(d['pos'], g['target'].astype(np.float32))
eng.rotate(d_obj, None)
eng.multislice(d_probe, want_pred=True)
eng.rotate_adjoint(d_grad, None)
e_pred = rel(eng.pred(), g['pred_64'])
e_loss = abs(eng.loss() - float(g['loss_64'])) / float(g['loss_64'])
e_grad = rel(d_grad.get()[::4, ::4, ::4], g['grad_64_sample'])
r = (float(g['ref32_pred_err']), float(g['ref32_loss_err']), float(g['ref32_grad_sample_err']))
print('%s depth 256 vs reference fp64: pred %.2e (x%.2f of reference fp32 %.2e), loss %.2e (x%.2f of %.2e), grad %.2e (x%.2f of %.2e)'
      % (os.environ.get('ADM_LIB_PATH', 'libadm.so'), e_pred, e_pred / r[0], r[0], e_loss, e_loss / r[1], r[1], e_grad, e_grad / r[2], r[2]))
