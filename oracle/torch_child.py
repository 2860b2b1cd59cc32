"""
TEST / BASELINE INFRASTRUCTURE ONLY -- never imported by the product (adorym_amd/).

oracle/torch_structured.py's loss_and_grad() run in a child interpreter.  This module imports no torch itself: the GPU tests
use it so that the process holding the libadm context stays free of PyTorch (a PyTorch-ROCm wheel loads its own copy of the
HIP runtime by an unversioned library name; with libadm.so loaded first the process ends up with two runtimes and aborts at
exit, after every test has passed).
"""
import numpy as np


def loss_and_grad_subprocess(obj_rot, pos, probe, h, k1, meas, threads=None, timeout=1800):
    """loss_and_grad() in a CHILD interpreter (inputs and results through an .npz in a temporary directory), for callers
    whose own process holds a libadm GPU context and must therefore not import torch (a PyTorch-ROCm wheel loads its own
    copy of the HIP runtime; two in one process abort at exit).  This function itself needs no torch."""
    import os
    import subprocess
    import sys
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, 'in.npz'), os.path.join(td, 'out.npz')
        np.savez(fin, obj_rot=obj_rot, pos=np.asarray(pos), probe=probe, h=h, k1=np.float64(k1), meas=meas,
                 threads=np.int64(threads or 0))
        r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'torch_structured.py'), fin, fout], capture_output=True, text=True, timeout=timeout)
        if r.returncode != 0:
            raise RuntimeError('torch_structured child failed (%d): %s' % (r.returncode, r.stderr[-600:]))
        f = np.load(fout)
        return float(f['loss']), f['grad']
