#!/usr/bin/env python3
"""Do the bandwidth-bound kernels of a step's tail run faster SIDE BY SIDE than one after the other?  (the
y-band pipeline of the tail was argued away, not measured.)  Config 3's shape, B positions of one angle (default 256):
the overlap-add, the rotation adjoint and the whole-object Adam pass are timed alone, back to back on one stream, and as two
concurrent streams (overlap-add on the main stream, the other two on the side stream, on buffers that do not depend on it).
    python tools/tail_overlap_probe.py [B]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adorym_amd as A                                  # noqa: E402
from adorym_amd import workloads as W                   # noqa: E402
from adorym_amd.dp import HipOps                        # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = W.c3_config()
ctx = A.Context(0)
eng = A.MultisliceEngine(ctx, cfg['obj_size'], cfg['probe_size'], cfg['probe_pos'], cfg['energy_ev'], cfg['psize_cm'], max_batch=B,
                         transmissions_only=True)
obj = ctx.array(W.random_guess(cfg['obj_size'], seed=1))
probe = ctx.array(W.probe_array(cfg))
pos = cfg['probe_pos'][(np.arange(B) + 100) % 529]
table = A.RotationTable(ctx, cfg['obj_size'], np.float32(0.6))
table.csr(eng.plan)
eng.set_batch(pos, np.ones((B, 72, 72), np.float32))
eng.rotate(obj, table, None)
eng.multislice(probe, accumulate=False)
n = obj.size
g, g2 = ctx.zeros(obj.shape), ctx.zeros(obj.shape)
x2, m, v = ctx.array(W.random_guess(cfg['obj_size'], seed=2)), ctx.zeros((n,)), ctx.zeros((n,))
ops = HipOps(ctx)
yr = eng.y_footprint(pos)


def oa():
    eng.accumulate_tiles()


def ra():
    eng.rotate_adjoint(g2, table, yr)


def adam():
    ops.adam(x2, g2, 0, m, v, 0, 0, n, 0, 1e-5, 0.9, 0.999, 1e-7, 0, None)


def timed(fn, reps=5):
    ctx.sync()
    ts = []
    for _ in range(reps):
        ctx.sync()
        t0 = time.perf_counter()
        fn()
        ctx.sync()
        ts.append(1e3 * (time.perf_counter() - t0))
    return float(np.median(ts))


def serial():
    oa(); ra(); adam()


def side_by_side():
    ctx.fork()
    ra(); adam()                # independent of the overlap-add in THIS probe (grad_rot rows of an earlier call)
    ctx.end_fork()
    oa()
    ctx.join()


for f in (oa, ra, adam, serial, side_by_side):
    f()
t = {f.__name__: timed(f) for f in (oa, ra, adam, serial, side_by_side)}
print('B = %d positions, planes %s: overlap-add %.3f ms, rotation adjoint %.3f, Adam (whole object) %.3f; one after the other %.3f; '
      'overlap-add beside (rotation adjoint + Adam) %.3f ms  [host-timed with a sync on both sides: +~0.02 ms each]'
      % (B, yr, t['oa'], t['ra'], t['adam'], t['serial'], t['side_by_side']))
