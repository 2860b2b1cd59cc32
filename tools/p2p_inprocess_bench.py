#!/usr/bin/env python3
"""The peer-to-peer exchange with R ranks as R CONTEXTS of ONE process on one GPU (raw pointers instead of IPC mappings:
adm_p2p_connect / adm_p2p_bind_object take plain device pointers): every rank has its own stream, flag block, object, gradient
and moment shard, and the host queues the R calls of adm_p2p_update one after the other -- rank 0's wait kernel spins until the
host has queued the last rank's signal.  No process per rank, so the GPU's hardware queues are not oversubscribed at R = 8 (as
they are with 8 processes on one GPU): this times the exchange's kernel-side cost at config 4's sizes (object 134 MB, shard
134 / R MB) with every "link" being local HBM, and checks the result against the rank-order sum.
    python tools/p2p_inprocess_bench.py [R] [reps]"""
import ctypes as C
import os
import sys
import time

import numpy as np

# HIP maps the streams of ONE process onto at most 4 hardware queues by default: with 2 R streams in this process a spinning wait
# kernel would block the stream that carries the signal it waits for (R = 8 timed out after 30 s).  One process per rank -- the
# product -- has two streams and never meets this.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '24')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adorym_amd as A                                  # noqa: E402
from adorym_amd._lib import check, OPT_ADAM             # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n = 2 * 256 ** 3
per = n // R
ctxs = [A.Context(0) for _ in range(R)]
lib = ctxs[0].lib
r = np.random.default_rng(0)
x0 = (r.standard_normal(n) * 1e-3).astype(np.float32)
xs = [c.array(x0) for c in ctxs]
g_host = [r.standard_normal(n).astype(np.float32) for _ in range(min(R, 2))]
gs = [c.array(g_host[q % len(g_host)] * np.float32(1 + q)) for q, c in enumerate(ctxs)]
ms = [c.zeros((per,)) for c in ctxs]
vs = [c.zeros((per,)) for c in ctxs]
flags, mails = [], []
for q, c in enumerate(ctxs):
    check(lib.adm_p2p_create(c.handle, q, R, 1 << 20))
    p = C.c_void_p()
    check(lib.adm_p2p_local(c.handle, 0, C.byref(p))); flags.append(p.value)
    check(lib.adm_p2p_local(c.handle, 1, C.byref(p))); mails.append(p.value)
arr = lambda ptrs: (C.c_void_p * R)(*[C.c_void_p(v) for v in ptrs])
for c in ctxs:
    check(lib.adm_p2p_connect(c.handle, arr(flags), arr(mails)))
    check(lib.adm_p2p_bind_object(c.handle, arr([x.ptr for x in xs]), arr([g.ptr for g in gs]), n))


def exchange(k):
    for q, c in enumerate(ctxs):
        check(lib.adm_p2p_update(c.handle, OPT_ADAM, ms[q].ptr, vs[q].ptr, q * per, (q + 1) * per, 0, n, k, 1e-4, 0.9, 0.999, 1e-7, 0, None))


exchange(0)
for c in ctxs:
    c.sync()
    check(lib.adm_p2p_status(c.handle))
# check: rank-order sum + Adam step 0 (m = (1-b1) g, v = (1-b2) g^2, bias-corrected: x - lr * g / (|g| + eps))
tot = g_host[0] * np.float32(1)
for q in range(1, R):
    tot = tot + g_host[q % len(g_host)] * np.float32(1 + q)
from oracle import adorym_oracle as O                   # checker only
want, _, _ = O.adam_step(x0, tot, np.zeros(n, np.float32), np.zeros(n, np.float32), 0, 1e-4)
got = [x.get() for x in xs]
err = max(float(np.abs(gq - want).max()) for gq in got)
same = all(np.array_equal(got[0], gq) for gq in got[1:])
ts = []
for k in range(1, reps + 1):
    for c in ctxs:
        c.sync()
    t0 = time.perf_counter()
    exchange(k)
    for c in ctxs:
        c.sync()
    ts.append(1e3 * (time.perf_counter() - t0))
for c in ctxs:
    check(lib.adm_p2p_status(c.handle))
t = float(np.median(ts))
local = 7 * per * 4 + (R - 1) * per * 4 * 2          # per rank: g, m, v, x read + m, v, x written on the shard, + peers' g read and x written
print('R = %d contexts on one GPU, object %d MB, shard %.1f MB: exchange %.3f ms (median of %d, host-timed incl. ~%d launches and two syncs); '
      '%.0f MB moved per rank -> %.2f TB/s aggregate; replicas identical: %s; max |x - oracle| %.2e'
      % (R, n * 4 >> 20, per * 4 / 2 ** 20, t, reps, 7 * R, local / 2 ** 20, R * local / (t * 1e-3) / 1e12, same, err))
for c in ctxs:
    check(lib.adm_p2p_destroy(c.handle))
