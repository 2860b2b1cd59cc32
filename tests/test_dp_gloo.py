"""CPU, world_size = 2 over gloo: the N>1 path of the data-parallel exchange + sharded update
(adorym_amd/dp.py + adorym_amd/comm.py) against a single-process update with the summed gradient --
the reference's semantics (`gradient.arr = comm.allreduce(gradient.arr)`, adorym/ptychography.py:1113-1114).
The element-wise kernels are replaced by a NumPy stand-in (tests may use the oracle); the product uses HipOps."""
import os
import socket
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class NumpyOps(object):
    """Same interface as adorym_amd.dp.HipOps on host buffers (torch CPU tensors / ndarrays)."""

    def wrap(self, tensor, n):
        return tensor.numpy()

    def alloc(self, n):
        return np.zeros(n, np.float32)

    def zero(self, buf):
        buf[...] = 0

    def copy(self, dst, dst_off, src, src_off, n):
        dst[dst_off:dst_off + n] = src[src_off:src_off + n]

    def adam(self, x, g, g_base, m, v, mv_base, lo, hi, i_batch, step_size, b1, b2, eps, flags, mask):
        from oracle import adorym_oracle as O
        xs, ms, vs = O.adam_step(x[lo:hi], g[lo - g_base:hi - g_base], m[lo - mv_base:hi - mv_base], v[lo - mv_base:hi - mv_base],
                                 i_batch, step_size, b1, b2, eps)
        if flags & 1:
            xs = np.clip(xs, 0, None)
        x[lo:hi] = xs; m[lo - mv_base:hi - mv_base] = ms; v[lo - mv_base:hi - mv_base] = vs

    def gd(self, x, g, g_base, lo, hi, step_size, flags, mask):
        x[lo:hi] = x[lo:hi] - np.float32(step_size) * g[lo - g_base:hi - g_base]


def _worker(rank, world, port, shape, seed, out_q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from torch_comm import TorchComm
    from adorym_amd.dp import DataParallelObject
    comm = TorchComm('gloo')
    try:
        st = DataParallelObject(NumpyOps(), comm, shape)
        n = st.n
        r = np.random.default_rng(seed)
        x0 = (r.standard_normal(n) * 1e-3).astype(np.float32)
        st.obj[:n] = x0
        for it in range(3):
            g_all = [np.random.default_rng(100 * it + k).standard_normal(n).astype(np.float32) for k in range(world)]
            st.zero_grad()
            st.grad[:n] += g_all[rank]                       # this rank's minibatch gradient
            st.exchange_and_update('adam', it, {'step_size': 1e-4}, flags=1)
        st.zero_grad()
        st.grad[:n] += 1.0
        st.exchange_and_update('gd', 0, {'step_size': 1e-5})
        out_q.put((rank, np.array(st.obj[:n]), st.lo, st.hi, comm.max_over_ranks(rank), comm.sum_over_ranks(1.0)))
    finally:
        comm.close()


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize('shape', [(4, 5, 6, 2), (3, 3, 3, 2)])     # second: n not divisible by 2*world -> padded shards
def test_reduce_scatter_adam_allgather_world2(shape):
    import torch.multiprocessing as mp
    from oracle import adorym_oracle as O
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, shape, 7, q)) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=120) for _ in procs]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    # single-process reference: Adam with the SUM of the per-rank gradients
    n = int(np.prod(shape))
    x = (np.random.default_rng(7).standard_normal(n) * 1e-3).astype(np.float32)
    m = np.zeros_like(x); v = np.zeros_like(x)
    for it in range(3):
        g = sum(np.random.default_rng(100 * it + k).standard_normal(n).astype(np.float32) for k in range(world))
        x, m, v = O.adam_step(x, g, m, v, it, step_size=1e-4)
        x = np.clip(x, 0, None)
    x = x - np.float32(1e-5) * np.float32(world)          # gd step with summed all-ones gradient
    los = sorted(r[2] for r in res)
    assert los[0] == 0 and all(r[3] > r[2] for r in res)
    for rank, obj, lo, hi, mx, sm in res:
        assert np.allclose(obj, x, rtol=1e-6, atol=1e-9), rank      # every rank holds the identical updated object
        assert mx == world - 1 and sm == world


# ---- the p2p branch of DataParallelObject on CPU: a stand-in transport with P2PComm's interface (bind_object / fused_update) whose
# "peer buffers" travel over gloo, with the fused kernel's arithmetic restated in NumPy: rank-order sum on [sum_lo, sum_hi), the
# owner's own buffer elsewhere, optimiser on the shard, result written into every replica
def _p2p_worker(rank, world, port, shape, seed, restricted, out_q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from torch_comm import TorchComm
    from adorym_amd.dp import DataParallelObject
    from oracle import adorym_oracle as O
    import torch

    class Buf(np.ndarray):          # an ndarray that answers to the two DeviceArray members the p2p branch touches
        @property
        def ptr(self):
            return self.ctypes.data

    class NumpyOpsP2P(NumpyOps):
        def alloc(self, n):
            return np.zeros(n, np.float32).view(Buf)

    class FakeP2P(TorchComm):
        def __init__(self):
            TorchComm.__init__(self, 'gloo')
            self.backend = 'p2p'

        def bind_object(self, obj, grad, n):
            self.x, self.g, self.n_pad = obj, grad, n

        def _gather_all(self, arr):
            parts = [torch.empty(len(arr), dtype=torch.float32) for _ in range(self.size)]
            self.dist.all_gather(parts, torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32)))
            return [p_.numpy() for p_ in parts]

        def fused_update(self, kind, m, v, lo, hi, sum_lo, sum_hi, i_batch, step_size, b1, b2, eps, flags, mask):
            gs = self._gather_all(np.asarray(self.g))               # "reading the peers' gradient buffers"
            i = np.arange(lo, hi)
            ins = (i >= sum_lo) & (i < sum_hi)
            acc = gs[0][lo:hi].copy()
            for q in range(1, self.size):
                acc = acc + gs[q][lo:hi]
            gsum = np.where(ins, acc, gs[self.rank][lo:hi]).astype(np.float32)
            x = np.asarray(self.x)
            if kind == 0:
                xs, ms, vs = O.adam_step(x[lo:hi], gsum, np.asarray(m)[:hi - lo], np.asarray(v)[:hi - lo], i_batch, step_size, b1, b2, eps)
                if flags & 1:
                    xs = np.clip(xs, 0, None)
                m[:hi - lo] = ms; v[:hi - lo] = vs
            else:
                xs = x[lo:hi] - np.float32(step_size) * gsum
            mine = np.zeros(self.n_pad // self.size, np.float32)
            mine[:hi - lo] = xs
            full = np.concatenate(self._gather_all(mine))           # "writing every replica"
            self.x[:] = full

    comm = FakeP2P()
    try:
        st = DataParallelObject(NumpyOpsP2P(), comm, shape)
        assert st.p2p and st.inplace
        n = st.n
        x0 = (np.random.default_rng(seed).standard_normal(n) * 1e-3).astype(np.float32)
        st.obj[:n] = x0
        t_lo, t_hi = (2 * (n // 8), 2 * (3 * n // 8)) if restricted else (0, n)
        for it in range(3):
            g_all = [np.random.default_rng(100 * it + k).standard_normal(n).astype(np.float32) for k in range(world)]
            reg = (np.arange(n) % 5 - 2).astype(np.float32) * np.float32(0.1)
            if restricted:
                st.grad[:] = np.nan                                   # nothing outside the touched range may be read from a peer
                st.grad[t_lo:t_hi] = g_all[rank][t_lo:t_hi]

                def reg_shard(lo, hi, a_lo, a_hi):
                    i = np.arange(lo, hi)
                    ins = (i >= a_lo) & (i < a_hi)
                    cur = np.asarray(st.grad[lo:hi]).copy()
                    st.grad[lo:hi] = np.where(ins, cur + world * reg[lo:hi], world * reg[lo:hi])
                st.exchange_and_update('adam', it, {'step_size': 1e-4}, flags=1, touched=(t_lo, t_hi), reg_shard=reg_shard)
            else:
                st.grad[:n] = g_all[rank] + reg
                st.exchange_and_update('adam', it, {'step_size': 1e-4}, flags=1, first=(0, 10))
        st.grad[:n] = 1.0
        st.exchange_and_update('gd', 0, {'step_size': 1e-5})
        out_q.put((rank, np.array(st.obj[:n])))
    finally:
        comm.close()


@pytest.mark.parametrize('restricted', [False, True])
def test_p2p_branch_of_the_sharded_update_world2(restricted):
    """DataParallelObject with a transport of P2PComm's shape (backend 'p2p', bind_object, fused_update): full exchange and the
    footprint-restricted one (peers' buffers are NaN outside the touched range) against the single-process update with the summed
    gradient (adorym/ptychography.py:1113-1129; every rank adds its regulariser term, forward_model.py:138-139)."""
    import torch.multiprocessing as mp
    from oracle import adorym_oracle as O
    shape = (4, 5, 6, 2)
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_p2p_worker, args=(r, world, port, shape, 7, restricted, q)) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=120) for _ in procs]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    n = int(np.prod(shape))
    x = (np.random.default_rng(7).standard_normal(n) * 1e-3).astype(np.float32)
    m = np.zeros_like(x); v = np.zeros_like(x)
    t_lo, t_hi = (2 * (n // 8), 2 * (3 * n // 8)) if restricted else (0, n)
    reg = (np.arange(n) % 5 - 2).astype(np.float32) * np.float32(0.1)
    for it in range(3):
        g = sum(np.random.default_rng(100 * it + k).standard_normal(n).astype(np.float32) for k in range(world))
        data = np.zeros(n, np.float32)
        data[t_lo:t_hi] = g[t_lo:t_hi]                     # (full exchange: the whole range)
        x, m, v = O.adam_step(x, data + world * reg, m, v, it, step_size=1e-4)
        x = np.clip(x, 0, None)
    x = x - np.float32(1e-5) * np.float32(world)
    for rank, obj in res:
        assert np.all(np.isfinite(obj))
        assert np.allclose(obj, x, rtol=1e-5, atol=1e-8), rank
    assert np.array_equal(res[0][1], res[1][1])


def test_shard_bounds_cover_and_align():
    from adorym_amd.comm import shard_bounds
    for n in (2, 10, 54, 2 * 256 ** 3):
        for size in (1, 2, 4, 8):
            segs = [shard_bounds(n, size, r) for r in range(size)]
            assert segs[0][0] == 0 and segs[-1][1] == n
            for (a, b), (c, d) in zip(segs[:-1], segs[1:]):
                assert b == c and a % 2 == 0 and b % 2 == 0


def test_rank_slices_of_global_batch_match_reference_rule():
    """adorym/ptychography.py:905-908: rank r takes positions [r*mb, (r+1)*mb) of the global batch, sorted."""
    from oracle import adorym_oracle as O
    b = O.epoch_task_list(0, 3, 10, 4, n_ranks=2)
    for k in range(len(b)):
        full = b[k] if len(b[k]) == 8 else np.concatenate([b[k], b[0][:8 - len(b[k])]])
        for r in range(2):
            th, ind = O.rank_batch(b, k, r, 4, 2)
            assert th == full[r * 4, 0] and np.array_equal(ind, np.sort(full[r * 4:(r + 1) * 4, 1]))


def test_deferred_update_equals_full_update():
    """exchange_and_update(first=(lo, hi)) + finish_update() (the single-GPU stream plan: the next minibatch's planes are
    updated first, the rest later on the side stream) is the same element-wise update as one full pass."""
    sys.path.insert(0, ROOT)
    from adorym_amd.comm import LocalComm
    from adorym_amd.dp import DataParallelObject
    shape = (6, 5, 4, 2)
    r = np.random.default_rng(3)
    ref = DataParallelObject(NumpyOps(), LocalComm(), shape)
    sp = DataParallelObject(NumpyOps(), LocalComm(), shape)
    n = ref.n
    x0 = (r.standard_normal(n) * 1e-3).astype(np.float32)
    ref.obj[:n] = x0
    sp.obj[:n] = x0
    plane = 5 * 4 * 2
    for it in range(4):
        g = r.standard_normal(n).astype(np.float32)
        for st in (ref, sp):
            st.zero_grad()                     # also flushes a deferred update
            st.grad[:n] += g
        ref.exchange_and_update('adam', it, {'step_size': 1e-4}, flags=1)
        lo, hi = (it % 3) * plane, ((it % 3) + 2) * plane
        sp.exchange_and_update('adam', it, {'step_size': 1e-4}, flags=1, first=(lo, hi))
        assert np.array_equal(sp.obj[lo:hi], ref.obj[lo:hi])           # the prioritised planes are final
        if it == 1:
            assert not np.array_equal(sp.obj[:n], ref.obj[:n])        # ... the rest is still pending
    sp.finish_update()
    assert np.array_equal(sp.obj[:n], ref.obj[:n])
    for a, b in zip(sp.moments, ref.moments):
        assert np.array_equal(a[:n], b[:n])


# ------------------------------------------------------------------------------------------------------------------
# The in-place exchange of the RCCL backend (adorym_amd/dp.py, `inplace`): the reduced shard lands in its slot of the gradient
# buffer and the updated shard is gathered from its slot of the object.  A stand-in with RcclComm's interface (backend
# 'rccl', buffers with .view(offset, shape)) runs the SAME DataParallelObject code on host buffers, its control plane being the product's TCP star.
class HostBuf(object):
    def __init__(self, a):
        self.a = a
        self.size = a.size

    def view(self, off, shape):
        return HostBuf(self.a[off:off + int(np.prod(shape))])

    def __getitem__(self, k):
        return self.a[k]

    def __setitem__(self, k, v):
        self.a[k] = v


class HostOps(NumpyOps):
    def wrap(self, t, n):
        return t

    def alloc(self, n):
        return HostBuf(np.zeros(n, np.float32))

    def zero(self, buf):
        buf.a[...] = 0

    def adam(self, x, g, g_base, m, v, mv_base, lo, hi, *a):
        NumpyOps.adam(self, x.a, g.a, g_base, m.a, v.a, mv_base, lo, hi, *a)

    def gd(self, x, g, g_base, lo, hi, *a):
        NumpyOps.gd(self, x.a, g.a, g_base, lo, hi, *a)


def _worker_inplace(rank, world, port, shape, seed, out_q, first=None):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    if first is not None:
        os.environ['ADM_OVERLAP_GATHER'] = '1'      # the two-part gather is opt-in at world size > 1
    from adorym_amd.comm import RcclComm
    from adorym_amd.dp import DataParallelObject

    class HostAsRccl(RcclComm):          # control plane = the real RcclComm code (TCP star); data plane = the star's array collectives on host buffers
        def reduce_scatter_sum(self, full, shard_out):
            t = self.group_.all_reduce_sum(full.a.copy())
            shard_out.a[...] = t[self.rank * shard_out.size:(self.rank + 1) * shard_out.size]

        def all_gather(self, full_out, shard_in):
            full_out.a[...] = self.group_.all_gather(np.ascontiguousarray(shard_in.a))

        def group(self):
            import contextlib
            return contextlib.nullcontext()      # no launch grouping on the host; the calls simply run one after the other

        def broadcast(self, dev, root):
            dev.a[...] = self.group_.broadcast(np.ascontiguousarray(dev.a).copy(), root)

    comm = HostAsRccl()
    try:
        st = DataParallelObject(HostOps(), comm, shape)
        assert st.inplace
        n = st.n
        r = np.random.default_rng(seed)
        st.obj.a[:n] = (r.standard_normal(n) * 1e-3).astype(np.float32)
        for it in range(3):
            g_all = [np.random.default_rng(100 * it + k).standard_normal(n).astype(np.float32) for k in range(world)]
            st.zero_grad()
            st.grad.a[:n] += g_all[rank]
            st.exchange_and_update('adam', it, {'step_size': 1e-4}, flags=1, first=first)
            if first is not None:
                whole = first[0] <= 0 and first[1] >= n          # everything = the plain all-gather, nothing deferred
                assert st.overlap_gather and st._gather_pending == (not whole)
                # only the planes in `first` are current everywhere now; the rest of the gather is pending
                early = np.array(st.obj.a[first[0]:first[1]])
                other = 1 - rank
                stale = np.array(st.obj.a[other * st.per:(other + 1) * st.per])      # the other rank's shard as we hold it
        if first is not None:
            st.finish_update()
            assert not st._gather_pending
            assert np.array_equal(early, st.obj.a[first[0]:first[1]])             # what was gathered first was already final
            if not whole:
                lo, hi = max(other * st.per, first[0]), min((other + 1) * st.per, first[1])
                outside = np.ones(st.per, bool)
                outside[max(0, lo - other * st.per):max(0, hi - other * st.per)] = False
                assert not np.array_equal(stale[outside], st.obj.a[other * st.per:(other + 1) * st.per][outside])   # ... the rest was not
        seeds = comm.bcast_object(4242 if rank == 0 else None, root=0)
        comm.barrier()
        out_q.put((rank, np.array(st.obj.a[:n]), comm.max_over_ranks(rank), comm.sum_over_ranks(1.0), seeds))
    finally:
        comm.close()


@pytest.mark.parametrize('shape,first', [((4, 5, 6, 2), None), ((3, 3, 3, 2), None), ((8, 5, 6, 2), (100, 300)), ((8, 5, 6, 2), (250, 400)),
                                         ((3, 3, 3, 2), (0, 54))])
def test_inplace_exchange_of_the_rccl_backend_world2(shape, first):
    """first=(lo, hi): the planes the next minibatches read are broadcast from their owners right after the update and the
    full all-gather is deferred to finish_update() (the driver queues it beside the next kernel); (0, n) = everything =
    the plain all-gather."""
    import torch.multiprocessing as mp
    from oracle import adorym_oracle as O
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_inplace, args=(r, world, port, shape, 7, q, first)) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=120) for _ in procs]
    [p.join(60) for p in procs]
    n = int(np.prod(shape))
    x = (np.random.default_rng(7).standard_normal(n) * 1e-3).astype(np.float32)
    m = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
    for it in range(3):
        g = sum(np.random.default_rng(100 * it + k).standard_normal(n).astype(np.float32) for k in range(world))
        x, m, v = O.adam_step(x, g, m, v, it, 1e-4)
        x = np.clip(x, 0, None)
    for rank, obj, mx, sm, seed in res:
        assert np.allclose(obj, x, rtol=2e-6, atol=1e-9)
        assert (mx, sm, seed) == (world - 1, float(world), 4242)
    assert np.array_equal(res[0][1], res[1][1])          # every rank holds the same object after the gather


def test_product_task_split_matches_reference_golden_both_ranks():
    """The DRIVER's own task-list code (adorym_amd/util.py epoch_task_list / rank_batch, called by reconstruct_ptychography)
    against golden F8 captured from the reference for 1 and 2 ranks (adorym/ptychography.py:791-847, 897-908), and rank 1's
    share against the reference rule."""
    sys.path.insert(0, ROOT)
    from adorym_amd.util import epoch_task_list, rank_batch
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'F8_tasks.npz'))
    n_theta, n_pos, mb = 5, 7, 3
    for n_ranks in (1, 2):
        seen_theta, seen_ind = [], []
        for e in (0, 1):
            batches = epoch_task_list(e, n_theta, n_pos, mb, n_ranks)
            assert len(batches) == int(g['r%d_e%d_ntask' % (n_ranks, e)])
            for k, b in enumerate(batches):
                assert np.array_equal(b, g['r%d_e%d_task_%d' % (n_ranks, e, k)])
            for k in range(len(batches)):
                th, ind = rank_batch(batches, k, 0, mb, n_ranks)
                seen_theta.append(th); seen_ind.append(ind)
                if n_ranks == 2:
                    full = batches[k]
                    assert len(full) == 2 * mb                      # topped up in place by rank_batch
                    th1, ind1 = rank_batch(batches, k, 1, mb, n_ranks)
                    assert th1 == full[mb, 0] and np.array_equal(ind1, np.sort(full[mb:2 * mb, 1]))
                    assert len(set(full[mb:, 0])) == 1              # all of a rank's pairs share one angle
        assert np.array_equal(np.array(seen_theta), g['r%d_rank0_theta' % n_ranks])
        assert np.array_equal(np.stack(seen_ind), g['r%d_rank0_ind' % n_ranks])
    # config 3's padded minibatch is the deterministic set SURVEY R17 describes
    last = epoch_task_list(0, 2, 529, 32, 1)[16]
    assert sorted(last[:, 1].tolist()) == list(range(0, 15)) + list(range(512, 529))


# ------------------------------------------------------------------------------------------ footprint-restricted exchange
def _worker_restricted(rank, world, port, shape, out_q):
    """Two updates on a [Y,X,Z,2] object: every rank's DATA gradient lives on its own footprint planes inside the union
    [t0, t1) of the ranks' footprints, the regulariser term (the same on every rank: it depends on the replicated object only)
    everywhere.  Full exchange: buffers = regulariser + data, reduce-scatter.  Restricted: buffers = data on [t0, t1) and
    garbage elsewhere, per-owner reductions over [t0, t1), regulariser added R-fold by the owner."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from torch_comm import TorchComm
    from adorym_amd.dp import DataParallelObject
    from oracle import adorym_oracle as O
    comm = TorchComm('gloo')
    try:
        Y, X, Z, _ = shape
        plane = X * Z * 2
        n = int(np.prod(shape))
        x0 = (np.random.default_rng(3).standard_normal(n) * 1e-3).astype(np.float32)
        a_d, a_b, gam = 1e-2, 1e-3, 1e-2

        def reg(x):
            o = x[:n].reshape(shape).astype(np.float32)
            return (O.l1_value_grad(o, a_d, a_b)[1] + O.tv_value_grad(o, gam)[1]).astype(np.float32).reshape(-1)

        res = {}
        for mode in ('full', 'restricted'):
            st = DataParallelObject(NumpyOps(), comm, shape)
            st.obj[:n] = x0
            for it in range(2):
                foot = [(1 + it, 4 + it), (3 + it, 6 + it)]                   # rank footprints (planes), overlapping
                t0, t1 = min(f[0] for f in foot) * plane, max(f[1] for f in foot) * plane
                data = np.zeros(n, np.float32)
                lo, hi = foot[rank][0] * plane, foot[rank][1] * plane
                data[lo:hi] = np.random.default_rng(50 * it + rank).standard_normal(hi - lo).astype(np.float32) * 1e-3
                if mode == 'full':
                    st.grad[:n] = reg(st.obj) + data
                    st.exchange_and_update('adam', it, {'step_size': 1e-4})
                else:
                    st.grad[:] = np.float32(np.nan)                            # whatever is outside [t0, t1) must not matter
                    st.grad[t0:t1] = data[t0:t1]
                    r_now = reg(st.obj) * np.float32(world)

                    def reg_shard(s_lo, s_hi, a_lo, a_hi, r_now=r_now, st=st):
                        s_hi = min(s_hi, n)
                        idx = np.arange(s_lo, s_hi)
                        add = (idx >= a_lo) & (idx < a_hi)
                        st.grad[idx[add]] += r_now[idx[add]]
                        st.grad[idx[~add]] = r_now[idx[~add]]
                    st.exchange_and_update('adam', it, {'step_size': 1e-4}, touched=(t0, t1), reg_shard=reg_shard)
            res[mode] = np.array(st.obj[:n])
        out_q.put((rank, res['full'], res['restricted']))
    finally:
        comm.close()


def test_restricted_exchange_equals_full_exchange_world2():
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    shape = (9, 4, 5, 2)            # n = 360: shard boundaries fall inside a plane (40 elements per plane)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_restricted, args=(r, world, port, shape, q)) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=120) for _ in procs]
    [p.join(timeout=60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for rank, full, restricted in res:
        assert np.all(np.isfinite(restricted))
        # same sums up to the order of two fp32 additions per element: Adam (lr 1e-4) turns that into <= a few 1e-10
        assert np.abs(restricted - full).max() <= 2e-3 * 1e-4, np.abs(restricted - full).max()
    assert np.array_equal(res[0][2], res[1][2])                  # replicas identical


def test_restricted_exchange_on_one_local_rank_completes_the_buffer():
    """exchange_and_update(touched=..., reg_shard=...) with ONE local rank (no communicator at all) used to take the
    ordinary path and let Adam consume the undefined part of the gradient buffer.  The contract is the same at every rank count:
    the buffer holds the data term on the touched planes only, the owner -- here the only rank -- completes it with the
    regulariser term before the update."""
    sys.path.insert(0, ROOT)
    from adorym_amd.comm import LocalComm
    from adorym_amd.dp import DataParallelObject
    from oracle import adorym_oracle as O
    shape = (6, 3, 4, 2)
    n = int(np.prod(shape))
    plane = n // shape[0]
    x0 = (np.random.default_rng(5).standard_normal(n) * 1e-3).astype(np.float32)
    reg = (np.random.default_rng(6).standard_normal(n) * 1e-3).astype(np.float32)
    data = np.zeros(n, np.float32)
    t0, t1 = 2 * plane, 5 * plane
    data[t0:t1] = (np.random.default_rng(7).standard_normal(t1 - t0) * 1e-3).astype(np.float32)
    st = DataParallelObject(NumpyOps(), LocalComm(), shape)
    assert not st.dist
    st.obj[:n] = x0
    st.grad[:] = np.float32(np.nan)                  # whatever is outside [t0, t1) must not matter
    st.grad[t0:t1] = data[t0:t1]
    calls = []

    def reg_shard(s_lo, s_hi, a_lo, a_hi):
        calls.append((s_lo, s_hi, a_lo, a_hi))
        idx = np.arange(s_lo, min(s_hi, n))
        add = (idx >= a_lo) & (idx < a_hi)
        st.grad[idx[add]] += reg[idx[add]]
        st.grad[idx[~add]] = reg[idx[~add]]

    st.exchange_and_update('adam', 0, {'step_size': 1e-4}, touched=(t0, t1), reg_shard=reg_shard)
    assert calls == [(0, n, t0, t1)]
    ref, _, _ = O.adam_step(x0, data + reg, np.zeros(n, np.float32), np.zeros(n, np.float32), 0, 1e-4)
    assert np.all(np.isfinite(st.obj[:n])) and np.allclose(st.obj[:n], ref, rtol=1e-6, atol=1e-10)
    with pytest.raises(ValueError):
        st.exchange_and_update('adam', 1, {'step_size': 1e-4}, touched=(t0, t1))
