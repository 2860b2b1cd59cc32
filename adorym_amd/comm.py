"""
Communication seam of the data-parallel (DP) mode.  The reference does
``gradient.arr = comm.allreduce(gradient.arr)`` with mpi4py pickles (adorym/ptychography.py:1113-1114)
and then applies the identical optimiser step on every rank.  Here, one process per GPU:

    reduce_scatter(sum) of the object gradient  ->  fused Adam on the owned shard (moments are
    sharded, ZeRO-1 style)  ->  all_gather of the updated object shards

Backends (all expose the same methods to adorym_amd/dp.py and the driver):
  LocalComm            1 rank, no torch import: the identity collectives of adorym/pseudo.py;
  RcclComm             the product's multi-GPU backend.  Data plane: RCCL over xGMI behind libadm's C ABI
                       (adm_reduce_scatter / adm_all_gather / adm_all_reduce / adm_broadcast, librccl dlopen'ed), enqueued on
                       the context's own streams, IN PLACE on libadm's own device buffers -- two communicators, one per
                       stream (main; side stream for the deferred part of the object all-gather).  Control plane: the
                       TCP star of adorym_amd/rendezvous.py (rank 0 listens next to MASTER_PORT): RCCL unique ids, seeds,
                       barriers, the resume agreement.  No framework anywhere in this path;
  HostStagedComm       validation backend (ADM_COMM=host): the same in-place contract on the same libadm buffers, but every
                       collective is staged through host memory (adm_d2h -> TCP star -> adm_h2d).  It exists so that the PRODUCT
                       -- driver, HIP kernels, sharded optimiser, two-part gather -- can run at world size > 1 with several
                       ranks on ONE GPU, where RCCL refuses duplicate devices.  Slow by construction; never the default;
  P2PComm              direct all-pairs exchange without RCCL (ADM_COMM=p2p): every rank maps the object and gradient buffers of
                       its peers (IPC handles handed over the TCP star) and ONE kernel per update sums the ranks' gradients of
                       the owned shard in rank order, applies the optimiser in registers and writes every replica
                       (adm_p2p_update: reduce-scatter + optimiser + all-gather in one pass); ranks are ordered against each
                       other by device-side flags, no host synchronisation.  Several ranks may share one GPU;
(No class here imports torch: the torch.distributed stand-in the CPU sharding tests use lives in tests/torch_comm.py.)
"""
import os
import numpy as np


class LocalComm(object):
    """Single rank: the identity collectives of adorym/pseudo.py:27-60."""
    rank = 0
    size = 1
    backend = 'local'

    def barrier(self):
        pass

    def shard_range(self, n):
        return 0, n

    def max_over_ranks(self, value):
        return value

    def sum_over_ranks(self, value):
        return value

    def bcast_object(self, obj, root=0):
        return obj


def shard_bounds(n, size, rank, align=2):
    """Contiguous shard [lo, hi) of a flat array of n elements; boundaries are multiples of `align`
    (2 keeps a voxel's (delta, beta) pair on one rank)."""
    per = -(-n // size)
    per = -(-per // align) * align
    lo = min(rank * per, n)
    hi = min(lo + per, n)
    return lo, hi


class RcclComm(object):
    """One process per GPU; data plane = RCCL behind the C ABI, control plane = the TCP star of adorym_amd/rendezvous.py
    (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT as set by torch.distributed.run, an mpirun wrapper or
    bench.py's own launcher) -- no framework in the product's multi-GPU path (the reference's seam: mpi4py,
    adorym/ptychography.py:39-50).  ``attach(ctx)`` must be called once with the rank's Context before the first device
    collective (the driver does)."""
    backend = 'rccl'

    def __init__(self, device_index=None, group=None):
        from .rendezvous import TcpGroup
        self.group_ = group if group is not None else TcpGroup.from_env()
        self.rank = self.group_.rank
        self.size = self.group_.size
        self.device_index = int(os.environ.get('LOCAL_RANK', '0')) if device_index is None else int(device_index)
        self.ctx = None

    def attach(self, ctx):
        """Create the RCCL communicators of this rank on ``ctx`` (collective over all ranks).  Every step that can fail on one
        rank only WITHOUT blocking the others is followed by an agreement over the control plane, so that either every rank
        returns with working communicators or every rank raises the same RuntimeError (callers such as bench.py then fall back
        TOGETHER).  Not covered: a rank that dies INSIDE ncclCommInitRank leaves its peers blocked in theirs until RCCL's own
        timeout -- a rendezvous cannot be made symmetric from one side.  Order: (1) each rank probes its own librccl (adm_comm_available) -> agree; (2) rank 0 creates the
        unique ids and ALWAYS broadcasts (None on failure) -> all ranks see the same outcome; (3) adm_comm_init (+ the
        side-stream communicator) -> agree, and tear down on disagreement."""
        import ctypes as C
        from ._lib import check
        if self.ctx is ctx:
            return self
        lib = ctx.lib
        why = ''
        try:
            check(lib.adm_comm_available())
            ok = 1.0
        except Exception as e:      # librccl missing / lacks a symbol: the most likely failure, identical on all ranks
            ok, why = 0.0, repr(e)
        if self.sum_over_ranks(ok) < self.size:
            raise RuntimeError('RCCL is not usable on every rank (this rank: %s)' % (why or 'ok'))
        ids = None
        if self.rank == 0:
            try:
                bufs = [C.create_string_buffer(128) for _ in range(2)]
                for b_ in bufs:
                    check(lib.adm_comm_unique_id(b_))
                ids = [bytes(b_.raw) for b_ in bufs]
            except Exception as e:
                why = repr(e)
        ids = self.bcast_object(ids, root=0)
        if ids is None:
            raise RuntimeError('rank 0 could not create the RCCL unique ids (%s)' % (why or 'see rank 0'))
        try:
            check(lib.adm_comm_init(ctx.handle, self.rank, self.size, C.create_string_buffer(ids[0], 128)))
            if os.environ.get('ADM_COMM_AUX', '1') == '1':
                check(lib.adm_comm_init_aux(ctx.handle, C.create_string_buffer(ids[1], 128)))
            ok = 1.0
        except Exception as e:
            ok, why = 0.0, repr(e)
        if self.sum_over_ranks(ok) < self.size:
            lib.adm_comm_destroy(ctx.handle)
            raise RuntimeError('RCCL communicator creation failed on some rank (this rank: %s)' % (why or 'ok'))
        self.ctx = ctx
        return self

    def stream_handle(self):
        return None             # the context owns its stream; RCCL is enqueued on it by libadm

    # ---- buffers the collectives touch: libadm device arrays ----
    def alloc(self, n, dtype=None):
        return self.ctx.zeros((int(n),))

    # ---- device collectives (asynchronous on the context's stream) ----
    def reduce_scatter_sum(self, full, shard_out):
        from ._lib import check
        check(self.ctx.lib.adm_reduce_scatter(self.ctx.handle, full.ptr, shard_out.ptr, shard_out.size))

    def all_gather(self, full_out, shard_in):
        from ._lib import check
        check(self.ctx.lib.adm_all_gather(self.ctx.handle, shard_in.ptr, full_out.ptr, shard_in.size))

    def all_reduce_device(self, dev):
        """In-place sum over ranks of a libadm device array (small parameter gradients: they never visit the host)."""
        from ._lib import check
        check(self.ctx.lib.adm_all_reduce(self.ctx.handle, dev.ptr, dev.size, 0))
        return dev

    def broadcast(self, dev, root):
        """In-place broadcast of a libadm device array (view) from rank ``root``."""
        from ._lib import check
        check(self.ctx.lib.adm_broadcast(self.ctx.handle, dev.ptr, dev.nbytes, int(root)))
        return dev

    def reduce(self, dev, root):
        """dev (a libadm device array / view) of rank ``root`` = sum over ranks of their ``dev``; the others' are left alone."""
        from ._lib import check
        check(self.ctx.lib.adm_reduce(self.ctx.handle, dev.ptr, dev.size, int(root)))
        return dev

    def group(self):
        """Context manager: the collectives issued inside are launched as one operation (ncclGroupStart / End)."""
        import contextlib
        from ._lib import check

        @contextlib.contextmanager
        def _g():
            check(self.ctx.lib.adm_comm_group_start(self.ctx.handle))
            try:
                yield
            finally:
                check(self.ctx.lib.adm_comm_group_end(self.ctx.handle))
        return _g()

    # ---- control plane (host) ----
    def barrier(self):
        if self.ctx is not None:
            self.ctx.sync()
        self.group_.barrier()

    def shard_range(self, n):
        return shard_bounds(n, self.size, self.rank)

    def max_over_ranks(self, value):
        return self.group_.max_over_ranks(value)

    def sum_over_ranks(self, value):
        return self.group_.sum_over_ranks(value)

    def bcast_object(self, obj, root=0):
        return self.group_.bcast_object(obj, root)

    def close(self, keep_group=False):
        """``keep_group``: leave the control plane up (bench.py hands it to the fallback transport)."""
        if self.ctx is not None:
            self.ctx.sync()
            self.ctx.lib.adm_comm_destroy(self.ctx.handle)
            self.ctx = None
        if not keep_group:
            self.group_.close()


class HostStagedComm(RcclComm):
    """VALIDATION backend: RcclComm's interface and in-place buffer contract with every device collective staged through
    host memory -- blocking adm_d2h, the control plane's array collective on the host copy (sums in rank order), blocking
    adm_h2d -- so that several ranks can share ONE GPU (RCCL refuses two ranks on one device).  The driver, the HIP kernels,
    the sharded optimiser and the two-part gather of DataParallelObject run unchanged; only the transport differs.  A
    collective issued between Context.fork() and end_fork() stages through the side stream, like RcclComm's would run on it.
    Selected with ADM_COMM=host."""
    backend = 'host'

    def attach(self, ctx):
        self.ctx = ctx
        return self

    def reduce_scatter_sum(self, full, shard_out):
        n = shard_out.size
        host = self.group_.all_reduce_sum(full.view(0, (n * self.size,)).get())
        shard_out.set(host[self.rank * n:(self.rank + 1) * n])

    def all_gather(self, full_out, shard_in):
        full_out.view(0, (shard_in.size * self.size,)).set(self.group_.all_gather(shard_in.get().reshape(-1)))

    def all_reduce_device(self, dev):
        dev.set(self.group_.all_reduce_sum(dev.get()))
        return dev

    def broadcast(self, dev, root):
        host = self.group_.broadcast(dev.get(), int(root))
        if self.rank != int(root):
            dev.set(host)
        return dev

    def reduce(self, dev, root):
        host = self.group_.reduce_sum(dev.get(), int(root))
        if self.rank == int(root):
            dev.set(host)
        return dev

    def group(self):
        import contextlib
        return contextlib.nullcontext()

    def close(self, keep_group=False):
        if self.ctx is not None:
            self.ctx.sync()
            self.ctx = None
        if not keep_group:
            self.group_.close()


class P2PComm(RcclComm):
    """Direct all-pairs exchange through IPC-mapped peer buffers (adm_p2p.hip); ADM_COMM=p2p.  Control plane = the TCP star
    (it carries the 64-byte IPC handles once, at attach / bind time); data plane = the ranks' own kernels reading and writing each
    other's device buffers, ordered by device-side flags.  The exchange of adorym/ptychography.py:1113-1129 is ONE call:
    ``fused_update`` (DataParallelObject uses it instead of reduce_scatter -> optimiser -> all_gather).  Sums are taken in
    rank order: the result equals HostStagedComm's and a serial sum of the ranks' buffers bit for bit.  Ranks may share a GPU
    (device index = LOCAL_RANK modulo the number of devices)."""
    backend = 'p2p'

    def __init__(self, device_index=None, group=None):
        RcclComm.__init__(self, device_index=device_index, group=group)
        if device_index is None:
            self.device_index = self.device_index % max(1, device_count())
        self._mapped = []           # peer allocations this rank has opened (closed in close())
        self.bound = None

    def _exchange_handles(self, arrays):
        """arrays: this rank's device pointers.  Returns, per array, the list over ranks of pointers valid in THIS process
        (own entry = own pointer).  Collective; either every rank returns or every rank raises."""
        import ctypes as C
        from ._lib import check, P2P_HANDLE_BYTES
        lib, h = self.ctx.lib, self.ctx.handle
        mine = np.zeros((len(arrays), P2P_HANDLE_BYTES), np.uint8)
        ok, why = 1.0, ''
        try:
            for k, ptr in enumerate(arrays):
                buf = C.create_string_buffer(P2P_HANDLE_BYTES)
                check(lib.adm_p2p_export(h, C.c_void_p(ptr), buf))
                mine[k] = np.frombuffer(buf.raw, np.uint8)
        except Exception as e:
            ok, why = 0.0, repr(e)
        if self.sum_over_ranks(ok) < self.size:
            raise RuntimeError('peer-to-peer transport: exporting the IPC handles failed on some rank (this rank: %s)' % (why or 'ok'))
        everyone = self.group_.all_gather(mine.reshape(1, -1)).reshape(self.size, len(arrays), P2P_HANDLE_BYTES)
        out = [[None] * self.size for _ in arrays]
        try:
            for q in range(self.size):
                for k, ptr in enumerate(arrays):
                    if q == self.rank:
                        out[k][q] = int(ptr)
                        continue
                    dp = C.c_void_p()
                    check(lib.adm_p2p_open(h, C.create_string_buffer(everyone[q, k].tobytes(), P2P_HANDLE_BYTES), C.byref(dp)))
                    self._mapped.append(dp.value)
                    out[k][q] = dp.value
        except Exception as e:
            ok, why = 0.0, repr(e)
        if self.sum_over_ranks(ok) < self.size:
            raise RuntimeError('peer-to-peer transport: mapping the peers\' buffers failed on some rank (this rank: %s)' % (why or 'ok'))
        return out

    @staticmethod
    def _ptr_array(ptrs):
        import ctypes as C
        return (C.c_void_p * len(ptrs))(*[C.c_void_p(p_) for p_ in ptrs])

    def attach(self, ctx):
        """Flag block + mailbox of this rank, exchanged with and mapped by every peer (collective)."""
        import ctypes as C
        from ._lib import check, P2P_MAX_RANKS
        if self.ctx is ctx:
            return self
        if self.size > P2P_MAX_RANKS:
            raise RuntimeError('peer-to-peer transport: at most %d ranks' % P2P_MAX_RANKS)
        self.ctx = ctx
        ok, why = 1.0, ''
        local = []
        try:
            check(ctx.lib.adm_p2p_create(ctx.handle, self.rank, self.size, int(os.environ.get('ADM_P2P_MAILBOX_BYTES', '0'))))
            for which in (0, 1):
                dp = C.c_void_p()
                check(ctx.lib.adm_p2p_local(ctx.handle, which, C.byref(dp)))
                local.append(dp.value)
        except Exception as e:
            ok, why = 0.0, repr(e)
        if self.sum_over_ranks(ok) < self.size:
            self.ctx = None
            ctx.lib.adm_p2p_destroy(ctx.handle)
            raise RuntimeError('peer-to-peer transport could not be created on every rank (this rank: %s)' % (why or 'ok'))
        if self.size > 1:
            flags, mail = self._exchange_handles(local)
            check(ctx.lib.adm_p2p_connect(ctx.handle, self._ptr_array(flags), self._ptr_array(mail)))
        self.group_.barrier()       # nobody signals into a flag block before every rank has zeroed and published its own
        return self

    def bind_object(self, obj, grad, n):
        """The object and gradient buffers of every rank, mapped here (collective; DataParallelObject calls it once)."""
        from ._lib import check
        if self.size > 1:
            xs, gs = self._exchange_handles([obj.ptr, grad.ptr])
        else:
            xs, gs = [obj.ptr], [grad.ptr]
        check(self.ctx.lib.adm_p2p_bind_object(self.ctx.handle, self._ptr_array(xs), self._ptr_array(gs), int(n)))
        self.bound = (obj, grad)

    # ---- device collectives (asynchronous on the context's stream) ----
    def fused_update(self, kind, m, v, lo, hi, sum_lo, sum_hi, i_batch, step_size, b1, b2, eps, flags, mask):
        """adm_p2p_update on the bound buffers: rank-order sum of the ranks' gradients on [lo, hi) n [sum_lo, sum_hi) (own
        gradient elsewhere), optimiser + constraints, result written to every replica."""
        from ._lib import check
        check(self.ctx.lib.adm_p2p_update(self.ctx.handle, int(kind), m.ptr if m is not None else None, v.ptr if v is not None else None,
                                          int(lo), int(hi), int(sum_lo), int(sum_hi), int(i_batch), float(step_size), float(b1), float(b2),
                                          float(eps), int(flags), mask.ptr if mask is not None else None))

    def all_reduce_device(self, dev):
        from ._lib import check
        check(self.ctx.lib.adm_p2p_all_reduce(self.ctx.handle, dev.ptr, dev.size))
        return dev

    def device_barrier(self):
        from ._lib import check
        check(self.ctx.lib.adm_p2p_barrier(self.ctx.handle))

    def check_status(self):
        """Raises if a wait on a peer timed out (the update kernels after it were skipped)."""
        from ._lib import check
        if self.ctx is not None:
            check(self.ctx.lib.adm_p2p_status(self.ctx.handle))

    def reduce_scatter_sum(self, full, shard_out):
        raise NotImplementedError('P2PComm: the exchange is fused with the update (fused_update)')

    all_gather = broadcast = reduce = reduce_scatter_sum

    def group(self):
        import contextlib
        return contextlib.nullcontext()

    def barrier(self):
        if self.ctx is not None:
            self.ctx.sync()
            self.check_status()
        self.group_.barrier()

    def close(self, keep_group=False):
        if self.ctx is not None:
            ctx = self.ctx
            ctx.sync()
            err = None
            try:
                self.check_status()
            except Exception as e:      # still tear down in step with the peers
                err = e
            try:
                self.group_.barrier()       # every rank has finished using its peers' buffers
                for ptr in self._mapped:
                    ctx.lib.adm_p2p_close(ctx.handle, ptr)
                self._mapped = []
                self.group_.barrier()       # every mapping of this rank's buffers is gone: they may be freed
            except Exception:
                pass
            ctx.lib.adm_p2p_destroy(ctx.handle)
            self.ctx = None
            self.bound = None
            if err is not None and not keep_group:
                self.group_.close()
                raise err
        if not keep_group:
            self.group_.close()


def device_count():
    """Number of GPUs libadm sees (adm_device_count: hipGetDeviceCount, no context is created)."""
    from . import _lib
    return int(_lib.load().adm_device_count())


def from_env():
    """LocalComm unless launched with WORLD_SIZE > 1 (torch.distributed.run, an mpirun wrapper, bench.py); then ADM_COMM selects the data plane:
    'rccl' (default: RCCL through the C ABI), 'p2p' (direct all-pairs exchange through IPC-mapped peer buffers, fused with
    the update; several ranks may share a GPU), 'host' (validation: staged through host memory, several ranks may share a GPU)."""
    if int(os.environ.get('WORLD_SIZE', '1')) > 1:
        kind = os.environ.get('ADM_COMM', 'rccl')
        if kind == 'p2p':
            return P2PComm()
        if kind not in ('rccl', 'host'):
            raise ValueError("ADM_COMM must be 'rccl', 'p2p' or 'host' (got '%s')" % kind)
        if kind == 'host':
            return HostStagedComm(device_index=int(os.environ.get('LOCAL_RANK', '0')) % max(1, device_count()))
        return RcclComm()
    return LocalComm()
