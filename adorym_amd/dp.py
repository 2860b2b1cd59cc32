"""
Data-parallel object state: the replicated object, its gradient buffer and the (sharded) optimiser
moments, plus the exchange + update step that replaces

    gradient.arr = comm.allreduce(gradient.arr)                 adorym/ptychography.py:1113-1114
    obj.arr = opt.apply_gradient(obj.arr, gradient, i, **opts)  adorym/ptychography.py:1120-1129
    constraints / mask                                          adorym/ptychography.py:1135-1158, 1210-1215

with  reduce_scatter(sum) -> fused Adam/GD + constraints on the owned shard -> all_gather.
The element-wise kernels come from an ``ops`` object: HipOps (libadm; the product) -- tests inject a
NumPy stand-in to exercise the sharding logic on CPU with the gloo backend.
"""
import os

import numpy as np

from . import _lib
from ._lib import check
from .device import DeviceArray


class HipOps(object):
    """Element-wise update kernels of libadm on flat fp32 device buffers."""

    def __init__(self, ctx):
        self.ctx = ctx

    def wrap(self, tensor, n):
        """Non-owning DeviceArray over a torch CUDA tensor (kept alive by the caller); libadm arrays pass through."""
        if isinstance(tensor, DeviceArray):
            return tensor
        return DeviceArray(self.ctx, (n,), np.float32, ptr=tensor.data_ptr())

    def alloc(self, n):
        return self.ctx.zeros((n,))

    def zero(self, buf):
        buf.zero_()

    def copy(self, dst, dst_off, src, src_off, n):
        check(self.ctx.lib.adm_d2d(self.ctx.handle, dst.ptr + 4 * dst_off, src.ptr + 4 * src_off, 4 * n))

    def adam(self, x, g, g_base, m, v, mv_base, lo, hi, i_batch, step_size, b1, b2, eps, flags, mask):
        """x indexed absolutely over [lo,hi); g at (i - g_base); m, v at (i - mv_base)."""
        check(self.ctx.lib.adm_adam_step(self.ctx.handle, x.ptr, g.ptr - 4 * g_base, m.ptr - 4 * mv_base, v.ptr - 4 * mv_base,
                                         lo, hi, int(i_batch), float(step_size), float(b1), float(b2), float(eps), int(flags),
                                         mask.ptr if mask is not None else None))

    def momentum(self, x, g, g_base, v, v_base, lo, hi, step_size, gamma, flags, mask):
        check(self.ctx.lib.adm_momentum_step(self.ctx.handle, x.ptr, g.ptr - 4 * g_base, v.ptr - 4 * v_base, lo, hi, float(step_size),
                                             float(gamma), int(flags), mask.ptr if mask is not None else None))

    def gd(self, x, g, g_base, lo, hi, step_size, flags, mask):
        check(self.ctx.lib.adm_gd_step(self.ctx.handle, x.ptr, g.ptr - 4 * g_base, lo, hi, float(step_size), int(flags),
                                       mask.ptr if mask is not None else None))


def constraint_flags(non_negativity=False, object_type='normal'):
    f = 0
    if non_negativity:
        f |= _lib.FLAG_NONNEG
    if object_type == 'absorption_only':
        f |= _lib.FLAG_ZERO_CH0
    if object_type == 'phase_only':
        f |= _lib.FLAG_ZERO_CH1
    return f


class DataParallelObject(object):
    """Replicated object [Y,X,Z,2] + gradient + sharded moments across ``comm.size`` ranks."""

    def __init__(self, ops, comm, shape, n_moments=2):
        self.ops = ops
        self.comm = comm
        self.shape = tuple(int(s) for s in shape)
        self.n = int(np.prod(self.shape))
        R = comm.size
        self.n_pad = -(-self.n // (2 * R)) * (2 * R)
        self.per = self.n_pad // R
        self.lo = comm.rank * self.per
        self.hi = min(self.lo + self.per, self.n)
        self._keep = []
        self.dist = R > 1 or getattr(comm, 'backend', 'local') != 'local'
        # RcclComm (and its host-staged validation twin) work on libadm's own buffers, in place: the reduced shard lands in
        # its slot of the gradient buffer and the updated shard is gathered from its slot of the object (no staging copies,
        # no torch tensors)
        self.inplace = self.dist and getattr(comm, 'backend', '') in ('rccl', 'host', 'p2p')
        # P2PComm: the ranks read each other's gradient buffers and write each other's objects directly; reduce-scatter,
        # optimiser and all-gather are ONE kernel per update (adm_p2p_update)
        self.p2p = self.dist and getattr(comm, 'backend', '') == 'p2p'
        if self.inplace:
            self.obj = ops.alloc(self.n_pad)
            self.grad = ops.alloc(self.n_pad)
            if self.p2p:
                comm.bind_object(self.obj, self.grad, self.n_pad)
        elif self.dist:
            t_obj, t_grad = comm.alloc(self.n_pad), comm.alloc(self.n_pad)
            self.t_obj, self.t_grad = t_obj, t_grad
            self.t_gshard, self.t_xshard = comm.alloc(self.per), comm.alloc(self.per)
            self.obj = ops.wrap(t_obj, self.n_pad)
            self.grad = ops.wrap(t_grad, self.n_pad)
            self.gshard = ops.wrap(self.t_gshard, self.per)
            self.xshard = ops.wrap(self.t_xshard, self.per)
        else:
            self.obj = ops.alloc(self.n_pad)
            self.grad = ops.alloc(self.n_pad)
        self.moments = [ops.alloc(self.per) for _ in range(n_moments)]   # ZeRO-1: only the owned shard
        # In-place exchange only: gather the planes the next minibatches read first, the rest beside the next kernel.
        # With more than one rank this two-part gather is OPT-IN (ADM_OVERLAP_GATHER=1, or bench.py after its own
        # equality check of the two paths on the running job): its gain depends on RCCL's kernel finding CUs beside the
        # multislice launch, which only a multi-GPU run can show; the plain all-gather is the default there.
        default = '1' if R == 1 else '0'
        self.overlap_gather = self.inplace and hasattr(comm, 'broadcast') and os.environ.get('ADM_OVERLAP_GATHER', default) == '1'
        # ADM_DEBUG_POISON=1: between exchange_and_update(first=...) and finish_update() the planes of OTHER ranks' shards
        # outside `first` are stale by contract; fill them with NaN so that a reader that skipped finish_update() shows up
        self.poison = os.environ.get('ADM_DEBUG_POISON', '0') == '1'
        self._gather_pending = False
        self.clock = None           # a device.PhaseClock while a caller (bench.py) wants per-phase device times

    def _tic(self, name):
        if self.clock is not None:
            self.clock.start(name)

    def _toc(self, name):
        if self.clock is not None:
            self.clock.stop(name)

    # gradient exchange + update ------------------------------------------------------------
    def _apply(self, optimizer, i_batch, options, flags, mask, g, g_base, lo, hi):
        if hi <= lo:
            return
        if optimizer == 'adam':
            self.ops.adam(self.obj, g, g_base, self.moments[0], self.moments[1], self.lo, lo, hi, i_batch,
                          options.get('step_size', 0.001), options.get('b1', 0.9), options.get('b2', 0.999),
                          options.get('eps', 1e-7), flags, mask)
        elif optimizer == 'gd':
            self.ops.gd(self.obj, g, g_base, lo, hi, options['step_size'], flags, mask)
        elif optimizer == 'momentum':
            self.ops.momentum(self.obj, g, g_base, self.moments[0], self.lo, lo, hi, options.get('step_size', 0.001),
                              options.get('gamma', 0.9), flags, mask)
        else:
            raise NotImplementedError("object optimizer '%s' is outside the accelerated path" % optimizer)

    def exchange_and_update(self, optimizer, i_batch, options, flags=0, mask=None, first=None, touched=None, reg_shard=None):
        """optimizer: 'adam' | 'gd' | 'momentum'.  options: dict(step_size=..., b1=..., ...) as the reference's options_dict.

        ``first=(lo, hi)`` (flat element range): the part of the object the NEXT minibatch reads (its y-planes).  Nothing
        outside ``first`` may be read before finish_update() -- which zero_grad() calls -- so that the caller can queue the
        remainder on the context's side stream after the next rotation, where it overlaps the next multislice kernel (which
        keeps only `minibatch` of the 256 CUs busy).  Same arithmetic, same result.
          * one rank: only that range is updated now, the rest of the (element-wise, order-independent) update is deferred;
          * several ranks, in-place RCCL exchange: every rank updates its whole shard (1/R of the object), then the part of
            EVERY shard inside ``first`` is broadcast from its owner (a fraction of the object, from a few owners) and the full
            all-gather is deferred to finish_update().  ``first`` must then be the SAME on every rank -- the union of the planes
            the next minibatches of ALL ranks read -- because it shapes a collective.

        ``touched=(lo, hi)`` + ``reg_shard`` (several ranks; opt-in): the footprint-restricted exchange.  The gradient buffers
        hold the DATA term only, and only on [lo, hi) -- the y-planes the global batch touched, the same range on every rank;
        everything else of the buffers is undefined.  Instead of reduce-scattering the whole buffer, the part of every shard
        inside [lo, hi) is summed onto its owner (one grouped launch of <= R reductions), and the owner then calls
        ``reg_shard(shard_lo, shard_hi, lo, hi)``, which ADDS the regulariser gradient -- identical on every rank in the
        reference, hence R-fold -- inside [lo, hi) and WRITES it elsewhere on the shard.  Same sums as the full exchange up to
        the order of two fp32 additions per element."""
        self.finish_update()
        if self.p2p:
            return self._exchange_and_update_p2p(optimizer, i_batch, options, flags, mask, touched, reg_shard)
        self._tic('reduce_scatter')
        if touched is not None:
            if reg_shard is None:
                raise ValueError('exchange_and_update: touched= needs reg_shard=')
            t_lo, t_hi = max(0, int(touched[0])), min(self.n, int(touched[1]))
            if self.dist:
                import contextlib
                with (self.comm.group() if hasattr(self.comm, 'group') else contextlib.nullcontext()):
                    for r in range(self.comm.size):
                        s_lo, s_hi = max(r * self.per, t_lo), min((r + 1) * self.per, t_hi)
                        if s_hi > s_lo:
                            if self.inplace:
                                self.comm.reduce(self.grad.view(s_lo, (s_hi - s_lo,)), r)
                            else:
                                self.comm.reduce_tensor(self.t_grad[s_lo:s_hi], r)
            # (one local rank: nothing to sum, but the contract is the same -- the buffer holds the data term on [t_lo, t_hi)
            # only and the owner, this rank, completes it with the regulariser term)
            reg_shard(self.lo, self.hi, t_lo, t_hi)
            g, g_base = self.grad, 0
        elif self.inplace:
            self.comm.reduce_scatter_sum(self.grad, self.grad.view(self.lo, (self.per,)))
            g, g_base = self.grad, 0
        elif self.dist:
            self.comm.reduce_scatter_sum(self.t_grad, self.t_gshard)
            g, g_base = self.gshard, self.lo
        else:
            g, g_base = self.grad, 0
        self._toc('reduce_scatter')
        self._tic('update')
        if first is not None and not self.dist and self.hi > self.lo:
            f_lo, f_hi = max(self.lo, int(first[0])), min(self.hi, int(first[1]))
            self._apply(optimizer, i_batch, options, flags, mask, g, g_base, f_lo, f_hi)
            self._deferred = (optimizer, i_batch, dict(options), flags, mask, g, g_base, f_lo, f_hi)
        else:
            self._apply(optimizer, i_batch, options, flags, mask, g, g_base, self.lo, self.hi)
        self._toc('update')
        f_lo, f_hi = (max(0, int(first[0])), min(self.n_pad, int(first[1]))) if first is not None else (0, self.n_pad)
        self._tic('first_gather')
        if self.inplace and self.overlap_gather and (f_lo > 0 or f_hi < self.n):
            import contextlib
            with (self.comm.group() if hasattr(self.comm, 'group') else contextlib.nullcontext()):   # one launch, all roots at once
                for r in range(self.comm.size):
                    s_lo, s_hi = max(r * self.per, f_lo), min((r + 1) * self.per, f_hi)
                    if s_hi > s_lo:
                        self.comm.broadcast(self.obj.view(s_lo, (s_hi - s_lo,)), r)
            if self.poison:
                self._poison_stale(f_lo, f_hi)
            self._gather_pending = True
        elif self.inplace:
            self.comm.all_gather(self.obj, self.obj.view(self.lo, (self.per,)))
        elif self.dist:
            self.ops.copy(self.xshard, 0, self.obj, self.lo, self.per)
            self.comm.all_gather(self.t_obj, self.t_xshard)
        self._toc('first_gather')

    def _exchange_and_update_p2p(self, optimizer, i_batch, options, flags, mask, touched, reg_shard):
        """The whole exchange as one kernel per rank (P2PComm.fused_update): for the owned shard, the ranks' gradient buffers are
        summed in rank order straight from the peers' memory, the optimiser runs in registers and the new values are written
        into every replica; the ranks' streams are barriered on entry and exit, so nothing is deferred (``first`` is moot).
        ``touched``: the buffers hold every rank's data term on [t_lo, t_hi) only, so only that range is summed over the ranks;
        the owner has completed ITS buffer on its shard beforehand (reg_shard: R-fold regulariser term added inside the range,
        written outside), and that is all the kernel reads outside the range."""
        kinds = {'adam': _lib.OPT_ADAM, 'gd': _lib.OPT_GD, 'momentum': _lib.OPT_MOMENTUM}
        if optimizer not in kinds:
            raise NotImplementedError("object optimizer '%s' is outside the accelerated path" % optimizer)
        s_lo, s_hi = 0, self.n_pad
        lo, hi = self.lo, max(self.lo, self.hi)
        if touched is not None:
            if reg_shard is None:
                raise ValueError('exchange_and_update: touched= needs reg_shard=')
            s_lo, s_hi = max(0, int(touched[0])), min(self.n, int(touched[1]))
            self._tic('reduce_scatter')
            reg_shard(self.lo, self.hi, s_lo, s_hi)
            self._toc('reduce_scatter')
        self._tic('fused_exchange')
        if optimizer == 'adam':
            self.comm.fused_update(kinds[optimizer], self.moments[0], self.moments[1], lo, hi, s_lo, s_hi, i_batch,
                                   options.get('step_size', 0.001), options.get('b1', 0.9), options.get('b2', 0.999),
                                   options.get('eps', 1e-7), flags, mask)
        elif optimizer == 'gd':
            self.comm.fused_update(kinds[optimizer], None, None, lo, hi, s_lo, s_hi, 0, options['step_size'], 0., 0., 0., flags, mask)
        else:
            self.comm.fused_update(kinds[optimizer], self.moments[0], None, lo, hi, s_lo, s_hi, 0, options.get('step_size', 0.001),
                                   options.get('gamma', 0.9), 0., 0., flags, mask)
        self._toc('fused_exchange')

    def _poison_stale(self, f_lo, f_hi):
        """Debug aid: NaN-fill what the contract of exchange_and_update(first=...) calls stale (other ranks' shards outside
        ``first``); the deferred all-gather of finish_update() overwrites it."""
        for r in range(self.comm.size):
            if r == self.comm.rank:
                continue
            for a, b in ((r * self.per, min((r + 1) * self.per, f_lo)), (max(r * self.per, f_hi), (r + 1) * self.per)):
                if b > a:
                    check(self.ops.ctx.lib.adm_memset(self.ops.ctx.handle, self.obj.ptr + 4 * a, 0xFF, 4 * (b - a)))

    def finish_update(self):
        """Apply the part of the last update that exchange_and_update(first=...) deferred (on the current stream)."""
        if self._gather_pending:            # a collective: every rank reaches this at the same point of its stream of calls
            self._gather_pending = False
            self._tic('deferred_gather')
            self.comm.all_gather(self.obj, self.obj.view(self.lo, (self.per,)))
            self._toc('deferred_gather')
        d = getattr(self, '_deferred', None)
        if d is not None:
            self._deferred = None
            optimizer, i_batch, options, flags, mask, g, g_base, f_lo, f_hi = d
            self._apply(optimizer, i_batch, options, flags, mask, g, g_base, self.lo, f_lo)
            self._apply(optimizer, i_batch, options, flags, mask, g, g_base, f_hi, self.hi)

    def zero_grad(self):
        self.finish_update()        # the deferred update still reads the gradient
        self.ops.zero(self.grad)
