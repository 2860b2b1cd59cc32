"""Thin Python objects over the libadm C ABI: Context (GPU + stream), DeviceArray (typed device
buffer), Plan (static geometry/physics) and Event.  NumPy is the only host-side array type."""
import ctypes as C
import numpy as np

from . import _lib
from ._lib import check


class Context(object):
    """One GPU + one HIP stream.  ``stream`` may be an existing hipStream_t handle (int), e.g.
    ``torch.cuda.current_stream().cuda_stream``, so that torch.distributed collectives and libadm
    kernels are ordered on the same stream."""

    def __init__(self, device=0, stream=None):
        self.lib = _lib.load()
        h = C.c_void_p()
        check(self.lib.adm_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h)))
        self.handle = h
        self.device = int(device)
        _lib.CREATED_CONTEXT = True

    def sync(self):
        check(self.lib.adm_ctx_sync(self.handle))

    def mem_info(self):
        """(free bytes, total bytes) of this context's GPU."""
        f, t = C.c_size_t(), C.c_size_t()
        check(self.lib.adm_mem_info(self.handle, C.byref(f), C.byref(t)))
        return int(f.value), int(t.value)

    def fork(self):
        """Following calls go to the side stream (after everything enqueued so far) until end_fork()."""
        check(self.lib.adm_ctx_fork(self.handle))

    def end_fork(self):
        check(self.lib.adm_ctx_end_fork(self.handle))

    def join(self):
        """Main stream waits for the side work enqueued between fork() and end_fork()."""
        check(self.lib.adm_ctx_join(self.handle))

    @property
    def stream(self):
        return self.lib.adm_ctx_stream(self.handle)

    def empty(self, shape, dtype=np.float32):
        return DeviceArray(self, shape, dtype)

    def zeros(self, shape, dtype=np.float32):
        a = DeviceArray(self, shape, dtype)
        a.zero_()
        return a

    def uploader(self, min_slot_bytes=1 << 20):
        """A context-wide pinned upload ring for small tables (rotation lookups, ...): asynchronous host-to-device."""
        ring = getattr(self, '_uploader', None)
        if ring is None or ring.slot_bytes < min_slot_bytes:
            if ring is not None:        # copies out of the smaller ring's pinned slots may still be in flight: it stays alive
                self.__dict__.setdefault('_retired_uploaders', []).append(ring)
            ring = self._uploader = UploadRing(self, max(int(min_slot_bytes), 1 << 20), n_slots=4)
        return ring

    def array(self, host, dtype=None):
        host = np.ascontiguousarray(host, dtype=dtype)
        a = DeviceArray(self, host.shape, host.dtype)
        a.set(host)
        return a

    def array_async(self, host, dtype=None, max_bytes=64 << 20):
        """Like array(), but the upload goes through a dedicated pinned two-slot ring and the stream is NOT drained (array() blocks
        until the copy has happened, i.e. until everything queued before it has run).  Arrays above ``max_bytes`` take the blocking
        way.  For bulk data that is needed a minibatch later (a resident dataset's next angle)."""
        host = np.ascontiguousarray(host, dtype=dtype)
        if host.nbytes > max_bytes:
            return self.array(host)
        a = DeviceArray(self, host.shape, host.dtype)
        ring = getattr(self, '_bulk_ring', None)
        if ring is None or ring.slot_bytes < host.nbytes:
            if ring is not None:        # copies out of its pinned slots may still be in flight: it stays alive
                self.__dict__.setdefault('_retired_uploaders', []).append(ring)
            ring = self._bulk_ring = UploadRing(self, host.nbytes, n_slots=2)
        ring.upload(a, host)
        return a

    def event(self):
        return Event(self)

    def close(self):
        if self.handle:
            self.lib.adm_ctx_destroy(self.handle)
            self.handle = None


class DeviceArray(object):
    """A typed, shaped view of device memory.  Owns the allocation unless built with ``ptr=``."""

    def __init__(self, ctx, shape, dtype=np.float32, ptr=None):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in (shape if np.ndim(shape) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.size = int(np.prod(self.shape)) if self.shape else 1
        self.nbytes = self.size * self.dtype.itemsize
        self._owns = ptr is None
        if ptr is None:
            p = C.c_void_p()
            check(ctx.lib.adm_malloc(ctx.handle, self.nbytes, C.byref(p)))
            self.ptr = p.value
        else:
            self.ptr = int(ptr)

    def zero_(self):
        check(self.ctx.lib.adm_memset(self.ctx.handle, self.ptr, 0, self.nbytes))
        return self

    def set(self, host):
        host = np.ascontiguousarray(host, dtype=self.dtype)
        if host.size != self.size:
            raise ValueError('size mismatch: device %s vs host %s' % (self.shape, host.shape))
        check(self.ctx.lib.adm_h2d(self.ctx.handle, self.ptr, host.ctypes.data, self.nbytes))
        return self

    def get(self):
        out = np.empty(self.shape, dtype=self.dtype)
        check(self.ctx.lib.adm_d2h(self.ctx.handle, out.ctypes.data, self.ptr, self.nbytes))
        return out

    def copy_from(self, other):
        if other.nbytes != self.nbytes:
            raise ValueError('size mismatch')
        check(self.ctx.lib.adm_d2d(self.ctx.handle, self.ptr, other.ptr, self.nbytes))
        return self

    def view(self, offset_elems, shape):
        """A non-owning sub-view starting ``offset_elems`` elements in."""
        return DeviceArray(self.ctx, shape, self.dtype, ptr=self.ptr + offset_elems * self.dtype.itemsize)

    def free(self):
        if self._owns and self.ptr:
            self.ctx.lib.adm_free(self.ctx.handle, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            if self._owns and self.ptr and self.ctx.handle:
                self.ctx.lib.adm_free(self.ctx.handle, self.ptr)
        except Exception:
            pass


class Event(object):
    def __init__(self, ctx):
        self.ctx = ctx
        h = C.c_void_p()
        check(ctx.lib.adm_event_create(ctx.handle, C.byref(h)))
        self.handle = h

    def record(self):
        check(self.ctx.lib.adm_event_record(self.ctx.handle, self.handle))
        return self

    def synchronize(self):
        check(self.ctx.lib.adm_event_sync(self.ctx.handle, self.handle))
        return self

    def elapsed_ms(self, end):
        ms = C.c_float()
        check(self.ctx.lib.adm_event_elapsed_ms(self.ctx.handle, self.handle, end.handle, C.byref(ms)))
        return ms.value

    def __del__(self):
        try:
            if getattr(self, 'handle', None) and self.ctx.handle:
                self.ctx.lib.adm_event_destroy(self.ctx.handle, self.handle)
                self.handle = None
        except Exception:
            pass


class PhaseClock(object):
    """HIP-event stopwatch for named phases queued on the context's streams: start(name) / stop(name) record events on the
    stream the context is enqueuing on at that moment (main, or the side stream inside fork()/end_fork()); totals() blocks
    on the recorded events and returns {name: (total ms, count)}.  Costs two event records per phase; nothing when unused."""

    def __init__(self, ctx):
        self.ctx = ctx
        self._open = {}
        self._pairs = []

    def start(self, name):
        self._open[name] = Event(self.ctx).record()

    def stop(self, name):
        self._pairs.append((name, self._open.pop(name), Event(self.ctx).record()))

    def totals(self):
        out = {}
        for name, e0, e1 in self._pairs:
            t, n = out.get(name, (0.0, 0))
            out[name] = (t + e0.elapsed_ms(e1), n + 1)
        self._pairs = []
        return out


class PinnedArray(object):
    """Page-locked host array (NumPy view) for asynchronous device-to-host copies."""

    def __init__(self, ctx, shape, dtype=np.float32):
        self.ctx = ctx
        self.shape = tuple(int(v) for v in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        h = C.c_void_p()
        check(ctx.lib.adm_host_alloc(ctx.handle, self.nbytes, C.byref(h)))
        self.handle = h
        buf = (C.c_char * max(self.nbytes, 1)).from_address(h.value)
        self.array = np.frombuffer(buf, dtype=self.dtype, count=int(np.prod(self.shape))).reshape(self.shape)

    def copy_to_async(self, dev, nbytes=None):
        """Queue host -> device of the first ``nbytes`` on the context's stream; this array must stay untouched until an
        event recorded afterwards has happened."""
        check(self.ctx.lib.adm_h2d_async(self.ctx.handle, dev.ptr, self.handle, int(nbytes if nbytes is not None else min(self.nbytes, dev.nbytes))))
        return self

    def copy_from_async(self, dev, nbytes=None):
        check(self.ctx.lib.adm_d2h_async(self.ctx.handle, self.handle, dev.ptr, int(nbytes if nbytes is not None else min(self.nbytes, dev.nbytes))))
        return self

    def __del__(self):
        try:
            if getattr(self, 'handle', None) and self.ctx.handle:      # (after Context.close() the block simply stays until the process ends)
                self.array = None
                self.ctx.lib.adm_host_free(self.ctx.handle, self.handle)
                self.handle = None
        except Exception:
            pass


class MappedArray(object):
    """Page-locked host memory that KERNELS write directly (it is device-accessible at ``ptr``): small results the host wants --
    per-position loss sums -- land where the host reads them, and no device-to-host copy is ever queued.  ``get()`` waits for
    the context's streams and returns a copy; ``view(offset, shape)`` is a sub-range with the same two members."""

    def __init__(self, ctx, shape, dtype=np.float32, _pinned=None, _offset=0):
        self.ctx = ctx
        self.shape = tuple(int(v) for v in (shape if np.ndim(shape) else (shape,)))
        self.dtype = np.dtype(dtype)
        self.size = int(np.prod(self.shape)) if self.shape else 1
        self.nbytes = self.size * self.dtype.itemsize
        self._pinned = _pinned if _pinned is not None else PinnedArray(ctx, (self.size,), self.dtype)
        self._offset = int(_offset)
        self.ptr = self._pinned.handle.value + self._offset * self.dtype.itemsize

    @property
    def host(self):
        """The live host view (valid to read once an event recorded behind the writing kernel has happened)."""
        return self._pinned.array.reshape(-1)[self._offset:self._offset + self.size].reshape(self.shape)

    def get(self):
        self.ctx.sync()
        return np.array(self.host)

    def view(self, offset_elems, shape):
        return MappedArray(self.ctx, shape, self.dtype, _pinned=self._pinned, _offset=self._offset + int(offset_elems))


class UploadRing(object):
    """Pinned staging ring owned by a context user: upload(dev, host_array) copies the array into the next pinned slot
    and queues an asynchronous host-to-device copy -- the host never waits for the stream (DeviceArray.set does).  A slot
    is reused only after the event recorded behind its copy has happened."""

    def __init__(self, ctx, slot_bytes, n_slots=4):
        self.ctx = ctx
        self.slot_bytes = int(slot_bytes)
        self.slots = [PinnedArray(ctx, (self.slot_bytes,), np.uint8) for _ in range(n_slots)]
        self.events = [None] * n_slots
        self.k = 0

    def upload(self, dev, host):
        host = np.ascontiguousarray(host, dtype=dev.dtype)
        nb = host.nbytes
        if nb > dev.nbytes:
            raise ValueError('upload larger than the device array')
        if nb > self.slot_bytes:
            dev.set(host) if nb == dev.nbytes else dev.view(0, host.shape).set(host)      # oversize: blocking path
            return
        k = self.k
        self.k = (k + 1) % len(self.slots)
        if self.events[k] is not None:
            self.events[k].synchronize()
        else:
            self.events[k] = Event(self.ctx)
        self.slots[k].array[:nb] = host.reshape(-1).view(np.uint8)
        self.slots[k].copy_to_async(dev, nb)
        self.events[k].record()


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Plan(object):
    """adm_plan: object/probe geometry, padding, physics constants and transfer functions."""

    def __init__(self, ctx, obj_size, probe_size, pads, k1, h, binning=1, n_modes=1, sign_convention=1,
                 det_mode=_lib.DET_FARFIELD, normalize_fft=False, h_free=None, loss_type=_lib.LOSS_LSQ, poisson_multiplier=1.0, unknown_type='delta_beta'):
        self.ctx = ctx
        d = _lib.PlanDesc()
        d.obj_y, d.obj_x, d.obj_z = [int(v) for v in obj_size]
        d.probe_y, d.probe_x = [int(v) for v in probe_size]
        (d.pad_y0, d.pad_y1), (d.pad_x0, d.pad_x1) = [(int(a), int(b)) for a, b in pads]
        d.binning, d.n_modes, d.sign_convention = int(binning), int(n_modes), int(sign_convention)
        d.det_mode, d.normalize_fft, d.k1 = int(det_mode), int(bool(normalize_fft)), float(k1)
        d.loss_type, d.poisson_multiplier = int(loss_type), float(poisson_multiplier)
        d.unknown_type = {'delta_beta': 0, 'real_imag': 1}[unknown_type]
        h = np.asarray(h)
        # h_real / h_imag are cast separately to fp32, as in adorym/propagate.py:202-204
        self._h = (np.ascontiguousarray(h.real, dtype=np.float32), np.ascontiguousarray(h.imag, dtype=np.float32))
        d.h_re, d.h_im = _fptr(self._h[0]), _fptr(self._h[1])
        if h_free is not None:
            hf = np.asarray(h_free)
            self._hf = (np.ascontiguousarray(hf.real, dtype=np.float32), np.ascontiguousarray(hf.imag, dtype=np.float32))
            d.hfree_re, d.hfree_im = _fptr(self._hf[0]), _fptr(self._hf[1])
        p = C.c_void_p()
        check(ctx.lib.adm_plan_create(ctx.handle, C.byref(d), C.byref(p)))
        self.handle = p
        self.desc = d
        self.obj_size = (d.obj_y, d.obj_x, d.obj_z)
        self.probe_size = (d.probe_y, d.probe_x)
        self.pads = ((d.pad_y0, d.pad_y1), (d.pad_x0, d.pad_x1))
        self.rot_shape = (d.obj_z, d.obj_y + d.pad_y0 + d.pad_y1, d.obj_x + d.pad_x0 + d.pad_x1, 2)
        assert int(np.prod(self.rot_shape)) == ctx.lib.adm_plan_rot_elems(p)

    def rotation_csr_scratch(self):
        """Device scratch of adm_rotation_csr_build, allocated once per plan (stream-ordered reuse)."""
        if getattr(self, '_csr_scratch', None) is None:
            self._csr_scratch = DeviceArray(self.ctx, (int(self.ctx.lib.adm_rotation_csr_scratch_bytes(self.handle)),), np.uint8)
        return self._csr_scratch

    def set_generic(self, on=True):
        """Route this plan through the any-size kernel (adm_ms_generic.hip) even if a tuned kernel exists for its probe size;
        call before the first workspace is sized."""
        check(self.ctx.lib.adm_plan_set_generic(self.handle, 1 if on else 0))

    def set_detector_kernels(self, kernels):
        """adm_plan_set_detector_kernels: position b of a launch is propagated to the detector with kernels[b % n] (complex
        [Py,Px] each; real / imaginary parts cast separately like h)."""
        k = np.stack([np.asarray(x) for x in kernels])
        re = np.ascontiguousarray(k.real, dtype=np.float32)
        im = np.ascontiguousarray(k.imag, dtype=np.float32)
        check(self.ctx.lib.adm_plan_set_detector_kernels(self.handle, len(k), re.ctypes.data, im.ctypes.data))

    def set_transmission_cache(self, on=True):
        """adm_rotate_fwd also stores the slice transmission of every voxel it writes and the multislice kernel multiplies
        with it instead of evaluating exp / sincos per covering position (delta_beta unknowns, binning 1; bit-identical)."""
        check(self.ctx.lib.adm_plan_set_transmission_cache(self.handle, int(on)))

    def workspace_bytes(self, batch):
        return int(self.ctx.lib.adm_plan_workspace_bytes(self.handle, int(batch)))

    def close(self):
        if self.handle:
            self.ctx.lib.adm_plan_destroy(self.handle)
            self.handle = None
