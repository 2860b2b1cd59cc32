"""
Fresnel transfer functions (host, setup time) and the MultisliceEngine that drives the HIP
kernels for one minibatch: rotate -> multislice forward + loss + adjoint -> rotate-adjoint.

Reference: adorym/propagate.py (get_kernel :62-81, gen_freq_mesh :54-60,
multislice_propagate_batch :131-288).
"""
import ctypes as C
import os
import numpy as np

from . import _lib
from ._lib import check
from .constants import PI
from .device import Context, DeviceArray, Plan, UploadRing
from .util import calculate_pad_len, rotation_lookup, build_rotation_adjoint_csr


def gen_freq_mesh(voxel_nm, shape):
    """adorym/propagate.py:54-60."""
    u = np.fft.fftfreq(shape[0])
    v = np.fft.fftfreq(shape[1])
    vv, uu = np.meshgrid(v, u)
    vv = vv / voxel_nm[1]
    uu = uu / voxel_nm[0]
    return uu, vv


def get_kernel(dist_nm, lmbda_nm, voxel_nm, grid_shape, fresnel_approx=True, sign_convention=1):
    """Unshifted Fresnel transfer function, complex128 (adorym/propagate.py:62-81)."""
    u, v = gen_freq_mesh(voxel_nm, grid_shape[0:2])
    if fresnel_approx:
        H = np.exp(-sign_convention * 1j * PI * lmbda_nm * dist_nm * (u ** 2 + v ** 2))
    else:
        quad = 1 - lmbda_nm ** 2 * (u ** 2 + v ** 2)
        quad_inner = np.clip(quad, 0, None)
        quad_mask = (quad > 0)
        H = np.exp(sign_convention * 1j * 2 * PI * dist_nm / lmbda_nm * np.sqrt(quad_inner))
        H = H * quad_mask
    return H


class RotationTable(object):
    """Per-angle rotation data on the device: the reference's fp16 lookup table (adorym/util.py:492-516) for the
    forward gather, and -- built lazily from it -- the CSR transpose used by the adjoint gather."""

    def __init__(self, ctx, obj_size, theta, coords_fp16=None):
        self.ctx = ctx
        self.obj_size = tuple(int(v) for v in obj_size)
        # the fp16 table itself is formed on the device (adm_rotation_table_build: same bits as util.rotation_lookup, which costs
        # 0.45 ms of NumPy per angle); a host copy exists only if one was handed in or somebody asks for it (``host``)
        self._theta = theta
        self._host = None if coords_fp16 is None else np.asarray(coords_fp16, dtype=np.float16)
        if self._host is None and os.environ.get('ADM_DEVICE_ROT_TABLE', '1') != '1':
            self._host = rotation_lookup(self.obj_size, theta)          # (the table from the host, as before round 6)
        # ONE allocation per angle: [fp16 table | ptr | src | w | lsrc | boxes]; the table goes up through the context's
        # pinned ring (no stream synchronisation when the driver meets a new angle in the middle of an epoch)
        _, X, Z = self.obj_size
        n = 4 * X * Z
        nblk = ((X + 15) // 16) * ((Z + 15) // 16)
        al = lambda v: (v + 255) & ~255
        sizes = [X * Z * 2 * 2, (X * Z + 1) * 4, n * 4, n * 4, n * 2, nblk * 16]
        offs = np.concatenate([[0], np.cumsum([al(v) for v in sizes])]).astype(int)
        self._arena = DeviceArray(ctx, (int(offs[-1]),), np.uint8)
        sub = lambda i, shape, dt: DeviceArray(ctx, shape, dt, ptr=self._arena.ptr + int(offs[i]))
        self.coords = sub(0, (X * Z, 2), np.uint16)
        self._parts = (sub(1, (X * Z + 1,), np.int32), sub(2, (n,), np.int32), sub(4, (n,), np.uint16), sub(3, (n,), np.float32),
                       sub(5, (nblk, 4), np.int32))          # ptr, src, lsrc, w, boxes
        if self._host is not None:
            ctx.uploader(sizes[0]).upload(self.coords, np.ascontiguousarray(self._host).view(np.uint16).reshape(X * Z, 2))
        else:
            th32 = np.float32(theta)
            check(ctx.lib.adm_rotation_table_build(ctx.handle, X, Z, float(np.cos(th32, dtype=np.float32)), float(np.sin(th32, dtype=np.float32)),
                                                   self.coords.ptr))
        self._csr = None
        self._csr_key = None
        th = float(theta)
        self.lanes_along_x = abs(np.cos(th)) > abs(np.sin(th))

    @property
    def ptr(self):
        return self.coords.ptr

    @property
    def host(self):
        """The table on the host (float16 [X*Z, 2]): the reference arithmetic in NumPy, evaluated on first use."""
        if self._host is None:
            self._host = rotation_lookup(self.obj_size, self._theta)
        return self._host

    def csr(self, plan):
        """(ptr, src, lsrc, w, boxes) device arrays of adm_rotate_adj_staged, built on the GPU (adm_rotation_csr_build,
        asynchronous: ~0.1 ms on the stream instead of 40-60 ms of NumPy per angle)."""
        key = (plan.rot_shape, plan.pads)
        if self._csr is None or self._csr_key != key:
            ctx = self.ctx
            parts = self._parts
            scratch = plan.rotation_csr_scratch()
            check(ctx.lib.adm_rotation_csr_build(plan.handle, self.coords.ptr, parts[0].ptr, parts[1].ptr, parts[2].ptr, parts[3].ptr,
                                                 parts[4].ptr, scratch.ptr, scratch.nbytes))
            self._csr = parts
            self._csr_key = key
        return self._csr

    def csr_host(self, plan):
        """The same tables from the host builder (tests compare the two)."""
        Zp, Yp, Xp, _ = plan.rot_shape
        return build_rotation_adjoint_csr(self.host, self.obj_size, Yp, Xp, plan.pads[1][0], staged=True)


class MultisliceEngine(object):
    """
    Device-resident state for the accelerated path of one reconstruction:

      obj [Y,X,Z,2] --adm_rotate_fwd--> obj_rot [Z][Yp][Xp][2] --adm_multislice_fwd_adj-->
      grad_rot --adm_rotate_adj--> grad_obj [Y,X,Z,2]

    ``probe_pos`` is the list of ALL probe positions (top-left corners, object coordinates); it
    fixes the zero padding of the rotated-frame buffers (adorym/util.py:1374-1406 applied once).
    """

    def __init__(self, ctx, obj_size, probe_size, probe_pos, energy_ev, psize_cm, free_prop_cm='inf', binning=1,
                 fresnel_approx=True, sign_convention=1, normalize_fft=False, kernel=None, scale_ri_by_k=True,
                 n_probe_modes=1, max_batch=None, loss_function_type='lsq', poisson_multiplier=1., unknown_type='delta_beta',
                 beamstop=None, generic=False, transmission_cache=True, transmissions_only=False):
        """``free_prop_cm``: 0 / None (exit wave), 'inf' (far field), a distance in cm (Fresnel propagation to the detector), or
        a SEQUENCE of n distances: position b of every launch is propagated to distance b % n (multi-distance data divided into
        sub-tiles, adorym/forward_model.py:999-1018 -- the caller lists every tile n times in a row, ``n_dists`` = n)."""
        self.ctx = ctx
        self.obj_size = tuple(int(v) for v in obj_size)
        self.probe_size = tuple(int(v) for v in probe_size)
        probe_pos = np.round(np.asarray(probe_pos)).astype(int).reshape(-1, 2)
        pads = calculate_pad_len(self.obj_size, probe_pos, self.probe_size)
        # adorym/propagate.py:143-153, 215
        voxel_nm = np.array([psize_cm] * 3) * 1.e7
        lmbda_nm = 1240. / energy_ev
        delta_nm = voxel_nm[-1]
        self.k1 = 2. * PI * delta_nm / lmbda_nm if scale_ri_by_k else 1.
        if kernel is None:
            kernel = get_kernel(delta_nm * binning, lmbda_nm, voxel_nm, self.probe_size, fresnel_approx=fresnel_approx,
                                sign_convention=sign_convention)
        h_free = None
        dists = None
        if not isinstance(free_prop_cm, str) and np.ndim(free_prop_cm) > 0:
            dists = [float(d_) for d_ in np.asarray(free_prop_cm).reshape(-1)]
            if len(dists) == 0 or any(d_ == 0 for d_ in dists):
                raise ValueError('free_prop_cm: a sequence of distances must be non-empty and non-zero')
            free_prop_cm = dists[0]
        self.n_dists = len(dists) if dists else 1
        if free_prop_cm in (0, None):
            det = _lib.DET_NONE
        elif isinstance(free_prop_cm, str) and free_prop_cm == 'inf':
            det = _lib.DET_FARFIELD
        else:
            det = _lib.DET_FRESNEL
            # fresnel_propagate always builds the Fresnel-approx kernel (adorym/propagate.py:537-546)
            h_free = get_kernel(float(free_prop_cm) * 1e7, lmbda_nm, voxel_nm, self.probe_size,
                                sign_convention=sign_convention)
        self.plan = Plan(ctx, self.obj_size, self.probe_size, pads, self.k1, kernel, binning=binning,
                         n_modes=n_probe_modes, sign_convention=sign_convention, det_mode=det,
                         normalize_fft=normalize_fft, h_free=h_free,
                         loss_type={'lsq': _lib.LOSS_LSQ, 'poisson': _lib.LOSS_POISSON}[loss_function_type],
                         poisson_multiplier=poisson_multiplier, unknown_type=unknown_type)
        if dists and len(dists) > 1:
            self.plan.set_detector_kernels([get_kernel(d_ * 1e7, lmbda_nm, voxel_nm, self.probe_size, sign_convention=sign_convention)
                                            for d_ in dists])
        if generic:
            self.plan.set_generic(True)       # the any-size kernel even where a tuned one exists (tests, A/B timing)
        # slice transmissions cached per rotated-frame voxel by rotate() (include/adm.h: adm_plan_set_transmission_cache)
        self.transmission_cache = bool(transmission_cache) and unknown_type == 'delta_beta' and binning == 1
        # transmissions_only: rotate() stores the slice transmissions and nothing else (nobody reads the rotated (delta, beta)
        # once the slice loop multiplies with cached numbers): half the stores of the rotation.  For callers that never look at
        # obj_rot themselves -- the driver and bench.py; tests and tools that read it keep the default.
        self.transmissions_only = bool(transmissions_only) and self.transmission_cache
        if self.transmission_cache:
            self.plan.set_transmission_cache(2 if self.transmissions_only else 1)
        self.unknown_type = unknown_type
        self.loss_function_type = loss_function_type
        self.pads = pads
        self.n_probe_modes = int(n_probe_modes)
        # beamstop (adorym/forward_model.py:128-136): detector pixels with beamstop >= 1e-5 enter the loss
        self.n_det = self.probe_size[0] * self.probe_size[1]
        if beamstop is not None:
            bs = np.ascontiguousarray(np.asarray(beamstop, dtype=np.float32).reshape(self.probe_size))
            check(ctx.lib.adm_plan_set_detector_mask(self.plan.handle, bs.ctypes.data))
            self.n_det = int((bs >= 1e-5).sum())
        self.obj_rot = ctx.zeros(self.plan.rot_shape)       # pads stay zero forever
        if unknown_type == 'real_imag':
            # pad_object pads a real_imag object with 1 + 0i (adorym/util.py:1338-1350): vacuum transmission
            fill = np.zeros(self.plan.rot_shape, np.float32)
            fill[..., 0] = 1.0
            self.obj_rot.set(fill)
            del fill
        self.grad_rot = ctx.zeros(self.plan.rot_shape)      # rows of the current batch are overwritten each call
        self.max_batch = 0
        self._accumulated = False
        self._all_pos_host = np.ascontiguousarray(probe_pos.astype(np.int32))
        self._all_pos_dev = ctx.array(self._all_pos_host)
        self._ws = self._pos = self._target = self._pred = self._loss = None
        if max_batch:
            self._reserve(max_batch)

    # -------------------------------------------------------------------------------- buffers
    def _reserve(self, batch):
        if batch <= self.max_batch:
            return
        Py, Px = self.probe_size
        self._ws = DeviceArray(self.ctx, (self.plan.workspace_bytes(batch),), np.uint8)
        self._pos = DeviceArray(self.ctx, (batch, 2), np.int32)
        self._target = DeviceArray(self.ctx, (batch, Py, Px), np.float32)
        self._pred = DeviceArray(self.ctx, (batch, Py, Px), np.float32)
        # The per-position loss sums are written by the kernel straight into page-locked HOST memory (MappedArray): no
        # device-to-host copy is ever queued (the 4.5 us blit kernel and its dependency gap sat in every minibatch of the
        # launch-bound paths).  Two buffers, written alternately by successive launches, each with an event that is recorded
        # LATER -- beside launch k+1, see loss_async -- and tells the host that launch k's sums have arrived.
        from .device import MappedArray, Event
        self._loss_pair = [MappedArray(self.ctx, (batch,), np.float32) for _ in range(2)]
        self._loss_events = [Event(self.ctx) for _ in range(2)]
        self._loss_tokens = [None, None]
        self._loss_k = 0
        self._loss = self._loss_pair[0]
        self._deferred_loss = None
        # pinned staging for the per-minibatch uploads (targets, positions): asynchronous, the host never drains the stream
        self._ring = UploadRing(self.ctx, batch * Py * Px * 4, n_slots=4)
        self.max_batch = batch

    def cacheless_plan(self):
        """A twin of the plan (same geometry) without the slice-transmission cache: rotations of arrays other than the object
        (the gradient resampling of rotate_out_of_loop) go through it and leave the cache of ``obj_rot`` alone."""
        if getattr(self, '_aux_plan', None) is None:
            d = self.plan.desc
            self._aux_plan = Plan(self.ctx, self.obj_size, self.probe_size, self.pads, d.k1, self.plan._h[0] + 1j * self.plan._h[1],
                                  binning=d.binning, n_modes=d.n_modes, sign_convention=d.sign_convention, det_mode=_lib.DET_NONE,
                                  unknown_type=self.unknown_type)
        return self._aux_plan

    def y_footprint(self, pos_batch):
        pos = np.round(np.asarray(pos_batch)).astype(int).reshape(-1, 2)
        lo = max(0, int(pos[:, 0].min()))
        hi = min(self.obj_size[0], int(pos[:, 0].max()) + self.probe_size[0])
        return lo, max(lo, hi)

    # -------------------------------------------------------------------------------- stages
    def rotate(self, obj, coords, y_range=None):
        """obj: DeviceArray [Y,X,Z,2]; coords: DeviceArray uint16 [X*Z,2] or None (no rotation)."""
        lo, hi = y_range if y_range is not None else (0, self.obj_size[0])
        cp = coords.ptr if coords is not None else None
        check(self.ctx.lib.adm_rotate_fwd(self.plan.handle, obj.ptr, cp, self.obj_rot.ptr, lo, hi))

    def rotate_adjoint(self, grad_obj, coords, y_range=None):
        """grad_obj += R^T grad_rot.  With a RotationTable the deterministic CSR gather is used; with a bare
        coordinate array (or None = identity) the scatter kernel with float atomics."""
        lo, hi = y_range if y_range is not None else (0, self.obj_size[0])
        if isinstance(coords, RotationTable):
            p, s, ls, w, b = coords.csr(self.plan)
            check(self.ctx.lib.adm_rotate_adj_staged(self.plan.handle, self.grad_rot.ptr, p.ptr, s.ptr, ls.ptr, w.ptr, b.ptr,
                                                     grad_obj.ptr, lo, hi))
        else:
            check(self.ctx.lib.adm_rotate_adj(self.plan.handle, self.grad_rot.ptr, coords.ptr if coords is not None else None,
                                              grad_obj.ptr, lo, hi))

    def set_batch(self, pos_batch, target):
        """Upload the probe positions [B,2] and target magnitudes [B,Py,Px] of the next minibatch
        (target may already be a DeviceArray)."""
        pos = np.ascontiguousarray(np.round(np.asarray(pos_batch)).astype(np.int32).reshape(-1, 2))
        B = len(pos)
        self._reserve(B)
        # all probe positions live on the device (uploaded once): a minibatch that is a contiguous run of the
        # scan list needs no host-to-device copy at all
        run = None
        if self._all_pos_host is not None and B <= len(self._all_pos_host):
            # (the same minibatches come back every epoch: the search over the scan list is done once per distinct batch)
            cache = self.__dict__.setdefault('_run_cache', {})
            key = pos.tobytes()
            run = cache.get(key, -1)
            if run == -1:
                run = None
                cand = np.flatnonzero((self._all_pos_host[:len(self._all_pos_host) - B + 1] == pos[0]).all(axis=1))
                for c in cand:
                    if np.array_equal(self._all_pos_host[c:c + B], pos):
                        run = int(c)
                        break
                if len(cache) > 65536:
                    cache.clear()
                cache[key] = run
        if run is not None:
            self._cur_pos = self._all_pos_dev.view(2 * run, (B, 2))
        else:
            self._cur_pos = self._pos.view(0, (B, 2))
            self._ring.upload(self._cur_pos, pos)
        self._pos_host = pos
        if isinstance(target, DeviceArray):
            self._cur_target = target
        else:
            self._cur_target = self._target.view(0, (B,) + self.probe_size)
            self._ring.upload(self._cur_target, np.asarray(target, dtype=np.float32))
        self._B = B
        return B

    def stage_target(self, target):
        """Upload the measured data [B,Py,Px] of the NEXT minibatch into one of two staging buffers on the stream the context is
        enqueuing on right now -- the drivers call it inside the side-stream region of the current minibatch -- and return the
        DeviceArray view to hand to that minibatch's set_batch().  The copy then runs beside the current multislice launch
        instead of in front of the next rotation on the main stream (0.66 MB per 32 positions: 13 us; 11 MB per fused angle: 0.22 ms
        of a PCIe-inclusive step)."""
        host = np.ascontiguousarray(target, dtype=np.float32)
        B = host.shape[0]
        bufs = self.__dict__.setdefault('_stage_bufs', [None, None])
        k = self.__dict__.get('_stage_k', 0) ^ 1
        self._stage_k = k
        need = max(B, self.max_batch or 0)
        if bufs[k] is None or bufs[k].shape[0] < B:
            bufs[k] = DeviceArray(self.ctx, (need,) + tuple(self.probe_size), np.float32)
        view = bufs[k].view(0, (B,) + tuple(self.probe_size))
        ring = self.__dict__.get('_stage_ring')
        if ring is None or ring.slot_bytes < host.nbytes:
            if ring is not None:
                self.__dict__.setdefault('_retired_rings', []).append(ring)
            ring = self._stage_ring = UploadRing(self.ctx, max(host.nbytes, need * int(np.prod(self.probe_size)) * 4), n_slots=2)
        ring.upload(view, host)
        return view

    def multislice(self, probe, grad_probe=None, want_grad=True, want_pred=False, grad_scale=None, accumulate=True,
                   shifts=None, shift_index=None, grad_shifts=None, probes_b=None):
        """Launch the fused kernel on the batch given to set_batch().  Returns nothing; read
        results with loss() / pred().

        ``shifts`` (DeviceArray float [n_entries,2] = (sy, sx)) switches to one Fourier-shifted probe set per position
        (adorym/forward_model.py:296-311); position b uses entry ``shift_index[b]`` (DeviceArray int32 [B]; None = b).
        With want_grad, ``grad_probe`` (+=) and ``grad_shifts`` (float [n_entries,2], +=) then receive the gradients
        taken through the shift.

        ``probes_b`` (DeviceArray [B, n_modes, Py, Px, 2]): one probe set per position handed over as it is (the windows of a
        full-field probe that the sub-tiles of multi-distance data see, adorym/forward_model.py:944-994); ``probe`` is ignored
        and no probe gradient is formed."""
        B = self._B
        Py, Px = self.probe_size
        if grad_scale is None:
            grad_scale = 2.0 / (B * self.n_det)       # d mean((pred-target)^2) / d pred
        lib = self.ctx.lib
        self._next_loss_buffer()
        if probes_b is not None:
            if shifts is not None or grad_probe is not None:
                raise ValueError('probes_b excludes shifts and grad_probe')
            if tuple(probes_b.shape) != (B, self.n_probe_modes, Py, Px, 2):
                raise ValueError('probes_b must be [%d, %d, %d, %d, 2], got %r' % (B, self.n_probe_modes, Py, Px, tuple(probes_b.shape)))
            check(lib.adm_multislice_fwd_adj_pp(
                self.plan.handle, self.obj_rot.ptr, probes_b.ptr, self._cur_pos.ptr, B, self._cur_target.ptr,
                1 if want_grad else 0, None, self._pred.ptr if want_pred else None, self._loss.ptr, float(grad_scale),
                self._ws.ptr, self._ws.nbytes))
        elif shifts is None:
            check(lib.adm_multislice_fwd_adj(
                self.plan.handle, self.obj_rot.ptr, probe.ptr, self._cur_pos.ptr, B, self._cur_target.ptr,
                1 if want_grad else 0, grad_probe.ptr if grad_probe is not None else None,
                self._pred.ptr if want_pred else None, self._loss.ptr, float(grad_scale), self._ws.ptr, self._ws.nbytes))
        else:
            M = self.n_probe_modes
            if getattr(self, '_probes_b', None) is None or self._probes_b.shape[0] < B:
                self._probes_b = DeviceArray(self.ctx, (max(B, self.max_batch), M, Py, Px, 2), np.float32)
                self._gprobes_b = DeviceArray(self.ctx, (max(B, self.max_batch), M, Py, Px, 2), np.float32)
            idx = shift_index.ptr if shift_index is not None else None
            need_adj = want_grad and (grad_probe is not None or grad_shifts is not None)
            check(lib.adm_probe_shift(self.plan.handle, probe.ptr, shifts.ptr, idx, B, self._probes_b.ptr))
            check(lib.adm_multislice_fwd_adj_pp(
                self.plan.handle, self.obj_rot.ptr, self._probes_b.ptr, self._cur_pos.ptr, B, self._cur_target.ptr,
                1 if want_grad else 0, self._gprobes_b.ptr if need_adj else None,
                self._pred.ptr if want_pred else None, self._loss.ptr, float(grad_scale), self._ws.ptr, self._ws.nbytes))
            if need_adj:
                if grad_shifts is None:
                    if getattr(self, '_gshift_dummy', None) is None or self._gshift_dummy.size != shifts.size:
                        self._gshift_dummy = DeviceArray(self.ctx, (shifts.size,), np.float32)
                    grad_shifts = self._gshift_dummy
                check(lib.adm_probe_shift_adj(self.plan.handle, probe.ptr, shifts.ptr, idx, B, self._gprobes_b.ptr,
                                              grad_probe.ptr if grad_probe is not None else None, grad_shifts.ptr))
        if want_grad and accumulate:
            self.accumulate_tiles()

    MAX_COVER = 64        # ADM_MAXCOVER of adm_object.hip: cover-list entries per rotated-frame pixel

    def _check_cover(self, pos):
        """How many tiles of ``pos`` cover the most-covered pixel (an upper bound of len(pos) is returned for small batches).  The
        overlap-add's lists keep at most MAX_COVER tiles per pixel: a batch beyond that takes the multi-pass form
        (accumulate_tiles).  Evaluated on the host BEFORE the launch, cached per position set."""
        if len(pos) <= self.MAX_COVER:
            return len(pos)
        import collections
        key = pos.tobytes()
        cache = self.__dict__.setdefault('_cover_ok', collections.OrderedDict())
        if key in cache:
            cache.move_to_end(key)
        else:
            while len(cache) >= 64:           # bounded: randomised scans meet a new position set every minibatch
                cache.popitem(last=False)
            Py, Px = self.probe_size
            y0 = pos[:, 0] - pos[:, 0].min()
            x0 = pos[:, 1] - pos[:, 1].min()
            d = np.zeros((int(y0.max()) + Py + 1, int(x0.max()) + Px + 1), np.int32)
            np.add.at(d, (y0, x0), 1)
            np.add.at(d, (y0 + Py, x0), -1)
            np.add.at(d, (y0, x0 + Px), -1)
            np.add.at(d, (y0 + Py, x0 + Px), 1)
            cache[key] = int(d.cumsum(0).cumsum(1).max())
        return cache[key]

    def build_cover(self):
        """Queue the cover lists of the overlap-add of the batch given to set_batch() NOW (they depend on the positions only):
        called inside Context.fork()/end_fork(), they are built beside the multislice launch and accumulate_tiles() -- after
        Context.join() -- skips its own build: one launch and one dependency gap less behind the kernel."""
        if self._check_cover(self._pos_host) > self.MAX_COVER:
            return                  # (the multi-pass overlap-add builds a list per pass)
        check(self.ctx.lib.adm_tile_cover_build(self.plan.handle, self._ws.ptr, self._ws.nbytes, self._cur_pos.ptr, self._B,
                                                self._pos_host.ctypes.data, 0, 0, 0))

    def accumulate_tiles(self):
        """Overlap-add the per-position tile gradients into the batch's rows of grad_rot.  A batch in which some pixel is covered
        by more than MAX_COVER tiles (a dense 2-D scan taken as one minibatch) is added in passes of MAX_COVER positions each."""
        lib, B = self.ctx.lib, self._B
        if self._check_cover(self._pos_host) > self.MAX_COVER:
            for k, lo in enumerate(range(0, B, self.MAX_COVER)):
                check(lib.adm_tile_grad_accumulate_range(self.plan.handle, self._ws.ptr, self._ws.nbytes, self._cur_pos.ptr, B,
                                                         self._pos_host.ctypes.data, self.grad_rot.ptr, lo, min(lo + self.MAX_COVER, B),
                                                         1 if k else 0))
            self._accumulated = True
            self._acc_parts = []            # (a pass of <= MAX_COVER positions cannot overflow a list: nothing to check)
            return
        check(lib.adm_tile_grad_accumulate(self.plan.handle, self._ws.ptr, self._ws.nbytes, self._cur_pos.ptr, B,
                                           self._pos_host.ctypes.data, self.grad_rot.ptr))
        self._accumulated = True
        self._acc_parts = [(self._ws, B)]

    N_CU = 256        # MI355X compute units = multislice workgroups resident at once

    def multislice_overlapped(self, probe, grad_probe=None, grad_scale=None, want_pred=False):
        """multislice(want_grad=True) + accumulate_tiles() for batches larger than the chip.  A batch of B > 256 positions runs
        in ceil(B/256) rounds of workgroups; here every round is its own launch (own workspace) and the overlap-add of round
        i runs on the side stream BESIDE the launch of round i+1 -- in particular beside a short last round that leaves most
        CUs idle.  Same sums as one launch + one overlap-add, up to the order of the additions per pixel."""
        B = self._B
        if B <= self.N_CU or self._check_cover(self._pos_host) > self.MAX_COVER:
            # (a batch denser than the cover lists hold is one launch followed by the multi-pass overlap-add)
            self.multislice(probe, grad_probe=grad_probe, want_grad=True, want_pred=want_pred, grad_scale=grad_scale)
            self.ctx.join()                   # side-stream work the caller queued before the launch (no-op if none)
            return
        Py, Px = self.probe_size
        if grad_scale is None:
            grad_scale = 2.0 / (B * self.n_det)
        # equal rounds (544 -> 182 + 181 + 181, not 256 + 256 + 32): a 32-position round costs 1.8 ms, most of a full one, and
        # the chip clocks higher with fewer CUs busy; 8.97 -> 8.44 ms per 544 positions (the same split cost nothing and gained
        # nothing before the kernel's load schedule; one round more: 9.37)
        n_rounds = -(-B // self.N_CU)
        sizes = [B // n_rounds + (1 if i < B % n_rounds else 0) for i in range(n_rounds)]
        bounds = [0] + [int(v) for v in np.cumsum(sizes)]
        parts = [(bounds[i], bounds[i + 1] - bounds[i]) for i in range(len(bounds) - 1)]
        if getattr(self, '_ws_parts', None) is None or len(self._ws_parts) < len(parts):
            need = self.plan.workspace_bytes(self.N_CU)
            self._ws_parts = [DeviceArray(self.ctx, (need,), np.uint8) for _ in parts]
        lib, h = self.ctx.lib, self.plan.handle
        self._next_loss_buffer()
        gp = grad_probe.ptr if grad_probe is not None else None
        pr = self._pred.ptr if want_pred else None
        y_lo = int(self._pos_host[:, 0].min())
        y_hi = int(self._pos_host[:, 0].max()) + Py
        self._acc_parts = []
        # (the whole batch's coverage is within MAX_COVER here, hence every round's)
        if len(parts) <= 4:
            # the cover lists of every round only need the positions: all built now, on the side stream beside the first
            # round, so that no round's overlap-add waits for its own list (the LAST one's build sat behind the last launch)
            self.ctx.fork()
            for i, (o, n) in enumerate(parts):
                check(lib.adm_tile_cover_build(h, self._ws_parts[i].ptr, self._ws_parts[i].nbytes, self._cur_pos.ptr + 8 * o, n,
                                               self._pos_host[o:o + n].ctypes.data, y_lo, y_hi, 1 if i else 0))
            self.ctx.end_fork()
        for i, (o, n) in enumerate(parts):
            ws = self._ws_parts[i]
            check(lib.adm_multislice_fwd_adj(h, self.obj_rot.ptr, probe.ptr, self._cur_pos.ptr + 8 * o, n,
                                             self._cur_target.ptr + 4 * o * Py * Px, 1, gp, (pr + 4 * o * Py * Px) if pr else None,
                                             self._loss.ptr + 4 * o, float(grad_scale), ws.ptr, ws.nbytes))
            self.ctx.fork()                   # side stream: waits for this round, then runs beside the next one
            check(lib.adm_tile_grad_accumulate_part(h, ws.ptr, ws.nbytes, self._cur_pos.ptr + 8 * o, n,
                                                    self._pos_host[o:o + n].ctypes.data, self.grad_rot.ptr, y_lo, y_hi, 1 if i else 0))
            self.ctx.end_fork()
            self._acc_parts.append((ws, n))
        self.ctx.join()
        self._accumulated = True

    def _check_overflow(self):
        if self._accumulated:
            for ws, n in getattr(self, '_acc_parts', []):
                if n > 64:                        # a pixel cannot be covered by more tiles than the part has
                    ov = C.c_int(0)
                    check(self.ctx.lib.adm_tile_grad_status(self.plan.handle, ws.ptr, ws.nbytes, n, C.byref(ov)))
                    if ov.value:
                        raise RuntimeError('tile overlap-add overflow: a pixel is covered by more than 64 tiles of this batch; '
                                           'use a smaller batch')
        self._accumulated = False

    def loss(self, last=None):
        """mean of the per-pixel loss terms over the batch (adorym/forward_model.py:88-103) -- blocks.
        ``last=n``: over the last n positions only (the final minibatch of a fused 'per angle' group)."""
        B = self._B
        self._check_overflow()
        sums = self._loss.view(0, (B,)).get().astype(np.float64)
        if last is not None:
            sums = sums[B - last:]
        return float(sums.sum() / (len(sums) * self.n_det))

    def _next_loss_buffer(self):
        """Switch to the other loss buffer for the launch about to be queued.  If the launch that wrote it last still has an
        unresolved token, its sums are taken off the buffer first (the host waits for that launch's event: it finished long
        ago), so no sum is ever overwritten before somebody could read it."""
        k = 1 - self._loss_k
        tok = self._loss_tokens[k]
        if tok is not None and tok['sums'] is None:
            self._resolve_loss(tok)
        self._loss_tokens[k] = None
        self._loss_k = k
        self._loss = self._loss_pair[k]

    def loss_async(self, last=None):
        """Register the read-back of the per-position loss sums of the batch just launched and return a token for
        loss_result(); the host is not blocked, so the next minibatch can be queued first.  The kernel writes the sums into
        page-locked host memory itself; what is DEFERRED to the next flush_loss_copy() is the event that says they are there
        -- the drivers call it inside the side-stream region of the next minibatch, where the marker costs the main stream
        nothing (on the main stream it sat between the optimiser kernel and the next rotation) -- or to loss_result(),
        whichever comes first."""
        if self._deferred_loss is not None:
            self.flush_loss_copy()
        token = {'k': self._loss_k, 'B': self._B, 'last': last, 'sums': None, 'recorded': False,
                 'buf': self._loss, 'ev': self._loss_events[self._loss_k]}      # (its own buffer and event: _reserve may replace the pair)
        self._loss_tokens[self._loss_k] = token
        self._deferred_loss = token
        return token

    def flush_loss_copy(self):
        """Record the event of the deferred loss read-back, if any, on the stream the context is enqueuing on right now (that
        stream must be behind the launch: the main stream, or the side stream after a fork)."""
        token = self._deferred_loss
        self._deferred_loss = None
        if token is not None and not token['recorded']:
            token['ev'].record()
            token['recorded'] = True

    def _resolve_loss(self, token):
        if not token['recorded']:
            if self._deferred_loss is token:
                self._deferred_loss = None
            token['ev'].record()
            token['recorded'] = True
        token['ev'].synchronize()
        token['sums'] = np.array(token['buf'].host[:token['B']], dtype=np.float64)

    def loss_result(self, token):
        if token['sums'] is None:
            self._resolve_loss(token)
        sums = token['sums']
        if token['last'] is not None:
            sums = sums[token['B'] - token['last']:]
        return float(sums.sum() / (len(sums) * self.n_det))

    def loss_sums(self, n):
        """The first n per-position sums of the last launch (blocks)."""
        return self._loss.view(0, (n,)).get().astype(np.float64)

    def pred(self):
        return self._pred.view(0, (self._B,) + self.probe_size).get()

    # -------------------------------------------------------------------------------- whole step
    def loss_and_grad(self, obj, grad_obj, coords, probe, pos_batch, target, grad_probe=None, footprint=True):
        """One minibatch: grad_obj += d loss / d obj; returns the (host) loss."""
        self.set_batch(pos_batch, target)
        yr = self.y_footprint(pos_batch)
        self.rotate(obj, coords, yr if footprint else None)
        self.multislice(probe, grad_probe=grad_probe)
        # outside the batch's y-footprint the gradient is identically zero (and grad_rot rows there are stale)
        self.rotate_adjoint(grad_obj, coords, yr)
        return self.loss()


class AngleBatch(object):
    """R rotation angles of an UNDIVIDED full-field dataset evaluated in one multislice launch (BASELINE config 2's "minibatch
    16": the reference forces minibatch_size = 1 for such data, adorym/ptychography.py:342-346, so 16 in flight are 16 ranks
    with one angle each whose gradients are summed, :1113-1114).  One angle is one workgroup, a chain of S dependent
    propagations that leaves 255 CUs idle; R angles are independent until the sum.

    No new kernel: the engine is built for a STACK of R objects along y -- plan geometry (R*Y, X, Z), probe position r at
    (r*Y + y0, x0).  Rotations act on every y plane separately with absolute plane numbers, so block r is rotated with angle
    r's table from the ONE real object by handing the kernel the object's address minus r blocks (plane r*Y + y of the stack
    then IS plane y of the object; nothing outside the block's planes is touched), and block r of the stacked gradient image
    is back-rotated -- the deterministic CSR gather, '+=' -- into the one real gradient buffer the same way, block after block
    on one stream.  Same sums as R separate evaluations accumulated into one buffer, in the same order per voxel."""

    def __init__(self, ctx, obj_size, probe_size, n_angles, energy_ev, psize_cm, probe_pos=(0, 0), **engine_kwargs):
        self.ctx = ctx
        self.obj_size = tuple(int(v) for v in obj_size)
        self.R = int(n_angles)
        Y, X, Z = self.obj_size
        y0, x0 = int(probe_pos[0]), int(probe_pos[1])
        Py = int(probe_size[0])
        # a probe that leaves its block along y would read the NEIGHBOURING angle's planes instead of vacuum and scatter its
        # gradient into the wrong block: the stack only stands in for R separate objects while every tile stays inside its own
        if self.R < 1 or y0 < 0 or y0 + Py > Y:
            raise ValueError('AngleBatch: the probe must lie inside the object along y (0 <= y0 and y0 + probe_y <= Y); got y0 = %d, '
                             'probe_y = %d, Y = %d' % (y0, Py, Y))
        self.pos = np.array([(r * Y + y0, x0) for r in range(self.R)], dtype=np.int64)
        engine_kwargs.setdefault('max_batch', self.R)
        self.engine = MultisliceEngine(ctx, (self.R * Y, X, Z), probe_size, self.pos, energy_ev, psize_cm, **engine_kwargs)
        if tuple(self.engine.plan.pads[0]) != (0, 0):
            raise ValueError('AngleBatch: unexpected y padding %s of the stacked geometry' % (self.engine.plan.pads[0],))
        self.block_bytes = Y * X * Z * 2 * 4

    def _shifted(self, arr, r):
        Y, X, Z = self.obj_size
        return DeviceArray(self.ctx, (self.R * Y, X, Z, 2), np.float32, ptr=arr.ptr - r * self.block_bytes)

    def rotate_all(self, obj, tables):
        """Block r of the stacked rotated object = ``obj`` rotated with tables[r], all R blocks in ONE launch (adm_rotate_fwd_stack):
        the table addresses go up through the pinned ring (R pointers), nothing else changes hands."""
        eng = self.engine
        if getattr(self, '_table_ptrs', None) is None:
            self._table_ptrs = DeviceArray(self.ctx, (self.R,), np.uint64)
        self.ctx.uploader().upload(self._table_ptrs, np.array([t.ptr for t in tables], dtype=np.uint64))
        check(self.ctx.lib.adm_rotate_fwd_stack(eng.plan.handle, obj.ptr, self._table_ptrs.ptr, self.R, eng.obj_rot.ptr))

    def rotate_adjoint_all(self, grad_obj, tables):
        """grad_obj += sum over r (ascending) of R_r^T (block r of the stacked gradient image), ONE launch
        (adm_rotate_adj_staged_stack): the same additions, in the same order, as R calls of rotate_adjoint."""
        eng = self.engine
        parts = [t.csr(eng.plan) for t in tables]             # (ptr, src, lsrc, w, boxes) device arrays per angle
        if getattr(self, '_adj_ptrs', None) is None:
            self._adj_ptrs = DeviceArray(self.ctx, (self.R, 5), np.uint64)
        self.ctx.uploader().upload(self._adj_ptrs, np.array([[p_[0].ptr, p_[1].ptr, p_[2].ptr, p_[3].ptr, p_[4].ptr] for p_ in parts],
                                                            dtype=np.uint64))
        # scratch for the angles' terms side by side (R copies of the real gradient), as long as that stays moderate
        # (ADM_STACK_SCRATCH_MB, 512): the R terms are then formed in parallel and added in angle order by a second launch
        if getattr(self, '_adj_scratch', None) is None:
            need = self.R * grad_obj.size * 4
            limit = float(os.environ.get('ADM_STACK_SCRATCH_MB', '512')) * 2 ** 20
            self._adj_scratch = DeviceArray(self.ctx, (self.R * grad_obj.size,), np.float32) if need <= limit else False
        sc = self._adj_scratch
        check(self.ctx.lib.adm_rotate_adj_staged_stack(eng.plan.handle, eng.grad_rot.ptr, self._adj_ptrs.ptr, self.R, grad_obj.ptr,
                                                       sc.ptr if sc is not False else None, sc.nbytes if sc is not False else 0))

    def loss_and_grad(self, obj, grad_obj, tables, probe, targets):
        """obj, grad_obj: DeviceArray [Y,X,Z,2]; tables: R RotationTables (built for obj_size); targets [R,Py,Px] magnitudes
        (host or device).  grad_obj += sum over angles of d(mean loss of angle r)/d obj.  Returns the R losses (blocking)."""
        eng = self.engine
        Y = self.obj_size[0]
        if len(tables) != self.R:
            raise ValueError('AngleBatch.loss_and_grad: %d rotation tables for %d angles' % (len(tables), self.R))
        n_t = targets.size if hasattr(targets, 'size') else np.asarray(targets).size
        if n_t != self.R * eng.n_det:
            raise ValueError('AngleBatch.loss_and_grad: targets must hold R x Py x Px = %d values, got %d' % (self.R * eng.n_det, n_t))
        # (the shifted base pointers below rely on the rotation kernels touching only the planes [r*Y, (r+1)*Y) they are given)
        eng.set_batch(self.pos, targets)
        self.rotate_all(obj, tables)
        n_det = eng.n_det
        eng.multislice(probe, grad_scale=2.0 / n_det)          # every angle is its own minibatch of one: mean over ITS pixels
        self.rotate_adjoint_all(grad_obj, tables)
        eng._check_overflow()
        sums = eng.loss_sums(self.R)
        return sums / n_det
