"""
Forward models with the reference's plugin interface (adorym/forward_model.py:28-175, 164-401):
``predict`` / ``get_data`` / ``loss`` / ``get_loss_function`` keep their names, argument order and
meaning; the arithmetic runs in libadm's HIP kernels through a MultisliceEngine.

Differences that follow from having no autograd: the closure returned by ``get_loss_function`` carries
a reference to its model (``calculate_loss.forward_model``) so that ``Differentiator.get_gradients`` can
call the hand-derived adjoint (``loss_and_gradients``) instead of ``torch.autograd.grad``.
"""
import inspect
import os
import numpy as np

from ._lib import check
from .device import DeviceArray
from .propagate import RotationTable
from .regularizers import combined_weights
from .util import calculate_pad_len


class ForwardModel(object):
    """adorym/forward_model.py:28-161."""

    def __init__(self, loss_function_type='lsq', distribution_mode=None, device=None, common_vars_dict=None,
                 raw_data_type='magnitude', simulation_mode=False):
        if distribution_mode is not None:
            raise NotImplementedError("distribution_mode '%s' is outside the accelerated path (DP only)" % distribution_mode)
        if loss_function_type not in ('lsq', 'poisson'):
            raise ValueError("loss_function_type must be 'lsq' or 'poisson'")
        if raw_data_type not in ('magnitude', 'intensity'):
            raise ValueError("raw_data_type must be 'magnitude' or 'intensity'")
        self.loss_function_type = loss_function_type
        self.argument_ls = []
        self.regularizer_dict = {}
        self.distribution_mode = distribution_mode
        self.device = device                      # adorym_amd.Context
        self.simulation_mode = simulation_mode
        self._current_loss = 0
        self._loss_thunk = None
        self.raw_data_type = raw_data_type
        self.i_call = 0
        self.common_vars = common_vars_dict
        if common_vars_dict is not None:
            for k in ('unknown_type', 'normalize_fft', 'sign_convention', 'rotate_out_of_loop', 'scale_ri_by_k',
                      'is_minus_logged', 'forward_algorithm', 'stdout_options', 'poisson_multiplier', 'common_probe_pos',
                      'binning', 'prj'):
                setattr(self, k, common_vars_dict.get(k))
        self.loss_args = {}
        self.reg_list = []

    @property
    def current_loss(self):
        """Loss of the last evaluation (adorym/forward_model.py:141-146).  The GPU path queues its read-back without
        blocking; the value is fetched the first time it is looked at."""
        if self._loss_thunk is not None:
            t, self._loss_thunk = self._loss_thunk, None
            self._current_loss = float(t())
        return self._current_loss

    @current_loss.setter
    def current_loss(self, v):
        self._loss_thunk = None
        self._current_loss = v

    def take_loss_thunk(self):
        """Detach the pending loss read-back (a callable returning the float) so that the caller can resolve it later,
        after more work has been queued; returns None if the loss is already on the host."""
        t, self._loss_thunk = self._loss_thunk, None
        if t is None:
            v = self._current_loss
            return lambda: v
        return t

    def update_loss_args(self, kwargs):
        self.loss_args = kwargs

    def add_regularizers(self, reg_list):
        self.reg_list = reg_list

    def get_argument_index(self, arg):
        for i, a in enumerate(self.argument_ls):
            if a == arg:
                return i
        raise ValueError('{} is not in the argument list.'.format(arg))

    # ---- the plugin contract's last-layer pieces (adorym/forward_model.py:75-147) -------------------------------------
    # A forward-model plugin whose predict() returns magnitudes on the HOST combines them with these exactly as the
    # reference's ForwardModel does.  The built-in models do not go through them: their loss is evaluated inside the
    # multislice kernel (loss_term, adm_ms_math.h) and the regulariser inside reg_grad_kernel.
    def get_regularization_value(self, obj, device=None):
        """Sum of the registered regularisers' values for ``obj`` (DeviceArray [Y,X,Z,2]); adorym/forward_model.py:75-80."""
        reg = float(self._regularize(obj, None)) if hasattr(self, '_regularize') else 0.0
        return reg

    def get_mismatch_loss(self, this_pred_batch, this_prj_batch):
        """LSQ / Poisson mismatch of predicted magnitudes and measured data (host arrays), adorym/forward_model.py:88-103."""
        pred = np.asarray(this_pred_batch, dtype=np.float32)
        meas = np.abs(np.asarray(this_prj_batch)).astype(np.float32)
        pm = np.float32(getattr(self, 'poisson_multiplier', None) or 1.)
        if self.loss_function_type == 'lsq':
            if self.raw_data_type == 'intensity':
                meas = np.sqrt(meas)
            return np.mean((pred - meas) ** 2)
        if self.raw_data_type == 'magnitude':
            meas = meas ** 2
        return np.mean(pred ** 2 * pm - meas * pm * np.log(pred ** 2 * pm))

    def loss(self, this_pred_batch, this_prj_batch, obj):
        """(Regularised) loss from predicted MAGNITUDES, with the beamstop selection of adorym/forward_model.py:121-147;
        stores ``current_loss``."""
        pred = np.asarray(this_pred_batch)
        meas = np.asarray(this_prj_batch)
        beamstop = self.common_vars.get('beamstop') if self.common_vars else None
        if beamstop is not None:
            keep = np.asarray(beamstop) >= 1e-5
            pred = pred[:, keep]
            meas = meas[:, keep]
        val = float(self.get_mismatch_loss(pred, meas))
        if len(self.reg_list) > 0:
            val += float(self.get_regularization_value(obj, device=self.device))
        self.current_loss = val
        return val

    def get_data(self, this_i_theta, this_ind_batch, theta_downsample=None, ds_level=1):
        """abs(prj[i_theta*theta_downsample, ind_batch]) (forward_model.py:113-119); sqrt of it for intensity
        data (the sqrt of get_mismatch_loss, :92-93, folded in here).  Returns a float32 host array."""
        if theta_downsample is None:
            theta_downsample = 1
        if ds_level not in (1, None):
            raise NotImplementedError('multiscale (ds_level > 1) is outside the accelerated path')
        t = np.abs(np.asarray(self.prj[int(this_i_theta) * theta_downsample, np.asarray(this_ind_batch)]))
        if self.loss_function_type == 'lsq':
            if self.raw_data_type == 'intensity':
                t = np.sqrt(t)
        else:   # Poisson compares intensities (forward_model.py:94-102): |prj|^2 for magnitude data, |prj| for intensity data
            if self.raw_data_type == 'magnitude':
                t = t ** 2
        return np.ascontiguousarray(t, dtype=np.float32)

    def predict(self, *args, **kwargs):
        raise NotImplementedError

    def get_loss_function(self):
        raise NotImplementedError


class PtychographyModel(ForwardModel):
    """
    adorym/forward_model.py:164-401.  Also serves the reference's SingleBatchFullfieldModel and
    SingleBatchPtychographyModel (:404-586): they compute the same function with fewer stacking steps.
    """

    def __init__(self, loss_function_type='lsq', distribution_mode=None, device=None, common_vars_dict=None,
                 raw_data_type='magnitude', simulation_mode=False, run_bfloat16=False, run_float64=False):
        super(PtychographyModel, self).__init__(loss_function_type, distribution_mode, device, common_vars_dict,
                                                raw_data_type, simulation_mode=simulation_mode)
        if run_bfloat16 or run_float64:
            raise NotImplementedError('the HIP path computes in fp32 (the reference default); bf16/fp64 runs are not provided')
        args = inspect.getfullargspec(self.predict).args
        args.pop(0)
        self.argument_ls = args
        self.engine = common_vars_dict['engine'] if common_vars_dict else None
        self._probe_dev = None
        self._grad_probe_dev = None
        self._reg_val = None
        # >1 when the driver fuses the minibatches of one angle ('per angle' update scheme) into one launch:
        # the batch then holds `batch_group` reference minibatches whose losses are summed
        self.batch_group = 1

    # ------------------------------------------------------------------ helpers
    def _check_static(self, probe_defocus_mm, probe_pos_offset, probe_pos_correction, prj_pos_offset):
        cv = self.common_vars
        for flag in ('optimize_probe_defocusing', 'optimize_probe_pos_offset', 'optimize_prj_pos_offset', 'optimize_tilt'):
            if cv.get(flag):
                raise NotImplementedError('%s is outside the accelerated path (SURVEY section 8 f2)' % flag)

    def _shift_args(self, probe_pos_correction, this_i_theta, this_ind_batch):
        """forward_model.py:297-311: the probes are Fourier-shifted per position when the corrections are being optimised or
        any of them exceeds 1e-3 (sic: positive values only).  Returns (shifts DeviceArray [n_theta*n_pos, 2], index
        DeviceArray int32 [B]) or (None, None)."""
        if probe_pos_correction is None:
            return None, None
        if isinstance(probe_pos_correction, DeviceArray):
            dev = probe_pos_correction
            active = bool(self.common_vars.get('optimize_all_probe_pos')) or getattr(self, '_static_shift_active', None)
            if active is None:
                active = self._static_shift_active = bool(np.any(dev.get() > 1e-3))
        else:
            host = np.ascontiguousarray(np.asarray(probe_pos_correction, dtype=np.float32))
            active = bool(self.common_vars.get('optimize_all_probe_pos')) or bool(np.any(host > 1e-3))
            if not active:
                return None, None
            key = (host.shape, host.tobytes())
            if getattr(self, '_corr_key', None) != key:
                self._corr_dev = self.device.array(host)
                self._corr_key = key
            dev = self._corr_dev
        if not active:
            return None, None
        n_pos = dev.shape[-2] if len(dev.shape) >= 2 else dev.size // 2
        idx = (int(this_i_theta) * n_pos + np.asarray(this_ind_batch, dtype=np.int64)).astype(np.int32)
        if len(idx) > 0 and int(idx[-1]) - int(idx[0]) == len(idx) - 1 and np.all(np.diff(idx) == 1):
            # a run of consecutive entries (the reference sorts a minibatch's indices, adorym/ptychography.py:907): a view of a
            # resident 0, 1, 2, ... array -- no upload, no copy kernel in the minibatch
            n_all = dev.size // 2
            if getattr(self, '_idx_all', None) is None or self._idx_all.size < n_all:
                self._idx_all = self.device.array(np.arange(n_all, dtype=np.int32))
            return dev, self._idx_all.view(int(idx[0]), (len(idx),))
        if getattr(self, '_idx_dev', None) is None or self._idx_dev.size < len(idx):
            self._idx_dev = DeviceArray(self.device, (max(len(idx), 64),), np.int32)
        view = self._idx_dev.view(0, (len(idx),))
        self.device.uploader().upload(view, idx)     # asynchronous (pinned ring): a blocking copy here drained the stream every minibatch
        return dev, view

    def _coords(self, this_i_theta):
        cv = self.common_vars
        if cv.get('two_d_mode') or self.rotate_out_of_loop:
            return None
        return cv['rotation_tables'](int(this_i_theta))

    # ------------------------------------------------------------------ rotate_out_of_loop (adorym/ptychography.py:917-947, 1063-1078)
    def rotate_outside(self, obj, this_i_theta):
        """obj.rotate_array(coords(theta), apply_to_arr_rot=False) of the reference: the WHOLE object is rotated to the angle
        outside the differentiated block.  Fills the engine's slice-major rotated buffer (what the multislice kernel reads,
        with its transmission cache) and ``arr_rot``, the same voxels in the object's own layout [Y,X,Z,2] (what the
        regularisers see and what the driver passes as ``obj``, like the reference's obj.arr_rot)."""
        eng = self.engine
        eng.rotate(obj, self.common_vars['rotation_tables'](int(this_i_theta)), None)
        if getattr(self, 'arr_rot', None) is None or self.arr_rot.shape != obj.shape:
            self.arr_rot = self.device.empty(obj.shape)
        self.arr_rot.zero_()
        check(self.device.lib.adm_rotate_adj(eng.plan.handle, eng.obj_rot.ptr, None, self.arr_rot.ptr, 0, eng.obj_size[0]))
        return self.arr_rot

    def resample_gradient(self, grad, this_i_theta):
        """gradient.rotate_array(coords(-theta), overwrite_arr=True): the accumulated gradient buffer, which is in the rotated
        frame, is resampled with the lookup table of -theta -- a bilinear interpolation like the forward rotation, not its
        transpose -- and overwritten.  Through a cache-less twin of the plan so that the slice-transmission cache of the
        rotated object stays valid; the engine's rotated-gradient buffer is free at this point and serves as scratch."""
        eng = self.engine
        key = int(this_i_theta)
        inv = self.__dict__.setdefault('_inv_tables', {})
        if key not in inv:
            inv[key] = RotationTable(self.device, eng.obj_size, -self.common_vars['theta_ls'][key])
        aux = eng.cacheless_plan()
        lib = self.device.lib
        check(lib.adm_rotate_fwd(aux.handle, grad.ptr, inv[key].ptr, eng.grad_rot.ptr, 0, eng.obj_size[0]))
        grad.zero_()
        check(lib.adm_rotate_adj(aux.handle, eng.grad_rot.ptr, None, grad.ptr, 0, eng.obj_size[0]))

    def _probe(self, probe_real, probe_imag):
        """probe_real/imag: host arrays [n_modes,Py,Px] or a DeviceArray [n_modes,Py,Px,2] passed as probe_real."""
        if isinstance(probe_real, DeviceArray):
            return probe_real
        pr = np.asarray(probe_real, dtype=np.float32)
        pi = np.asarray(probe_imag, dtype=np.float32)
        host = np.ascontiguousarray(np.stack([pr, pi], -1))
        if self._probe_dev is None or self._probe_dev.shape != host.shape:
            self._probe_dev = self.device.empty(host.shape)
        self._probe_dev.set(host)
        return self._probe_dev

    def _run(self, obj, probe_real, probe_imag, this_i_theta, this_pos_batch, target, want_grad, grad_obj=None,
             want_probe_grad=False, want_pred=False, probe_pos_correction=None, this_ind_batch=None, want_shift_grad=False,
             side_hook=None, regularize=True, init_grad=False):
        """One evaluation.  Stream plan (same as bench.py): rotation on the main stream; then, on the context's side
        stream, ``side_hook()`` (the driver zeroes the gradient buffer / finishes the previous update there) and the
        regulariser gradient -- they only need the object -- while the multislice chain, which occupies `minibatch` of
        the 256 CUs, runs on the main stream; joined before the back-rotation adds into ``grad_obj``."""
        eng = self.engine
        ctx = self.device
        probe = self._probe(probe_real, probe_imag)
        coords = self._coords(this_i_theta)
        eng.set_batch(this_pos_batch, target)
        yr = eng.y_footprint(this_pos_batch)
        rool = bool(self.rotate_out_of_loop) and not self.common_vars.get('two_d_mode')
        if not (rool and obj is getattr(self, 'arr_rot', None)):
            # (rotate_out_of_loop: ``obj`` IS the rotated object; the driver hands over arr_rot, whose slice-major twin the
            # engine already holds from rotate_outside(); any other array is loaded as it is, coords = None)
            eng.rotate(obj, coords, yr)
        ctx.fork()
        eng.flush_loss_copy()       # the previous minibatch's loss read-back: on the side stream, beside this kernel
        self._flush_reg_copy()      # ... and its regulariser value (before the kernel below overwrites it)
        if side_hook is not None:
            side_hook()
        # init_grad: grad_obj is uninitialised -- the regulariser kernel writes it (one pass) or it is zero-filled
        rx = getattr(self, 'restricted_planes', None) if want_grad else None
        if rx is not None and not init_grad:
            # the restricted exchange defers the regulariser gradient to the shard owners (R-fold, after the reduction): an
            # evaluation that ACCUMULATES into a buffer initialised elsewhere would add it here as well and count it twice
            raise RuntimeError('restricted_planes is set but the gradient buffer is not initialised by this evaluation '
                               '(init_grad=False): the regulariser term would be counted twice')
        if rx is not None:
            # footprint-restricted exchange (DataParallelObject.exchange_and_update(touched=...)): the gradient buffer carries
            # the DATA term only, on the planes the global batch touches -- zero those; the regulariser's gradient is added by
            # the shard owners after the reduction, only its VALUE (for the loss log) is evaluated here
            plane = int(np.prod(eng.obj_size[1:])) * 2 * 4
            check(ctx.lib.adm_memset(ctx.handle, grad_obj.ptr + rx[0] * plane, 0, (rx[1] - rx[0]) * plane))
            self._reg_pending = self._regularize_launch(obj, None, False, value_only=True) if regularize else False
        else:
            self._reg_pending = self._regularize_launch(obj, grad_obj if want_grad else None, init_grad and want_grad) if regularize else False
            if init_grad and want_grad and not regularize:
                grad_obj.zero_()
        if want_grad and isinstance(coords, RotationTable):
            # first minibatch of an angle: its rotation-adjoint tables are built here, on the side stream beside the
            # multislice kernel (0.4 ms of emit + sort), not on the main stream when the back-rotation asks for them
            coords.csr(eng.plan)
        nxt = self.__dict__.pop('prefetch_table', None)
        if want_grad and isinstance(nxt, RotationTable):
            # ... and the NEXT angle's (the driver hands its table over during the current angle's last minibatch): built here, beside
            # this kernel, instead of in front of the next angle's first launch with the GPU idle behind the host
            nxt.csr(eng.plan)
        B = len(np.asarray(this_pos_batch).reshape(-1, 2))
        early_cover = want_grad and B <= eng.N_CU
        if early_cover:
            eng.build_cover()       # the overlap-add's cover lists only need the positions: built beside the kernel
        nb = self.__dict__.pop('next_batch', None)
        if nb is not None and want_grad:
            # the NEXT evaluation's measured data (the driver says which): host -> device now, beside this kernel
            t_next = self._target(nb[0], nb[1])
            if isinstance(t_next, np.ndarray):
                self._staged = ((int(nb[0]), np.asarray(nb[1]).tobytes()), eng.stage_target(t_next))
        ctx.end_fork()
        gp = None
        if want_probe_grad:
            if self._grad_probe_dev is None or self._grad_probe_dev.shape != probe.shape:
                self._grad_probe_dev = self.device.empty(probe.shape)
            gp = self._grad_probe_dev.zero_()
        shifts, idx = self._shift_args(probe_pos_correction, this_i_theta, this_ind_batch)
        gsh = None
        if want_shift_grad:
            if shifts is None:
                raise RuntimeError('gradient w.r.t. probe_pos_correction requested but no corrections were passed')
            if getattr(self, '_grad_shift_dev', None) is None or self._grad_shift_dev.shape != shifts.shape:
                self._grad_shift_dev = self.device.empty(shifts.shape)
            gsh = self._grad_shift_dev.zero_()
        mb = B // self.batch_group
        gs = 2.0 / (mb * eng.n_det)     # each reference minibatch is a mean over ITS positions (and the kept detector pixels)
        if want_grad and shifts is None and B > eng.N_CU:
            # no join here: the overlapped launch forks again, and the side stream is in order, so its overlap-adds queue
            # behind the regulariser kernel while the first round of workgroups already runs beside it
            eng.multislice_overlapped(probe, grad_probe=gp, grad_scale=gs, want_pred=want_pred)
        else:
            eng.multislice(probe, grad_probe=gp, want_grad=want_grad, want_pred=want_pred, grad_scale=gs, shifts=shifts,
                           shift_index=idx, grad_shifts=gsh, accumulate=False)
            ctx.join()              # (the side stream's work is a fraction of the kernel's time: the join does not wait)
            if want_grad:
                eng.accumulate_tiles()
        self._last_mb = mb
        if want_grad:
            eng.rotate_adjoint(grad_obj, coords, yr)
        return gp, gsh

    def _regularize_launch(self, obj, grad_obj, init_grad=False, value_only=False):
        """Queue the regulariser kernels (value into a device scalar, gradient added to grad_obj if given) on the
        current stream.  ``init_grad``: grad_obj holds garbage and is initialised here (regulariser kernel in 'set' mode,
        or a zero fill).  ``value_only``: no gradient at all (plain L1 / TV).  Returns True if a value is pending."""
        from .regularizers import ReweightedL1Regularizer
        self._flush_reg_copy()      # the previous value leaves before this launch overwrites it
        ad, ab, gm = combined_weights(self.reg_list)
        rw = [r for r in self.reg_list if isinstance(r, ReweightedL1Regularizer)]
        plain = ad != 0 or ab != 0 or gm != 0
        if init_grad and not plain:
            grad_obj.zero_()
        if not plain and not rw:
            return False
        if self._reg_val is None:
            self._reg_val = self.device.zeros((1,))
        self._reg_val.zero_()
        if value_only:
            if rw:
                raise NotImplementedError('value-only regulariser evaluation with a reweighted L1 term')
            k = float(self.batch_group)
            check(self.device.lib.adm_reg_grad(self.engine.plan.handle, obj.ptr, ad * k, ab * k, gm * k, None, self._reg_val.ptr))
            return True
        if grad_obj is None:
            if getattr(self, '_scratch_grad', None) is None or self._scratch_grad.size != obj.size:
                self._scratch_grad = self.device.empty((obj.size,))
            grad_obj = self._scratch_grad
        k = float(self.batch_group)     # every fused minibatch adds the regulariser once (forward_model.py:138-139)
        if plain:
            fn = self.device.lib.adm_reg_grad_set if init_grad else self.device.lib.adm_reg_grad
            check(fn(self.engine.plan.handle, obj.ptr, ad * k, ab * k, gm * k, grad_obj.ptr, self._reg_val.ptr))
        for r in rw:
            if r.weight_l1 is None:
                raise RuntimeError('ReweightedL1Regularizer: update_l1_weight() has not been called')
            check(self.device.lib.adm_reg_grad_weighted(self.engine.plan.handle, obj.ptr, r.weight_l1.ptr, float(r.alpha_d or 0.) * k,
                                                        float(r.alpha_b or 0.) * k, grad_obj.ptr, self._reg_val.ptr))
        return True

    def _reg_value_async(self, pending):
        """Register the read-back of the regulariser value; returns a callable giving the float.  Like the data term's sums
        (MultisliceEngine.loss_async) the 4-byte copy itself is DEFERRED: it is queued by the next evaluation inside its
        side-stream region, before the regulariser kernel that overwrites the value -- on the main stream it sat between the
        back-rotation and the optimiser kernel with its dependency gap (~12 us per minibatch) -- or by the callable, whichever
        comes first."""
        if not pending:
            return lambda: 0.0
        from .device import PinnedArray, Event
        if getattr(self, '_reg_pinned', None) is None:
            self._reg_pinned = [PinnedArray(self.device, (1,)) for _ in range(2)]
            self._reg_events = [Event(self.device) for _ in range(2)]
            self._reg_slot = 0
        self._flush_reg_copy()
        self._reg_slot ^= 1
        k = self._reg_slot
        self._reg_deferred = k
        g = float(self.batch_group)

        def value():
            if getattr(self, '_reg_deferred', None) == k:
                self._flush_reg_copy()
            self._reg_events[k].synchronize()
            return float(self._reg_pinned[k].array[0]) / g
        return value

    def _flush_reg_copy(self):
        """Queue the deferred regulariser-value read-back, if any, on the stream the context is enqueuing on right now."""
        k = getattr(self, '_reg_deferred', None)
        if k is not None:
            self._reg_deferred = None
            self._reg_pinned[k].copy_from_async(self._reg_val, 4)
            self._reg_events[k].record()

    def _regularize(self, obj, grad_obj):
        """Adds the regulariser gradient to grad_obj (if given) and returns the regulariser value (blocking)."""
        return self._reg_value_async(self._regularize_launch(obj, grad_obj))()

    def _queue_loss(self, data_loss_fn=None):
        """current_loss <- data term + regulariser, both read back lazily."""
        tok = self.engine.loss_async(last=self._last_mb) if data_loss_fn is None else None
        regv = self._reg_value_async(getattr(self, '_reg_pending', False))
        eng = self.engine
        self._loss_thunk = (lambda: eng.loss_result(tok) + regv()) if data_loss_fn is None else (lambda: data_loss_fn() + regv())

    def _resident_enabled(self):
        if getattr(self, '_resident', None) is None:
            shp = getattr(self.prj, 'shape', None)
            limit = float(os.environ.get('ADM_RESIDENT_DATA_MB', '1024')) * 2 ** 20
            self._resident = {} if (shp is not None and len(shp) == 4 and 4.0 * np.prod(shp) <= limit) else False
        return self._resident is not False

    def prefetch_data(self, i_theta):
        """Resident datasets only: the processed data of angle ``i_theta`` on the device, uploaded ASYNCHRONOUSLY through a pinned
        staging ring the first time the angle is asked for (a blocking copy here drains the stream: 1.3 ms of idle GPU at every
        first-touch angle).  The driver asks one minibatch before the angle begins, so the host-side preparation (|prj|, a copy
        into pinned memory: ~2 ms for 529 x 72 x 72 values) overlaps the current angle's last launch.  Returns the DeviceArray,
        or None when the dataset is streamed."""
        if not self._resident_enabled():
            return None
        td = self.common_vars.get('theta_downsample') or 1
        key = int(i_theta) * td
        dev = self._resident.get(key)
        if dev is None:
            host = self.get_data(i_theta, np.arange(self.prj.shape[1]), theta_downsample=td, ds_level=self.common_vars.get('ds_level', 1))
            up = getattr(self.device, 'array_async', None) or self.device.array
            dev = up(np.ascontiguousarray(host, dtype=np.float32))
            self._resident[key] = dev
        return dev

    def _target(self, this_i_theta, this_ind_batch):
        """The minibatch's measured data as the loss wants it (get_data).  Small datasets -- 2-D ptychography above all, where every
        epoch revisits the same angle -- are kept RESIDENT on the device, one processed [n_pos, Py, Px] array per angle uploaded the
        first time the angle is met; a minibatch that is a run of consecutive positions (the reference sorts a minibatch's indices,
        adorym/ptychography.py:907) is then a view of it: no host work, no copy.  Larger datasets (config 3: 5.5 GB) keep streaming
        their minibatches through the pinned ring, where the host work hides behind a 2 ms GPU step.  ADM_RESIDENT_DATA_MB (1024)."""
        td = self.common_vars.get('theta_downsample') or 1
        ind = np.asarray(this_ind_batch)
        if self._resident_enabled() and len(ind) > 0 and int(ind[-1]) - int(ind[0]) == len(ind) - 1 and np.all(np.diff(ind) == 1):
            key = int(this_i_theta) * td
            dev = self._resident.get(key)
            if dev is None:
                dev = self.prefetch_data(this_i_theta)
            n_px = dev.shape[1] * dev.shape[2]
            return dev.view(int(ind[0]) * n_px, (len(ind), dev.shape[1], dev.shape[2]))
        return self.get_data(this_i_theta, this_ind_batch, theta_downsample=td, ds_level=self.common_vars.get('ds_level', 1))

    # ------------------------------------------------------------------ reference interface
    def predict(self, obj, probe_real, probe_imag, probe_defocus_mm, probe_pos_offset, this_i_theta, this_pos_batch, prj,
                probe_pos_correction, this_ind_batch, tilt_ls, prj_pos_offset):
        """Predicted detector magnitudes [minibatch, Py, Px] (host float32), adorym/forward_model.py:179-387."""
        self._check_static(probe_defocus_mm, probe_pos_offset, probe_pos_correction, prj_pos_offset)
        B = len(this_pos_batch)
        zeros = np.zeros((B,) + tuple(self.engine.probe_size), np.float32)
        self._run(obj, probe_real, probe_imag, this_i_theta, this_pos_batch, zeros, want_grad=False, want_pred=True,
                  probe_pos_correction=probe_pos_correction, this_ind_batch=this_ind_batch, regularize=False)
        self.i_call += 1
        return self.engine.pred()

    def get_loss_function(self):
        def calculate_loss(obj, probe_real, probe_imag, probe_defocus_mm, probe_pos_offset, this_i_theta, this_pos_batch, prj,
                           probe_pos_correction, this_ind_batch, tilt_ls, prj_pos_offset):
            self._check_static(probe_defocus_mm, probe_pos_offset, probe_pos_correction, prj_pos_offset)
            target = self._target(this_i_theta, this_ind_batch)
            self._run(obj, probe_real, probe_imag, this_i_theta, this_pos_batch, target, want_grad=False,
                      probe_pos_correction=probe_pos_correction, this_ind_batch=this_ind_batch)
            self._queue_loss()
            return self.current_loss
        calculate_loss.forward_model = self
        return calculate_loss

    def loss_and_gradients(self, opt_args_ls, grad_obj, obj, probe_real, probe_imag, probe_defocus_mm, probe_pos_offset,
                           this_i_theta, this_pos_batch, prj, probe_pos_correction, this_ind_batch, tilt_ls, prj_pos_offset,
                           _side_hook=None, _init_grad=False):
        """
        The hand-derived replacement of ``torch.autograd.grad(loss, [args in opt_args_ls])``
        (adorym/wrappers.py:300-331): accumulates d loss/d obj into ``grad_obj`` (DeviceArray) and returns
        the gradients ordered like opt_args_ls: index 0 -> grad_obj, probe_real/probe_imag indices -> host arrays.
        """
        self._check_static(probe_defocus_mm, probe_pos_offset, probe_pos_correction, prj_pos_offset)
        staged = self.__dict__.pop('_staged', None)
        if staged is not None and staged[0] == (int(this_i_theta), np.asarray(this_ind_batch).tobytes()):
            target = staged[1]          # uploaded during the previous evaluation (PtychographyModel._run, next_batch)
        else:
            target = self._target(this_i_theta, this_ind_batch)
        i_pr, i_pi = self.get_argument_index('probe_real'), self.get_argument_index('probe_imag')
        i_pc = self.get_argument_index('probe_pos_correction')
        want_probe = (i_pr in opt_args_ls) or (i_pi in opt_args_ls)
        gp, gsh = self._run(obj, probe_real, probe_imag, this_i_theta, this_pos_batch, target, want_grad=True, grad_obj=grad_obj,
                            want_probe_grad=want_probe, probe_pos_correction=probe_pos_correction, this_ind_batch=this_ind_batch,
                            want_shift_grad=i_pc in opt_args_ls, side_hook=_side_hook, init_grad=_init_grad)
        self._queue_loss()          # self.current_loss fetches it on first access
        out = []
        for i in opt_args_ls:
            if i == 0:
                out.append(grad_obj)
            elif i in (i_pr, i_pi):
                # both entries reference ONE interleaved device array [n_modes,Py,Px,2] (real, imag): no host round trip
                out.append(gp)
            elif i == i_pc:
                out.append(gsh)             # DeviceArray, dense like probe_pos_correction (zero outside this minibatch)
            else:
                raise NotImplementedError("gradient w.r.t. '%s' is outside the accelerated path" % self.argument_ls[i])
        return tuple(out)


SingleBatchFullfieldModel = PtychographyModel
SingleBatchPtychographyModel = PtychographyModel


class SparseMultisliceModel(ForwardModel):
    def __init__(self, *a, **k):
        raise NotImplementedError('SparseMultisliceModel is outside the accelerated path (not in BASELINE configs)')


class MultiDistModel(PtychographyModel):
    """
    adorym/forward_model.py:809-1092 for one undivided field of view (the reference's n_blocks == 1 branch, config 5):
    the S = 1 object is illuminated by the (plane) probe and Fresnel-propagated to every distance of ``free_prop_cm``
    (fresnel_propagate_wrapped); the loss compares with the measured holograms, optionally registered by the affine
    matrices ``prj_affine_ls`` (w.affine_transform).  Gradients w.r.t. obj, probe, free_prop_cm and prj_affine_ls come from
    the hand-derived adjoint in adm_holo_fwd_adj.  ``common_vars_dict['holo_engine']`` is an adorym_amd.HolographyEngine.

    Data divided into sub-tiles and / or propagated with a safe zone (n_blocks > 1 or safe_zone_width > 0, :884-1034):
    ``common_vars_dict['tile_engine']`` is a MultisliceEngine built for the sequence of distances -- tile = sub-hologram + 2 safe
    zones, one detector-plane Fresnel kernel per distance in the plan (adm_plan_set_detector_kernels), detector mask = the
    sub-hologram's window -- and a minibatch of tiles at all distances is ONE launch of the multislice kernel with one probe
    window per entry (adm_multislice_fwd_adj_pp); object gradient only.
    """

    def __init__(self, loss_function_type='lsq', distribution_mode=None, device=None, common_vars_dict=None,
                 raw_data_type='magnitude', simulation_mode=False, run_bfloat16=False, run_float64=False):
        super(MultiDistModel, self).__init__(loss_function_type, distribution_mode, device, common_vars_dict, raw_data_type,
                                             simulation_mode=simulation_mode, run_bfloat16=run_bfloat16, run_float64=run_float64)
        if loss_function_type != 'lsq':
            raise NotImplementedError('MultiDistModel: only the LSQ loss is on the accelerated path')
        self.holo = common_vars_dict['holo_engine'] if common_vars_dict else None
        self.tile_engine = common_vars_dict.get('tile_engine') if common_vars_dict else None
        self._data_key = None
        self._data_dev = None
        self._small = {}

    def _check(self, safe_zone_width, ctf_lg_kappa, probe_pos_correction):
        cv = self.common_vars
        for flag in ('optimize_probe_defocusing', 'optimize_probe_pos_offset', 'optimize_prj_pos_offset', 'optimize_ctf_lg_kappa'):
            if cv.get(flag):
                raise NotImplementedError('%s with MultiDistModel is outside the accelerated path' % flag)
        if cv.get('optimize_all_probe_pos'):
            # one (sy, sx) per distance applied to the measured holograms (forward_model.py:1075-1085)
            if cv.get('optimize_prj_affine'):
                raise NotImplementedError('optimize_all_probe_pos together with optimize_prj_affine is outside the accelerated path')
            if self.tile_engine is not None:
                raise NotImplementedError('optimize_all_probe_pos with multi-distance data divided into sub-tiles is outside the accelerated path')
            if probe_pos_correction is None:
                raise ValueError('optimize_all_probe_pos needs probe_pos_correction [n_dists, 2]')
        if not cv.get('two_d_mode'):
            raise NotImplementedError('MultiDistModel is accelerated for two_d_mode (one object slice) only')
        if self.tile_engine is not None:
            if int(safe_zone_width or 0) != int(cv.get('safe_zone_width') or 0):
                raise ValueError('safe_zone_width differs from the width the tile engine was built for')
            for flag in ('optimize_free_prop', 'optimize_prj_affine'):
                if cv.get(flag):
                    raise NotImplementedError('%s with multi-distance data divided into sub-tiles is outside the accelerated path' % flag)
        elif safe_zone_width not in (0, None):
            raise NotImplementedError('safe_zone_width > 0 needs the tile engine (the driver builds it)')

    # ------------------------------------------------------------------ sub-tiles + safe zone (forward_model.py:884-1034)
    # One engine serves all distances: its plan holds the n_dists detector-plane kernels and a launch lists every tile of the
    # minibatch n_dists times in a row ("tile-major": tile j at every distance, then tile j + 1), so entry b belongs to tile
    # b // n_dists and distance b % n_dists.  The reference's order -- predictions and data -- is distance-major
    # [i_dist * n_blocks + tile] (:1019-1021, 1051-1054); only predict() has to put its output back into it.
    def _tile_probes(self, probe_real, probe_imag, pos):
        """One probe window [1, T, T] per launch entry, cut from the full-field probe padded with 1 + 0i
        (forward_model.py:916-925, 944-994).  Line :1005 passes ``subprobe_imag_ls_ls[k][i_mode, :, :]`` (no leading ':'): with
        one mode that is the IMAGINARY window of the first tile of the n_dp_batch chunk, used for the whole chunk -- kept, it is
        what the reference computes (golden F18 pins it with a probe that varies over the field).  The probe is not optimised on
        this path, so the windows of a batch are built once per distinct batch and stay on the device."""
        cv = self.common_vars
        szw = int(cv.get('safe_zone_width') or 0)
        eng = self.tile_engine
        T, nd = eng.probe_size, eng.n_dists
        # which probe the cached windows were cut from: a device array by identity (nothing updates it on this path: the driver
        # refuses optimize_probe with tiles), host arrays by content
        if isinstance(probe_real, DeviceArray):
            ident = ('dev', probe_real.ptr, tuple(probe_real.shape))
        else:
            host = np.stack([np.asarray(probe_real, np.float32), np.asarray(probe_imag, np.float32)], -1)
            ident = ('host', host.shape, host.tobytes())
        if getattr(self, '_probe_ident', None) != ident:
            if isinstance(probe_real, DeviceArray):
                host = probe_real.get()
            self._probe_host = host.reshape(host.shape[-3], host.shape[-2], 2)
            self._probe_cache = {}
            self._probe_ident = ident
        key = pos.tobytes()
        dev = self._probe_cache.get(key)
        if dev is None:
            pr, pi = self._probe_host[..., 0], self._probe_host[..., 1]
            pad = np.zeros((2, 2), int)
            if szw > 0:
                pad = calculate_pad_len(pr.shape, pos - szw, T)
                pr = np.pad(pr, [tuple(pad[0]), tuple(pad[1])], mode='constant', constant_values=1)
                pi = np.pad(pi, [tuple(pad[0]), tuple(pad[1])], mode='constant', constant_values=0)
            n_dp = int(cv.get('n_dp_batch') or len(pos))
            host = np.empty((len(pos), 1, T[0], T[1], 2), np.float32)
            for j, p in enumerate(pos):
                y, x = int(p[0] + pad[0, 0] - szw), int(p[1] + pad[1, 0] - szw)
                j0 = (j // n_dp) * n_dp
                y0, x0 = int(pos[j0, 0] + pad[0, 0] - szw), int(pos[j0, 1] + pad[1, 0] - szw)
                host[j, 0, :, :, 0] = pr[y:y + T[0], x:x + T[1]]
                host[j, 0, :, :, 1] = pi[y0:y0 + T[0], x0:x0 + T[1]]
            if len(self._probe_cache) > 4096:
                self._probe_cache.clear()
            dev = self._probe_cache[key] = self.device.array(np.repeat(host, nd, axis=0))
        return dev

    def _tile_frames(self, this_i_theta, tiles):
        """The measured holograms of ``tiles`` at every distance, tile-major, each inside a zero frame of the tile's size (the
        safe zone carries no data and no weight in the loss): |prj[theta, i_dist * n_blocks + tile]| through get_data
        (forward_model.py:1049-1056)."""
        cv = self.common_vars
        szw = int(cv.get('safe_zone_width') or 0)
        eng = self.tile_engine
        T, nd = eng.probe_size, eng.n_dists
        n_blocks = self.prj.shape[1] // nd
        tiles = np.asarray(tiles)
        ind = (tiles[:, None] + n_blocks * np.arange(nd)[None, :]).reshape(-1)
        t = self.get_data(this_i_theta, ind, theta_downsample=cv.get('theta_downsample') or 1, ds_level=cv.get('ds_level', 1))
        if szw == 0:
            return t
        full = np.zeros((len(ind), T[0], T[1]), np.float32)
        full[:, szw:szw + t.shape[1], szw:szw + t.shape[2]] = t
        return full

    def _tile_targets(self, this_i_theta, this_ind_batch):
        """Datasets whose framed copy fits ADM_RESIDENT_DATA_MB live on the device, one tile-major [n_blocks * n_dists, Ty, Tx] array
        per angle: a minibatch that is a run of consecutive tiles (the reference sorts a minibatch's indices,
        adorym/ptychography.py:907) is a view of it -- no host work, no copy."""
        eng = self.tile_engine
        T, nd = eng.probe_size, eng.n_dists
        n_blocks = self.prj.shape[1] // nd
        ind = np.asarray(this_ind_batch)
        if getattr(self, '_tiles_resident', None) is None:
            limit = float(os.environ.get('ADM_RESIDENT_DATA_MB', '1024')) * 2 ** 20
            self._tiles_resident = {} if 4.0 * self.prj.shape[0] * self.prj.shape[1] * T[0] * T[1] <= limit else False
        if self._tiles_resident is not False and len(ind) > 0 and int(ind[-1]) - int(ind[0]) == len(ind) - 1 and np.all(np.diff(ind) == 1):
            key = int(this_i_theta)
            dev = self._tiles_resident.get(key)
            if dev is None:
                dev = self._tiles_resident[key] = self.device.array(self._tile_frames(this_i_theta, np.arange(n_blocks)))
            return dev.view(int(ind[0]) * nd * T[0] * T[1], (len(ind) * nd, T[0], T[1]))
        return self._tile_frames(this_i_theta, ind)

    def _run_tiled(self, obj, probe_real, probe_imag, this_i_theta, this_pos_batch, this_ind_batch, want_grad, grad_obj=None,
                   want_pred=False):
        """ONE launch over (tiles of the minibatch) x (distances).  The loss is the mean over distances x tiles x sub-hologram
        pixels (forward_model.py:1019-1029, 1087) -- the engine's mean over its n_dists * B entries and the kept detector pixels."""
        szw = int(self.common_vars.get('safe_zone_width') or 0)
        eng = self.tile_engine
        nd = eng.n_dists
        pos = np.ascontiguousarray(np.round(np.asarray(this_pos_batch)).astype(np.int64).reshape(-1, 2))
        B = len(pos)
        probes_b = self._tile_probes(probe_real, probe_imag, pos)
        tpos = np.repeat(pos - szw, nd, axis=0)
        eng.set_batch(tpos, self._tile_targets(this_i_theta, this_ind_batch))
        yr = eng.y_footprint(tpos)
        eng.rotate(obj, None, yr)
        if want_grad:
            # the overlap-add's cover lists only need the positions: built on the side stream, beside the launch
            self.device.fork()
            eng.flush_loss_copy()
            eng.build_cover()
            self.device.end_fork()
        eng.multislice(None, want_grad=want_grad, want_pred=want_pred, probes_b=probes_b, accumulate=False)
        if want_grad:
            self.device.join()
            eng.accumulate_tiles()
            eng.rotate_adjoint(grad_obj, None, yr)
        if want_pred:
            sy, sx = self.prj.shape[-2:]
            p = eng.pred()[:, szw:szw + sy, szw:szw + sx]
            return np.ascontiguousarray(p.reshape(B, nd, sy, sx).transpose(1, 0, 2, 3).reshape(nd * B, sy, sx))
        tok = eng.loss_async()
        return lambda: eng.loss_result(tok)

    def _dev(self, name, value, shape):
        """Small parameter arrays: pass DeviceArrays through, upload (and cache) host values."""
        if isinstance(value, DeviceArray):
            return value
        host = np.ascontiguousarray(np.asarray(value, dtype=np.float32).reshape(shape))
        cur = self._small.get(name)
        if cur is None or cur[0].tobytes() != host.tobytes():
            self._small[name] = (host, self.device.array(host))
        return self._small[name][1]

    def _data(self, this_i_theta):
        """All holograms of this angle, raw (the sqrt for intensity data happens after the affine registration)."""
        td = self.common_vars.get('theta_downsample') or 1
        key = int(this_i_theta) * td
        if self._data_key != key:
            t = np.ascontiguousarray(np.abs(np.asarray(self.prj[key])), dtype=np.float32)
            self._data_dev = self.device.array(t)
            self._data_key = key
        return self._data_dev

    def _spectrum(self, this_i_theta):
        """FFT2(|data_d|) of this angle's holograms, kept on the device beside them (the shift refinement multiplies it with a
        phase ramp per distance and minibatch)."""
        data = self._data(this_i_theta)
        if getattr(self, '_spec_key', None) != self._data_key:
            self._spec_dev = self.holo.data_spectrum(data)
            self._spec_key = self._data_key
        return self._spec_dev

    def _run(self, obj, probe_real, probe_imag, this_i_theta, free_prop_cm, prj_affine_ls, want_grad, grad_obj=None, grads=None,
             want_pred=False, overwrite=False, probe_pos_correction=None):
        nd = self.holo.n_dists
        probe = self._probe(probe_real, probe_imag)          # [1, ny, nx, 2]
        dists = self._dev('free_prop_cm', free_prop_cm, (nd,))
        if self.common_vars.get('optimize_all_probe_pos'):   # forward_model.py:1075-1085
            g = grads or {}
            if g.get('dists') is not None:
                raise NotImplementedError('optimize_all_probe_pos together with optimize_free_prop is outside the accelerated path')
            shifts = self._dev('probe_pos_correction', probe_pos_correction, (nd, 2))
            self.holo.forward_adjoint_shifted(obj, probe, dists, self._spectrum(this_i_theta), shifts, want_grad=want_grad, grad_obj=grad_obj,
                                              grad_probe=g.get('probe'), grad_shifts=g.get('shifts'), want_pred=want_pred, overwrite=overwrite)
            return
        aff = None
        if self.common_vars.get('optimize_prj_affine'):      # forward_model.py:1063-1070
            aff = self._dev('prj_affine_ls', prj_affine_ls, (nd, 2, 3))
        g = grads or {}
        self.holo.forward_adjoint(obj, probe, dists, self._data(this_i_theta), affine=aff, want_grad=want_grad, grad_obj=grad_obj,
                                  grad_probe=g.get('probe'), grad_dists=g.get('dists'), grad_affine=g.get('affine') if aff is not None else None,
                                  want_pred=want_pred, overwrite=overwrite)

    def predict(self, obj, probe_real, probe_imag, probe_defocus_mm, probe_pos_offset, this_i_theta, this_pos_batch, prj,
                probe_pos_correction, this_ind_batch, free_prop_cm, safe_zone_width, prj_affine_ls, ctf_lg_kappa, prj_pos_offset):
        """Detected magnitudes [n_dists, ny, nx] (host float32), adorym/forward_model.py:819-1034."""
        self._check(safe_zone_width, ctf_lg_kappa, probe_pos_correction)
        self.i_call += 1
        if self.tile_engine is not None:
            return self._run_tiled(obj, probe_real, probe_imag, this_i_theta, this_pos_batch, this_ind_batch, want_grad=False, want_pred=True)
        self._run(obj, probe_real, probe_imag, this_i_theta, free_prop_cm, prj_affine_ls, want_grad=False, want_pred=True,
                  probe_pos_correction=probe_pos_correction)
        return self.holo.pred()

    def get_loss_function(self):
        def calculate_loss(obj, probe_real, probe_imag, probe_defocus_mm, probe_pos_offset, this_i_theta, this_pos_batch, prj,
                           probe_pos_correction, this_ind_batch, free_prop_cm, safe_zone_width, prj_affine_ls, ctf_lg_kappa,
                           prj_pos_offset):
            self._check(safe_zone_width, ctf_lg_kappa, probe_pos_correction)
            if self.tile_engine is not None:
                datav = self._run_tiled(obj, probe_real, probe_imag, this_i_theta, this_pos_batch, this_ind_batch, want_grad=False)
                self.current_loss = float(datav() + self._regularize(obj, None))
                return self.current_loss
            self._run(obj, probe_real, probe_imag, this_i_theta, free_prop_cm, prj_affine_ls, want_grad=False,
                      probe_pos_correction=probe_pos_correction)
            self.current_loss = float(self.holo.loss() + self._regularize(obj, None))
            return self.current_loss
        calculate_loss.forward_model = self
        return calculate_loss

    def loss_and_gradients(self, opt_args_ls, grad_obj, obj, probe_real, probe_imag, probe_defocus_mm, probe_pos_offset,
                           this_i_theta, this_pos_batch, prj, probe_pos_correction, this_ind_batch, free_prop_cm, safe_zone_width,
                           prj_affine_ls, ctf_lg_kappa, prj_pos_offset, _side_hook=None, _init_grad=False):
        """Replacement of torch.autograd.grad over MultiDistModel's loss: gradients ordered like opt_args_ls; index 0 ->
        grad_obj (accumulated in place), probe_real/probe_imag -> one interleaved DeviceArray, free_prop_cm -> DeviceArray
        [n_dists], prj_affine_ls -> DeviceArray [n_dists,2,3]."""
        self._check(safe_zone_width, ctf_lg_kappa, probe_pos_correction)
        if _side_hook is not None:
            _side_hook()
        if self.tile_engine is not None:
            if list(opt_args_ls) != [0]:
                raise NotImplementedError('multi-distance data divided into sub-tiles: only the object gradient is on the accelerated path')
            # the regulariser kernel initialises the gradient buffer ('set' mode, or a zero fill), the launches add to it
            regv = self._reg_value_async(self._regularize_launch(obj, grad_obj, init_grad=bool(_init_grad)))
            datav = self._run_tiled(obj, probe_real, probe_imag, this_i_theta, this_pos_batch, this_ind_batch, want_grad=True, grad_obj=grad_obj)
            self._loss_thunk = lambda: datav() + regv()
            return (grad_obj,)
        fa = getattr(self, 'fused_adam', None)
        if fa is not None:
            # the driver has established that this minibatch's update is plain Adam on exactly these gradients (one rank, no
            # regulariser / constraint / mask): gradient launch group and optimiser steps are ONE call, nothing is returned
            nd = self.holo.n_dists
            probe = self._probe(probe_real, probe_imag)
            dists = self._dev('free_prop_cm', free_prop_cm, (nd,))
            aff = self._dev('prj_affine_ls', prj_affine_ls, (nd, 2, 3)) if self.common_vars.get('optimize_prj_affine') else None
            if (fa['dists'] is not None and not isinstance(free_prop_cm, DeviceArray)) or \
                    (fa['affine'] is not None and not isinstance(prj_affine_ls, DeviceArray)):
                raise RuntimeError('fused holography update: the optimised parameters must live on the device')
            self.holo.forward_adjoint_adam(obj, probe, dists, self._data(this_i_theta), fa['obj_mv'], fa['step_obj'], fa['i_batch'], affine=aff,
                                           dists_mv=fa['dists'][:2] if fa['dists'] else None, step_dists=fa['dists'][2] if fa['dists'] else 0.,
                                           affine_mv=fa['affine'][:2] if fa['affine'] else None, step_affine=fa['affine'][2] if fa['affine'] else 0.,
                                           affine_pin=fa['pin'], b1=fa['b1'], b2=fa['b2'], eps=fa['eps'])
            datav = self.holo.loss_async()
            self._loss_thunk = lambda: datav()
            return tuple(None for _ in opt_args_ls)
        # _init_grad: the object-gradient buffer holds garbage -> the engine overwrites every gradient ('=' instead of '+=': no
        # zero fills on a path that is bound by the number of launches); otherwise it accumulates into zeroed small buffers
        nd = self.holo.n_dists
        idx = {n: self.get_argument_index(n) for n in ('probe_real', 'probe_imag', 'free_prop_cm', 'prj_affine_ls', 'probe_pos_correction')}
        grads = {}
        if idx['probe_real'] in opt_args_ls or idx['probe_imag'] in opt_args_ls:
            probe = self._probe(probe_real, probe_imag)
            if self._grad_probe_dev is None or self._grad_probe_dev.shape != probe.shape:
                self._grad_probe_dev = self.device.empty(probe.shape)
            grads['probe'] = self._grad_probe_dev
        if idx['free_prop_cm'] in opt_args_ls:
            if getattr(self, '_gd', None) is None:
                self._gd = self.device.empty((nd,))
            grads['dists'] = self._gd if _init_grad else self._gd.zero_()
        if idx['prj_affine_ls'] in opt_args_ls:
            if getattr(self, '_ga', None) is None:
                self._ga = self.device.empty((nd, 2, 3))
            grads['affine'] = self._ga if _init_grad else self._ga.zero_()
        if idx['probe_pos_correction'] in opt_args_ls:
            if getattr(self, '_gs', None) is None:
                self._gs = self.device.empty((nd, 2))
            grads['shifts'] = self._gs.zero_()           # (adm_holo_shift_grad accumulates)
        self._run(obj, probe_real, probe_imag, this_i_theta, free_prop_cm, prj_affine_ls, want_grad=True, grad_obj=grad_obj, grads=grads,
                  overwrite=bool(_init_grad), probe_pos_correction=probe_pos_correction)
        # loss and regulariser value are read back lazily (the driver looks at them after the next minibatch has been queued)
        regv = self._reg_value_async(self._regularize_launch(obj, grad_obj))
        datav = self.holo.loss_async()
        self._loss_thunk = lambda: datav() + regv()
        out = []
        for i in opt_args_ls:
            if i == 0:
                out.append(grad_obj)
            elif i in (idx['probe_real'], idx['probe_imag']):
                out.append(grads['probe'])
            elif i == idx['free_prop_cm']:
                out.append(grads['dists'])
            elif i == idx['prj_affine_ls']:
                out.append(grads['affine'])
            elif i == idx['probe_pos_correction']:
                out.append(grads['shifts'])
            else:
                raise NotImplementedError("gradient w.r.t. '%s' is outside the accelerated path" % self.argument_ls[i])
        return tuple(out)
