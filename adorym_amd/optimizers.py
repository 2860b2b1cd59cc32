"""
Optimisers with the reference's plugin contract (adorym/optimizers.py:32-126, 242-252, 275-337, 432-464):

    opt = AdamOptimizer(name, output_folder, distribution_mode, options_dict)
    opt.create_container(whole_object_size, use_checkpoint, device_obj)
    x = opt.apply_gradient(x, gradient, i_batch, **opt.options_dict)

``x`` / ``gradient`` are DeviceArrays (flat views are fine); the update runs in the fused HIP kernel
adm_adam_step / adm_gd_step in place and returns ``x``.  The moment arrays live in
``params_whole_array_dict`` like the reference's.
"""
import numpy as np

from ._lib import check
from .device import DeviceArray
from .array_ops import Gradient


class Optimizer(object):
    """adorym/optimizers.py:32-126 (distribution_mode=None subset)."""

    def __init__(self, name, output_folder='.', params_list=(), distribution_mode=None, options_dict=None, forward_model=None):
        if distribution_mode is not None:
            raise NotImplementedError("distribution_mode '%s' is outside the accelerated path (DP only)" % distribution_mode)
        self.name = name
        self.forward_model = forward_model
        self.output_folder = output_folder
        self.params_list = params_list
        self.params_whole_array_dict = {}
        self.i_batch = 0
        self.index_in_grad_returns = None
        self.distribution_mode = distribution_mode
        self.options_dict = options_dict if options_dict is not None else {}
        self.grads = None
        self.whole_object_size = None
        self.ctx = None

    def __str__(self):
        s = self.__class__.__name__ + '; '
        for k in self.options_dict.keys():
            s = s + k + ': ' + str(self.options_dict[k]) + '; '
        return s

    def create_container(self, whole_object_size, use_checkpoint=False, device_obj=None, use_numpy=False, dtype='float32'):
        """``device_obj`` is an adorym_amd.Context."""
        self.whole_object_size = whole_object_size
        self.ctx = device_obj
        self.create_param_arrays(whole_object_size, device=device_obj)

    def create_param_arrays(self, whole_object_size, device=None, use_numpy=False):
        self.whole_object_size = whole_object_size
        for param_name in self.params_list:
            self.params_whole_array_dict[param_name] = device.zeros(tuple(whole_object_size))

    def set_index_in_grad_return(self, ind):
        self.index_in_grad_returns = ind

    def convert_gradient(self, gradient):
        return gradient.arr if isinstance(gradient, Gradient) else gradient


class AdamOptimizer(Optimizer):
    """adorym/optimizers.py:265-337: m/v moments, bias correction with (i_batch+1), eps=1e-7."""

    def __init__(self, name, output_folder='.', distribution_mode=None, options_dict=None, forward_model=None):
        super(AdamOptimizer, self).__init__(name, output_folder=output_folder, params_list=['m', 'v'],
                                            distribution_mode=distribution_mode, options_dict=options_dict,
                                            forward_model=forward_model)

    def apply_gradient(self, x, gradient, i_batch, step_size=0.001, b1=0.9, b2=0.999, eps=1e-7, flags=0, mask=None,
                       update_batch_count=True, **kwargs):
        g = self.convert_gradient(gradient)
        m, v = self.params_whole_array_dict['m'], self.params_whole_array_dict['v']
        ctx = x.ctx
        check(ctx.lib.adm_adam_step(ctx.handle, x.ptr, g.ptr, m.ptr, v.ptr, 0, x.size, int(i_batch), float(step_size),
                                    float(b1), float(b2), float(eps), int(flags), mask.ptr if mask is not None else None))
        if update_batch_count:
            self.i_batch += 1
        return x


class GDOptimizer(Optimizer):
    """adorym/optimizers.py:432-464 (step halving schedule :452-460)."""

    def __init__(self, name, output_folder='.', distribution_mode=None, options_dict=None, forward_model=None):
        super(GDOptimizer, self).__init__(name, output_folder=output_folder, params_list=[], distribution_mode=distribution_mode,
                                          options_dict=options_dict, forward_model=forward_model)

    @staticmethod
    def scheduled_step(i_batch, step_size, dynamic_rate=True, first_downrate_iteration=92):
        if dynamic_rate:
            threshold_iteration = first_downrate_iteration
            i = 1
            while threshold_iteration < i_batch:
                threshold_iteration += first_downrate_iteration * 2 ** i
                i += 1
                step_size /= 2.
        return step_size

    def apply_gradient(self, x, gradient, i_batch, step_size=0.001, dynamic_rate=True, first_downrate_iteration=92, flags=0,
                       mask=None, **kwargs):
        g = self.convert_gradient(gradient)
        step = self.scheduled_step(i_batch, step_size, dynamic_rate, first_downrate_iteration)
        ctx = x.ctx
        check(ctx.lib.adm_gd_step(ctx.handle, x.ptr, g.ptr, 0, x.size, float(step), int(flags),
                                  mask.ptr if mask is not None else None))
        return x


class MomentumOptimizer(Optimizer):
    """adorym/optimizers.py:366-411: v = gamma*v + step*g; x = x - v."""

    def __init__(self, name, output_folder='.', distribution_mode=None, options_dict=None, forward_model=None):
        super(MomentumOptimizer, self).__init__(name, output_folder=output_folder, params_list=['v'],
                                                distribution_mode=distribution_mode, options_dict=options_dict,
                                                forward_model=forward_model)

    def apply_gradient(self, x, gradient, i_batch, step_size=0.001, gamma=0.9, flags=0, mask=None, **kwargs):
        g = self.convert_gradient(gradient)
        v = self.params_whole_array_dict['v']
        ctx = x.ctx
        check(ctx.lib.adm_momentum_step(ctx.handle, x.ptr, g.ptr, v.ptr, 0, x.size, float(step_size), float(gamma), int(flags),
                                        mask.ptr if mask is not None else None))
        return x


def apply_small_params(ctx, items, i_batch):
    """The small optimisable parameters of one minibatch (adorym/optimizers.py:1022-1083: probe, probe_pos_correction,
    free_prop_cm, prj_affine_ls) in ONE launch (adm_adam_step_small) when every one of them is driven by a plain AdamOptimizer with
    the same (b1, b2, eps); otherwise one by one through the optimisers' own apply_gradient.  ``items``: dicts with
    opt, x, g (DeviceArrays), and optionally center_cols (drift guard, :1046-1048), pin (DeviceArray copied over the first
    entries of x, :1067-1073), zero_grad (the gradient accumulator is zero-filled once used).  Same arithmetic either way."""
    import ctypes as C
    from ._lib import SmallParam, SMALL_PARAMS_MAX
    if not items:
        return
    keys = {(float(it['opt'].options_dict.get('b1', 0.9)), float(it['opt'].options_dict.get('b2', 0.999)),
             float(it['opt'].options_dict.get('eps', 1e-7))) for it in items}
    plain = all(type(it['opt']) is AdamOptimizer and set(it['opt'].options_dict) <= {'step_size', 'b1', 'b2', 'eps'} for it in items)
    if plain and len(keys) == 1 and len(items) <= SMALL_PARAMS_MAX:
        # the descriptor array is the same minibatch after minibatch (same buffers, same step sizes): build it once per combination
        sig = tuple((it['x'].ptr, it['g'].ptr, it['opt'].params_whole_array_dict['m'].ptr, it['opt'].params_whole_array_dict['v'].ptr,
                     it['x'].size, float(it['opt'].options_dict.get('step_size', 0.001)), int(it.get('center_cols', 0)),
                     bool(it.get('zero_grad')), it['pin'].ptr if it.get('pin') is not None else 0) for it in items)
        cache = ctx.__dict__.setdefault('_small_param_descs', {})
        arr = cache.get(sig)
        if arr is None:
            if len(cache) > 64:
                cache.clear()
            arr = (SmallParam * len(items))()
            for k, it in enumerate(items):
                o = it['opt']
                pin = it.get('pin')
                arr[k] = SmallParam(x=it['x'].ptr, g=it['g'].ptr, m=o.params_whole_array_dict['m'].ptr, v=o.params_whole_array_dict['v'].ptr,
                                    n=it['x'].size, step_size=float(o.options_dict.get('step_size', 0.001)),
                                    center_cols=int(it.get('center_cols', 0)), zero_grad=1 if it.get('zero_grad') else 0,
                                    pin=pin.ptr if pin is not None else None, pin_n=pin.size if pin is not None else 0)
            cache[sig] = arr
        for it in items:
            it['opt'].i_batch += 1
        b1, b2, eps = next(iter(keys))
        check(ctx.lib.adm_adam_step_small(ctx.handle, arr, len(items), int(i_batch), b1, b2, eps))
        return
    for it in items:
        it['opt'].apply_gradient(it['x'], it['g'], i_batch, **it['opt'].options_dict)
        if it.get('center_cols'):
            check(ctx.lib.adm_center_rows(ctx.handle, it['x'].ptr, it['x'].size // it['center_cols'], it['center_cols']))
        if it.get('pin') is not None:
            check(ctx.lib.adm_d2d(ctx.handle, it['x'].ptr, it['pin'].ptr, it['pin'].nbytes))
        if it.get('zero_grad'):
            it['g'].zero_()


def _unsupported(name):
    class _U(Optimizer):
        def __init__(self, *a, **k):
            raise NotImplementedError('%s is outside the accelerated path (SURVEY section 2, component 5)' % name)
    _U.__name__ = name
    return _U


CurveballOptimizer = _unsupported('CurveballOptimizer')
CGOptimizer = _unsupported('CGOptimizer')
ScipyOptimizer = _unsupported('ScipyOptimizer')
