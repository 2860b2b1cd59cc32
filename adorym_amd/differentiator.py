"""
Differentiator with the reference's seam (adorym/differentiator.py:28-42): ``create_loss_node(loss, opt_args_ls)``
then ``get_gradients(**kwargs)`` -> tuple of gradients ordered like ``opt_args_ls``.  The reference calls
``torch.autograd.grad`` here (adorym/wrappers.py:300-331); this build calls the forward model's hand-derived
adjoint (HIP kernels).  No autograd, no tracing.
"""


class Differentiator(object):

    def __init__(self):
        self.loss_object = None
        self.opt_args_ls = []
        self.loss_args = {}
        self._grad_buf = None

    def create_loss_node(self, loss, opt_args_ls=None):
        """``loss``: the closure returned by ForwardModel.get_loss_function() (it carries ``.forward_model``).
        ``opt_args_ls``: indices into the predict() argument list to differentiate against."""
        if not hasattr(loss, 'forward_model'):
            raise TypeError('the loss function must come from an adorym_amd ForwardModel.get_loss_function()')
        self.loss_object = loss
        self.opt_args_ls = list(opt_args_ls) if opt_args_ls is not None else [0]

    def get_gradients(self, _accumulate_into=None, _side_hook=None, _init_grad=False, **kwargs):
        """Returns (d loss/d obj, ...) like the reference.  The object gradient is a DeviceArray: by default a
        buffer owned by this Differentiator that is overwritten on every call; pass ``_accumulate_into`` to add
        the gradient straight into the caller's accumulation buffer (what ptychography.py:1063-1066 does next)."""
        fm = self.loss_object.forward_model
        obj = kwargs['obj']
        if _accumulate_into is None:
            if self._grad_buf is None or self._grad_buf.size != obj.size:
                self._grad_buf = obj.ctx.empty((obj.size,))
            self._grad_buf.zero_()
            target = self._grad_buf
        else:
            target = _accumulate_into
        if _side_hook is not None:      # work the model may queue on the side stream next to the multislice chain
            kwargs['_side_hook'] = _side_hook
        if _init_grad:                  # the accumulation buffer holds garbage: the model initialises it (no separate zero fill)
            kwargs['_init_grad'] = True
        return fm.loss_and_gradients(self.opt_args_ls, target, **kwargs)

    def get_l_h_hessian_and_h_x_jacobian_mvps(self, *args, **kwargs):
        raise NotImplementedError('Gauss-Newton products serve the Curveball optimizer only; outside the accelerated path')
