"""DP-mode containers of the reference (adorym/array_ops.py:45-345): the object function, its gradient
buffer and the finite-support mask, as device arrays."""
import numpy as np


class LargeArray(object):
    def __init__(self, full_size, distribution_mode=None, monochannel=False, output_folder=None, device=None):
        if distribution_mode is not None:
            raise NotImplementedError("distribution_mode '%s' is outside the accelerated path (DP only)" % distribution_mode)
        self.full_size = list(full_size)
        self.distribution_mode = distribution_mode
        self.monochannel = monochannel
        self.output_folder = output_folder
        self.arr = None
        self.device = device


class ObjectFunction(LargeArray):
    """adorym/array_ops.py:165-251.  ``arr`` is a DeviceArray [Y,X,Z,2]."""

    def __init__(self, full_size, distribution_mode=None, output_folder=None, ds_level=1, object_type='normal', device=None):
        super(ObjectFunction, self).__init__(full_size, distribution_mode, False, output_folder, device)
        self.ds_level = ds_level
        self.object_type = object_type

    @staticmethod
    def initial_values(shape, initial_guess=None, random_guess_means_sigmas=(8.7e-7, 5.1e-8, 1e-7, 1e-8),
                       object_type='normal', non_negativity=False, unknown_type='delta_beta'):
        """initialize_object_for_dp (adorym/util.py:71-125), delta_beta branch.  Uses the legacy global
        NumPy RNG like the reference (the caller seeds it)."""
        if initial_guess is None:
            delta = np.random.normal(size=shape, loc=random_guess_means_sigmas[0], scale=random_guess_means_sigmas[2])
            beta = np.random.normal(size=shape, loc=random_guess_means_sigmas[1], scale=random_guess_means_sigmas[3])
        else:
            delta = np.array(initial_guess[0], dtype=np.float64)
            beta = np.array(initial_guess[1], dtype=np.float64)
        if unknown_type == 'real_imag':
            # the guess is (magnitude, phase) and is converted to (real, imag) (adorym/util.py:106-122)
            if object_type == 'phase_only':
                delta[...] = 1
            elif object_type == 'absorption_only':
                beta[...] = 0
            cplx = delta * np.exp(1j * beta)
            return np.stack([cplx.real, cplx.imag], -1).astype(np.float32)
        if object_type == 'phase_only':
            beta[...] = 0
        elif object_type == 'absorption_only':
            delta[...] = 0
        if non_negativity:
            delta[delta < 0] = 0
            beta[beta < 0] = 0
        return np.stack([delta, beta], -1).astype(np.float32)

    def initialize_array_with_values(self, obj_delta, obj_beta, device=None):
        dev = device or self.device
        self.arr = dev.array(np.stack([obj_delta, obj_beta], -1), np.float32)


class Gradient(ObjectFunction):
    """adorym/array_ops.py:289-301."""

    def __init__(self, obj, forward_model=None):
        super(Gradient, self).__init__(obj.full_size, obj.distribution_mode, obj.output_folder, obj.ds_level, obj.object_type,
                                       obj.device)
        self.forward_model = forward_model


class Mask(LargeArray):
    """adorym/array_ops.py:304-345: monochannel support mask [Y,X,Z] (device float array)."""

    def __init__(self, full_size, finite_support_mask_path=None, distribution_mode=None, output_folder=None, ds_level=1,
                 device=None):
        super(Mask, self).__init__(full_size, distribution_mode, True, output_folder, device)
        self.mask = None
        self.finite_support_mask_path = finite_support_mask_path
        self.ds_level = ds_level

    def initialize_array_with_values(self, mask, device=None):
        dev = device or self.device
        self.mask = dev.array(np.asarray(mask), np.float32)
