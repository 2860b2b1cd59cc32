"""Regulariser descriptors with the reference's constructor signatures (adorym/regularizers.py).
The values / gradients are evaluated on the GPU by adm_reg_grad (L1 + TV fused in one pass; the plan's
unknown_type selects the delta_beta or the real_imag definition, regularizers.py:30-46 / 95-110)."""


class Regularizer(object):
    """adorym/regularizers.py:5-15."""

    def __init__(self, unknown_type='delta_beta'):
        if unknown_type not in ('delta_beta', 'real_imag'):
            raise ValueError("unknown_type must be 'delta_beta' or 'real_imag'")
        self.unknown_type = unknown_type

    def weights(self):
        """(alpha_d, alpha_b, gamma) contribution of this term."""
        return 0., 0., 0.


class L1Regularizer(Regularizer):
    """alpha_d * mean|delta| + alpha_b * mean|beta| (adorym/regularizers.py:18-46)."""

    def __init__(self, alpha_d, alpha_b, unknown_type='delta_beta'):
        super(L1Regularizer, self).__init__(unknown_type)
        self.alpha_d = alpha_d
        self.alpha_b = alpha_b

    def weights(self):
        return float(self.alpha_d or 0.), float(self.alpha_b or 0.), 0.


class TVRegularizer(Regularizer):
    """gamma * (TV(delta) + TV(beta)), periodic anisotropic TV / voxel count
    (adorym/regularizers.py:86-110 -> adorym/util.py:1427-1440)."""

    def __init__(self, gamma, unknown_type='delta_beta'):
        super(TVRegularizer, self).__init__(unknown_type)
        self.gamma = gamma

    def weights(self):
        return 0., 0., float(self.gamma or 0.)


class ReweightedL1Regularizer(Regularizer):
    """alpha_d*mean(w_d|delta|) + alpha_b*mean(w_b|beta|) with weights refreshed by the driver
    (adorym/regularizers.py:49-84, adorym/ptychography.py:995-1000); for unknown_type='real_imag', with
    wm = w_re^2 + w_im^2: alpha_d*mean(wm*| |o| - mean|o| |) + alpha_b*mean(wm*|arg o|) (regularizers.py:73-82).
    ``weight_l1`` is a device array [Y,X,Z,2]; the arithmetic is adm_reg_grad_weighted's."""

    def __init__(self, alpha_d, alpha_b, unknown_type='delta_beta'):
        super(ReweightedL1Regularizer, self).__init__(unknown_type)
        self.alpha_d = alpha_d
        self.alpha_b = alpha_b
        self.weight_l1 = None

    def update_l1_weight(self, weight_l1):
        self.weight_l1 = weight_l1


class CorrRegularizer(Regularizer):
    def __init__(self, gamma, unknown_type='delta_beta'):
        raise NotImplementedError('CorrRegularizer is outside the accelerated path (unused by every config)')


class GradCorrRegularizer(Regularizer):
    def __init__(self, gamma, unknown_type='delta_beta'):
        raise NotImplementedError('GradCorrRegularizer is outside the accelerated path (unused by every config)')


def combined_weights(reg_list):
    ad = ab = gm = 0.
    for r in reg_list:
        if isinstance(r, ReweightedL1Regularizer):
            continue
        a, b, g = r.weights()
        ad += a; ab += b; gm += g
    return ad, ab, gm
