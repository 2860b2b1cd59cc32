"""Process-wide switches that scripts written for the reference may read or set (`adorym.global_settings.backend`, the
precision flags).  This package has one backend, 'hip', and computes in fp32: reconstruct_ptychography() rejects the
bf16 / fp64 requests instead of consulting these flags."""

_DEFAULTS = dict(backend='hip', xpu=False, run_bf16=False, run_fp64=False)
globals().update(_DEFAULTS)


def reset():
    """Restore the defaults (tests)."""
    globals().update(_DEFAULTS)
