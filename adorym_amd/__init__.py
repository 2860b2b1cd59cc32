"""
adorym_amd -- MI355X-native implementation of Adorym's multislice forward model, hand-derived
adjoint and object update (the hot path of mdw771/adorym), behind the reference's Python API.

Compute runs in hand-written HIP kernels (libadm.so, C ABI in include/adm.h) loaded with ctypes.
There is no CPU fallback: importing the device layer without the built library raises.
"""
from .constants import PI  # noqa: F401
from . import global_settings  # noqa: F401
from .device import Context, DeviceArray, Plan, Event  # noqa: F401
from .propagate import MultisliceEngine, get_kernel, gen_freq_mesh  # noqa: F401

__version__ = '0.1.0'
