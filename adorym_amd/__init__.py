"""
adorym_amd -- MI355X-native implementation of Adorym's multislice forward model, hand-derived
adjoint and object update (the hot path of mdw771/adorym), behind the reference's Python API.

Compute runs in hand-written HIP kernels (libadm.so, C ABI in include/adm.h) loaded with ctypes.
There is no CPU fallback: importing the device layer without the built library raises.
"""
from .constants import PI  # noqa: F401
from . import global_settings  # noqa: F401
from .device import Context, DeviceArray, Plan, Event, PinnedArray, UploadRing  # noqa: F401
from .propagate import MultisliceEngine, RotationTable, AngleBatch, get_kernel, gen_freq_mesh  # noqa: F401
from .holography import HolographyEngine  # noqa: F401

__version__ = '0.1.0'

# ---- the reference's public names for this path (adorym/__init__.py:1-11 star-exports) ----
from .forward_model import (ForwardModel, PtychographyModel, SingleBatchFullfieldModel,  # noqa: F401,E402
                            SingleBatchPtychographyModel, SparseMultisliceModel, MultiDistModel)
from .differentiator import Differentiator  # noqa: F401,E402
from .optimizers import (Optimizer, AdamOptimizer, GDOptimizer, MomentumOptimizer, CurveballOptimizer,  # noqa: F401,E402
                         CGOptimizer, ScipyOptimizer)
from .regularizers import (Regularizer, L1Regularizer, TVRegularizer, ReweightedL1Regularizer,  # noqa: F401,E402
                           CorrRegularizer, GradCorrRegularizer)
from .array_ops import ObjectFunction, Gradient, Mask  # noqa: F401,E402
from .ptychography import reconstruct_ptychography  # noqa: F401,E402
