"""
Multi-distance near-field holography on the GPU: the engine behind MultiDistModel
(adorym/forward_model.py:809-1092) for one undivided field of view and one object slice.  See include/adm.h
(adm_holo_*) and csrc/adm_holo.hip.
"""
import ctypes as C
import numpy as np

from ._lib import HoloDesc, check
from .constants import PI
from .device import DeviceArray


class HolographyEngine(object):
    def __init__(self, ctx, field_size, n_dists, energy_ev, psize_cm, sign_convention=1, unknown_type='real_imag',
                 raw_data_type='intensity', scale_ri_by_k=True):
        self.ctx = ctx
        self.ny, self.nx = int(field_size[0]), int(field_size[1])
        self.n_dists = int(n_dists)
        lmbda_nm = 1240. / energy_ev
        voxel_nm = psize_cm * 1e7
        k1 = 2. * PI * voxel_nm / lmbda_nm if scale_ri_by_k else 1.
        desc = HoloDesc(ny=self.ny, nx=self.nx, n_dists=self.n_dists, lambda_nm=lmbda_nm, voxel_nm_y=voxel_nm, voxel_nm_x=voxel_nm,
                        sign_convention=int(sign_convention), unknown_type=1 if unknown_type == 'real_imag' else 0,
                        raw_intensity=1 if raw_data_type == 'intensity' else 0, k1=float(k1))
        h = C.c_void_p()
        check(ctx.lib.adm_holo_create(ctx.handle, C.byref(desc), C.byref(h)))
        self.handle = h
        # the per-distance loss sums are written by the last kernel of a launch group straight into page-locked HOST memory (two
        # slots, alternating): no device-to-host copy is queued, the host reads the slot once an event recorded behind the launch
        # has happened (the 4.6 us blit kernel and its dependency gap were 5 % of a config-5 minibatch)
        from .device import PinnedArray, Event
        self._pinned = [PinnedArray(ctx, (max(self.n_dists, 16),)) for _ in range(2)]
        self._events = [Event(ctx) for _ in range(2)]
        self._slot = 0
        self._pred = None

    def __del__(self):
        try:
            if getattr(self, 'handle', None) and self.ctx.handle:
                # (the handle keeps a pointer to the context: destroying it after Context.close() would touch freed memory)
                self.ctx.lib.adm_holo_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def forward_adjoint(self, obj, probe, dists_cm, data, affine=None, want_grad=True, grad_obj=None, grad_probe=None,
                        grad_dists=None, grad_affine=None, want_pred=False, overwrite=False):
        """All arguments are DeviceArrays: obj [ny,nx,(1,)2], probe [ny,nx,2], dists_cm [n_dists], data [n_dists,ny,nx] (raw),
        affine [n_dists,2,3] or None.  Gradients are accumulated (+=) except grad_probe (overwritten); ``overwrite=True``:
        all of them are overwritten (no zero fills needed)."""
        if want_pred and self._pred is None:
            self._pred = DeviceArray(self.ctx, (self.n_dists, self.ny, self.nx), np.float32)
        p = lambda a: a.ptr if a is not None else None
        self._slot ^= 1
        check(self.ctx.lib.adm_holo_fwd_adj(self.handle, obj.ptr, probe.ptr, dists_cm.ptr, p(affine), data.ptr, (2 if overwrite else 1) if want_grad else 0,
                                            p(grad_obj), p(grad_probe), p(grad_dists), p(grad_affine),
                                            self._pred.ptr if want_pred else None, self._pinned[self._slot].handle))

    def forward_adjoint_adam(self, obj, probe, dists_cm, data, obj_mv, step_obj, i_batch, affine=None, dists_mv=None, step_dists=0.,
                             affine_mv=None, step_affine=0., affine_pin=None, b1=0.9, b2=0.999, eps=1e-7, want_pred=False):
        """forward_adjoint(overwrite=True) FUSED with the Adam steps of the object (always), the distances (``dists_mv`` = (m, v)
        DeviceArrays, or None) and the affine matrices (``affine_mv``, with ``affine_pin`` = DeviceArray copied over the first
        entries afterwards): adm_holo_fwd_adj_adam.  obj / dists_cm / affine are updated in place; no gradient is stored."""
        from ._lib import HoloAdam
        if want_pred and self._pred is None:
            self._pred = DeviceArray(self.ctx, (self.n_dists, self.ny, self.nx), np.float32)
        p = lambda a: a.ptr if a is not None else None
        o = HoloAdam(m_obj=obj_mv[0].ptr, v_obj=obj_mv[1].ptr, step_obj=float(step_obj),
                     m_dists=p(dists_mv[0]) if dists_mv else None, v_dists=p(dists_mv[1]) if dists_mv else None, step_dists=float(step_dists),
                     m_affine=p(affine_mv[0]) if affine_mv else None, v_affine=p(affine_mv[1]) if affine_mv else None,
                     step_affine=float(step_affine), affine_pin=p(affine_pin), affine_pin_n=affine_pin.size if affine_pin is not None else 0,
                     i_batch=int(i_batch), b1=float(b1), b2=float(b2), eps=float(eps))
        self._slot ^= 1
        check(self.ctx.lib.adm_holo_fwd_adj_adam(self.handle, obj.ptr, probe.ptr, dists_cm.ptr, p(affine), data.ptr, C.byref(o),
                                                 self._pred.ptr if want_pred else None, self._pinned[self._slot].handle))

    # ---- per-distance shift refinement of the measured holograms (optimize_all_probe_pos, adorym/forward_model.py:1075-1085) ----
    def data_spectrum(self, data):
        """FFT2(|data_d|) of the raw holograms [n_dists, ny, nx] (DeviceArray), transposed ([d][kx][ky]); once per dataset."""
        spec = DeviceArray(self.ctx, (self.n_dists, self.nx, self.ny, 2), np.float32)
        check(self.ctx.lib.adm_holo_data_spectrum(self.handle, data.ptr, spec.ptr))
        return spec

    def forward_adjoint_shifted(self, obj, probe, dists_cm, spectrum, shifts, want_grad=True, grad_obj=None, grad_probe=None,
                                grad_shifts=None, want_pred=False, overwrite=False):
        """forward_adjoint() against the holograms Fourier-shifted by ``shifts`` [n_dists, 2] = (sy, sx) (realign_image_fourier, real
        part kept): the registered targets are formed from ``spectrum`` (data_spectrum) in front of the launch group and, with
        ``grad_shifts`` (+=), dL/dshifts behind it."""
        if getattr(self, '_targets', None) is None:
            self._targets = DeviceArray(self.ctx, (self.n_dists, self.ny, self.nx), np.float32)
            self._cot = DeviceArray(self.ctx, (self.n_dists, self.ny, self.nx), np.float32)
        lib = self.ctx.lib
        check(lib.adm_holo_shift_targets(self.handle, spectrum.ptr, shifts.ptr, self._targets.ptr))
        want_sg = want_grad and grad_shifts is not None
        check(lib.adm_holo_set_registration(self.handle, self._cot.ptr if want_sg else None, 1))
        try:
            self.forward_adjoint(obj, probe, dists_cm, self._targets, want_grad=want_grad, grad_obj=grad_obj, grad_probe=grad_probe,
                                 want_pred=want_pred, overwrite=overwrite)
        finally:
            check(lib.adm_holo_set_registration(self.handle, None, 0))
        if want_sg:
            check(lib.adm_holo_shift_grad(self.handle, self._cot.ptr, spectrum.ptr, shifts.ptr, grad_shifts.ptr))

    def shifted_targets(self):
        """The registered holograms T_d of the last forward_adjoint_shifted() (host copy; tests)."""
        return self._targets.get()

    def loss(self):
        """mean over (distance, pixel) of the squared residual of the last launch -- blocks."""
        return self.loss_async()()

    def loss_async(self):
        """Record an event behind the last launch and return a callable that gives its loss: the host does not wait, the driver
        resolves it after the next minibatch has been queued (the next launch writes the OTHER slot)."""
        k = self._slot
        self._events[k].record()
        norm = float(self.n_dists * self.ny * self.nx)

        def value():
            self._events[k].synchronize()
            return float(self._pinned[k].array[:self.n_dists].astype(np.float64).sum() / norm)
        return value

    def pred(self):
        return self._pred.get()
