"""
CPU oracle for the Adorym multislice forward + hand-adjoint hot path.

*** TEST INFRASTRUCTURE ONLY ***  Only tests/, __graft_entry__.smoke() and the
``cpu_baseline`` leg of bench.py may import this module -- and there only as the
checker / the timed CPU baseline, never as the product path.  The product
(adorym_amd/) must fail loudly if the HIP extension is missing; it never falls
back to this file.

What this is: a plain-NumPy restatement of the reference algorithm
(mdw771/adorym, PyTorch-CPU backend) for the path named in BASELINE.json:
rotation lookup + bilinear gather, zero-pad + tile extraction, multislice
Fresnel propagation, far-field / near-field magnitude loss, the hand-derived
adjoint that replaces ``torch.autograd.grad``, L1/TV regularisers, Adam / GD
updates and the DP-mode batching order.  Each function cites the reference
``file:line`` it follows (paths relative to the reference repo root).

Parity pinning: the reference's own test-suite holds no golden vectors for
this path (one smoke test without assertions whose data file is absent), so
this oracle is pinned against outputs of the *imported reference itself*
(PyTorch CPU, fp64 and fp32) captured in this container by
``tests/golden/gen_goldens.py`` and committed as ``tests/golden/*.npz``;
``tests/test_oracle_vs_golden.py`` checks every function below against them.

The arithmetic type is selectable (``dtype='float64'`` = the fp64 oracle,
``'float32'`` = the timed CPU baseline that mirrors the reference's default).
"""
from __future__ import annotations

import math
import numpy as np

# adorym/constants.py:90 -- the reference uses this truncated value everywhere on the
# path (propagate.py:20 star-imports it); ptychography.py:52 redefines PI = 3.1415927 but
# only uses it as the default theta_end.
PI = 3.14159265359


def _cdtype(dtype):
    return np.complex128 if np.dtype(dtype) == np.float64 else np.complex64


# --------------------------------------------------------------------------------------
# R4  Fresnel transfer function
# --------------------------------------------------------------------------------------
def gen_freq_mesh(voxel_nm, shape):
    """adorym/propagate.py:54-60."""
    u = np.fft.fftfreq(shape[0])
    v = np.fft.fftfreq(shape[1])
    vv, uu = np.meshgrid(v, u)
    vv = vv / voxel_nm[1]
    uu = uu / voxel_nm[0]
    return uu, vv


def get_kernel(dist_nm, lmbda_nm, voxel_nm, grid_shape, fresnel_approx=True, sign_convention=1):
    """adorym/propagate.py:62-81.  Returns complex128 [Py, Px] (unshifted)."""
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


# --------------------------------------------------------------------------------------
# R1  rotation lookup table (fp16)
# --------------------------------------------------------------------------------------
def rotation_coords(array_size, theta, dtype='float32'):
    """
    adorym/util.py:446-477 + 492-516 (save_rotation_lookup), axis=0.

    Returns the fp16 table ``[X*Z, 2]`` of source coordinates ``(x_old, z_old)`` for every
    rotated-frame voxel, flat index ``x*Z + z``.  The reference evaluates this in torch at the
    dtype chosen by w.create_variable / w.create_constant (wrappers.py:121-176): float32 by
    default, float64 under run_float64 -- and stores float16 either way.
    NOTE the reference subtracts the *other* axis' centre (util.py:459-460) -- reproduced.
    """
    dt = np.dtype(dtype).type
    Y, X, Z = array_size
    cx = dt((X - 1) / 2)
    cz = dt((Z - 1) / 2)
    # coords_ls[0]: this_axis=1 (X): repeat over Z, minus image_center[other_axis=2]
    xc = np.repeat(np.arange(X), Z).astype(np.float64) - (Z - 1) / 2
    # coords_ls[1]: this_axis=2 (Z): tile over X, minus image_center[other_axis=1]
    zc = np.tile(np.arange(Z), X).astype(np.float64) - (X - 1) / 2
    coord_new = np.stack([xc, zc]).astype(dt)
    th = dt(theta)
    c = np.cos(th, dtype=dt)
    s = np.sin(th, dtype=dt)
    # torch.matmul [2,2]x[2,N]: two-term dot products, one rounding per product and per add
    # (no FMA contraction on the CPU path).
    old0 = (c * coord_new[0]).astype(dt) + ((-s) * coord_new[1]).astype(dt)
    old1 = (s * coord_new[0]).astype(dt) + (c * coord_new[1]).astype(dt)
    coord1_old = (old0 + cx).astype(dt)   # + image_center[1]
    coord2_old = (old1 + cz).astype(dt)   # + image_center[2]
    return np.stack([coord1_old, coord2_old], axis=1).astype(np.float16)


def _sample_setup(coords_fp16, X, Z, dtype):
    """
    Coordinate pipeline of apply_rotation -> w.grid_sample (util.py:536-552,
    wrappers.py:1105-1147) followed by torch's grid_sampler (align_corners=False,
    padding_mode='border', bilinear).

    coords fp16 -> float64 (util.py:539-540) -> normalised in float64
    ``-1 + 2 g / s + 1 / s`` (wrappers.py:1137; s = arr.shape[2:] = (X, Z) applied to the
    flipped (z, x) pair -- harmless for cubic arrays, reproduced as written) -> cast to the
    array dtype (wrappers.py:1141) -> un-normalised ``((g + 1) * size - 1) / 2`` in that dtype.
    """
    dt = np.dtype(dtype)
    c = coords_fp16.astype(np.float64)
    x_old = c[:, 0]
    z_old = c[:, 1]
    # after tc.flip: grid[..., 0] = z_old (torch "x" = W axis = Z), grid[..., 1] = x_old (H axis = X)
    gz = (-1 + 2. * z_old / X + 1. / X).astype(dt)   # divided by arr_shape[0] == X (sic)
    gx = (-1 + 2. * x_old / Z + 1. / Z).astype(dt)   # divided by arr_shape[1] == Z (sic)
    iz = ((gz + dt.type(1)) * dt.type(Z) - dt.type(1)) / dt.type(2)
    ix = ((gx + dt.type(1)) * dt.type(X) - dt.type(1)) / dt.type(2)
    iz = np.minimum(dt.type(Z - 1), np.maximum(iz, dt.type(0)))
    ix = np.minimum(dt.type(X - 1), np.maximum(ix, dt.type(0)))
    iz0 = np.floor(iz)
    ix0 = np.floor(ix)
    tz = (iz - iz0).astype(dt)
    tx = (ix - ix0).astype(dt)
    iz0 = iz0.astype(np.int64)
    ix0 = ix0.astype(np.int64)
    iz1 = iz0 + 1
    ix1 = ix0 + 1
    one = dt.type(1)
    # torch weights: nw = (ix_se - ix) * (iy_se - iy) ... with x = W (z here), y = H (x here)
    w00 = (one - tx) * (one - tz)   # (ix0, iz0)
    w01 = (one - tx) * tz           # (ix0, iz1)
    w10 = tx * (one - tz)           # (ix1, iz0)
    w11 = tx * tz                   # (ix1, iz1)
    v_x1 = ix1 <= X - 1
    v_z1 = iz1 <= Z - 1
    return ix0, ix1, iz0, iz1, w00, w01, w10, w11, v_x1, v_z1


def rotate_fwd(obj, coords_fp16, dtype=None):
    """R2: apply_rotation (util.py:536-552).  obj [Y,X,Z,C] -> rotated [Y,X,Z,C]."""
    dtype = obj.dtype if dtype is None else dtype
    Y, X, Z, C = obj.shape
    ix0, ix1, iz0, iz1, w00, w01, w10, w11, vx1, vz1 = _sample_setup(coords_fp16, X, Z, dtype)
    ix1c = np.minimum(ix1, X - 1)
    iz1c = np.minimum(iz1, Z - 1)
    o = obj.astype(dtype, copy=False)
    w01 = np.where(vz1, w01, 0)
    w10 = np.where(vx1, w10, 0)
    w11 = np.where(vx1 & vz1, w11, 0)
    out = (o[:, ix0, iz0, :] * w00[None, :, None] + o[:, ix0, iz1c, :] * w01[None, :, None]
           + o[:, ix1c, iz0, :] * w10[None, :, None] + o[:, ix1c, iz1c, :] * w11[None, :, None])
    return out.reshape(Y, X, Z, C).astype(dtype, copy=False)


def rotate_adj(grad_rot, coords_fp16, dtype=None):
    """R2 adjoint (autograd of grid_sampler_2d): scatter-add of the four weighted corners."""
    dtype = grad_rot.dtype if dtype is None else dtype
    Y, X, Z, C = grad_rot.shape
    ix0, ix1, iz0, iz1, w00, w01, w10, w11, vx1, vz1 = _sample_setup(coords_fp16, X, Z, dtype)
    g = grad_rot.reshape(Y, X * Z, C).astype(dtype, copy=False)
    out = np.zeros((Y, X * Z, C), dtype=dtype)
    flat = lambda a, b: a * Z + b
    for (ia, ib, ww, valid) in ((ix0, iz0, w00, None), (ix0, iz1, w01, vz1),
                                (ix1, iz0, w10, vx1), (ix1, iz1, w11, vx1 & vz1)):
        if valid is None:
            idx = flat(ia, ib)
            contrib = g * ww[None, :, None]
        else:
            idx = flat(ia[valid], ib[valid])
            contrib = g[:, valid, :] * ww[valid][None, :, None]
        for y in range(Y):
            np.add.at(out[y], idx, contrib[y])
    return out.reshape(Y, X, Z, C)


# --------------------------------------------------------------------------------------
# R3  padding + tile extraction
# --------------------------------------------------------------------------------------
def calculate_pad_len(this_obj_size, probe_pos, probe_size):
    """adorym/util.py:1374-1406 (both unknown_type branches are identical)."""
    probe_pos = np.asarray(probe_pos)
    pad_arr = np.array([[0, 0], [0, 0]])
    if min(probe_pos[:, 0]) < 0:
        pad_arr[0, 0] = -int(min(probe_pos[:, 0]))
    if max(probe_pos[:, 0]) + probe_size[0] > this_obj_size[0]:
        pad_arr[0, 1] = int(max(probe_pos[:, 0])) + probe_size[0] - this_obj_size[0]
    if min(probe_pos[:, 1]) < 0:
        pad_arr[1, 0] = -int(min(probe_pos[:, 1]))
    if max(probe_pos[:, 1]) + probe_size[1] > this_obj_size[1]:
        pad_arr[1, 1] = int(max(probe_pos[:, 1])) + probe_size[1] - this_obj_size[1]
    return pad_arr


def extract_tiles(obj_rot, pos_batch, probe_size, unknown_type='delta_beta'):
    """pad_object (util.py:1327-1351; delta_beta: zero pad, real_imag: pad with 1 + 0i) + the tile stack of
    forward_model.py:313-331.  Returns tiles [B, Py, Px, S, 2] and pad_arr."""
    pos = np.round(np.asarray(pos_batch)).astype(int)            # forward_model.py:248
    pad = calculate_pad_len(obj_rot.shape[:3], pos, probe_size)
    o = np.pad(obj_rot, [tuple(pad[0]), tuple(pad[1]), (0, 0), (0, 0)], mode='constant')
    if unknown_type == 'real_imag':
        inside = np.zeros(o.shape[:2], bool)
        inside[pad[0, 0]:o.shape[0] - pad[0, 1], pad[1, 0]:o.shape[1] - pad[1, 1]] = True
        o[~inside, :, 0] = 1
    tiles = np.stack([o[p[0] + pad[0, 0]: p[0] + pad[0, 0] + probe_size[0],
                        p[1] + pad[1, 0]: p[1] + pad[1, 0] + probe_size[1]] for p in pos])
    return tiles, pad


def scatter_tiles_adj(grad_tiles, pos_batch, obj_shape):
    """Adjoint of extract_tiles: overlap-add the tile gradients, dropping the pad region."""
    pos = np.round(np.asarray(pos_batch)).astype(int)
    B, Py, Px = grad_tiles.shape[:3]
    Y, X = obj_shape[:2]
    pad = calculate_pad_len(obj_shape[:3], pos, (Py, Px))
    buf = np.zeros((Y + pad[0].sum(), X + pad[1].sum()) + tuple(grad_tiles.shape[3:]), dtype=grad_tiles.dtype)
    for b, p in enumerate(pos):
        buf[p[0] + pad[0, 0]: p[0] + pad[0, 0] + Py, p[1] + pad[1, 0]: p[1] + pad[1, 0] + Px] += grad_tiles[b]
    return buf[pad[0, 0]: pad[0, 0] + Y, pad[1, 0]: pad[1, 0] + X]


# --------------------------------------------------------------------------------------
# R5-R8  multislice forward, detector-plane transform, loss
# --------------------------------------------------------------------------------------
class Physics(object):
    """Static parameters of multislice_propagate_batch (propagate.py:131-153, 196-207)."""

    def __init__(self, probe_size, energy_ev, psize_cm, free_prop_cm='inf', binning=1,
                 fresnel_approx=True, sign_convention=1, normalize_fft=False, kernel=None,
                 scale_ri_by_k=True, unknown_type='delta_beta'):
        self.probe_size = tuple(int(p) for p in probe_size)
        self.energy_ev = float(energy_ev)
        self.psize_cm = float(psize_cm)
        self.voxel_nm = np.array([psize_cm] * 3) * 1.e7            # propagate.py:146
        self.lmbda_nm = 1240. / energy_ev                            # propagate.py:148
        self.delta_nm = self.voxel_nm[-1]
        self.binning = int(binning)
        self.unknown_type = unknown_type      # 'delta_beta' | 'real_imag' (propagate.py:236-249)
        self.sigma = int(sign_convention)
        self.normalize_fft = bool(normalize_fft)
        self.free_prop_cm = free_prop_cm
        self.k1 = 2. * PI * self.delta_nm / self.lmbda_nm if scale_ri_by_k else 1.   # propagate.py:215
        if kernel is None:
            kernel = get_kernel(self.delta_nm * binning, self.lmbda_nm, self.voxel_nm, self.probe_size,
                                fresnel_approx=fresnel_approx, sign_convention=sign_convention)
        self.h = np.asarray(kernel)
        if free_prop_cm in (0, None):
            self.det_mode = 'none'
            self.h_free = None
        elif isinstance(free_prop_cm, str) and free_prop_cm == 'inf':
            self.det_mode = 'far'
            self.h_free = None
        else:
            # fresnel_propagate (propagate.py:537-553) always uses the Fresnel-approx kernel
            self.det_mode = 'fresnel'
            self.h_free = get_kernel(free_prop_cm * 1e7, self.lmbda_nm, self.voxel_nm, self.probe_size,
                                     sign_convention=sign_convention)

    def h_cast(self, dtype):
        """h_real/h_imag are cast separately to the working dtype (propagate.py:202-204)."""
        dt = np.dtype(dtype)
        return (self.h.real.astype(dt) + 1j * self.h.imag.astype(dt)).astype(_cdtype(dt))

    def h_free_cast(self, dtype):
        dt = np.dtype(dtype)
        return (self.h_free.real.astype(dt) + 1j * self.h_free.imag.astype(dt)).astype(_cdtype(dt))


def _fftshift2(a):
    """w.fftshift (wrappers.py:796-813) == np.fft.fftshift on the last two axes."""
    return np.fft.fftshift(a, axes=(-2, -1))


def _detector_fwd(psi, phys, dtype):
    """propagate.py:263-280 (+ wrappers.py:725-770)."""
    norm = 'ortho' if phys.normalize_fft else None
    if phys.det_mode == 'none':
        return psi
    if phys.det_mode == 'far':
        if phys.sigma == 1:
            return _fftshift2(np.fft.fft2(psi, norm=norm))
        return _fftshift2(np.fft.ifft2(psi, norm=norm))
    return np.fft.ifft2(np.fft.fft2(psi) * phys.h_free_cast(dtype))


def _detector_adj(G, phys, dtype):
    """Adjoint of _detector_fwd for the convention G := dL/dRe + i dL/dIm."""
    N = G.shape[-2] * G.shape[-1]
    if phys.det_mode == 'none':
        return G
    if phys.det_mode == 'far':
        G = np.fft.ifftshift(G, axes=(-2, -1))
        if phys.normalize_fft:
            return np.fft.ifft2(G, norm='ortho') if phys.sigma == 1 else np.fft.fft2(G, norm='ortho')
        if phys.sigma == 1:
            return np.fft.ifft2(G) * N        # (unnormalised fft2)^H
        return np.fft.fft2(G) / N             # (ifft2 with 1/N)^H
    return np.fft.ifft2(np.fft.fft2(G) * np.conj(phys.h_free_cast(dtype)))


def _slice_sums(tiles, i_step, binning):
    S = tiles.shape[3]
    lo = i_step * binning
    hi = min(lo + binning, S)
    if hi - lo == 1:
        return tiles[:, :, :, lo, 0], tiles[:, :, :, lo, 1], lo, hi
    return tiles[:, :, :, lo:hi, 0].sum(axis=3), tiles[:, :, :, lo:hi, 1].sum(axis=3), lo, hi


def _modulator(delta_s, beta_s, phys, dtype):
    """delta_beta: w.exp_complex(-k1*beta, -sigma*k1*delta) (propagate.py:241, wrappers.py:600-608);
    real_imag: the slice itself, c = re + i*im (propagate.py:243-247; binning > 1 would multiply slices)."""
    dt = np.dtype(dtype)
    if phys.unknown_type == 'real_imag':
        assert phys.binning == 1
        return (delta_s + 1j * beta_s).astype(_cdtype(dt))
    k1 = dt.type(phys.k1)
    e = np.exp(-k1 * beta_s)
    ph = -dt.type(phys.sigma) * k1 * delta_s
    return (e * np.cos(ph) + 1j * (e * np.sin(ph))).astype(_cdtype(dt))


def multislice_forward(tiles, probe, phys, dtype='float64', keep=False):
    """
    multislice_propagate_batch, non-projection delta_beta branch (propagate.py:195-270).

    tiles [B,Py,Px,S,2]; probe complex [Py,Px] (one mode) or [B,Py,Px] (one, e.g. Fourier-shifted, probe per
    position, forward_model.py:296-311).  Returns the detector-plane
    complex field [B,Py,Px] (and, with keep=True, the list of post-modulation fields psi'_s).
    """
    dt = np.dtype(dtype)
    cdt = _cdtype(dt)
    tiles = tiles.astype(dt, copy=False)
    B = tiles.shape[0]
    S = tiles.shape[3]
    n_steps = int(np.ceil(S / phys.binning))
    h = phys.h_cast(dt)
    probe = np.asarray(probe)
    psi = probe.astype(cdt).copy() if probe.ndim == 3 else np.broadcast_to(probe.astype(cdt), (B,) + probe.shape).copy()
    kept = []
    for i in range(n_steps):
        d, b, lo, hi = _slice_sums(tiles, i, phys.binning)
        c = _modulator(d, b, phys, dt)
        psi = (psi * c).astype(cdt)
        if keep:
            kept.append(psi)
        if i < n_steps - 1:
            psi = np.fft.ifft2(np.fft.fft2(psi) * h).astype(cdt)
    out = _detector_fwd(psi, phys, dt).astype(cdt)
    return (out, kept) if keep else out


def predict(tiles, probes, phys, dtype='float64'):
    """forward_model.py:337-375: |Psi| for one mode, sqrt(sum_m |Psi_m|^2) for several."""
    probes = np.asarray(probes)
    if probes.ndim == 2:
        probes = probes[None]
    fields = [multislice_forward(tiles, p, phys, dtype) for p in probes]
    if len(fields) == 1:
        return np.abs(fields[0]), fields
    inten = sum((f.real ** 2 + f.imag ** 2) for f in fields)
    return np.sqrt(inten), fields


def target_magnitude(meas, raw_data_type='magnitude'):
    """forward_model.py:113-119 + 88-93: abs(prj) (and sqrt for intensity data)."""
    t = np.abs(meas)
    return np.sqrt(t) if raw_data_type == 'intensity' else t


def mismatch_loss(pred, meas, loss_function_type='lsq', raw_data_type='magnitude', poisson_multiplier=1.):
    """ForwardModel.get_mismatch_loss (forward_model.py:88-103)."""
    if loss_function_type == 'lsq':
        return np.mean((pred - target_magnitude(meas, raw_data_type)) ** 2)
    a = np.abs(meas) ** 2 if raw_data_type == 'magnitude' else np.abs(meas)
    return np.mean(pred ** 2 * poisson_multiplier - a * poisson_multiplier * np.log(pred ** 2 * poisson_multiplier))


def _dloss_dpred(pred, meas, loss_function_type, raw_data_type, poisson_multiplier):
    n = pred.size
    if loss_function_type == 'lsq':
        return 2. * (pred - target_magnitude(meas, raw_data_type)) / n
    a = np.abs(meas) ** 2 if raw_data_type == 'magnitude' else np.abs(meas)
    return (2. * pred * poisson_multiplier - a * poisson_multiplier * 2. / pred) / n


def fourier_shift_phase(shape, shift, dtype):
    """The multiplier of realign_image_fourier (util.py:380-390): exp(-2 PI i (fx*shift[1] + fy*shift[0])), PI = 3.14159265359,
    frequencies and the argument evaluated in ``dtype`` like the reference's tensors."""
    dt = np.dtype(dtype)
    fy = np.fft.fftfreq(shape[0], 1).astype(dt)[:, None]
    fx = np.fft.fftfreq(shape[1], 1).astype(dt)[None, :]
    arg = (dt.type(-2) * dt.type(PI)) * (fx * dt.type(shift[1]) + fy * dt.type(shift[0]))
    return (np.cos(arg) + 1j * np.sin(arg)).astype(_cdtype(dt))


def fourier_shift(probes, shift, dtype='float64'):
    """realign_image_fourier(probe_real, probe_imag, shift, axes=(1, 2)) on complex probes [..., Py, Px]."""
    cdt = _cdtype(np.dtype(dtype))
    ph = fourier_shift_phase(probes.shape[-2:], shift, dtype)
    return np.fft.ifft2((np.fft.fft2(np.asarray(probes).astype(cdt)).astype(cdt) * ph).astype(cdt)).astype(cdt)


def forward_adjoint_tiles(tiles, probes, meas, phys, dtype='float64', loss_function_type='lsq',
                          raw_data_type='magnitude', poisson_multiplier=1., shifts=None, beamstop=None):
    """
    Loss + hand-derived gradient w.r.t. the tiles and the probe(s).  Replaces
    ``torch.autograd.grad`` (wrappers.py:322) over forward_model.py:337-375 +
    propagate.py:195-270 + forward_model.py:88-103.  SURVEY section 3.4.

    Returns loss, pred [B,Py,Px], grad_tiles [B,Py,Px,S,2], grad_probes complex [M,Py,Px]
    (real part = dL/dprobe_real, imag part = dL/dprobe_imag).

    ``beamstop`` [Py,Px] (optional): only detector pixels with beamstop >= 1e-5 enter the loss (forward_model.py:128-136).

    ``shifts`` [B,2] (optional): sub-pixel probe position corrections; position b sees every mode Fourier-shifted by
    shifts[b] (forward_model.py:296-311).  A fifth return value dL/dshifts [B,2] is appended and the probe gradient
    is taken through the shift.
    """
    dt = np.dtype(dtype)
    cdt = _cdtype(dt)
    probes = np.asarray(probes)
    if probes.ndim == 2:
        probes = probes[None]
    tiles = tiles.astype(dt, copy=False)
    B, Py, Px, S, _ = tiles.shape
    n_steps = int(np.ceil(S / phys.binning))
    h = phys.h_cast(dt)
    fields, kepts = [], []
    if shifts is not None:
        phases = np.stack([fourier_shift_phase(probes.shape[-2:], sh, dt) for sh in shifts])      # [B,Py,Px]
        spectra = np.fft.fft2(probes.astype(cdt)).astype(cdt)                                      # [M,Py,Px]
    for m, p in enumerate(probes):
        if shifts is not None:
            p = np.fft.ifft2((spectra[m][None] * phases).astype(cdt)).astype(cdt)                  # [B,Py,Px]
        f, k = multislice_forward(tiles, p, phys, dt, keep=True)
        fields.append(f)
        kepts.append(k)
    if len(fields) == 1:
        pred = np.abs(fields[0])
    else:
        pred = np.sqrt(sum((f.real ** 2 + f.imag ** 2) for f in fields))
    meas = np.asarray(meas).astype(dt, copy=False)
    if beamstop is None:
        loss = mismatch_loss(pred, meas, loss_function_type, raw_data_type, poisson_multiplier)
        dldp = _dloss_dpred(pred, meas, loss_function_type, raw_data_type, poisson_multiplier).astype(dt)
    else:
        keep = np.asarray(beamstop) >= 1e-5
        loss = mismatch_loss(pred[:, keep], meas[:, keep], loss_function_type, raw_data_type, poisson_multiplier)
        dldp = np.zeros_like(pred)
        dldp[:, keep] = _dloss_dpred(pred[:, keep], meas[:, keep], loss_function_type, raw_data_type, poisson_multiplier)
    grad_tiles = np.zeros_like(tiles)
    grad_probes = np.zeros(probes.shape, dtype=cdt)
    k1 = dt.type(phys.k1)
    sg = dt.type(phys.sigma)
    for m, (f, kept) in enumerate(zip(fields, kepts)):
        with np.errstate(divide='ignore', invalid='ignore'):
            unit = np.where(pred > 0, f / pred, 0) if len(fields) == 1 else f / pred
        G = _detector_adj((dldp * unit).astype(cdt), phys, dt).astype(cdt)
        for i in range(n_steps - 1, -1, -1):
            d, b, lo, hi = _slice_sums(tiles, i, phys.binning)
            c = _modulator(d, b, phys, dt)
            if phys.unknown_type == 'real_imag':
                zc = G * np.conj(kept[i] / c)          # G_c = G_psi' * conj(psi), psi = psi'/c the pre-modulation field
                gd, gb = zc.real.astype(dt), zc.imag.astype(dt)
            else:
                z = np.conj(G) * kept[i]
                gb = (-k1 * z.real).astype(dt)
                gd = (sg * k1 * z.imag).astype(dt)
            grad_tiles[:, :, :, lo:hi, 0] += gd[..., None]
            grad_tiles[:, :, :, lo:hi, 1] += gb[..., None]
            G = (G * np.conj(c)).astype(cdt)
            if i > 0:
                G = np.fft.ifft2(np.fft.fft2(G) * np.conj(h)).astype(cdt)
        if shifts is None:
            grad_probes[m] = G.sum(axis=0)
        else:
            # p_b = IFFT2(Phi_b * F), F = FFT2(p):  dL/dp = IFFT2(sum_b conj(Phi_b) FFT2(G_b));
            # dL/ds = 2 PI sum_k f_k Im(conj(Ghat_k) Phi_k F_k),  Ghat = FFT2(G_b) / (Py Px)
            Gh = np.fft.fft2(G)
            grad_probes[m] = np.fft.ifft2((np.conj(phases) * Gh).sum(axis=0))
            t = np.imag(np.conj(Gh / (Py * Px)) * phases * spectra[m][None])
            fy = np.fft.fftfreq(Py, 1)[:, None]
            fx = np.fft.fftfreq(Px, 1)[None, :]
            if m == 0:
                grad_shifts = np.zeros((B, 2), dtype=dt)
            grad_shifts[:, 0] += 2 * PI * (t * fy).sum(axis=(1, 2))
            grad_shifts[:, 1] += 2 * PI * (t * fx).sum(axis=(1, 2))
    if shifts is not None:
        return loss, pred, grad_tiles, grad_probes, grad_shifts
    return loss, pred, grad_tiles, grad_probes


def forward_adjoint_object(obj, coords_fp16, probes, pos_batch, meas, phys, dtype='float64', **loss_kw):
    """
    The whole differentiable block of PtychographyModel.get_loss_function
    (forward_model.py:264-353, 389-401) and its gradient w.r.t. ``obj`` [Y,X,Z,2] and the
    probes.  ``coords_fp16=None`` means no rotation (two_d_mode / rotate_out_of_loop).
    """
    dt = np.dtype(dtype)
    obj = obj.astype(dt, copy=False)
    obj_rot = rotate_fwd(obj, coords_fp16, dt) if coords_fp16 is not None else obj
    probes = np.asarray(probes)
    psize = probes.shape[-2:]
    tiles, _ = extract_tiles(obj_rot, pos_batch, psize, phys.unknown_type)
    loss, pred, gt, gp = forward_adjoint_tiles(tiles, probes, meas, phys, dt, **loss_kw)
    g_rot = scatter_tiles_adj(gt, pos_batch, obj.shape)
    g_obj = rotate_adj(g_rot, coords_fp16, dt) if coords_fp16 is not None else g_rot
    return loss, pred, g_obj, gp


# --------------------------------------------------------------------------------------
# R9  regularisers (value and gradient)
# --------------------------------------------------------------------------------------
def l1_value_grad(obj, alpha_d, alpha_b):
    """L1Regularizer.get_value (regularizers.py:30-46), delta_beta branch."""
    V = obj[..., 0].size
    val = 0.
    g = np.zeros_like(obj)
    if alpha_d not in (None, 0):
        val += alpha_d * np.mean(np.abs(obj[..., 0]))
        g[..., 0] = alpha_d * np.sign(obj[..., 0]) / V
    if alpha_b not in (None, 0):
        val += alpha_b * np.mean(np.abs(obj[..., 1]))
        g[..., 1] = alpha_b * np.sign(obj[..., 1]) / V
    return val, g


def tv_value_grad(obj, gamma):
    """TVRegularizer.get_value (regularizers.py:95-110) -> total_variation_3d (util.py:1427-1440):
    sum over the three axes of sum|roll(a,1,ax) - a| / a.size, periodic, per channel."""
    val = 0.
    g = np.zeros_like(obj)
    V = obj[..., 0].size
    for ch in range(2):
        a = obj[..., ch]
        for ax in range(3):
            d = np.roll(a, 1, axis=ax) - a            # d[i] = a[i-1] - a[i]
            val += gamma * np.sum(np.abs(d)) / V
            s = np.sign(d)
            # d/da[i] of |a[i-1]-a[i]| = -s[i];  of |a[i]-a[i+1]| = +s[i+1]
            g[..., ch] += gamma * (np.roll(s, -1, axis=ax) - s) / V
    return val, g


def _tv_core(a):
    """value*V and d/da of sum_axes sum|roll(a,1,ax) - a| for one scalar field."""
    val = 0.
    g = np.zeros_like(a)
    for ax in range(3):
        d = np.roll(a, 1, axis=ax) - a
        val += np.sum(np.abs(d))
        sgn = np.sign(d)
        g += np.roll(sgn, -1, axis=ax) - sgn
    return val, g


def tv_value_grad_ri(obj, gamma):
    """TVRegularizer, real_imag branch (regularizers.py:105-110): gamma * (TV(r^2 + i^2) + TV(arctan2(i, r)))."""
    r, i = obj[..., 0], obj[..., 1]
    V = r.size
    u = r ** 2 + i ** 2
    ph = np.arctan2(i, r)
    vu, gu = _tv_core(u)
    vp, gp = _tv_core(ph)
    g = np.zeros_like(obj)
    g[..., 0] = gamma * (gu * 2 * r - gp * i / u) / V
    g[..., 1] = gamma * (gu * 2 * i + gp * r / u) / V
    return gamma * (vu + vp) / V, g


def l1_value_grad_ri(obj, alpha_d, alpha_b):
    """L1Regularizer, real_imag branch (regularizers.py:38-45): alpha_d*mean| |o| - mean|o| | + alpha_b*mean|arg o|."""
    r, i = obj[..., 0], obj[..., 1]
    V = r.size
    val = 0.
    g = np.zeros_like(obj)
    om = np.sqrt(r ** 2 + i ** 2)
    if alpha_d not in (None, 0):
        dev = om - om.mean()
        val += alpha_d * np.mean(np.abs(dev))
        sg = np.sign(dev)
        gom = alpha_d * (sg - sg.mean()) / V            # d/d om_j of mean_k |om_k - mean(om)|
        g[..., 0] += gom * r / om
        g[..., 1] += gom * i / om
    if alpha_b not in (None, 0):
        ph = np.arctan2(i, r)
        val += alpha_b * np.mean(np.abs(ph))
        gph = alpha_b * np.sign(ph) / V
        g[..., 0] += -gph * i / om ** 2
        g[..., 1] += gph * r / om ** 2
    return val, g


# --------------------------------------------------------------------------------------
# R13-R15  optimisers and constraints
# --------------------------------------------------------------------------------------
def adam_step(x, g, m, v, i_batch, step_size=0.001, b1=0.9, b2=0.999, eps=1e-7):
    """AdamOptimizer.apply_gradient math (optimizers.py:309-318), in x's dtype."""
    dt = x.dtype.type
    m = dt(b1) * m
    m = m + dt(1 - b1) * g
    v = dt(b2) * v
    v = v + dt(1 - b2) * (g ** 2)
    mhat = m / dt(1 - b1 ** (i_batch + 1))
    vhat = v / dt(1 - b2 ** (i_batch + 1))
    d = dt(step_size) * mhat / (np.sqrt(vhat) + dt(eps))
    return (x - d).astype(x.dtype), m.astype(x.dtype), v.astype(x.dtype)


def momentum_step(x, g, v, step_size=0.001, gamma=0.9):
    """MomentumOptimizer.apply_gradient (optimizers.py:376-411): v = gamma*v + step*g; x = x - v."""
    dt = x.dtype.type
    v = dt(gamma) * v + dt(step_size) * g
    return (x - v).astype(x.dtype), v.astype(x.dtype)


def reweighted_l1_weight(obj):
    """ptychography.py:995-1000 (DP branch): max(obj) / (|obj| + 1e-4 * mean(obj)), over both channels jointly."""
    return obj.max() / (np.abs(obj) + 1e-4 * obj.mean())


def reweighted_l1_value_grad(obj, weight, alpha_d, alpha_b):
    """ReweightedL1Regularizer.get_value (regularizers.py:64-71), weights are constants (no_grad)."""
    V = obj[..., 0].size
    val = 0.
    g = np.zeros_like(obj)
    if alpha_d not in (None, 0):
        val += alpha_d * np.mean(weight[..., 0] * np.abs(obj[..., 0]))
        g[..., 0] = alpha_d * weight[..., 0] * np.sign(obj[..., 0]) / V
    if alpha_b not in (None, 0):
        val += alpha_b * np.mean(weight[..., 1] * np.abs(obj[..., 1]))
        g[..., 1] = alpha_b * weight[..., 1] * np.sign(obj[..., 1]) / V
    return val, g


def reweighted_l1_value_grad_ri(obj, weight, alpha_d, alpha_b):
    """ReweightedL1Regularizer.get_value for unknown_type='real_imag' (regularizers.py:73-82): with wm = w_re^2 + w_im^2,
    alpha_d * mean(wm * | |o| - mean|o| |) + alpha_b * mean(wm * |atan2(im, re)|); the weights are constants (no_grad).
    Gradient w.r.t. (re, im) by the chain rule, sign(0) = 0 like torch's abs."""
    r, i = obj[..., 0], obj[..., 1]
    V = r.size
    wm = weight[..., 0] ** 2 + weight[..., 1] ** 2
    val = 0.
    g = np.zeros_like(obj)
    u = r ** 2 + i ** 2
    if alpha_d not in (None, 0):
        om = np.sqrt(u)
        dev = om - om.mean()
        val += alpha_d * np.mean(wm * np.abs(dev))
        s = wm * np.sign(dev)
        gom = alpha_d * (s - s.mean()) / V
        g[..., 0] += gom * r / om
        g[..., 1] += gom * i / om
    if alpha_b not in (None, 0):
        ph = np.arctan2(i, r)
        val += alpha_b * np.mean(wm * np.abs(ph))
        gph = alpha_b * wm * np.sign(ph) / V
        g[..., 0] += -gph * i / u
        g[..., 1] += gph * r / u
    return val, g


def gd_step_size(i_batch, step_size, dynamic_rate=True, first_downrate_iteration=92):
    """GDOptimizer.apply_gradient schedule (optimizers.py:452-460)."""
    if dynamic_rate:
        threshold_iteration = first_downrate_iteration
        i = 1
        while threshold_iteration < i_batch:
            threshold_iteration += first_downrate_iteration * 2 ** i
            i += 1
            step_size /= 2.
    return step_size


def gd_step(x, g, i_batch, step_size=0.001, dynamic_rate=True, first_downrate_iteration=92):
    return (x - x.dtype.type(gd_step_size(i_batch, step_size, dynamic_rate, first_downrate_iteration)) * g).astype(x.dtype)


def apply_constraints(x, non_negativity=False, object_type='normal', mask=None):
    """ptychography.py:1135-1158 (delta_beta) + array_ops.py:239-251."""
    if non_negativity:
        x = np.clip(x, 0, None)
    if object_type == 'absorption_only':
        x = x.copy(); x[..., 0] *= 0
    if object_type == 'phase_only':
        x = x.copy(); x[..., 1] *= 0
    if mask is not None:
        x = x * mask[..., None].astype(x.dtype)
    return x


# --------------------------------------------------------------------------------------
# R16  probe initialisation
# --------------------------------------------------------------------------------------
def generate_gaussian_map(size, mag_max, mag_sigma, phase_max, phase_sigma):
    """adorym/util.py:189-195."""
    py = np.arange(size[0]) - (size[0] - 1.) / 2
    px = np.arange(size[1]) - (size[1] - 1.) / 2
    pxx, pyy = np.meshgrid(px, py)
    map_mag = mag_max * np.exp(-(pxx ** 2 + pyy ** 2) / (2 * mag_sigma ** 2))
    map_phase = phase_max * np.exp(-(pxx ** 2 + pyy ** 2) / (2 * phase_sigma ** 2))
    return map_mag, map_phase


def gaussian_probe(size, mag_sigma, phase_sigma, phase_max):
    """initialize_probe, 'gaussian' branch (util.py:201-206)."""
    mag, ph = generate_gaussian_map(size, 1, mag_sigma, phase_max, phase_sigma)
    return mag * np.cos(ph) + 1j * mag * np.sin(ph)


# --------------------------------------------------------------------------------------
# R17  DP-mode task list
# --------------------------------------------------------------------------------------
def generate_disk(shape, radius):
    """util.py:1484-1492: soft-edged disk, clip(radius - r, 0, 1) with r measured from the array centre."""
    radius = int(radius)
    x = np.arange(shape[1]) - (shape[1] - 1) / 2
    y = np.arange(shape[0]) - (shape[0] - 1) / 2
    xx, yy = np.meshgrid(x, y)
    return np.clip(radius - np.sqrt(xx ** 2 + yy ** 2), 0, 1)


def aperture_defocus_probe(probe_size, aperture_radius, probe_defocus_cm, lmbda_nm, psize_cm, beamstop_radius=0, sign_convention=1):
    """initialize_probe, 'aperture_defocus' branch (util.py:205-222): disk aperture (minus a beamstop disk) Fresnel-
    propagated by the defocus distance with get_kernel / convolve_with_transfer_function, fp64."""
    mag = generate_disk(probe_size, aperture_radius)
    if beamstop_radius > 0:
        mag = mag * (1 - generate_disk(probe_size, beamstop_radius))
    h = get_kernel(probe_defocus_cm * 1e7, lmbda_nm, [psize_cm * 1e7] * 3, probe_size, sign_convention=sign_convention)
    return np.fft.ifft2(np.fft.fft2(mag.astype(np.complex128)) * h)


def probe_ifft_guess(data, raw_data_type='intensity', sign_convention=1):
    """create_probe_initial_guess_ptycho (util.py:300-333, no beamstop, no noise): the mean measured magnitude over all angles and
    positions, taken back from the detector plane -- ifftshift, inverse (forward for sign_convention -1) FFT2, ifftshift."""
    dat = np.asarray(data)
    if raw_data_type == 'intensity':
        dat = np.sqrt(dat)
    wavefront = np.mean(np.abs(dat), axis=(0, 1))
    wavefront = np.fft.ifft2(np.fft.ifftshift(wavefront)) if sign_convention == 1 else np.fft.fft2(np.fft.ifftshift(wavefront))
    return np.fft.ifftshift(wavefront)


def rescale_probe(probe, data_first_angle, raw_data_type='magnitude', normalize_fft=False, sign_convention=1):
    """initialize_probe, rescale_intensity tail (util.py:254-281).  ``probe`` complex [Py,Px] or [M,Py,Px];
    ``data_first_angle`` = exchange/data[0:1].  Reproduces the reference's `len(probe_real) == 3` test (the per-mode
    normalisation only triggers for exactly three modes... or a 3-row probe)."""
    dat = np.asarray(data_first_angle, dtype=np.float64)
    if raw_data_type == 'magnitude':
        dat = dat ** 2
    npix = np.prod(probe.shape[-2:])
    target = np.sum(np.mean(np.abs(dat), axis=(0, 1)))
    if not normalize_fft:
        target = target / npix if sign_convention == 1 else target * npix
    current = np.sum(probe.real ** 2 + probe.imag ** 2)
    if len(probe) == 3:
        current /= probe.shape[0]
    return probe * np.sqrt(target / current)


def split_tasks(arr, split_size):
    """adorym/util.py:1629-1635."""
    res = []
    ind = 0
    while ind < len(arr):
        res.append(arr[ind:min(ind + split_size, len(arr))])
        ind += split_size
    return res


def epoch_task_list(i_epoch, n_theta, n_pos, minibatch_size, n_ranks=1, update_scheme='immediate',
                    randomize_probe_pos=False, two_d_mode=False):
    """
    ptychography.py:791-847 (distribution_mode=None, common_probe_pos=True).  Uses the legacy
    global NumPy RNG exactly like the reference (np.random.seed(i_epoch); shuffle; choice).
    Returns the list of global batches, each an int array [<= n_ranks*mb, 2] of (i_theta, i_pos).
    """
    n_tot_per_batch = minibatch_size * n_ranks
    np.random.seed(i_epoch)
    if not two_d_mode:
        theta_ind_ls = np.arange(n_theta)
        np.random.shuffle(theta_ind_ls)
    else:
        theta_ind_ls = np.array([0])
    ind_list_rand = None
    for i, i_theta in enumerate(theta_ind_ls):
        spots_ls = range(n_pos)
        if randomize_probe_pos:
            spots_ls = np.random.choice(spots_ls, len(spots_ls), replace=False)
        if update_scheme == 'immediate' and n_pos % minibatch_size != 0:
            spots_ls = np.append(spots_ls, np.random.choice(spots_ls[:-n_pos % minibatch_size],
                                                            minibatch_size - (n_pos % minibatch_size),
                                                            replace=False))
        elif update_scheme == 'per angle' and n_pos % n_tot_per_batch != 0:
            spots_ls = np.append(spots_ls, np.random.choice(spots_ls[:-n_pos % n_tot_per_batch],
                                                            n_tot_per_batch - (n_pos % n_tot_per_batch),
                                                            replace=False))
        if i == 0:
            ind_list_rand = np.zeros([len(theta_ind_ls) * len(spots_ls), 2], dtype='int32')
        temp = np.stack([np.array([i_theta] * len(spots_ls)), spots_ls], axis=1)
        ind_list_rand[i * len(spots_ls):(i + 1) * len(spots_ls), :] = temp
    return split_tasks(ind_list_rand, n_tot_per_batch)


def rank_batch(batches, i_batch, rank, minibatch_size, n_ranks=1):
    """ptychography.py:901-908: short last batch is topped up from batch 0; each rank takes its
    contiguous slice; tile indices are sorted."""
    n_tot = minibatch_size * n_ranks
    b = batches[i_batch]
    if len(b) < n_tot:
        b = np.concatenate([b, batches[0][:n_tot - len(b)]])
    i_theta = int(b[rank * minibatch_size, 0])
    ind = np.sort(b[rank * minibatch_size:(rank + 1) * minibatch_size, 1])
    return i_theta, ind


# --------------------------------------------------------------------------------------
# end-to-end DP-mode reconstruction (oracle of reconstruct_ptychography, hot path only)
# --------------------------------------------------------------------------------------
def reconstruct(prj, obj_init, probe, probe_pos, theta_ls, phys, n_epochs=1, minibatch_size=1,
                optimizer='adam', learning_rate=1e-5, alpha_d=None, alpha_b=None, gamma=None,
                update_scheme='immediate', optimizer_batch_number_increment='angle',
                non_negativity=False, object_type='normal', mask=None, dtype='float64',
                n_ranks=1, two_d_mode=False, raw_data_type='magnitude', gd_options=None,
                return_trace=False, optimize_probe=False, probe_learning_rate=1e-5, rank_local_counters=False,
                rotate_out_of_loop=False):
    """
    Control flow of reconstruct_ptychography (ptychography.py:783-1295) restricted to
    distribution_mode=None, shared probe, AD path.  ``n_ranks>1`` emulates `mpirun -n R`: per-rank
    gradients (each with its own regulariser term, forward_model.py:138) are summed (ptychography.py:1113-1114),
    and so are the probe gradients (optimizers.py:1022-1032).

    Every rank of the reference holds its OWN replica of the object, the optimiser moments and the step counter
    i_opt_batch, and decides `is_last_batch_of_this_theta` with the angle of ITS share of the global batch
    (ptychography.py:904-910, 1266-1271).  When a global batch straddles two angles the ranks' counters drift apart, the
    Adam bias corrections differ and the replicas are no longer identical (rank 0's is the one written out).
    ``rank_local_counters=True`` restates exactly that (pinned against golden F14 'immediate'); the default keeps ONE
    counter for all ranks -- rank 0's view of the global batch -- which is what a sharded (single-copy) update can do and
    what adorym_amd does; the two coincide whenever no global batch straddles angles (pinned: F14 'immediate6*',
    'perangle', 'probe6').  Returns rank 0's object.

    ``rotate_out_of_loop`` (ptychography.py:917-947, 1011, 1063-1078; forward_model.py:266-271): the object is rotated to
    the angle OUTSIDE the differentiated block, once per change of angle (so the minibatches of one angle that follow an
    'immediate' update still see the object as it was when the angle began); loss, regularisers and gradient are taken
    w.r.t. that rotated array; the ACCUMULATED gradient buffer is then resampled with the lookup table of -theta
    (gradient.rotate_array(..., overwrite_arr=True): an interpolation, not the transpose of the forward gather) after
    every minibatch -- in 'per angle' mode what was accumulated earlier is therefore resampled again with each further
    minibatch (the reference's own TODO at :1075); restated literally.  Golden F15.
    """
    dt = np.dtype(dtype)
    n_rep = n_ranks if rank_local_counters else 1
    obj0 = np.stack([obj_init[0], obj_init[1]], -1).astype(dt)
    objs = [obj0.copy() for _ in range(n_rep)]
    ms = [np.zeros_like(obj0) for _ in range(n_rep)]
    vs = [np.zeros_like(obj0) for _ in range(n_rep)]
    probe = np.asarray(probe)
    pst0 = np.stack([probe.real, probe.imag], -1).astype(dt)            # [..., 2] like the reference's stacked probe
    psts = [pst0.copy() for _ in range(n_rep)]
    pms = [np.zeros_like(pst0) for _ in range(n_rep)]
    pvs = [np.zeros_like(pst0) for _ in range(n_rep)]
    n_theta = len(theta_ls)
    n_pos = len(probe_pos)
    probe_pos_int = np.round(np.asarray(probe_pos)).astype(int)
    tables = {}
    losses = []
    first_grad = None
    gd_options = gd_options or {'dynamic_rate': True, 'first_downrate_iteration': 20}
    for i_epoch in range(n_epochs):
        batches = epoch_task_list(i_epoch, n_theta, n_pos, minibatch_size, n_ranks, update_scheme,
                                  two_d_mode=two_d_mode)
        n_batch = len(batches)
        i_opt = [0] * n_rep   # starting_epoch * n_batch + starting_batch (ptychography.py:848), no checkpoint
        grad_acc = None
        gp_acc = None
        current_theta = [-1] * n_rep        # ptychography.py:855
        arr_rot = [None] * n_rep
        for i_batch in range(n_batch):
            g_sum = None
            gp_sum = None
            thetas = []
            for rank in range(n_ranks):
                rep = rank if rank_local_counters else 0
                obj = objs[rep]
                pc = psts[rep][..., 0] + 1j * psts[rep][..., 1]
                i_theta, ind = rank_batch(batches, i_batch, rank, minibatch_size, n_ranks)
                thetas.append(i_theta)
                coords = None
                if not two_d_mode:
                    if i_theta not in tables:
                        tables[i_theta] = rotation_coords(obj.shape[:3], theta_ls[i_theta], dt)
                    coords = tables[i_theta]
                pos = probe_pos_int[ind]
                meas = np.abs(prj[i_theta, ind])
                if rotate_out_of_loop and not two_d_mode:
                    if i_theta != current_theta[rep]:
                        arr_rot[rep] = rotate_fwd(obj, coords, dt)
                        current_theta[rep] = i_theta
                    obj, coords = arr_rot[rep], None
                loss, pred, g, gp = forward_adjoint_object(obj, coords, pc, pos, meas, phys, dt,
                                                           raw_data_type=raw_data_type)
                if alpha_d not in (None, 0) or alpha_b not in (None, 0):
                    rv, rg = l1_value_grad(obj, alpha_d, alpha_b)
                    loss += rv; g = g + rg
                if gamma not in (None, 0):
                    rv, rg = tv_value_grad(obj, gamma)
                    loss += rv; g = g + rg
                if rank == 0:
                    loss_rank0 = float(loss)
                g_sum = g if g_sum is None else g_sum + g
                gp = np.stack([gp.real, gp.imag], -1).reshape(pst0.shape).astype(dt)
                gp_sum = gp if gp_sum is None else gp_sum + gp
            if first_grad is None:
                first_grad = g_sum.copy()
            grad_acc = g_sum if grad_acc is None else grad_acc + g_sum
            gp_acc = gp_sum if gp_acc is None else gp_acc + gp_sum
            if rotate_out_of_loop and not two_d_mode:
                if n_ranks != 1:
                    raise NotImplementedError('rotate_out_of_loop is restated for one rank')
                key = ('inv', thetas[0])
                if key not in tables:
                    tables[key] = rotation_coords(obj0.shape[:3], -theta_ls[thetas[0]], dt)
                grad_acc = rotate_fwd(grad_acc, tables[key], dt)        # ptychography.py:1069-1078
            logged = False
            for rep in range(n_rep):
                cur_theta = thetas[rep] if rank_local_counters else int(batches[i_batch][0, 0])
                last_of_theta = i_batch == n_batch - 1 or int(batches[i_batch + 1][0, 0]) != cur_theta
                if not (update_scheme == 'per angle' and not last_of_theta):
                    if optimizer == 'adam':
                        objs[rep], ms[rep], vs[rep] = adam_step(objs[rep], grad_acc.astype(dt), ms[rep], vs[rep], i_opt[rep],
                                                                step_size=learning_rate)
                    else:
                        objs[rep] = gd_step(objs[rep], grad_acc.astype(dt), i_opt[rep], step_size=learning_rate, **gd_options)
                    objs[rep] = apply_constraints(objs[rep], non_negativity, object_type, mask)
                    if optimize_probe:
                        psts[rep], pms[rep], pvs[rep] = adam_step(psts[rep], gp_acc.astype(dt), pms[rep], pvs[rep], i_opt[rep],
                                                                  step_size=probe_learning_rate)
                    # the convergence log is only written when the loop body is not cut short by the
                    # 'per angle' `continue` (ptychography.py:1095-1099 vs :1261)
                    logged = logged or rep == 0
                if optimizer_batch_number_increment == 'angle':
                    if last_of_theta:
                        i_opt[rep] += 1
                else:
                    i_opt[rep] += 1
            if logged:
                losses.append(loss_rank0)
                grad_acc = None
                gp_acc = None
    if return_trace:
        if optimize_probe:
            return objs[0], losses, first_grad, psts[0][..., 0] + 1j * psts[0][..., 1]
        return objs[0], losses, first_grad
    return objs[0]


def reconstruct_2d(prj, obj_init, probes, probe_pos, phys, n_epochs=1, minibatch_size=1, learning_rate=1e-3,
                   gamma=None, alpha_d=None, alpha_b=None, raw_data_type='magnitude', optimize_probe=False,
                   probe_learning_rate=1e-3, optimize_all_probe_pos=False, all_probe_pos_learning_rate=1e-2,
                   dtype='float64', return_trace=False):
    """
    reconstruct_ptychography in two_d_mode with the config-1 feature set (ptychography.py:783-1295 +
    optimizers.py:1000-1049): complex-transmission or delta/beta object [Y,X,1,2] updated by Adam, optional Adam
    on the probe modes and on the per-position sub-pixel corrections (re-centred after every update), regularisers
    of either unknown type.  ``obj_init`` = (channel0, channel1) already in the unknown's representation;
    ``probes`` complex [M,Py,Px]; ``probe_pos`` float [n_pos,2] (non-integer parts become the initial corrections).
    """
    dt = np.dtype(dtype)
    cdt = _cdtype(dt)
    obj = np.stack([obj_init[0], obj_init[1]], -1).astype(dt)
    m, v = np.zeros_like(obj), np.zeros_like(obj)
    probes = np.asarray(probes).astype(cdt)
    pst = np.stack([probes.real, probes.imag], -1).astype(dt)
    pm, pv = np.zeros_like(pst), np.zeros_like(pst)
    probe_pos = np.asarray(probe_pos, dtype=float)
    n_pos = len(probe_pos)
    pos_int = np.round(probe_pos).astype(int)
    corr = np.tile(probe_pos - pos_int, [1, 1, 1]).astype(dt)               # [n_theta=1, n_pos, 2]
    cm, cv = np.zeros_like(corr), np.zeros_like(corr)
    ri = phys.unknown_type == 'real_imag'
    losses, first_grad = [], None
    for i_epoch in range(n_epochs):
        batches = epoch_task_list(i_epoch, 1, n_pos, minibatch_size, 1, 'immediate', two_d_mode=True)
        n_batch = len(batches)
        i_opt_batch = 0
        for i_batch in range(n_batch):
            _, ind = rank_batch(batches, i_batch, 0, minibatch_size, 1)
            use_shift = optimize_all_probe_pos or np.any(corr > 1e-3)
            tiles, _ = extract_tiles(obj, pos_int[ind], probes.shape[-2:], phys.unknown_type)
            pc = (pst[..., 0] + 1j * pst[..., 1]).astype(cdt)
            res = forward_adjoint_tiles(tiles, pc, prj[0, ind], phys, dt, raw_data_type=raw_data_type,
                                        shifts=corr[0, ind] if use_shift else None)
            loss, gt, gp = res[0], res[2], res[3]
            g = scatter_tiles_adj(gt, pos_int[ind], obj.shape)
            if alpha_d not in (None, 0) or alpha_b not in (None, 0):
                rv, rg = (l1_value_grad_ri if ri else l1_value_grad)(obj, alpha_d, alpha_b)
                loss += rv; g = g + rg
            if gamma not in (None, 0):
                rv, rg = (tv_value_grad_ri if ri else tv_value_grad)(obj, gamma)
                loss += rv; g = g + rg
            if first_grad is None:
                first_grad = g.copy()
            obj, m, v = adam_step(obj, g.astype(dt), m, v, i_opt_batch, step_size=learning_rate)
            if optimize_probe:
                pst, pm, pv = adam_step(pst, np.stack([gp.real, gp.imag], -1).astype(dt), pm, pv, i_opt_batch,
                                        step_size=probe_learning_rate)
            if optimize_all_probe_pos:
                gc = np.zeros_like(corr)
                gc[0, ind] = res[4]
                corr, cm, cv = adam_step(corr, gc, cm, cv, i_opt_batch, step_size=all_probe_pos_learning_rate)
                corr = corr - corr.mean(axis=(0, 1))
            losses.append(float(loss))
            if i_batch == n_batch - 1:
                i_opt_batch += 1
    out = dict(obj=obj, probes=pst[..., 0] + 1j * pst[..., 1], pos_corr=corr, losses=losses, first_grad=first_grad)
    return out


# --------------------------------------------------------------------------------------
# f1  multi-distance holography (MultiDistModel, forward_model.py:809-1092), one undivided tile, S = 1
# --------------------------------------------------------------------------------------
def affine_sample(img, theta, want_grad_theta=False, cot=None):
    """w.affine_transform (wrappers.py:1158-1174) = F.affine_grid (align_corners=False) + F.grid_sample (bilinear,
    padding_mode='border', align_corners=False) for one image [H,W] and one 2x3 matrix.  With ``cot`` (cotangent of the
    output) also returns d<cot, out>/dtheta [2,3] as torch's grid_sampler backward does (zero through clamped coordinates)."""
    H, W = img.shape

    def base(n):
        # torch: linspace(-1, 1, n) * (n - 1) / n; the linspace is evaluated from both ends with a FUSED multiply-add
        # (fma(step, i, -1) below the middle, fma(-step, n-1-i, 1) above).  Reproduced to the rounding because at the
        # identity transform the sampling points sit ON pixel centres, where floor() decides which one-sided
        # derivative the backward pass sees.  The fma is emulated in the next wider type.
        dt_ = img.dtype.type
        wide = np.longdouble if img.dtype == np.float64 else np.float64
        step = wide(dt_(2) / dt_(n - 1))
        i = np.arange(n)
        lo = (wide(-1) + step * i.astype(wide)).astype(img.dtype)
        hi = (wide(1) - step * (n - 1 - i).astype(wide)).astype(img.dtype)
        r = np.where(i < n // 2, lo, hi)
        return (r * dt_(n - 1)) / dt_(n)
    X, Y = np.meshgrid(base(W), base(H))
    gx = theta[0, 0] * X + theta[0, 1] * Y + theta[0, 2]
    gy = theta[1, 0] * X + theta[1, 1] * Y + theta[1, 2]
    ix = ((gx + 1) * W - 1) / 2
    iy = ((gy + 1) * H - 1) / 2
    mx = ((ix > 0) & (ix < W - 1)).astype(img.dtype)          # clip_coordinates_set_grad: 0 at and beyond the borders
    my = ((iy > 0) & (iy < H - 1)).astype(img.dtype)
    ix = np.clip(ix, 0, W - 1)
    iy = np.clip(iy, 0, H - 1)
    x0 = np.floor(ix).astype(int); y0 = np.floor(iy).astype(int)
    wx = ix - x0; wy = iy - y0
    x1 = np.minimum(x0 + 1, W - 1); y1 = np.minimum(y0 + 1, H - 1)
    # corners outside the image get weight 0 in torch (within_bounds test): x0+1 == W only happens with wx == 0
    v00, v01, v10, v11 = img[y0, x0], img[y0, x1], img[y1, x0], img[y1, x1]
    out = v00 * (1 - wx) * (1 - wy) + v01 * wx * (1 - wy) + v10 * (1 - wx) * wy + v11 * wx * wy
    if cot is None:
        return out
    dix = ((v01 - v00) * (1 - wy) + (v11 - v10) * wy) * mx * (W / 2) * cot
    diy = ((v10 - v00) * (1 - wx) + (v11 - v01) * wx) * my * (H / 2) * cot
    g = np.array([[np.sum(dix * X), np.sum(dix * Y), np.sum(dix)], [np.sum(diy * X), np.sum(diy * Y), np.sum(diy)]])
    return out, g


def fourier_shift_real(img, shift, dtype='float64', want_derivatives=False):
    """realign_image_fourier (util.py:380-397) of a real image with a zero imaginary part, REAL part of the result -- what
    MultiDistModel keeps of it (forward_model.py:1075-1085: ``this_prj_batch_idist, _ = realign_image_fourier(...)``):
    Re IFFT2( FFT2(img) * exp(-2 PI i (fx * shift[1] + fy * shift[0])) ), fx / fy = np.fft.fftfreq along columns / rows.
    With want_derivatives also d/dshift[0], d/dshift[1]."""
    dt = np.dtype(dtype)
    cdt = _cdtype(dt)
    H, W = img.shape
    fy = np.fft.fftfreq(H, 1)[:, None].astype(dt)
    fx = np.fft.fftfreq(W, 1)[None, :].astype(dt)
    arg = (dt.type(-2 * PI) * (fx * dt.type(shift[1]) + fy * dt.type(shift[0]))).astype(dt)
    S = (np.fft.fft2(np.asarray(img, dtype=dt)).astype(cdt) * (np.cos(arg) + 1j * np.sin(arg)).astype(cdt)).astype(cdt)
    out = np.fft.ifft2(S).real.astype(dt)
    if not want_derivatives:
        return out
    dy = np.fft.ifft2(S * (dt.type(-2 * PI) * 1j * fy)).real.astype(dt)
    dx = np.fft.ifft2(S * (dt.type(-2 * PI) * 1j * fx)).real.astype(dt)
    return out, dy, dx


def holo_forward_adjoint(obj, probe, dists_cm, affine, data, energy_ev, psize_cm, raw_data_type='intensity',
                         sign_convention=1, dtype='float64', shifts=None):
    """Loss and gradients of the multi-distance chain for a real_imag object [N,N,1,2] under a complex probe [N,N]:
    psi = probe*o;  Psi_d = IFFT2(FFT2(psi) * exp(-i sigma PI lambda d (u^2+v^2)))  (propagate.py:84-103, 556-568);
    loss = mean_d,pixels (|Psi_d| - sqrt|affine(data_d, A_d)|)^2.  Returns loss, pred, target, grad_obj [N,N,1,2],
    grad_probe (complex), grad_dists_cm [n_d], grad_affine [n_d,2,3].

    ``shifts`` [n_d, 2] (optimize_all_probe_pos with multi-distance data, forward_model.py:1075-1085): every registered hologram
    is Fourier-shifted by its own (sy, sx), real part kept, before the loss; an eighth return value dL/dshifts [n_d, 2] is
    appended (the affine gradient is then not formed: the reference demos refine one or the other)."""
    dt = np.dtype(dtype)
    cdt = _cdtype(dt)
    N0, N1 = obj.shape[:2]
    lm = 1240. / energy_ev
    u, v = gen_freq_mesh(np.array([psize_cm * 1e7] * 3), [N0, N1])
    uv2 = (u ** 2 + v ** 2).astype(dt)
    o = (obj[:, :, 0, 0] + 1j * obj[:, :, 0, 1]).astype(cdt)
    psi = (probe.astype(cdt) * o).astype(cdt)
    F = np.fft.fft2(psi).astype(cdt)
    nd = len(dists_cm)
    n_tot = nd * N0 * N1
    loss = 0.
    preds, tgts = [], []
    g_dists = np.zeros(nd, dtype=dt)
    g_aff = np.zeros((nd, 2, 3), dtype=dt)
    g_sh = np.zeros((nd, 2), dtype=dt)
    GF = np.zeros_like(F)
    for i in range(nd):
        d_nm = dt.type(dists_cm[i]) * dt.type(1e7)
        arg = (dt.type(-sign_convention * PI * lm) * d_nm * uv2).astype(dt)
        Hd = (np.cos(arg) + 1j * np.sin(arg)).astype(cdt)
        Psi = np.fft.ifft2(F * Hd).astype(cdt)
        pred = np.abs(Psi)
        samp = affine_sample(np.asarray(data[i], dtype=dt), np.asarray(affine[i], dtype=dt))
        if shifts is not None:
            samp, d_sy, d_sx = fourier_shift_real(samp, shifts[i], dt, want_derivatives=True)
        tgt = np.sqrt(np.abs(samp)) if raw_data_type == 'intensity' else np.abs(samp)
        diff = pred - tgt
        loss += np.sum(diff ** 2) / n_tot
        with np.errstate(divide='ignore', invalid='ignore'):
            G = np.where(pred > 0, (2 / n_tot) * diff * Psi / pred, 0).astype(cdt)
            dtgt = -(2 / n_tot) * diff
            cot = dtgt * (np.sign(samp) / (2 * np.sqrt(np.abs(samp))) if raw_data_type == 'intensity' else np.sign(samp))
        cot = np.where(np.isfinite(cot), cot, 0)
        if shifts is not None:
            g_sh[i] = [np.sum(cot * d_sy), np.sum(cot * d_sx)]
        else:
            _, g_aff[i] = affine_sample(np.asarray(data[i], dtype=dt), np.asarray(affine[i], dtype=dt), cot=cot)
        Gh = np.fft.fft2(G) / (N0 * N1)
        dH = (-1j * sign_convention * PI * lm) * uv2 * Hd
        g_dists[i] = np.real(np.sum(np.conj(Gh) * dH * F)) * 1e7
        GF += np.conj(Hd) * Gh
        preds.append(pred); tgts.append(tgt)
    g_psi = np.fft.ifft2(GF) * (N0 * N1)
    g_o = g_psi * np.conj(probe)
    g_obj = np.zeros_like(obj, dtype=dt)
    g_obj[:, :, 0, 0] = g_o.real
    g_obj[:, :, 0, 1] = g_o.imag
    if shifts is not None:
        return loss, np.stack(preds), np.stack(tgts), g_obj, g_psi * np.conj(o), g_dists, g_aff, g_sh
    return loss, np.stack(preds), np.stack(tgts), g_obj, g_psi * np.conj(o), g_dists, g_aff


def reconstruct_multidist(data, obj_init, probe, dists_cm, energy_ev, psize_cm, n_epochs=1, learning_rate=1e-2,
                          optimize_free_prop=False, free_prop_learning_rate=1e-1, optimize_prj_affine=False,
                          prj_affine_learning_rate=1e-3, raw_data_type='intensity', dtype='float64', optimize_all_probe_pos=False,
                          all_probe_pos_learning_rate=1e-2):
    """reconstruct_ptychography for multi-distance data of one undivided tile (two_d_mode, minibatch 1, one minibatch per
    epoch => the Adam step counter is 0 in every epoch, ptychography.py:848): Adam on the object, optionally on the
    distances and the affine matrices (matrix 0 is pinned to the identity after every update, optimizers.py:1062-1075), or on
    one (sy, sx) per distance (``optimize_all_probe_pos``: probe_pos_correction [n_dists, 2] from zero, re-centred after every
    update, optimizers.py:1039-1049)."""
    dt = np.dtype(dtype)
    obj = np.stack([obj_init[0], obj_init[1]], -1).astype(dt)
    m, v = np.zeros_like(obj), np.zeros_like(obj)
    dists = np.asarray(dists_cm, dtype=dt).copy()
    dm, dv = np.zeros_like(dists), np.zeros_like(dists)
    aff = np.tile(np.array([[1., 0, 0], [0, 1., 0]], dtype=dt), [len(dists), 1, 1])
    am, av = np.zeros_like(aff), np.zeros_like(aff)
    losses, first_grad = [], None
    sh = np.zeros((len(dists), 2), dtype=dt)
    sm, sv = np.zeros_like(sh), np.zeros_like(sh)
    trace = []
    for i_epoch in range(n_epochs):
        res = holo_forward_adjoint(obj, probe, dists, aff, data, energy_ev, psize_cm, raw_data_type, 1, dt,
                                   shifts=sh if optimize_all_probe_pos else None)
        loss, g, gd, ga = res[0], res[3], res[5], res[6]
        if first_grad is None:
            first_grad = g.copy()
        obj, m, v = adam_step(obj, g, m, v, 0, step_size=learning_rate)
        if optimize_all_probe_pos:
            sh, sm, sv = adam_step(sh, res[7], sm, sv, 0, step_size=all_probe_pos_learning_rate)
            sh = sh - sh.mean(axis=0)
            trace.append(sh.copy())
        if optimize_free_prop:
            dists, dm, dv = adam_step(dists, gd, dm, dv, 0, step_size=free_prop_learning_rate)
        if optimize_prj_affine:
            aff, am, av = adam_step(aff, ga, am, av, 0, step_size=prj_affine_learning_rate)
            aff[0] = np.array([[1., 0, 0], [0, 1., 0]], dtype=dt)
        losses.append(float(loss))
    return dict(obj=obj, dists=dists, affine=aff, losses=losses, first_grad=first_grad, shifts=sh, shift_trace=trace)


# --------------------------------------------------------------------------------------
# f1  multi-distance holography divided into sub-tiles with a safe zone (MultiDistModel, forward_model.py:884-1034, n_blocks > 1)
# --------------------------------------------------------------------------------------
def multidist_subprobes(probe, pos_batch, sub_size, szw, n_dp_batch=20):
    """forward_model.py:916-925, 944-994: the full-field probe (same size as the object, ptychography.py:312-314) is padded with
    1 + 0i so that every (sub + 2 szw) window of the batch lies inside it (calculate_pad_len on the windows of THIS batch), and
    position j sees the window at pos_j - szw.  Line :1005 hands multislice_propagate_batch ``subprobe_imag_ls_ls[k][i_mode, :, :]``
    (no leading ':'): with one probe mode that is the imaginary window of the FIRST position of the n_dp_batch chunk, broadcast
    over the chunk -- restated as it is.  Returns complex [B, Ty, Tx]."""
    probe = np.asarray(probe)
    pos = np.round(np.asarray(pos_batch)).astype(int).reshape(-1, 2)
    T = (sub_size[0] + 2 * szw, sub_size[1] + 2 * szw)
    if szw > 0:
        pad = calculate_pad_len(probe.shape[-2:], pos - szw, T)
        pr = np.pad(probe.real, [tuple(pad[0]), tuple(pad[1])], mode='constant', constant_values=1)
        pi = np.pad(probe.imag, [tuple(pad[0]), tuple(pad[1])], mode='constant', constant_values=0)
    else:
        pad = np.zeros((2, 2), int)
        pr, pi = probe.real, probe.imag
    out = np.zeros((len(pos),) + T, dtype=np.result_type(probe.dtype, np.complex64))
    for c0 in range(0, len(pos), n_dp_batch):
        chunk = range(c0, min(c0 + n_dp_batch, len(pos)))
        for n_, j in enumerate(chunk):
            y, x = pos[j, 0] + pad[0, 0] - szw, pos[j, 1] + pad[1, 0] - szw
            if n_ == 0:
                im0 = pi[y:y + T[0], x:x + T[1]]
            out[j] = pr[y:y + T[0], x:x + T[1]] + 1j * im0
    return out


def multidist_tiles_forward_adjoint(obj, probe, pos_batch, sub_size, szw, dists_cm, meas, energy_ev, psize_cm,
                                    unknown_type='real_imag', raw_data_type='magnitude', n_dp_batch=20, dtype='float64',
                                    sign_convention=1, scale_ri_by_k=True):
    """MultiDistModel.predict + get_loss_function for data divided into sub-tiles (forward_model.py:884-1092, n_blocks > 1,
    optimize_free_prop / optimize_prj_affine off) and the gradient w.r.t. the object that ``torch.autograd.grad`` returns.
    Tile j of the batch: the object window [pos_j - szw, pos_j + sub + szw) of the object padded like pad_object (:912), lit by
    its probe window, modulated slice by slice and Fresnel-propagated to every distance (fresnel_propagate, propagate.py:282-288);
    the safe zone is cut off the magnitudes (:1027-1029) and the mean squared mismatch is taken over ALL distances and tiles
    (:1054-1058, data order [i_dist * n_blocks + tile]).  ``meas`` [n_dists * B, sub_y, sub_x], distance-major.
    Returns loss, pred [n_dists * B, sub_y, sub_x], grad_obj [Y, X, S, 2]."""
    dt = np.dtype(dtype)
    cdt = _cdtype(dt)
    obj = np.asarray(obj).astype(dt, copy=False)
    pos = np.round(np.asarray(pos_batch)).astype(int).reshape(-1, 2)
    B = len(pos)
    T = (sub_size[0] + 2 * szw, sub_size[1] + 2 * szw)
    tiles, _ = extract_tiles(obj, pos - szw, T, unknown_type)
    probes_b = multidist_subprobes(probe, pos, sub_size, szw, n_dp_batch).astype(cdt)
    S = tiles.shape[3]
    fields, kepts, physs = [], [], []
    for d in dists_cm:
        phys = Physics(T, energy_ev, psize_cm, free_prop_cm=float(d), sign_convention=sign_convention, scale_ri_by_k=scale_ri_by_k,
                       unknown_type=unknown_type)
        f, k = multislice_forward(tiles, probes_b, phys, dt, keep=True)
        fields.append(f); kepts.append(k); physs.append(phys)
    cy, cx = slice(szw, szw + sub_size[0]), slice(szw, szw + sub_size[1])
    pred = np.concatenate([np.abs(f)[:, cy, cx] for f in fields], 0).astype(dt)
    meas = np.asarray(meas).astype(dt, copy=False)
    loss = mismatch_loss(pred, meas, 'lsq', raw_data_type, 1.)
    dldp = _dloss_dpred(pred, meas, 'lsq', raw_data_type, 1.).astype(dt)
    grad_tiles = np.zeros_like(tiles)
    for i, (f, kept, phys) in enumerate(zip(fields, kepts, physs)):
        full = np.zeros((B,) + T, dtype=dt)
        full[:, cy, cx] = dldp[i * B:(i + 1) * B]
        mag = np.abs(f)
        with np.errstate(divide='ignore', invalid='ignore'):
            unit = np.where(mag > 0, f / mag, 0)
        G = _detector_adj((full * unit).astype(cdt), phys, dt).astype(cdt)
        h = phys.h_cast(dt)
        k1, sg = dt.type(phys.k1), dt.type(phys.sigma)
        for s in range(S - 1, -1, -1):               # the sweep of forward_adjoint_tiles, binning 1
            d_, b_, lo, hi = _slice_sums(tiles, s, 1)
            c = _modulator(d_, b_, phys, dt)
            if unknown_type == 'real_imag':
                zc = G * np.conj(kept[s] / c)
                gd, gb = zc.real.astype(dt), zc.imag.astype(dt)
            else:
                z = np.conj(G) * kept[s]
                gb = (-k1 * z.real).astype(dt)
                gd = (sg * k1 * z.imag).astype(dt)
            grad_tiles[:, :, :, lo:hi, 0] += gd[..., None]
            grad_tiles[:, :, :, lo:hi, 1] += gb[..., None]
            G = (G * np.conj(c)).astype(cdt)
            if s > 0:
                G = np.fft.ifft2(np.fft.fft2(G) * np.conj(h)).astype(cdt)
    return loss, pred, scatter_tiles_adj(grad_tiles, pos - szw, obj.shape)


def reconstruct_multidist_tiles(prj, obj_init, probe, probe_pos, sub_size, szw, dists_cm, energy_ev, psize_cm, n_epochs=1,
                                minibatch_size=4, learning_rate=1e-2, unknown_type='real_imag', raw_data_type='magnitude',
                                n_dp_batch=20, dtype='float64', n_ranks=1):
    """reconstruct_ptychography on multi-distance data divided into n_blocks sub-tiles (two_d_mode: one angle; the tiles are the
    'probe positions' of the task list, ptychography.py:791-912): Adam on the object.  ``prj`` [1, n_dists * n_blocks, sub, sub].
    ``n_ranks`` > 1: `mpirun -n R` -- every rank evaluates its slice of the global batch (:905-909), the gradients are summed
    (:1113-1114), one Adam step; ``losses`` are rank 0's, ``losses_by_rank`` everyone's."""
    dt = np.dtype(dtype)
    obj = np.stack([obj_init[0], obj_init[1]], -1).astype(dt)
    m, v = np.zeros_like(obj), np.zeros_like(obj)
    pos = np.asarray(probe_pos, dtype=float)
    n_blocks = len(pos)
    nd = len(dists_cm)
    losses, first_grad, first_pred = [[] for _ in range(n_ranks)], None, None
    inds = [[] for _ in range(n_ranks)]
    for i_epoch in range(n_epochs):
        batches = epoch_task_list(i_epoch, 1, n_blocks, minibatch_size, n_ranks, 'immediate', two_d_mode=True)
        i_opt_batch = 0
        for i_batch in range(len(batches)):
            g = None
            for r in range(n_ranks):
                _, ind = rank_batch(batches, i_batch, r, minibatch_size, n_ranks)
                full = np.concatenate([ind + i * n_blocks for i in range(nd)])          # forward_model.py:1051-1054
                loss, pred, gr = multidist_tiles_forward_adjoint(obj, probe, pos[ind], sub_size, szw, dists_cm, prj[0, full], energy_ev,
                                                                 psize_cm, unknown_type, raw_data_type, n_dp_batch, dt)
                g = gr if g is None else g + gr
                losses[r].append(float(loss))
                inds[r].append(ind)
                if first_pred is None:
                    first_pred = pred.copy()
            if first_grad is None:
                first_grad = g.copy()
            obj, m, v = adam_step(obj, g.astype(dt), m, v, i_opt_batch, step_size=learning_rate)
            if i_batch == len(batches) - 1:
                i_opt_batch += 1
    return dict(obj=obj, losses=losses[0], losses_by_rank=losses, batches_by_rank=inds, first_grad=first_grad, first_pred=first_pred)
