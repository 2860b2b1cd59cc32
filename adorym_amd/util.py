"""Host-side setup helpers of the hot path (run once per reconstruction / per angle, NumPy only).
Each function names the reference code whose behaviour it reproduces."""
import numpy as np

from .constants import PI  # noqa: F401


def rotation_lookup(array_size, theta, dtype='float32'):
    """
    fp16 source-coordinate table ``[X*Z, 2]`` for rotating about axis 0 by ``theta``:
    reference adorym/util.py:446-477 (coordinate stack, 2x2 rotation) and :492-516
    (save_rotation_lookup: float32 torch arithmetic, stored as float16).  The reference subtracts
    the *other* axis' centre for each coordinate (util.py:459-460); kept, it only matters for
    non-cubic objects.  Arithmetic is float32 with one rounding per multiply and per add so the
    table is bit-identical to the reference's.
    """
    dt = np.dtype(dtype).type
    _, X, Z = [int(v) for v in array_size]
    xc = (np.repeat(np.arange(X), Z).astype(np.float64) - (Z - 1) / 2).astype(dt)
    zc = (np.tile(np.arange(Z), X).astype(np.float64) - (X - 1) / 2).astype(dt)
    th = dt(theta)
    c, s = np.cos(th, dtype=dt), np.sin(th, dtype=dt)
    x_old = ((c * xc).astype(dt) + ((-s) * zc).astype(dt) + dt((X - 1) / 2)).astype(dt)
    z_old = ((s * xc).astype(dt) + (c * zc).astype(dt) + dt((Z - 1) / 2)).astype(dt)
    return np.stack([x_old, z_old], axis=1).astype(np.float16)


def calculate_pad_len(this_obj_size, probe_pos, probe_size, unknown_type='delta_beta'):
    """adorym/util.py:1374-1406: padding so that every tile of ``probe_pos`` fits."""
    probe_pos = np.asarray(probe_pos)
    pad_arr = np.array([[0, 0], [0, 0]])
    for ax in (0, 1):
        lo = min(probe_pos[:, ax])
        hi = max(probe_pos[:, ax])
        if lo < 0:
            pad_arr[ax, 0] = -int(lo)
        if hi + probe_size[ax] > this_obj_size[ax]:
            pad_arr[ax, 1] = int(hi) + probe_size[ax] - this_obj_size[ax]
    return pad_arr


def split_tasks(arr, split_size):
    """adorym/util.py:1629-1635."""
    res = []
    ind = 0
    while ind < len(arr):
        res.append(arr[ind:min(ind + split_size, len(arr))])
        ind += split_size
    return res


def generate_gaussian_map(size, mag_max, mag_sigma, phase_max, phase_sigma):
    """adorym/util.py:189-195."""
    py = np.arange(size[0]) - (size[0] - 1.) / 2
    px = np.arange(size[1]) - (size[1] - 1.) / 2
    pxx, pyy = np.meshgrid(px, py)
    map_mag = mag_max * np.exp(-(pxx ** 2 + pyy ** 2) / (2 * mag_sigma ** 2))
    map_phase = phase_max * np.exp(-(pxx ** 2 + pyy ** 2) / (2 * phase_sigma ** 2))
    return map_mag, map_phase


def mag_phase_to_real_imag(mag, phase):
    """adorym/util.py:1596-1598."""
    a = mag * np.exp(1j * phase)
    return a.real, a.imag


def generate_disk(shape, radius):
    """adorym/util.py:1484-1492: soft-edged disk clip(radius - r, 0, 1) about the array centre."""
    radius = int(radius)
    x = np.arange(shape[1]) - (shape[1] - 1) / 2
    y = np.arange(shape[0]) - (shape[0] - 1) / 2
    xx, yy = np.meshgrid(x, y)
    return np.clip(radius - np.sqrt(xx ** 2 + yy ** 2), 0, 1)


def _fresnel_propagate_host(probe_real, probe_imag, dist_nm, lmbda_nm, voxel_nm, sign_convention=1):
    """adorym/propagate.py:537-553 on the host in float64 (the reference runs this one-off initialisation through its NumPy
    backend as well, override_backend='autograd')."""
    from .propagate import get_kernel
    h = get_kernel(dist_nm, lmbda_nm, voxel_nm, probe_real.shape[-2:], sign_convention=sign_convention)
    out = np.fft.ifft2(np.fft.fft2(probe_real + 1j * probe_imag) * h)
    return out.real, out.imag


def initialize_probe(probe_size, probe_type, pupil_function=None, probe_initial=None, rescale_intensity=False,
                     extra_defocus_cm=None, sign_convention=1, data_first_angle=None, **kwargs):
    """adorym/util.py:198-283: 'gaussian', 'aperture_defocus', 'plane', 'supplied'/'fixed' (+ pupil, extra defocus,
    intensity rescaling).  ``data_first_angle`` = exchange/data[0:1] (the reference re-opens the HDF5 file for it).
    Returns (probe_real, probe_imag) as float64 arrays."""
    if probe_type == 'gaussian':
        mag, phase = generate_gaussian_map(probe_size, 1, kwargs['probe_mag_sigma'], kwargs['probe_phase_max'],
                                           kwargs['probe_phase_sigma'])
        probe_real, probe_imag = mag_phase_to_real_imag(mag, phase)
    elif probe_type == 'aperture_defocus':
        mag = generate_disk(probe_size, kwargs['aperture_radius'])
        beamstop_radius = kwargs.get('beamstop_radius', 0)
        if beamstop_radius > 0:
            mag = mag * (1 - generate_disk(probe_size, beamstop_radius))
        probe_real, probe_imag = _fresnel_propagate_host(mag, np.zeros_like(mag), kwargs['probe_defocus_cm'] * 1e7, kwargs['lmbda_nm'],
                                                         [kwargs['psize_cm'] * 1e7] * 3, sign_convention)
    elif probe_type in ('supplied', 'fixed'):
        probe_real, probe_imag = mag_phase_to_real_imag(np.asarray(probe_initial[0]), np.asarray(probe_initial[1]))
    elif probe_type == 'plane':
        probe_real = np.ones(probe_size)
        probe_imag = np.zeros(probe_size)
    elif probe_type == 'ifft':
        # util.py:225-236 -> create_probe_initial_guess_ptycho (:300-333): the probe estimated from ALL measured data.  (Its
        # beamstop branch is never reached from the driver: `beamstop` is a named argument of reconstruct_ptychography and not
        # part of the **kwargs that adorym/ptychography.py:609-618 hands to initialize_probe.)
        if kwargs.get('data_all') is None:
            raise ValueError("probe_type 'ifft' needs the measured data (data_all)")
        guess = create_probe_initial_guess_ptycho(kwargs['data_all'], raw_data_type=kwargs.get('raw_data_type', 'intensity'),
                                                  sign_convention=sign_convention)
        probe_real, probe_imag = guess.real, guess.imag
    else:
        raise ValueError("Invalid wavefront type. Choose from 'plane', 'fixed', 'supplied'.")
    if pupil_function is not None:
        probe_real = probe_real * pupil_function
        probe_imag = probe_imag * pupil_function
    if extra_defocus_cm is not None:
        probe_real, probe_imag = _fresnel_propagate_host(probe_real, probe_imag, extra_defocus_cm * 1e7, kwargs['lmbda_nm'],
                                                         [kwargs['psize_cm'] * 1e7] * 3, sign_convention)
    if rescale_intensity:
        # util.py:254-281 (including its `len(probe_real) == 3` test for the per-mode normalisation)
        dat = np.asarray(data_first_angle, dtype=np.float64)
        if kwargs['raw_data_type'] == 'magnitude':
            dat = dat ** 2
        npix = np.prod(np.shape(probe_real)[-2:])
        intensity_target = np.sum(np.mean(np.abs(dat), axis=(0, 1)))
        if not kwargs['normalize_fft']:
            intensity_target = intensity_target / npix if sign_convention == 1 else intensity_target * npix
        intensity_current = np.sum(probe_real ** 2 + probe_imag ** 2)
        if len(probe_real) == 3:
            intensity_current /= probe_real.shape[0]
        s_ = np.sqrt(intensity_target / intensity_current)
        probe_real = probe_real * s_
        probe_imag = probe_imag * s_
    return probe_real, probe_imag


def create_probe_initial_guess_ptycho(data, raw_data_type='intensity', sign_convention=1):
    """adorym/util.py:300-333 without beamstop and noise: mean over (angle, position) of the measured magnitudes, back from the
    detector plane.  ``data``: the whole exchange/data array-like [n_theta, n_pos, Py, Px] (read in one piece, as the reference does)."""
    dat = np.asarray(data[...] if hasattr(data, 'shape') and not isinstance(data, np.ndarray) else data)
    if raw_data_type == 'intensity':
        dat = np.sqrt(dat)
    wavefront = np.mean(np.abs(dat), axis=(0, 1))
    if sign_convention == 1:
        wavefront = np.fft.ifft2(np.fft.ifftshift(wavefront))
    else:
        wavefront = np.fft.fft2(np.fft.ifftshift(wavefront))
    return np.fft.ifftshift(wavefront)


def build_rotation_adjoint_csr(coords_fp16, obj_size, Yp, Xp, pad_x0, staged=False):
    """
    Transpose of the bilinear sampling operator of one angle, as CSR over object-plane voxels t = x*Z + z
    (for adm_rotate_adj_csr).  The coordinate pipeline is w.grid_sample's (adorym/wrappers.py:1137-1141) followed by
    torch's grid_sampler (border padding, align_corners=False): fp16 table -> float64 normalise -> float32
    un-normalise -> clamp -> floor / weights, all in float32 like the forward kernel.
    Returns (ptr int32 [X*Z+1], src int32 [nnz], w float32 [nnz]); src = z'*(Yp*Xp) + pad_x0 + x'.
    """
    _, X, Z = [int(v) for v in obj_size]
    f32 = np.float32
    c = np.asarray(coords_fp16).astype(np.float64)
    gz = (-1 + 2. * c[:, 1] / X + 1. / X).astype(f32)       # (sic) divided by arr_shape[0] == X
    gx = (-1 + 2. * c[:, 0] / Z + 1. / Z).astype(f32)
    iz = ((gz + f32(1)) * f32(Z) - f32(1)) / f32(2)
    ix = ((gx + f32(1)) * f32(X) - f32(1)) / f32(2)
    iz = np.minimum(f32(Z - 1), np.maximum(iz, f32(0)))
    ix = np.minimum(f32(X - 1), np.maximum(ix, f32(0)))
    fz, fx = np.floor(iz), np.floor(ix)
    tz, tx = (iz - fz).astype(f32), (ix - fx).astype(f32)
    z0, x0 = fz.astype(np.int64), fx.astype(np.int64)
    vz, vx = (z0 + 1 <= Z - 1), (x0 + 1 <= X - 1)
    one = f32(1)
    p = np.arange(X * Z, dtype=np.int64)
    src_all = (p % Z) * (Yp * Xp) + pad_x0 + (p // Z)
    tg, sr, ww = [], [], []
    for (dx, dz, w, valid) in ((0, 0, (one - tx) * (one - tz), None), (0, 1, (one - tx) * tz, vz),
                               (1, 0, tx * (one - tz), vx), (1, 1, tx * tz, vx & vz)):
        m = (w != 0) if valid is None else (valid & (w != 0))
        tg.append(((x0 + dx) * Z + (z0 + dz))[m])
        sr.append(src_all[m])
        ww.append(w[m])
    tg, sr, ww = np.concatenate(tg), np.concatenate(sr), np.concatenate(ww)
    order = np.lexsort((sr, tg))                      # by target, then by source: fixed summation order
    tg, sr, ww = tg[order], sr[order], ww[order]
    ptr = np.zeros(X * Z + 1, dtype=np.int64)
    np.cumsum(np.bincount(tg, minlength=X * Z), out=ptr[1:])
    if not staged:
        return ptr.astype(np.int32), sr.astype(np.int32), ww.astype(np.float32)
    # LDS-staged form (adm_rotate_adj_staged): per 16 x 16 patch of targets the bounding box of its sources
    nbx, nbz = (X + 15) // 16, (Z + 15) // 16
    blk = ((tg % Z) // 16) * nbx + (tg // Z) // 16
    zs, xs = sr // (Yp * Xp), sr % (Yp * Xp) - pad_x0
    big = np.iinfo(np.int64).max
    x0 = np.full(nbx * nbz, big); z0 = np.full(nbx * nbz, big)
    x1 = np.full(nbx * nbz, -1); z1 = np.full(nbx * nbz, -1)
    np.minimum.at(x0, blk, xs); np.minimum.at(z0, blk, zs)
    np.maximum.at(x1, blk, xs); np.maximum.at(z1, blk, zs)
    empty = x1 < 0
    x0[empty] = z0[empty] = 0
    bw = np.where(empty, 1, x1 - x0 + 1); bh = np.where(empty, 1, z1 - z0 + 1)
    too_big = bw * bh > 4096          # (rim patches, onto which border clamping folds a corner of the rotated frame, reach ~3000 at 256^3)
    bw[too_big] = 0
    lsrc = np.where(too_big[blk], 0, (zs - z0[blk]) * bw[blk] + (xs - x0[blk]))
    boxes = np.stack([x0, z0, bw, bh], -1).astype(np.int32)
    return ptr.astype(np.int32), sr.astype(np.int32), lsrc.astype(np.uint16), ww.astype(np.float32), boxes


def epoch_task_list(i_epoch, n_theta, n_pos, minibatch_size, n_ranks=1, update_scheme='immediate', randomize_probe_pos=False,
                    fixed_theta=None):
    """The (i_theta, i_pos) task list of one epoch, split into global batches of n_ranks * minibatch_size pairs
    (adorym/ptychography.py:791-847, split_tasks util.py:1629-1635).  Same legacy-RNG draw order as the reference
    (np.random.seed(i_epoch); shuffle of the angle indices; the permutation that pads the positions of an angle to a
    multiple of the batch), so the batches are the reference's bit for bit.  ``fixed_theta``: two_d_mode's single angle."""
    n_tot_per_batch = minibatch_size * n_ranks
    np.random.seed(i_epoch)
    if fixed_theta is None:
        theta_ind_ls = np.arange(n_theta)
        np.random.shuffle(theta_ind_ls)
    else:
        theta_ind_ls = np.array([fixed_theta])
    ind_list_rand = None
    for i, i_theta in enumerate(theta_ind_ls):
        spots_ls = range(n_pos)
        if randomize_probe_pos:
            spots_ls = np.random.choice(spots_ls, len(spots_ls), replace=False)
        if update_scheme == 'immediate' and n_pos % minibatch_size != 0:
            spots_ls = np.append(spots_ls, np.random.choice(spots_ls[:-n_pos % minibatch_size],
                                                            minibatch_size - (n_pos % minibatch_size), replace=False))
        elif update_scheme == 'per angle' and n_pos % n_tot_per_batch != 0:
            spots_ls = np.append(spots_ls, np.random.choice(spots_ls[:-n_pos % n_tot_per_batch],
                                                            n_tot_per_batch - (n_pos % n_tot_per_batch), replace=False))
        if i == 0:
            ind_list_rand = np.zeros([len(theta_ind_ls) * len(spots_ls), 2], dtype='int32')
        temp = np.stack([np.array([i_theta] * len(spots_ls)), spots_ls], axis=1)
        ind_list_rand[i * len(spots_ls):(i + 1) * len(spots_ls), :] = temp
    return split_tasks(ind_list_rand, n_tot_per_batch)


def rank_batch(ind_list_rand, i_batch, rank, minibatch_size, n_ranks=1):
    """What rank ``rank`` processes in global batch ``i_batch`` (adorym/ptychography.py:897-908): a short last batch is
    topped up from the first one (in place, like the reference), the rank takes pairs [rank*mb, (rank+1)*mb), all of one
    angle, position indices sorted.  Returns (i_theta, sorted position indices)."""
    n_tot_per_batch = minibatch_size * n_ranks
    if len(ind_list_rand[i_batch]) < n_tot_per_batch:
        n_supp = n_tot_per_batch - len(ind_list_rand[i_batch])
        ind_list_rand[i_batch] = np.concatenate([ind_list_rand[i_batch], ind_list_rand[0][:n_supp]])
    b = ind_list_rand[i_batch]
    i_theta = int(b[rank * minibatch_size, 0])
    return i_theta, np.sort(b[rank * minibatch_size:(rank + 1) * minibatch_size, 1])
