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


def initialize_probe(probe_size, probe_type, pupil_function=None, probe_initial=None, **kwargs):
    """adorym/util.py:198-283, the branches that need no data file: 'gaussian', 'plane',
    'supplied'/'fixed'.  Returns (probe_real, probe_imag) as float64 arrays."""
    if probe_type == 'gaussian':
        mag, phase = generate_gaussian_map(probe_size, 1, kwargs['probe_mag_sigma'], kwargs['probe_phase_max'],
                                           kwargs['probe_phase_sigma'])
        probe_real, probe_imag = mag_phase_to_real_imag(mag, phase)
    elif probe_type in ('supplied', 'fixed'):
        probe_real, probe_imag = mag_phase_to_real_imag(np.asarray(probe_initial[0]), np.asarray(probe_initial[1]))
    elif probe_type == 'plane':
        probe_real = np.ones(probe_size)
        probe_imag = np.zeros(probe_size)
    elif probe_type in ('aperture_defocus', 'ifft'):
        raise NotImplementedError("probe_type '%s' is outside the accelerated path (SURVEY section 8 f2)" % probe_type)
    else:
        raise ValueError("Invalid wavefront type. Choose from 'plane', 'fixed', 'supplied'.")
    if pupil_function is not None:
        probe_real = probe_real * pupil_function
        probe_imag = probe_imag * pupil_function
    return probe_real, probe_imag
