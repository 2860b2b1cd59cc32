"""Synthetic workloads in the shapes of BASELINE.json's configs (no reference data files exist in
the tree).  Host-side NumPy generators only."""
import numpy as np

from .util import initialize_probe


def c3_config():
    """Config 3: 256^3 multislice ptychotomography (demos/multislice_ptycho_256_theta.py:52-93 of the
    reference; BASELINE.json fixes minibatch 32, binning 1)."""
    ys = np.arange(23) * 12 - 36
    return dict(
        name='C3 multislice ptychotomography 256^3 (cone_256_foam_ptycho shape)',
        obj_size=(256, 256, 256), probe_size=(72, 72),
        probe_pos=np.array([(y, x) for y in ys for x in ys], dtype=np.int64),
        energy_ev=5000., psize_cm=1.e-7, free_prop_cm='inf', binning=1, minibatch_size=32,
        n_theta=500, theta_st=0., theta_end=2 * np.pi,
        alpha_d=1e-9 * 1.7e7, alpha_b=1e-10 * 1.7e7, gamma=1e-9 * 1.7e7, learning_rate=5e-5,
        probe=dict(probe_type='gaussian', probe_mag_sigma=6, probe_phase_sigma=6, probe_phase_max=0.5))


def c2_config():
    """Config 2: 64^3 full-field multislice tomography (tests/test_multislice_tomography_64.py:20-65)."""
    return dict(
        name='C2 multislice tomography 64^3 (adhesin shape)',
        obj_size=(64, 64, 64), probe_size=(64, 64), probe_pos=np.array([(0, 0)], dtype=np.int64),
        energy_ev=800., psize_cm=0.67e-7, free_prop_cm=0, binning=1, minibatch_size=1,
        n_theta=50, theta_st=0., theta_end=2 * np.pi,
        alpha_d=1.e-9 * 64 ** 3, alpha_b=1.e-10 * 64 ** 3, gamma=0., learning_rate=1e-7,
        probe=dict(probe_type='plane'))


def probe_array(cfg):
    pr, pi = initialize_probe(cfg['probe_size'], **cfg['probe'])
    return np.stack([pr, pi], -1).astype(np.float32)


def foam_object(shape, seed=0, delta_max=3e-4, beta_ratio=0.05, n_bubbles=120):
    """A cone filled with spherical voids ("foam"): delta in [0, delta_max], beta = beta_ratio*delta."""
    Y, X, Z = shape
    r = np.random.default_rng(seed)
    y, x, z = np.meshgrid(np.arange(Y), np.arange(X), np.arange(Z), indexing='ij', sparse=True)
    rad = 0.42 * X * (1.0 - 0.7 * y / max(Y - 1, 1))
    body = ((x - X / 2) ** 2 + (z - Z / 2) ** 2) <= rad ** 2
    vol = body.astype(np.float32)
    for _ in range(n_bubbles):
        cy, cx, cz = r.uniform(0, Y), r.uniform(0.2 * X, 0.8 * X), r.uniform(0.2 * Z, 0.8 * Z)
        rr = r.uniform(0.02, 0.07) * X
        y0, y1 = int(max(0, cy - rr)), int(min(Y, cy + rr + 1))
        sl = (slice(y0, y1),)
        d2 = (y[sl] - cy) ** 2 + (x - cx) ** 2 + (z - cz) ** 2
        vol[y0:y1][d2 <= rr ** 2] = 0.0
    delta = (delta_max * vol).astype(np.float32)
    return np.stack([delta, beta_ratio * delta], -1)


def random_guess(shape, seed=0, means_sigmas=(8.7e-7, 5.1e-8, 1e-7, 1e-8)):
    """The reference's default initial guess (adorym/util.py:83-86)."""
    r = np.random.default_rng(seed)
    d = r.normal(means_sigmas[0], means_sigmas[2], shape).astype(np.float32)
    b = r.normal(means_sigmas[1], means_sigmas[3], shape).astype(np.float32)
    return np.stack([d, b], -1)
