"""The drop-in boundary against the reference's own demo scripts (demos/*.py): every demo's keyword surface -- the flags, plugin
objects and output switches it passes to reconstruct_ptychography, including the ones that are plain pass-through for this
build ('center', 'xpu', 'n_batch_per_update', 'full_intermediate', ...) -- on small synthetic data, through the product on the GPU.
Numerical parity of each feature set is pinned elsewhere (goldens F6, F11, F12, F14-F20); what is asserted here is that a user
who switches the import runs every demo without an exception, gets finite numbers and the reference's output files."""
import os

import numpy as np
import pytest

import cases
from oracle import adorym_oracle as O      # checker only (synthesises the measured data)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def A():
    import adorym_amd
    return adorym_amd


def _ptycho_data(obj, pos, probe, energy, psize, free_prop_cm='inf', raw='intensity', theta=None):
    """|predicted field| (or its square) of the oracle for a 2-D / 3-D object at the given positions."""
    P = probe.shape[-2:]
    phys = O.Physics(P, energy, psize, free_prop_cm=free_prop_cm)
    out = []
    for th in (theta if theta is not None else [None]):
        o = obj if th is None else O.rotate_fwd(obj, O.rotation_coords(obj.shape[:3], th))
        tiles, _ = O.extract_tiles(o, np.round(pos).astype(int), P)
        pred, _ = O.predict(tiles, probe, phys)
        out.append(pred ** 2 if raw == 'intensity' else pred)
    return np.stack(out).astype(np.float32)


def _finite(st):
    assert len(st['losses']) >= 1 and np.all(np.isfinite(st['losses']))
    assert np.all(np.isfinite(st['delta'])) and np.all(np.isfinite(st['beta']))


def test_2d_ptychography_experimental_data(A, tmp_path):
    """demos/2d_ptychography_experimental_data.py:38-90: Optimizer OBJECTS for object / probe / positions, an aperture + defocus probe
    with a central stop spread over 5 modes, intensity rescaling, random initial guess, real_imag unknowns, all output switches."""
    r = cases.rng(2201)
    Y, X, P = 44, 40, 16
    pos = np.array([(y, x) for y in range(-2, 30, 4) for x in range(-2, 26, 4)], dtype=float) + r.uniform(-0.4, 0.4, (56, 2))
    obj = np.stack([1 - 0.2 * cases.smooth_field((Y, X, 1), 371), 0.2 * cases.smooth_field((Y, X, 1), 372)], -1)
    probe = O.aperture_defocus_probe((P, P), 4, 0.002, 1240. / 8801., 1.3e-6, beamstop_radius=1)
    tiles, _ = O.extract_tiles(obj, np.round(pos).astype(int), (P, P), 'real_imag')
    phys = O.Physics((P, P), 8801., 1.3e-6, unknown_type='real_imag')
    prj = (O.predict(tiles, probe, phys)[0] ** 2)[None].astype(np.float32)
    out = str(tmp_path)
    o_obj = A.AdamOptimizer('obj', output_folder=out, distribution_mode=None, options_dict={'step_size': 1e-3})
    o_probe = A.AdamOptimizer('probe', output_folder=out, distribution_mode=None, options_dict={'step_size': 1e-3, 'eps': 1e-7})
    o_pos = A.AdamOptimizer('probe_pos_correction', output_folder=out, distribution_mode=None, options_dict={'step_size': 1e-2})
    st = A.reconstruct_ptychography(
        fname=prj, probe_pos=pos, theta_st=0, theta_end=0, n_epochs=2, obj_size=(Y, X, 1), two_d_mode=True, energy_ev=8801.121930115722,
        psize_cm=1.3e-06, minibatch_size=35, output_folder='test', cpu_only=False, save_path=out, use_checkpoint=False,
        n_epoch_final_pass=None, save_intermediate=True, full_intermediate=True, initial_guess=None,
        random_guess_means_sigmas=(1., 0., 0.001, 0.002), n_dp_batch=350, probe_type='aperture_defocus', n_probe_modes=5, aperture_radius=4,
        beamstop_radius=1, probe_defocus_cm=0.002, rescale_probe_intensity=True, free_prop_cm='inf', backend='pytorch',
        raw_data_type='intensity', beamstop=None, optimizer=o_obj, optimize_probe=True, optimizer_probe=o_probe,
        optimize_all_probe_pos=True, optimizer_all_probe_pos=o_pos, save_history=True, update_scheme='immediate', unknown_type='real_imag',
        save_stdout=True, loss_function_type='lsq', normalize_fft=False, return_state=True)
    _finite(st)
    assert st['probe_real'].shape == (5, P, P)
    for f in ('obj_mag_ds_1.tiff', 'obj_phase_ds_1.tiff', 'probe_mag_ds_1.tiff', 'probe_phase_ds_1.tiff', 'summary.txt'):
        assert os.path.exists(os.path.join(out, 'test', f)), f
    assert os.path.isdir(os.path.join(out, 'test', 'intermediate', 'object'))


@pytest.mark.parametrize('probe_type', ['supplied', 'ifft'])
def test_2d_ptychography_dense_scan_position_correction_and_probe_optimization(A, tmp_path, probe_type):
    """demos/2d_ptychography_w_position_correction.py ('supplied' probe, cpu_only=True: ignored with a warning) and
    2d_ptychography_w_probe_optimization.py ('ifft' probe, optimize_probe): a phase-only object, the whole dense scan as ONE minibatch
    (more than 64 tiles on a pixel), sub-pixel position refinement, pass-through keywords 'center', 'probe_size', 'finite_support_mask'."""
    N, P = 48, 16
    pos = np.array([(y, x) for y in np.arange(-4, 36, 2) for x in np.arange(-4, 36, 2)])          # 400 positions, 2 pixels apart
    obj = np.stack([3e-6 * cases.smooth_field((N, N, 1), 381), np.zeros((N, N, 1))], -1)
    pm, pp = cases.smooth_field((P, P, 1), 382)[..., 0] + 0.5, 0.3 * cases.smooth_field((P, P, 1), 383)[..., 0]
    prj = _ptycho_data(obj, pos.astype(float), (pm * np.exp(1j * pp))[None], 5000., 1e-7, raw='magnitude')
    kw = dict(probe_type='supplied', probe_initial=[pm, pp], cpu_only=True) if probe_type == 'supplied' else \
        dict(probe_type='ifft', probe_initial=None, optimize_probe=True, cpu_only=False)
    with pytest.warns(UserWarning) if probe_type == 'supplied' else _nullcontext():
        st = A.reconstruct_ptychography(
            fname=prj, theta_st=0, theta_end=0, theta_downsample=1, n_epochs=2, obj_size=(N, N, 1), alpha_d=0, alpha_b=0, gamma=0,
            probe_size=(P, P), learning_rate=4e-3, center=512, energy_ev=5000, psize_cm=1.e-7, minibatch_size=400, n_batch_per_update=1,
            output_folder='recon', save_path=str(tmp_path), multiscale_level=1, n_epoch_final_pass=None, save_intermediate=True,
            full_intermediate=True, initial_guess=None, n_dp_batch=20, object_type='phase_only', probe_pos=pos, forward_algorithm='fresnel',
            finite_support_mask=None, free_prop_cm='inf', optimizer='adam', two_d_mode=True, distribution_mode=None, use_checkpoint=False,
            backend='pytorch', optimize_all_probe_pos=True, save_history=True, raw_data_type='magnitude', return_state=True, **kw)
    _finite(st)
    assert np.all(st['beta'] == 0)                                       # phase_only: the absorption channel is zeroed after every update
    assert np.abs(st['probe_pos_correction']).max() > 0


class _nullcontext(object):
    def __enter__(self): return None
    def __exit__(self, *a): return False


def test_multislice_ptycho_theta(A, tmp_path):
    """demos/multislice_ptycho_256_theta.py:52-93 at 32^3: Gaussian probe, L1 + TV, minibatch 2 / n_dp_batch 1, use_checkpoint=True with
    no checkpoint on disk (starts from scratch and writes one), a second invocation resuming from the first's result with
    reweighted_l1=True (the demo's epoch > 0 branch), pass-through 'xpu' / 'center' / run_bfloat16=False."""
    N, P, n_theta = 32, 12, 3
    pos = [(y, x) for y in np.arange(3) * 8 - 2 for x in np.arange(3) * 8 - 2]
    obj = np.stack([2e-6 * cases.smooth_field((N, N, N), 391), 2e-7 * cases.smooth_field((N, N, N), 392)], -1)
    mag, ph = O.generate_gaussian_map((P, P), 1, 3, 0.5, 3)
    theta = np.linspace(0, 2 * np.pi, n_theta, dtype='float32')
    prj = _ptycho_data(obj, np.array(pos, dtype=float), (mag * np.exp(1j * ph))[None], 5000., 1e-7, raw='magnitude', theta=theta)
    base = dict(fname=prj, theta_st=0, theta_end=2 * np.pi, theta_downsample=None, n_epochs=1, obj_size=(N, N, N), alpha_d=1e-9 * 1.7e7,
                alpha_b=1e-10 * 1.7e7, gamma=1e-9 * 1.7e7, probe_size=(P, P), learning_rate=5e-5, center=16, energy_ev=5000, psize_cm=1.e-7,
                minibatch_size=2, n_batch_per_update=1, cpu_only=False, save_path=str(tmp_path), multiscale_level=1,
                n_epoch_final_pass=None, save_intermediate=False, full_intermediate=False, n_dp_batch=1, probe_type='gaussian',
                probe_mag_sigma=3, probe_phase_sigma=3, probe_phase_max=0.5, forward_algorithm='fresnel', probe_pos=pos,
                finite_support_mask=None, free_prop_cm='inf', optimizer='adam', distribution_mode=None, use_checkpoint=True, backend='pytorch',
                run_bfloat16=False, run_float64=False, xpu=None, return_state=True)
    st0 = A.reconstruct_ptychography(output_folder='epoch_0', initial_guess=None, reweighted_l1=False, **base)
    _finite(st0)
    assert os.path.isdir(os.path.join(str(tmp_path), 'epoch_0', 'checkpoint'))
    st1 = A.reconstruct_ptychography(output_folder='epoch_1', initial_guess=[st0['delta'], st0['beta']], reweighted_l1=True, **base)
    _finite(st1)


def test_multislice_tomography(A, tmp_path):
    """demos/multislice_tomography_64.py:36-76 at 24^3: undivided full-field data (one 'position', minibatch 1), near-field detector at
    0 cm, an L1Regularizer OBJECT in `regularizers`, reweighted_l1, a finite-support mask from a TIFF, theta_downsample."""
    N, n_theta = 24, 20
    obj = np.stack([1e-5 * cases.smooth_field((N, N, N), 395), 1e-6 * cases.smooth_field((N, N, N), 396)], -1)
    theta = np.linspace(0, 2 * np.pi, n_theta, dtype='float32')
    prj = _ptycho_data(obj, np.zeros((1, 2)), np.ones((1, N, N), complex), 800., 0.67e-7, free_prop_cm=0, raw='magnitude', theta=theta)
    from adorym_amd._io import write_tiff
    zz, yy, xx = np.meshgrid(*[np.arange(N) - N / 2 + 0.5] * 3, indexing='ij')
    os.makedirs(os.path.join(str(tmp_path), 'mask'), exist_ok=True)
    mask_path = os.path.join(str(tmp_path), 'mask', 'mask.tiff')
    write_tiff(((zz ** 2 + yy ** 2 + xx ** 2) < (0.45 * N) ** 2).astype(np.float32), mask_path[:-5], dtype='float32')
    reg_l1 = A.L1Regularizer(alpha_d=1.e-9 * N ** 3, alpha_b=1.e-10 * N ** 3)
    st = A.reconstruct_ptychography(
        fname=prj, theta_st=0, theta_end=2 * np.pi, theta_downsample=10, n_epochs=2, regularizers=[reg_l1], obj_size=(N, N, N),
        probe_size=(N, N), learning_rate=1e-7, center=12, energy_ev=800, psize_cm=0.67e-7, minibatch_size=1, n_batch_per_update=1,
        output_folder='test', cpu_only=False, save_path=str(tmp_path), multiscale_level=1, n_epoch_final_pass=None, save_intermediate=True,
        full_intermediate=True, initial_guess=None, n_dp_batch=1, fresnel_approx=True, probe_type='plane', probe_initial=None,
        forward_algorithm='fresnel', object_type='normal', probe_pos=[(0, 0)], finite_support_mask_path=mask_path, free_prop_cm=0,
        optimize_probe_defocusing=False, probe_defocusing_learning_rate=1e-7, distribution_mode=None, optimizer='adam', use_checkpoint=False,
        binning=1, reweighted_l1=True, backend='pytorch', return_state=True)
    _finite(st)
    assert len(st['losses']) == 2 * 2                      # 20 angles / theta_downsample 10 = 2 angles per epoch
    outside = ((zz ** 2 + yy ** 2 + xx ** 2) >= (0.45 * N) ** 2)
    assert np.all(st['delta'][outside] == 0)               # the finite-support mask is applied after every update


def test_manual_scripts_multislice_ptycho_gd_binning_checkpoints(A, tmp_path):
    """tests/manual_scripts/test_multislice_ptycho_256_theta.py:40-78 and test_multislice_ptycho_64.py:36-82 of the reference at 32^3:
    the GD optimiser with its step-halving schedule, binning = 8 (4 modulation steps), theta_downsample, a minibatch that does not
    divide the scan, the forward model handed over as a CLASS, store_checkpoint=True, the old `shared_file_object` keyword."""
    N, P, n_theta = 32, 12, 20
    pos = [(y, x) for y in np.arange(3) * 8 - 2 for x in np.arange(3) * 8 - 2]
    obj = np.stack([2e-6 * cases.smooth_field((N, N, N), 397), 2e-7 * cases.smooth_field((N, N, N), 398)], -1)
    mag, ph = O.generate_gaussian_map((P, P), 1, 3, 0.5, 3)
    theta = np.linspace(0, 2 * np.pi, n_theta, dtype='float32')
    prj = _ptycho_data(obj, np.array(pos, dtype=float), (mag * np.exp(1j * ph))[None], 5000., 1e-7, raw='magnitude', theta=theta)
    st = A.reconstruct_ptychography(
        fname=prj, theta_st=0, theta_end=2 * np.pi, theta_downsample=10, n_epochs=2, obj_size=(N, N, N), alpha_d=0, alpha_b=0, gamma=0,
        probe_size=(P, P), learning_rate=1e-5, center=16, energy_ev=5000, psize_cm=1.e-7, minibatch_size=4, n_batch_per_update=1,
        output_folder='epoch_0', cpu_only=False, use_checkpoint=False, store_checkpoint=True, save_path=str(tmp_path), multiscale_level=1,
        n_epoch_final_pass=None, save_intermediate=True, full_intermediate=True, initial_guess=None, n_dp_batch=23, probe_type='gaussian',
        forward_algorithm='fresnel', forward_model=A.PtychographyModel, probe_pos=pos, finite_support_mask=None, probe_mag_sigma=3,
        probe_phase_sigma=3, probe_phase_max=0.5, reweighted_l1=False, optimizer='gd', free_prop_cm='inf', backend='pytorch', binning=8,
        shared_file_object=False, object_type='normal', optimize_probe_defocusing=False, probe_defocusing_learning_rate=1e-7,
        probe_learning_rate=1e-3, probe_learning_rate_init=1e-3, debug=False, update_scheme='immediate', distribution_mode=None,
        return_state=True)
    _finite(st)
    assert len(st['losses']) == 2 * 2 * 3                 # 2 epochs x 2 angles x ceil(9 / 4) minibatches
    assert os.path.isdir(os.path.join(str(tmp_path), 'epoch_0', 'checkpoint'))
