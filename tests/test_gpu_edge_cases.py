"""Edge cases of the C ABI and the host layer on the GPU: error behaviour, ragged / overhanging inputs,
forward-only and binned paths, determinism of the gradient path, full-field (config-2 shape) run."""
import ctypes as C
import os
import numpy as np
import pytest

import cases
from oracle import adorym_oracle as O      # checker only

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def A():
    import adorym_amd
    return adorym_amd


@pytest.fixture(scope='module')
def ctx(A):
    c = A.Context(0)
    yield c
    c.close()


def c2(z):
    return np.stack([z.real, z.imag], -1).astype(np.float32)


def rel(a, b):
    return np.linalg.norm(np.asarray(a, np.float64) - np.asarray(b, np.float64)) / np.linalg.norm(np.asarray(b, np.float64))


def test_c_abi_error_codes_and_messages(A, ctx):
    from adorym_amd import _lib
    lib = ctx.lib
    # null / invalid arguments return ADM_ERR_INVALID and leave a message; nothing throws or crashes
    assert lib.adm_rotate_fwd(None, None, None, None, 0, 1) == _lib.ADM_ERR_INVALID
    assert b'null' in lib.adm_last_error()
    eng = A.MultisliceEngine(ctx, (16, 16, 4), (16, 16), np.array([(0, 0)]), 5000., 1e-7)
    obj = ctx.zeros((16, 16, 4, 2))
    assert lib.adm_rotate_fwd(eng.plan.handle, obj.ptr, None, eng.obj_rot.ptr, 5, 3) == _lib.ADM_ERR_INVALID      # y_lo > y_hi
    assert lib.adm_rotate_fwd(eng.plan.handle, obj.ptr, None, eng.obj_rot.ptr, 0, 99) == _lib.ADM_ERR_INVALID     # beyond Y
    assert lib.adm_rotate_fwd(eng.plan.handle, obj.ptr, None, eng.obj_rot.ptr, 4, 4) == _lib.ADM_OK               # empty range: no-op
    probe = ctx.zeros((16, 16, 2)); pos = ctx.array(np.zeros((1, 2), np.int32)); tgt = ctx.zeros((1, 16, 16)); loss = ctx.zeros((1,))
    rc = lib.adm_multislice_fwd_adj(eng.plan.handle, eng.obj_rot.ptr, probe.ptr, pos.ptr, 0, tgt.ptr, 0, None, None, loss.ptr, 1.0, None, 0)
    assert rc == _lib.ADM_ERR_INVALID and b'batch' in lib.adm_last_error()                                          # empty batch
    rc = lib.adm_multislice_fwd_adj(eng.plan.handle, eng.obj_rot.ptr, probe.ptr, pos.ptr, 1, tgt.ptr, 1, None, None, loss.ptr, 1.0, None, 0)
    assert rc == _lib.ADM_ERR_INVALID and b'workspace' in lib.adm_last_error()                                      # gradient without scratch
    with pytest.raises(ValueError):
        A.Plan(ctx, (16, 16, 4), (16, 16), ((0, 0), (0, 0)), 1.0, np.ones((16, 16), complex), binning=0)
    with pytest.raises(ValueError):
        A.Plan(ctx, (16, 16, 4), (16, 16), ((0, 0), (0, 0)), 1.0, np.ones((16, 16), complex), sign_convention=2)
    d = ctx.zeros((4,))
    with pytest.raises(ValueError):
        d.set(np.zeros(5, np.float32))                                                                              # size mismatch on upload


def test_tiles_hanging_over_every_edge_and_duplicates(A, ctx):
    """Positions far outside on all four sides (zero padding, adorym/util.py:1327-1406) and a duplicated position
    (the reference pads short spot lists with repeats, ptychography.py:816-819): gradients add up."""
    r = cases.rng(61)
    N, P, S = 20, 12, 6
    obj = np.stack([1e-3 * r.uniform(size=(N, N, S)), 1e-4 * r.uniform(size=(N, N, S))], -1)
    pos = np.array([(-11, -11), (-11, 19), (19, -11), (19, 19), (4, 4), (4, 4)])
    probe = (0.5 + r.uniform(0, 1, (P, P))) * np.exp(1j * r.uniform(-np.pi, np.pi, (P, P)))
    phys = O.Physics((P, P), 5000., 1e-7)
    target = np.abs(r.standard_normal((len(pos), P, P))) * 5
    loss_o, _, g_o, _ = O.forward_adjoint_object(obj, None, probe, pos, target, phys, 'float64')
    eng = A.MultisliceEngine(ctx, (N, N, S), (P, P), pos, 5000., 1e-7)
    assert eng.pads.tolist() == [[11, 11], [11, 11]]
    d_grad = ctx.zeros(obj.shape)
    loss = eng.loss_and_grad(ctx.array(obj, np.float32), d_grad, None, ctx.array(c2(probe)), pos, target)
    assert abs(loss - loss_o) <= 1e-5 * abs(loss_o)
    assert rel(d_grad.get(), g_o) < 1e-4


@pytest.mark.regression
def test_gradient_path_is_bitwise_reproducible(A, ctx):
    """No atomics on the object-gradient path (tile overlap-add and CSR rotation adjoint are gathers)."""
    r = cases.rng(62)
    N, P, S = 32, 16, 8
    obj = np.stack([1e-3 * r.uniform(size=(N, N, S)), 1e-4 * r.uniform(size=(N, N, S))], -1).astype(np.float32)
    pos = np.array([(y, x) for y in (-4, 4, 12) for x in (-4, 4, 12, 20)])
    probe = ctx.array(c2((0.5 + r.uniform(0, 1, (P, P))) * np.exp(1j * r.uniform(-np.pi, np.pi, (P, P)))))
    target = np.abs(r.standard_normal((len(pos), P, P))).astype(np.float32) * 4
    eng = A.MultisliceEngine(ctx, (N, N, S), (P, P), pos, 5000., 1e-7)
    tab = A.RotationTable(ctx, (N, N, S), np.float32(0.9))
    d_obj = ctx.array(obj)
    outs = []
    for _ in range(3):
        g = ctx.zeros(obj.shape)
        eng.loss_and_grad(d_obj, g, tab, probe, pos, target)
        outs.append(g.get())
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


def test_fullfield_config2_shape_driver_vs_oracle(A, ctx, tmp_path):
    """Config 2 of BASELINE.json: 64^3 full-field multislice tomography, one 64x64 'position' at (0,0), near field
    (free_prop_cm=0), plane probe, minibatch 1, L1 regulariser, finite-support mask, Adam -- the shape of the
    reference's only pytest (tests/test_multislice_tomography_64.py:20-65), on 6 angles."""
    N, n_theta = 64, 6
    truth = np.stack([2e-5 * cases.smooth_field((N, N, N), 71), 2e-7 * cases.smooth_field((N, N, N), 72)], -1)
    theta_ls = np.linspace(0, 2 * np.pi, n_theta, dtype='float32')
    phys = O.Physics((N, N), 800., 0.67e-7, free_prop_cm=0)
    probe = np.ones((N, N), complex)
    prj = np.zeros((n_theta, 1, N, N))
    for i, th in enumerate(theta_ls):
        rot = O.rotate_fwd(truth, O.rotation_coords((N, N, N), th), 'float64')
        prj[i, 0] = np.abs(O.multislice_forward(rot[None], probe, phys, 'float64'))[0]
    r = cases.rng(73)
    guess = [r.normal(8.7e-7, 1e-7, (N, N, N)), r.normal(5.1e-8, 1e-8, (N, N, N))]
    yy, xx, zz = np.meshgrid(*[np.arange(N)] * 3, indexing='ij')
    mask = (((xx - 31.5) ** 2 + (zz - 31.5) ** 2) < 28 ** 2).astype(np.float32)
    kw = dict(n_epochs=1, minibatch_size=1, optimizer='adam', learning_rate=1e-7, alpha_d=1.e-9 * 64 ** 3, alpha_b=1.e-10 * 64 ** 3,
              gamma=0)
    st = A.reconstruct_ptychography(fname=prj.astype(np.float32), obj_size=(N, N, N), probe_pos=[(0, 0)], theta_st=0, theta_end=2 * np.pi,
                                    n_theta=n_theta, energy_ev=800., psize_cm=0.67e-7, free_prop_cm=0, probe_type='plane',
                                    initial_guess=guess, finite_support_mask_path=mask, save_path=str(tmp_path), output_folder='ff',
                                    store_checkpoint=False, use_checkpoint=False, return_state=True, **kw)
    ref, losses, _ = O.reconstruct(prj.astype(np.float32).astype(np.float64), guess, probe, np.array([(0., 0.)]), theta_ls, phys,
                                   mask=mask, dtype='float64', return_trace=True, n_epochs=1, minibatch_size=1, optimizer='adam',
                                   learning_rate=1e-7, alpha_d=kw['alpha_d'], alpha_b=kw['alpha_b'], gamma=None)
    x = np.stack([st['delta'], st['beta']], -1)
    assert np.allclose(st['losses'], losses, rtol=1e-4)
    assert np.sqrt(np.mean((x - ref) ** 2)) < 1e-5                     # BASELINE's absolute bar
    d = np.abs(x - ref)
    assert (d > 3e-8).mean() < 2e-3, (d > 3e-8).mean()                 # lr = 1e-7 steps: all but sign-flip voxels agree
    assert np.all(x[mask == 0] == 0)                                   # finite-support mask applied


@pytest.mark.parametrize('P', [8, 12, 16, 18, 24, 27, 32, 36, 64, 72])
def test_every_compiled_probe_size_vs_oracle(A, ctx, P):
    """All FFT factorisations N = R1*R2 shipped in libadm (radices 2, 3, 4, 8, 9), forward and gradient."""
    r = cases.rng(80 + P)
    S, B = 5, 2
    obj = np.stack([2e-3 * r.uniform(size=(B * P, P, S)), 2e-4 * r.uniform(size=(B * P, P, S))], -1)
    pos = np.array([(b * P, 0) for b in range(B)])
    probe = (0.5 + r.uniform(0, 1, (P, P))) * np.exp(1j * r.uniform(-np.pi, np.pi, (P, P)))
    phys = O.Physics((P, P), 5000., 1e-7)
    truth = np.stack([2e-3 * r.uniform(size=(B * P, P, S)), 2e-4 * r.uniform(size=(B * P, P, S))], -1)   # independent of the guess
    target = np.abs(O.multislice_forward(O.extract_tiles(truth, pos, (P, P))[0], probe, phys, 'float64'))
    loss_o, pred_o, g_o, gp_o = O.forward_adjoint_object(obj, None, probe, pos, target, phys, 'float64')
    _, _, g32, _ = O.forward_adjoint_object(obj.astype(np.float32), None, probe, pos, target, phys, 'float32')
    eng = A.MultisliceEngine(ctx, (B * P, P, S), (P, P), pos, 5000., 1e-7)
    d_grad = ctx.zeros(obj.shape)
    d_gp = ctx.zeros((P, P, 2))
    eng.set_batch(pos, target)
    eng.rotate(ctx.array(obj, np.float32), None)
    eng.multislice(ctx.array(c2(probe)), grad_probe=d_gp, want_pred=True)
    eng.rotate_adjoint(d_grad, None)
    assert rel(eng.pred(), pred_o) < 2e-6
    assert abs(eng.loss() - loss_o) <= 2e-5 * abs(loss_o)
    e, e32 = rel(d_grad.get(), g_o), rel(g32, g_o)
    assert e < 1e-4 and e <= 3 * e32 + 1e-5, (P, e, e32)
    assert rel(d_gp.get(), c2(gp_o[0])) < 1e-4


def test_two_d_mode_config1_shape_driver_vs_oracle(A, ctx, tmp_path):
    """Config 1 of BASELINE.json in miniature: 2-D single-slice ptychography (obj_size[-1] == 1 => two_d_mode, no
    rotation), several probe modes, intensity data, overlapping raster scan, minibatch > 1."""
    r = cases.rng(91)
    Y, X, P, M = 60, 52, 16, 3
    truth = np.stack([2e-2 * cases.smooth_field((Y, X, 1), 92), 2e-3 * cases.smooth_field((Y, X, 1), 93)], -1)
    guess = [np.full((Y, X, 1), 1e-2), np.full((Y, X, 1), 1e-3)]
    pos = np.array([(y, x) for y in range(-4, 50, 8) for x in range(-4, 42, 8)], dtype=float)
    base = (0.5 + r.uniform(0, 1, (P, P))) * np.exp(1j * r.uniform(-np.pi, np.pi, (P, P)))
    probes = np.stack([base * (0.6 ** m) * np.exp(1j * m * 0.3 * r.uniform(-1, 1, (P, P))) for m in range(M)])
    phys = O.Physics((P, P), 8000., 1e-6)
    tiles, _ = O.extract_tiles(truth, pos, (P, P))
    inten = O.predict(tiles, probes, phys, 'float64')[0] ** 2                 # raw_data_type='intensity'
    prj = inten[None].astype(np.float32)
    st = A.reconstruct_ptychography(fname=prj, obj_size=(Y, X, 1), probe_pos=pos, energy_ev=8000., psize_cm=1e-6, free_prop_cm='inf',
                                    raw_data_type='intensity', n_probe_modes=M, probe_type='supplied',
                                    probe_initial=[np.abs(probes), np.angle(probes)], initial_guess=guess, minibatch_size=7,
                                    n_epochs=2, optimizer='adam', learning_rate=1e-4, gamma=0, alpha_d=0, alpha_b=0,
                                    save_path=str(tmp_path), output_folder='twod', store_checkpoint=False, use_checkpoint=False,
                                    return_state=True)
    ref, losses, _ = O.reconstruct(np.sqrt(prj.astype(np.float64)) ** 2, guess, probes, pos, np.zeros(1, 'float32'), phys, n_epochs=2,
                                   minibatch_size=7, optimizer='adam', learning_rate=1e-4, dtype='float64', two_d_mode=True,
                                   raw_data_type='intensity', return_trace=True)
    x = np.stack([st['delta'], st['beta']], -1)
    assert len(st['losses']) == len(losses) and np.allclose(st['losses'], losses, rtol=2e-4)
    upd = np.linalg.norm(ref - np.stack(guess, -1))
    assert np.linalg.norm(x - ref) < 5e-3 * upd, np.linalg.norm(x - ref) / upd


def test_two_d_real_imag_driver_vs_oracle(A, ctx, tmp_path):
    """Config 1's unknown type: 2-D ptychography with a complex-transmission object (unknown_type='real_imag'),
    guess given as (magnitude, phase) and converted like the reference (adorym/util.py:118-122)."""
    r = cases.rng(97)
    Y, X, P = 44, 40, 16
    mag_t = 1 - 0.3 * cases.smooth_field((Y, X, 1), 98)
    ph_t = 0.8 * cases.smooth_field((Y, X, 1), 99)
    truth = np.stack([mag_t * np.cos(ph_t), mag_t * np.sin(ph_t)], -1)
    guess = [np.full((Y, X, 1), 0.9), np.full((Y, X, 1), 0.1)]
    pos = np.array([(y, x) for y in range(-4, 36, 8) for x in range(-4, 32, 8)], dtype=float)
    probe = (0.5 + r.uniform(0, 1, (P, P))) * np.exp(1j * r.uniform(-np.pi, np.pi, (P, P)))
    phys = O.Physics((P, P), 8000., 1e-6, unknown_type='real_imag')
    tiles, _ = O.extract_tiles(truth, pos, (P, P), 'real_imag')
    prj = np.abs(O.multislice_forward(tiles, probe, phys, 'float64'))[None].astype(np.float32)
    st = A.reconstruct_ptychography(fname=prj, obj_size=(Y, X, 1), probe_pos=pos, energy_ev=8000., psize_cm=1e-6, free_prop_cm='inf',
                                    unknown_type='real_imag', probe_type='supplied', probe_initial=[np.abs(probe), np.angle(probe)],
                                    initial_guess=guess, minibatch_size=5, n_epochs=2, optimizer='adam', learning_rate=1e-3, gamma=0,
                                    alpha_d=0, alpha_b=0, save_path=str(tmp_path), output_folder='ri', store_checkpoint=False,
                                    use_checkpoint=False, return_state=True)
    g0 = guess[0] * np.exp(1j * guess[1])
    ref, losses, _ = O.reconstruct(prj.astype(np.float64), [g0.real, g0.imag], probe, pos, np.zeros(1, 'float32'), phys, n_epochs=2,
                                   minibatch_size=5, optimizer='adam', learning_rate=1e-3, dtype='float64', two_d_mode=True, return_trace=True)
    x = np.stack([st['delta'], st['beta']], -1)
    assert np.allclose(st['losses'], losses, rtol=2e-4)
    upd = np.linalg.norm(ref - np.stack([g0.real, g0.imag], -1))
    assert np.linalg.norm(x - ref) < 5e-3 * upd
    assert sorted(f for f in os.listdir(st['output_folder']) if f.startswith('obj_')) == ['obj_mag_ds_1.tiff', 'obj_phase_ds_1.tiff']


# ------------------------------------------------------------------------------------ f2 row: sub-pixel probe positions
@pytest.mark.parametrize('name,kind,fp', [('db_s5_far', 'delta_beta', 'inf'), ('ri_s1_far', 'real_imag', 'inf'),
                                          ('ri_s3_near', 'real_imag', 0)])
def test_position_gradients_vs_reference(A, ctx, name, kind, fp):
    """Fourier-shifted probes per position + gradients w.r.t. shifts, probe modes and tiles against the reference's
    autograd (golden F11; fp64 values, compared with the 3x rule against the reference's own fp32 run)."""
    f = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'F11_c1.npz'))
    tiles, probe, shifts, meas = f[name + '_tiles'], f[name + '_probe'], f[name + '_shifts'], f[name + '_meas']
    B, P, _, S, _ = tiles.shape
    M = probe.shape[0]
    # lay the tiles side by side in a [P, B*P, S] object
    obj = np.concatenate(list(tiles), axis=1)
    pos = np.array([[0, b * P] for b in range(B)])
    eng = A.MultisliceEngine(ctx, (P, B * P, S), (P, P), pos, cases.ENERGY_EV, cases.PSIZE_CM, free_prop_cm=fp, n_probe_modes=M,
                             max_batch=B, unknown_type=kind)
    eng.set_batch(pos, meas.astype(np.float32))
    obj_d = ctx.array(obj.astype(np.float32))
    eng.rotate(obj_d, None, None)
    probe_d = ctx.array(np.stack([probe.real, probe.imag], -1).astype(np.float32))
    sh_d = ctx.array(shifts.astype(np.float32))
    gp = ctx.zeros(probe_d.shape)
    gs = ctx.zeros(sh_d.shape)
    eng.multislice(probe_d, grad_probe=gp, want_pred=True, shifts=sh_d, grad_shifts=gs)
    g_obj = ctx.zeros(obj_d.shape)
    eng.rotate_adjoint(g_obj, None, None)
    loss = eng.loss()
    t64, t32 = name + '_64', name + '_32'
    assert abs(loss - f[t64 + '_loss']) < 2e-5 * abs(f[t64 + '_loss'])
    pred = eng.pred()
    assert np.linalg.norm(pred - f[t64 + '_pred']) < 2e-6 * np.linalg.norm(f[t64 + '_pred'])
    gt = np.stack(np.split(g_obj.get(), B, axis=1))
    gph = gp.get()
    for mine, key in ((gt, '_grad_tiles'), (gph[..., 0] + 1j * gph[..., 1], '_grad_probe'), (gs.get(), '_grad_shifts')):
        ref64, ref32 = f[t64 + key], f[t32 + key]
        err = np.linalg.norm(mine - ref64) / np.linalg.norm(ref64)
        err_ref = np.linalg.norm(ref32 - ref64) / np.linalg.norm(ref64)
        assert err < max(1e-4, 3 * err_ref), (key, err, err_ref)


def test_probe_shift_kernel_vs_reference(A, ctx):
    f = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'F11_c1.npz'))
    probe = f['shift_probe']
    M, P = probe.shape[:2]
    eng = A.MultisliceEngine(ctx, (P, P, 1), (P, P), np.zeros((1, 2)), cases.ENERGY_EV, cases.PSIZE_CM, n_probe_modes=M, max_batch=3)
    shifts = np.stack([f['shift%d_s' % k] for k in range(3)]).astype(np.float32)
    out = ctx.empty((3, M, P, P, 2))
    from adorym_amd._lib import check
    probe_d = ctx.array(np.stack([probe.real, probe.imag], -1).astype(np.float32))
    shifts_d = ctx.array(shifts)
    check(ctx.lib.adm_probe_shift(eng.plan.handle, probe_d.ptr, shifts_d.ptr, None, 3, out.ptr))
    o = out.get()
    for k in range(3):
        ref = f['shift%d_64' % k]
        assert np.abs(o[k, ..., 0] + 1j * o[k, ..., 1] - ref).max() < 3e-6 * np.abs(ref).max()


def test_real_imag_regularisers_vs_reference(A, ctx):
    f = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'F11_c1.npz'))
    obj = f['reg_obj']
    Y, X, Z = obj.shape[:3]
    eng = A.MultisliceEngine(ctx, (Y, X, Z), (8, 8), np.zeros((1, 2)), cases.ENERGY_EV, cases.PSIZE_CM, max_batch=1, unknown_type='real_imag')
    from adorym_amd._lib import check
    x = ctx.array(obj.astype(np.float32))
    for (ad, ab, gm), key in (((0., 0., 0.7), 'tv'), ((0.8, 0.3, 0.), 'l1')):
        g = ctx.zeros(x.shape)
        val = ctx.zeros((1,))
        check(ctx.lib.adm_reg_grad(eng.plan.handle, x.ptr, ad, ab, gm, g.ptr, val.ptr))
        assert abs(val.get()[0] - f['reg_%s_val' % key]) < 2e-6 * abs(f['reg_%s_val' % key])
        ref = f['reg_%s_grad' % key]
        assert np.abs(g.get() - ref).max() < 1e-5 * np.abs(ref).max()


def test_center_rows(A, ctx):
    from adorym_amd._lib import check
    r = cases.rng(5)
    x = r.standard_normal((1000, 2)).astype(np.float32) + 3
    d = ctx.array(x)
    check(ctx.lib.adm_center_rows(ctx.handle, d.ptr, 1000, 2))
    assert np.abs(d.get() - (x - x.mean(0))).max() < 1e-5


def test_config1_shape_driver_vs_reference(A, ctx, tmp_path):
    """The whole config-1 feature set through reconstruct_ptychography on the GPU against the REFERENCE driver's own run
    (golden F11): 2-D, real_imag unknowns, two probe modes, intensity data, probe rescaling, Adam on object + probe +
    sub-pixel position corrections, TV regulariser on |o|^2 and arg o."""
    f = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'F11_c1.npz'))
    Cc = cases.C1MINI
    inp = cases.c1mini_inputs()
    st = A.reconstruct_ptychography(
        fname=f['e2e_prj'], obj_size=(Cc['Y'], Cc['X'], 1), probe_pos=inp['pos_nominal'], theta_st=0, theta_end=0, n_theta=1,
        two_d_mode=True, energy_ev=Cc['energy_ev'], psize_cm=Cc['psize_cm'], free_prop_cm='inf', minibatch_size=Cc['minibatch_size'],
        n_epochs=2, initial_guess=[inp['guess'][0], inp['guess'][1]], probe_type='supplied',
        probe_initial=[inp['probe_guess'][0], inp['probe_guess'][1]], n_probe_modes=Cc['M'], rescale_probe_intensity=True,
        raw_data_type='intensity', optimize_probe=True, probe_learning_rate=1e-3, optimize_all_probe_pos=True,
        all_probe_pos_learning_rate=1e-2, unknown_type='real_imag', gamma=1e-6, optimizer='adam', learning_rate=1e-3,
        n_dp_batch=Cc['n_dp_batch'], save_path=str(tmp_path), output_folder='c1', store_checkpoint=False, use_checkpoint=False,
        return_state=True)
    assert np.allclose(st['losses'], f['e2e_losses_64'], rtol=5e-4)
    x = np.stack([st['delta'], st['beta']], -1)
    g0 = inp['guess'][0] * np.exp(1j * inp['guess'][1])
    upd = np.linalg.norm(f['e2e_obj_64'] - np.stack([g0.real, g0.imag], -1))

    def rel(a, b, scale):
        return np.linalg.norm(a - b) / scale
    e_obj, e_obj_ref = rel(x, f['e2e_obj_64'], upd), rel(f['e2e_obj_32'], f['e2e_obj_64'], upd)
    assert e_obj < max(5e-3, 3 * e_obj_ref), (e_obj, e_obj_ref)
    p = st['probe_real'] + 1j * st['probe_imag']
    pn = np.linalg.norm(f['e2e_probe_64'])
    assert rel(p, f['e2e_probe_64'], pn) < max(1e-4, 3 * rel(f['e2e_probe_32'], f['e2e_probe_64'], pn))
    cn = np.linalg.norm(f['e2e_pos_corr_64'])
    assert rel(st['probe_pos_correction'], f['e2e_pos_corr_64'], cn) < max(2e-2, 3 * rel(f['e2e_pos_corr_32'], f['e2e_pos_corr_64'], cn))


# ------------------------------------------------------------------------------------ f1 row: multi-distance holography
def test_multidistance_gradients_vs_reference(A, ctx):
    """adm_holo_fwd_adj against the reference's autograd through MultiDistModel's chain (golden F12): loss, prediction,
    gradients w.r.t. object, probe, the propagation distances and the affine registration matrices."""
    from adorym_amd.holography import HolographyEngine
    f = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'F12_multidist.npz'))
    Cc = cases.C5MINI
    inp = cases.c5mini_inputs()
    N = Cc['N']
    eng = HolographyEngine(ctx, (N, N), 3, Cc['energy_ev'], Cc['psize_cm'])
    obj = ctx.array(f['guess'].astype(np.float32))
    probe = ctx.array(np.stack([f['probe'].real, f['probe'].imag], -1).astype(np.float32))
    dists = ctx.array(inp['dists_guess'].astype(np.float32))
    aff = ctx.array(f['aff_guess'].astype(np.float32))
    data = ctx.array(f['data'].astype(np.float32))
    g_obj, g_probe = ctx.zeros(obj.shape), ctx.zeros(probe.shape)
    g_d, g_a = ctx.zeros((3,)), ctx.zeros((3, 2, 3))
    eng.forward_adjoint(obj, probe, dists, data, affine=aff, grad_obj=g_obj, grad_probe=g_probe, grad_dists=g_d, grad_affine=g_a,
                        want_pred=True)
    assert abs(eng.loss() - f['loss_64']) < 2e-5 * abs(f['loss_64'])
    assert np.linalg.norm(eng.pred() - f['pred_64']) < 2e-6 * np.linalg.norm(f['pred_64'])
    gp = g_probe.get()
    for mine, key in ((g_obj.get(), 'grad_obj'), (gp[..., 0] + 1j * gp[..., 1], 'grad_probe'), (g_d.get(), 'grad_dists'),
                      (g_a.get(), 'grad_affine')):
        r64, r32 = f[key + '_64'], f[key + '_32']
        err = np.linalg.norm(mine - r64) / np.linalg.norm(r64)
        err_ref = np.linalg.norm(r32 - r64) / np.linalg.norm(r64)
        assert err < max(1e-4, 3 * err_ref), (key, err, err_ref)


@pytest.mark.parametrize('N', [16, 64, 128, 512, 1024])
def test_large_field_transforms_vs_oracle(A, ctx, N):
    """The batched row-FFT (Stockham radix 8/4/2) behind the holography path at every pass structure, through the
    public entry point: plane probe, object = random complex field, one distance, data = 0 => pred = |propagated field|."""
    from adorym_amd.holography import HolographyEngine
    r = cases.rng(700 + N)
    o = (1 + 0.3 * r.standard_normal((N, N))) * np.exp(1j * r.uniform(-1, 1, (N, N)))
    eng = HolographyEngine(ctx, (N, N), 2, 8000., 5e-5)
    obj = ctx.array(np.stack([o.real, o.imag], -1).astype(np.float32))
    probe = ctx.array(np.stack([np.ones((N, N)), np.zeros((N, N))], -1).astype(np.float32))
    dists = np.array([3.0, 7.5])
    data = ctx.zeros((2, N, N))
    eng.forward_adjoint(obj, probe, ctx.array(dists.astype(np.float32)), data, want_grad=False, want_pred=True)
    _, pred, _, _, _, _, _ = O.holo_forward_adjoint(np.stack([o.real, o.imag], -1)[:, :, None, :], np.ones((N, N), complex), dists,
                                                    np.tile(np.array([[1., 0, 0], [0, 1., 0]]), [2, 1, 1]), np.zeros((2, N, N)), 8000., 5e-5)
    assert np.linalg.norm(eng.pred() - pred) < 3e-6 * np.linalg.norm(pred)


def test_config5_shape_driver_vs_reference(A, ctx, tmp_path):
    """Multi-distance holography through reconstruct_ptychography against the REFERENCE driver's own run (golden F12):
    object + propagation distances + affine registration matrices optimised together (config-5 feature set)."""
    f = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'F12_multidist.npz'))
    Cc = cases.C5MINI
    inp = cases.c5mini_inputs()
    N = Cc['N']
    st = A.reconstruct_ptychography(
        fname=f['data'][None], obj_size=(N, N, 1), probe_pos=np.array([[0., 0.]]), theta_st=0, theta_end=0, n_theta=1, two_d_mode=True,
        energy_ev=Cc['energy_ev'], psize_cm=Cc['psize_cm'], free_prop_cm=np.array(inp['dists_guess']), minibatch_size=1, n_epochs=4,
        initial_guess=[inp['guess'][0], inp['guess'][1]], probe_type='plane', raw_data_type='intensity', unknown_type='real_imag',
        gamma=0, alpha_d=0, alpha_b=0, optimizer='adam', learning_rate=1e-2, optimize_free_prop=True, free_prop_learning_rate=1e-1,
        optimize_prj_affine=True, prj_affine_learning_rate=1e-3, n_dp_batch=1, randomize_probe_pos=True, save_path=str(tmp_path),
        output_folder='c5', store_checkpoint=False, use_checkpoint=False, return_state=True)
    assert np.allclose(st['losses'], f['e2e_losses_64'], rtol=2e-3)
    d64, d32 = f['e2e_dists_64'], f['e2e_dists_32']
    assert np.abs(st['free_prop_cm'] - d64).max() < max(2e-3, 3 * np.abs(d32 - d64).max())
    a64, a32 = f['e2e_affine_64'], f['e2e_affine_32']
    assert np.abs(st['prj_affine_ls'] - a64).max() < max(2e-4, 3 * np.abs(a32 - a64).max())
    assert np.array_equal(st['prj_affine_ls'][0], np.array([[1., 0, 0], [0, 1., 0]], np.float32))
    x = np.stack([st['delta'], st['beta']], -1)
    g0 = inp['guess'][0] * np.exp(1j * inp['guess'][1])
    upd = np.linalg.norm(f['e2e_obj_64'] - np.stack([g0.real, g0.imag], -1))
    e, e_ref = np.linalg.norm(x - f['e2e_obj_64']) / upd, np.linalg.norm(f['e2e_obj_32'] - f['e2e_obj_64']) / upd
    assert e < max(5e-3, 3 * e_ref), (e, e_ref)


@pytest.mark.parametrize('B', [300, 600])
@pytest.mark.regression
def test_overlapped_split_launch_matches_single_launch(A, ctx, B):
    """A batch larger than the chip (B > 256 workgroups) launched as equal rounds (150 + 150, 200 + 200 + 200) with every
    round's overlap-add beside the next round (multislice_overlapped) gives the same tile-gradient sums and losses as one
    launch + one overlap-add."""
    r = cases.rng(314)
    Y, X, S, P = (60, 64, 4, 16) if B == 300 else (90, 96, 4, 16)     # (a pixel may be covered by at most 64 tiles of a launch)
    pos = np.stack([r.integers(-6, Y - 8, B), r.integers(-6, X - 8, B)], 1)
    pos[B - 10:] = pos[:10]                                # duplicates, like the padded minibatches of a fused angle
    eng = A.MultisliceEngine(ctx, (Y, X, S), (P, P), pos, cases.ENERGY_EV, cases.PSIZE_CM, max_batch=B)
    obj = ctx.array(np.stack([r.uniform(0, 2e-3, (Y, X, S)), r.uniform(0, 2e-4, (Y, X, S))], -1).astype(np.float32))
    probe = ctx.array((r.standard_normal((1, P, P, 2))).astype(np.float32))
    meas = (np.abs(r.standard_normal((B, P, P))) * 10).astype(np.float32)
    out = []
    for overlapped in (False, True):
        eng.set_batch(pos, meas)
        eng.rotate(obj, None, None)
        eng.grad_rot.zero_()
        gp = ctx.zeros(probe.shape)
        if overlapped:
            eng.multislice_overlapped(probe, grad_probe=gp)
        else:
            eng.multislice(probe, grad_probe=gp)
        g = ctx.zeros(obj.shape)
        eng.rotate_adjoint(g, None, None)
        out.append((eng.loss(), g.get(), eng._loss.get().copy(), gp.get()))
    assert out[0][0] == out[1][0]
    assert np.array_equal(out[0][2], out[1][2])
    assert np.abs(out[0][1] - out[1][1]).max() <= 2e-6 * np.abs(out[0][1]).max()
    assert np.abs(out[0][3] - out[1][3]).max() <= 1e-5 * np.abs(out[0][3]).max()


def test_beamstop_vs_reference(A, ctx):
    """Beamstop mask (forward_model.py:128-136) through the kernel against the reference's autograd (golden F13)."""
    f = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'F13_beamstop.npz'))
    name = 'p12_s9_far_pos'
    c = cases.tile_case_inputs(name)
    meas = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'F23_' + name + '.npz'))['meas']
    tiles = c['guess']
    B, P, _, S, _ = tiles.shape
    obj = np.concatenate(list(tiles), axis=1)
    pos = np.array([[0, b * P] for b in range(B)])
    eng = A.MultisliceEngine(ctx, (P, B * P, S), (P, P), pos, cases.ENERGY_EV, cases.PSIZE_CM, free_prop_cm=c['free_prop_cm'],
                             binning=c['binning'], max_batch=B, beamstop=f['beamstop'])
    eng.set_batch(pos, meas.astype(np.float32))
    obj_d = ctx.array(obj.astype(np.float32))
    eng.rotate(obj_d, None, None)
    probe = c['probes']
    probe_d = ctx.array(np.stack([probe.real, probe.imag], -1).astype(np.float32))
    gp = ctx.zeros(probe_d.shape)
    eng.multislice(probe_d, grad_probe=gp)
    g_obj = ctx.zeros(obj_d.shape)
    eng.rotate_adjoint(g_obj, None, None)
    assert abs(eng.loss() - f['loss_64']) < 2e-5 * abs(f['loss_64'])
    gt = np.stack(np.split(g_obj.get(), B, axis=1))
    gph = gp.get()
    for mine, key in ((gt, 'grad_tiles'), (gph[0, ..., 0] + 1j * gph[0, ..., 1], 'grad_probe')):
        r64, r32 = f[key + '_64'], f[key + '_32']
        err = np.linalg.norm(mine - r64) / np.linalg.norm(r64)
        err_ref = np.linalg.norm(r32 - r64) / np.linalg.norm(r64)
        assert err < max(1e-4, 3 * err_ref), (key, err, err_ref)


def test_config1_full_size_minibatch_vs_oracle(A, ctx):
    """BASELINE config 1 at its full size (618 x 606 x 1 real_imag object, 5 incoherent modes, minibatch 35; P = 64 as the
    data file is absent): loss and gradients w.r.t. object, probe modes and sub-pixel positions against the fp64 oracle."""
    r = cases.rng(6181)
    Y, X, P, M, B = 618, 606, 64, 5, 35
    energy, psize = 8801.121930115722, 1.32789376566526e-06
    pos = np.stack([r.integers(-20, Y - 44, B), r.integers(-20, X - 44, B)], 1).astype(float) + r.uniform(-0.4, 0.4, (B, 2))
    pos_int = np.round(pos).astype(int)
    mag = 1 - 0.3 * r.uniform(size=(Y, X, 1)); ph = 0.6 * r.uniform(-1, 1, (Y, X, 1))
    obj = np.stack([mag * np.cos(ph), mag * np.sin(ph)], -1)
    probe = (r.standard_normal((M, P, P)) + 1j * r.standard_normal((M, P, P))) * np.array([1, .5, .3, .2, .1])[:, None, None]
    phys = O.Physics((P, P), energy, psize, free_prop_cm='inf', unknown_type='real_imag')
    tiles, _ = O.extract_tiles(obj, pos_int, (P, P), 'real_imag')
    shifts = pos - pos_int
    meas = O.predict(tiles, probe, phys, 'float64')[0] * (1 + 0.2 * r.uniform(-1, 1, (B, P, P)))
    loss, pred, gt, gp, gs = O.forward_adjoint_tiles(tiles, probe, meas, phys, 'float64', shifts=shifts)
    g_ref = O.scatter_tiles_adj(gt, pos_int, obj.shape)
    eng = A.MultisliceEngine(ctx, (Y, X, 1), (P, P), pos_int, energy, psize, free_prop_cm='inf', n_probe_modes=M, max_batch=B,
                             unknown_type='real_imag')
    eng.set_batch(pos_int, meas.astype(np.float32))
    obj_d = ctx.array(obj.astype(np.float32))
    eng.rotate(obj_d, None, None)
    probe_d = ctx.array(np.stack([probe.real, probe.imag], -1).astype(np.float32))
    sh_d = ctx.array(shifts.astype(np.float32))
    gpd, gsd, g = ctx.zeros(probe_d.shape), ctx.zeros(sh_d.shape), ctx.zeros(obj_d.shape)
    eng.multislice(probe_d, grad_probe=gpd, shifts=sh_d, grad_shifts=gsd)
    eng.rotate_adjoint(g, None, None)
    assert abs(eng.loss() - loss) < 2e-5 * abs(loss)
    gph = gpd.get()
    for mine, ref, tol in ((g.get(), g_ref, 1e-4), (gph[..., 0] + 1j * gph[..., 1], gp, 1e-4), (gsd.get(), gs, 2e-4)):
        assert np.linalg.norm(mine - ref) < tol * np.linalg.norm(ref), np.linalg.norm(mine - ref) / np.linalg.norm(ref)


def test_config5_full_size_vs_oracle(A, ctx):
    """BASELINE config 5 at its full size (512 x 512, 4 distances): loss, prediction and gradients w.r.t. the object, the
    distances and the affine matrices against the fp64 oracle."""
    from adorym_amd.holography import HolographyEngine
    r = cases.rng(5125)
    N, nd = 512, 4
    energy, psize = 17050., 1e-4
    mag = 1 - 0.2 * cases.smooth_field((N, N, 1), 5126); ph = 0.5 * cases.smooth_field((N, N, 1), 5127)
    obj = np.stack([mag * np.cos(ph), mag * np.sin(ph)], -1)
    dists = np.array([40., 60., 90., 140.])
    aff = np.tile(np.array([[1., 0, 0], [0, 1., 0]]), [nd, 1, 1]) + 0.01 * r.uniform(-1, 1, (nd, 2, 3))
    data = (1 + 0.1 * cases.smooth_field((nd, N, N), 5128)) ** 2
    loss, pred, _, g_obj, _, g_d, g_a = O.holo_forward_adjoint(obj, np.ones((N, N), complex), dists, aff, data, energy, psize)
    # the affine-matrix gradient is ill-conditioned in fp32 at this size (sampling coordinates up to 512 carry 6e-5 px of
    # rounding, and the sum over 262 144 pixels cancels): the oracle evaluated in fp32 sets the scale (3x rule)
    g_a32 = O.holo_forward_adjoint(obj.astype(np.float32), np.ones((N, N), np.complex64), dists, aff, data, energy, psize,
                                   dtype='float32')[6]
    e_a32 = np.linalg.norm(g_a32 - g_a) / np.linalg.norm(g_a)
    eng = HolographyEngine(ctx, (N, N), nd, energy, psize)
    obj_d = ctx.array(obj.astype(np.float32))
    probe_d = ctx.array(np.stack([np.ones((N, N)), np.zeros((N, N))], -1).astype(np.float32))
    g, gd, ga = ctx.zeros(obj_d.shape), ctx.zeros((nd,)), ctx.zeros((nd, 2, 3))
    eng.forward_adjoint(obj_d, probe_d, ctx.array(dists.astype(np.float32)), ctx.array(data.astype(np.float32)),
                        affine=ctx.array(aff.astype(np.float32)), grad_obj=g, grad_dists=gd, grad_affine=ga, want_pred=True)
    assert abs(eng.loss() - loss) < 5e-5 * abs(loss)
    assert np.linalg.norm(eng.pred() - pred) < 3e-6 * np.linalg.norm(pred)
    for mine, ref, tol in ((g.get(), g_obj, 2e-4), (gd.get(), g_d, 2e-3), (ga.get(), g_a, max(2e-3, 3 * e_a32))):
        assert np.linalg.norm(mine - ref) < tol * np.linalg.norm(ref), np.linalg.norm(mine - ref) / np.linalg.norm(ref)


@pytest.mark.parametrize('shape', [(16, 16), (32, 128), (256, 64), (1024, 16), (16, 2048)])
def test_holography_gradients_at_every_line_geometry(A, ctx, shape):
    """The gradient path of the holography kernels (K3 with gradient, K4, K5) at field sizes that exercise every lines-per-block
    geometry of the block-cooperative transposed stores / loads of round 6 (128 lines per block at N = 16 ... one at N = 2048),
    square and not: loss, prediction, object / distance / affine gradients against the fp64 oracle
    (adorym/forward_model.py:809-1092 restated in oracle.holo_forward_adjoint)."""
    from adorym_amd.holography import HolographyEngine
    ny, nx = shape
    nd = 3
    r = cases.rng(9000 + ny * 7 + nx)
    energy, psize = 17050., 1e-4
    mag = 1 - 0.2 * cases.smooth_field((ny, nx, 1), 9100 + ny + nx); ph = 0.5 * cases.smooth_field((ny, nx, 1), 9200 + ny + nx)
    obj = np.stack([mag * np.cos(ph), mag * np.sin(ph)], -1)
    dists = np.array([3., 5., 9.])
    aff = np.tile(np.array([[1., 0, 0], [0, 1., 0]]), [nd, 1, 1]) + 0.01 * r.uniform(-1, 1, (nd, 2, 3))
    data = (1 + 0.1 * cases.smooth_field((nd, ny, nx), 9300 + ny + nx)) ** 2
    loss, pred, _, g_obj, _, g_d, g_a = O.holo_forward_adjoint(obj, np.ones((ny, nx), complex), dists, aff, data, energy, psize)
    g_a32 = O.holo_forward_adjoint(obj.astype(np.float32), np.ones((ny, nx), np.complex64), dists, aff, data, energy, psize, dtype='float32')[6]
    e_a32 = np.linalg.norm(g_a32 - g_a) / np.linalg.norm(g_a)
    eng = HolographyEngine(ctx, (ny, nx), nd, energy, psize)
    obj_d = ctx.array(obj.astype(np.float32))
    probe_d = ctx.array(np.stack([np.ones((ny, nx)), np.zeros((ny, nx))], -1).astype(np.float32))
    g, gd, ga = ctx.zeros(obj_d.shape), ctx.zeros((nd,)), ctx.zeros((nd, 2, 3))
    eng.forward_adjoint(obj_d, probe_d, ctx.array(dists.astype(np.float32)), ctx.array(data.astype(np.float32)),
                        affine=ctx.array(aff.astype(np.float32)), grad_obj=g, grad_dists=gd, grad_affine=ga, want_pred=True)
    assert abs(eng.loss() - loss) < 5e-5 * abs(loss)
    assert np.linalg.norm(eng.pred() - pred) < 5e-6 * np.linalg.norm(pred)
    for mine, ref, tol in ((g.get(), g_obj, 2e-4), (gd.get(), g_d, 2e-3), (ga.get(), g_a, max(2e-3, 3 * e_a32))):
        assert np.linalg.norm(mine - ref) < tol * np.linalg.norm(ref), (shape, np.linalg.norm(mine - ref) / np.linalg.norm(ref))


@pytest.mark.parametrize('which', ['all', 'object_only', 'object_and_dists'])
@pytest.mark.regression
def test_holography_fused_adam_equals_the_separate_update_bitwise(A, ctx, which):
    """adm_holo_fwd_adj_adam -- the Adam steps of object, distances and affine matrices inside the launch group's last kernel, no
    gradient stored -- against adm_holo_fwd_adj (overwriting) followed by the one small-parameter Adam launch: three minibatches,
    every array and every moment equal bit for bit, the loss too (adorym/ptychography.py:1120-1129, optimizers.py:1062-1083)."""
    from adorym_amd.holography import HolographyEngine
    from adorym_amd.optimizers import AdamOptimizer, apply_small_params
    ny, nx, nd = 128, 256, 3
    r = cases.rng(4242)
    energy, psize = 17050., 1e-4
    obj_h = np.stack([1 + 0.05 * r.standard_normal((ny, nx, 1)), 0.05 * r.standard_normal((ny, nx, 1))], -1).astype(np.float32)
    d_h = np.array([3., 5., 9.], np.float32)
    a_h = (np.tile(np.array([[1., 0, 0], [0, 1., 0]]), [nd, 1, 1]) + 0.01 * r.uniform(-1, 1, (nd, 2, 3))).astype(np.float32)
    data = ctx.array(((1 + 0.1 * r.standard_normal((nd, ny, nx))) ** 2).astype(np.float32))
    probe = ctx.array(np.stack([np.ones((1, ny, nx)), np.zeros((1, ny, nx))], -1).astype(np.float32))
    ident = ctx.array(np.array([[1., 0, 0], [0, 1., 0]], np.float32))
    use_d, use_a = which != 'object_only', which == 'all'
    out = []
    for fused in (False, True):
        eng = HolographyEngine(ctx, (ny, nx), nd, energy, psize)
        obj, dists, aff = ctx.array(obj_h), ctx.array(d_h), ctx.array(a_h)
        o_obj = AdamOptimizer('obj', options_dict={'step_size': 1e-2}); o_obj.create_param_arrays(list(obj.shape), device=ctx)
        o_d = AdamOptimizer('free_prop_cm', options_dict={'step_size': 1e-1}); o_d.create_param_arrays([nd], device=ctx)
        o_a = AdamOptimizer('prj_affine_ls', options_dict={'step_size': 1e-3}); o_a.create_param_arrays(list(aff.shape), device=ctx)
        mv = lambda o: (o.params_whole_array_dict['m'], o.params_whole_array_dict['v'])
        g, gd, ga = ctx.empty(obj.shape), ctx.empty((nd,)), ctx.empty(aff.shape)
        losses = []
        for k in range(3):
            if fused:
                eng.forward_adjoint_adam(obj, probe, dists, data, mv(o_obj), 1e-2, k, affine=aff,
                                         dists_mv=mv(o_d) if use_d else None, step_dists=1e-1,
                                         affine_mv=mv(o_a) if use_a else None, step_affine=1e-3, affine_pin=ident if use_a else None)
            else:
                eng.forward_adjoint(obj, probe, dists, data, affine=aff, grad_obj=g, grad_dists=gd if use_d else None,
                                    grad_affine=ga if use_a else None, overwrite=True)
                items = [dict(opt=o_obj, x=obj.view(0, (obj.size,)), g=g.view(0, (g.size,)))]
                if use_d:
                    items.append(dict(opt=o_d, x=dists, g=gd))
                if use_a:
                    items.append(dict(opt=o_a, x=aff, g=ga, pin=ident))
                apply_small_params(ctx, items, k)
            losses.append(eng.loss())
        out.append(dict(obj=obj.get(), d=dists.get(), a=aff.get(), m=[t.get() for o in (o_obj, o_d, o_a) for t in mv(o)], losses=losses))
    sep, fus = out
    assert np.abs(fus['obj'] - obj_h).max() > 1e-3 and (not use_d or np.abs(fus['d'] - d_h).max() > 1e-3)
    assert np.array_equal(sep['obj'], fus['obj']) and np.array_equal(sep['d'], fus['d']) and np.array_equal(sep['a'], fus['a'])
    for u, v in zip(sep['m'], fus['m']):
        assert np.array_equal(u, v)
    assert sep['losses'] == fus['losses']
    if use_a:
        assert np.array_equal(fus['a'][0], np.array([[1., 0, 0], [0, 1., 0]], np.float32))       # matrix 0 stays pinned to the identity


@pytest.mark.parametrize('delay', [0, 2])
@pytest.mark.regression
def test_config5_driver_fused_update_equals_separate_update_bitwise(A, ctx, tmp_path, monkeypatch, delay):
    """The config-5 feature set through reconstruct_ptychography with the Adam steps inside the gradient launch group
    (adm_holo_fwd_adj_adam: the default where the update is plain Adam on one rank) and with the separate small-parameter launch
    (ADM_HOLO_FUSED_ADAM=0): the same losses, object, distances and affine matrices, bit for bit -- also when the distances and
    matrices only start moving after `other_params_update_delay` minibatches (adorym/optimizers.py:1000-1083)."""
    from adorym_amd import holography as H
    f = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'F12_multidist.npz'))
    Cc = cases.C5MINI
    inp = cases.c5mini_inputs()
    N = Cc['N']
    calls = {'fused': 0}
    orig = H.HolographyEngine.forward_adjoint_adam

    def counting(self, *a, **k):
        calls['fused'] += 1
        return orig(self, *a, **k)

    monkeypatch.setattr(H.HolographyEngine, 'forward_adjoint_adam', counting)
    out = []
    for fusedv in ('0', '1'):
        monkeypatch.setenv('ADM_HOLO_FUSED_ADAM', fusedv)
        calls['fused'] = 0
        st = A.reconstruct_ptychography(
            fname=f['data'][None], obj_size=(N, N, 1), probe_pos=np.array([[0., 0.]]), theta_st=0, theta_end=0, n_theta=1, two_d_mode=True,
            energy_ev=Cc['energy_ev'], psize_cm=Cc['psize_cm'], free_prop_cm=np.array(inp['dists_guess']), minibatch_size=1, n_epochs=4,
            initial_guess=[inp['guess'][0], inp['guess'][1]], probe_type='plane', raw_data_type='intensity', unknown_type='real_imag',
            gamma=0, alpha_d=0, alpha_b=0, optimizer='adam', learning_rate=1e-2, optimize_free_prop=True, free_prop_learning_rate=1e-1,
            optimize_prj_affine=True, prj_affine_learning_rate=1e-3, n_dp_batch=1, randomize_probe_pos=True, save_path=str(tmp_path),
            output_folder='c5_' + fusedv, store_checkpoint=False, use_checkpoint=False, return_state=True, other_params_update_delay=delay)
        assert calls['fused'] == (4 if fusedv == '1' else 0)
        out.append(st)
    a, b = out
    assert a['losses'] == b['losses']
    for k in ('delta', 'beta', 'free_prop_cm', 'prj_affine_ls'):
        assert np.array_equal(a[k], b[k]), k
    assert np.abs(b['free_prop_cm'] - np.array(inp['dists_guess'])).max() > 1e-3


# ------------------------------------------------------------------------------------ f1 row: per-distance shift refinement
def test_shifted_holograms_gradients_vs_reference(A, ctx):
    """optimize_all_probe_pos with multi-distance data (adorym/forward_model.py:1075-1085): the registered targets
    Re IFFT2(FFT2(|data|) Phi(shift)), the loss and the gradients w.r.t. the object and the per-distance shifts against the
    reference's autograd (golden F19) under the 3x rule."""
    from adorym_amd.holography import HolographyEngine
    f = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'F19_multidist_shifts.npz'))
    Cc = cases.C5MINI
    N = Cc['N']
    eng = HolographyEngine(ctx, (N, N), 3, Cc['energy_ev'], Cc['psize_cm'])
    obj = ctx.array(f['guess'].astype(np.float32))
    probe = ctx.array(np.stack([np.ones((N, N)), np.zeros((N, N))], -1).astype(np.float32))
    dists = ctx.array(np.array(Cc['dists_cm'], np.float32))
    shifts = ctx.array(f['shift_guess'].astype(np.float32))
    spec = eng.data_spectrum(ctx.array(f['data'].astype(np.float32)))
    want = np.fft.fft2(np.abs(f['data'].astype(np.float64)))                       # [d][ky][kx]
    got = spec.get()
    got = (got[..., 0] + 1j * got[..., 1]).transpose(0, 2, 1)
    assert np.linalg.norm(got - want) < 2e-6 * np.linalg.norm(want)
    g_obj, g_sh = ctx.zeros(obj.shape), ctx.zeros((3, 2))
    eng.forward_adjoint_shifted(obj, probe, dists, spec, shifts, grad_obj=g_obj, grad_shifts=g_sh, want_pred=True)
    t64 = f['target_64']
    assert np.linalg.norm(eng.shifted_targets() - t64) < max(2e-6, 3 * np.linalg.norm(f['target_32'] - t64) / np.linalg.norm(t64)) * np.linalg.norm(t64)
    assert abs(eng.loss() - f['loss_64']) < 2e-5 * abs(f['loss_64'])
    assert np.linalg.norm(eng.pred() - f['pred_64']) < 2e-6 * np.linalg.norm(f['pred_64'])
    for mine, key in ((g_obj.get(), 'grad_obj'), (g_sh.get(), 'grad_shifts')):
        r64, r32 = f[key + '_64'], f[key + '_32']
        err = np.linalg.norm(mine - r64) / np.linalg.norm(r64)
        err_ref = np.linalg.norm(r32 - r64) / np.linalg.norm(r64)
        print('%s: %.2e (reference fp32: %.2e)' % (key, err, err_ref))
        assert err < max(1e-4, 3 * err_ref), (key, err, err_ref)
    # forward only: same loss, nothing written
    eng.forward_adjoint_shifted(obj, probe, dists, spec, shifts, want_grad=False)
    assert abs(eng.loss() - f['loss_64']) < 2e-5 * abs(f['loss_64'])


@pytest.mark.parametrize('shape', [(16, 64), (128, 32), (256, 256), (1024, 2048)])
def test_shifted_holograms_vs_oracle_at_other_sizes(A, ctx, shape):
    """The shift stage at non-square fields and other line lengths (lines-per-block geometries of the transforms)."""
    from adorym_amd.holography import HolographyEngine
    ny, nx = shape
    r = cases.rng(1900 + ny + nx)
    nd = 2
    dists = np.array([35., 70.])
    o = (1 + 0.1 * r.standard_normal((ny, nx))) * np.exp(1j * 0.2 * r.standard_normal((ny, nx)))
    obj_h = np.stack([o.real, o.imag], -1)[:, :, None, :]
    data = (1 + 0.2 * r.standard_normal((nd, ny, nx))) ** 2
    sh = r.uniform(-1.5, 1.5, (nd, 2))
    eng = HolographyEngine(ctx, (ny, nx), nd, 17050., 1e-4)
    obj = ctx.array(obj_h.astype(np.float32))
    probe = ctx.array(np.stack([np.ones((ny, nx)), np.zeros((ny, nx))], -1).astype(np.float32))
    spec = eng.data_spectrum(ctx.array(data.astype(np.float32)))
    g_obj, g_sh = ctx.zeros(obj.shape), ctx.zeros((nd, 2))
    eng.forward_adjoint_shifted(obj, probe, ctx.array(dists.astype(np.float32)), spec, ctx.array(sh.astype(np.float32)), grad_obj=g_obj, grad_shifts=g_sh)
    ident = np.tile(np.array([[1., 0, 0], [0, 1., 0]]), [nd, 1, 1])
    res = O.holo_forward_adjoint(obj_h, np.ones((ny, nx), complex), dists, ident, data.astype(np.float32).astype(np.float64), 17050., 1e-4, shifts=sh)
    assert abs(eng.loss() - res[0]) < 2e-5 * abs(res[0])
    assert np.linalg.norm(np.sqrt(np.abs(eng.shifted_targets())) - res[2]) < 1e-5 * np.linalg.norm(res[2])
    assert np.linalg.norm(g_obj.get() - res[3]) < 1e-4 * np.linalg.norm(res[3])
    # (for uncorrelated data the shift gradient is a sum of 10^4 ... 10^6 cancelling terms: the yardstick is the fp32 oracle's own error)
    r32 = O.holo_forward_adjoint(obj_h, np.ones((ny, nx), complex), dists, ident, data.astype(np.float32).astype(np.float64), 17050., 1e-4, shifts=sh,
                                 dtype='float32')
    e, e_ref = np.linalg.norm(g_sh.get() - res[7]), np.linalg.norm(r32[7] - res[7])
    assert e < max(2e-4 * np.linalg.norm(res[7]), 3 * e_ref), (g_sh.get(), res[7], r32[7])


def test_shift_refinement_driver_vs_reference(A, ctx, tmp_path):
    """demos/2d_multidist_holography_w_position_correction.py at a small size through reconstruct_ptychography against the REFERENCE
    driver's own run (golden F19): the object and one (sy, sx) per distance refined together, the corrections re-centred after
    every update (optimizers.py:1039-1049)."""
    f = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'F19_multidist_shifts.npz'))
    Cc = cases.C5MINI
    inp = cases.c5mini_inputs()
    N = Cc['N']
    st = A.reconstruct_ptychography(
        fname=f['data'][None], obj_size=(N, N, 1), probe_pos=np.array([[0., 0.]]), theta_st=0, theta_end=0, n_theta=1, two_d_mode=True,
        energy_ev=Cc['energy_ev'], psize_cm=Cc['psize_cm'], free_prop_cm=np.array(Cc['dists_cm']), minibatch_size=1, n_epochs=5,
        initial_guess=[inp['guess'][0], inp['guess'][1]], probe_type='plane', raw_data_type='intensity', unknown_type='real_imag',
        gamma=0, alpha_d=0, alpha_b=0, optimizer='adam', learning_rate=1e-2, optimize_all_probe_pos=True, all_probe_pos_learning_rate=1e-1,
        n_dp_batch=1, randomize_probe_pos=True, safe_zone_width=0, save_path=str(tmp_path), output_folder='sh', store_checkpoint=False,
        use_checkpoint=False, return_state=True)
    l64, l32 = f['e2e_losses_64'], f['e2e_losses_32']
    assert np.all(np.abs(np.array(st['losses']) - l64) <= np.maximum(2e-4 * np.abs(l64), 3 * np.abs(l32 - l64))), (st['losses'], l64)
    s64, s32 = f['e2e_shifts_64'], f['e2e_shifts_32']
    print('shifts', st['probe_pos_correction'], 'reference', s64)
    assert np.abs(st['probe_pos_correction'] - s64).max() < max(2e-3, 3 * np.abs(s32 - s64).max())
    assert abs(st['probe_pos_correction'].mean(axis=0)).max() < 1e-6          # re-centred
    x = np.stack([st['delta'], st['beta']], -1)
    g0 = inp['guess'][0] * np.exp(1j * inp['guess'][1])
    upd = np.linalg.norm(f['e2e_obj_64'] - np.stack([g0.real, g0.imag], -1))
    e, e_ref = np.linalg.norm(x - f['e2e_obj_64']) / upd, np.linalg.norm(f['e2e_obj_32'] - f['e2e_obj_64']) / upd
    assert e < max(5e-3, 3 * e_ref), (e, e_ref)


def test_dense_scan_as_one_minibatch_with_probe_from_data_vs_oracle(A, ctx, tmp_path):
    """The feature set of demos/2d_ptychography_w_probe_optimization.py at a small size: 2-D ptychography, the WHOLE dense scan as one
    minibatch (20 x 20 positions one pixel apart under a 16 x 16 probe: up to 256 tiles on a pixel -> the multi-pass overlap-add),
    probe_type='ifft' (estimated from the data), Adam on object + probe + sub-pixel position corrections.  Against the fp64 oracle
    under the 3x rule with the oracle's own fp32 run as the yardstick (the reference's pieces behind the oracle are pinned by F11 / F20)."""
    r = cases.rng(2100)
    N, P = 40, 16
    pos = np.array([(y, x) for y in range(2, 22) for x in range(2, 22)], dtype=float) + r.uniform(-0.3, 0.3, (400, 2))
    truth = np.stack([2e-6 * cases.smooth_field((N, N, 1), 361), 2e-7 * cases.smooth_field((N, N, 1), 362)], -1)
    pm, pp = cases.smooth_field((P, P, 1), 363)[..., 0] + 0.5, 0.3 * cases.smooth_field((P, P, 1), 364)[..., 0]
    probe_t = pm * np.exp(1j * pp)
    phys = O.Physics((P, P), 8000., 1e-5)
    tiles, _ = O.extract_tiles(truth, np.round(pos).astype(int), (P, P))
    pred, _ = O.predict(tiles, probe_t, phys)
    prj = (pred ** 2)[None].astype(np.float32)                                   # intensity data
    guess = [np.full((N, N, 1), 5e-7), np.full((N, N, 1), 5e-8)]
    kw = dict(n_epochs=3, minibatch_size=400, learning_rate=1e-7, raw_data_type='intensity', optimize_probe=True, probe_learning_rate=1e-3,
              optimize_all_probe_pos=True, all_probe_pos_learning_rate=1e-2)
    st = A.reconstruct_ptychography(
        fname=prj, obj_size=(N, N, 1), probe_pos=pos, theta_st=0, theta_end=0, n_theta=1, two_d_mode=True, energy_ev=8000., psize_cm=1e-5,
        free_prop_cm='inf', initial_guess=guess, probe_type='ifft', gamma=0, alpha_d=0, alpha_b=0, optimizer='adam', save_path=str(tmp_path),
        output_folder='dense', store_checkpoint=False, use_checkpoint=False, return_state=True, **kw)
    p0 = O.probe_ifft_guess(prj, 'intensity', 1)
    runs = {}
    for dt in ('float64', 'float32'):
        runs[dt] = O.reconstruct_2d(prj.astype(np.float64), guess, p0[None], pos, phys, dtype=dt, **kw)
    o64, o32 = runs['float64'], runs['float32']
    assert np.allclose(st['losses'], o64['losses'], rtol=max(2e-4, 3 * np.abs(np.array(o32['losses']) / np.array(o64['losses']) - 1).max()))
    x = np.stack([st['delta'], st['beta']], -1)
    upd = np.linalg.norm(o64['obj'] - np.stack(guess, -1))
    e, e_ref = np.linalg.norm(x - o64['obj']) / upd, np.linalg.norm(o32['obj'] - o64['obj']) / upd
    print('dense scan: object %.2e of the update from the fp64 oracle (oracle fp32: %.2e)' % (e, e_ref))
    assert upd > 0 and e < max(5e-3, 3 * e_ref), (e, e_ref)
    p = (st['probe_real'] + 1j * st['probe_imag'])[0]
    pn = np.linalg.norm(o64['probes'][0])
    assert np.linalg.norm(p - o64['probes'][0]) / pn < max(1e-4, 3 * np.linalg.norm(o32['probes'][0] - o64['probes'][0]) / pn)
    cn = np.linalg.norm(o64['pos_corr'])
    assert np.linalg.norm(st['probe_pos_correction'] - o64['pos_corr']) / cn < max(2e-2, 3 * np.linalg.norm(o32['pos_corr'] - o64['pos_corr']) / cn)
