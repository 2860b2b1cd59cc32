"""
Parity of the HIP path (through the C ABI) against the reference's golden vectors and the
pinned CPU oracle.  All tests here need a real MI355X:  pytest -m gpu
Tolerances follow SURVEY.md section 8(c):
  forward |Psi|  : rel-L2 vs the fp64 reference <= 2e-6 (small S) / 5e-6 (S >= 32: 1e-7 * sqrt(#FFTs) growth)
  loss           : rel <= 1e-5
  gradient       : rel-L2 vs the fp64 reference <= 1e-4 and <= 3x the reference's own fp32 error (+ floor 1e-5)
  Adam / GD      : rel 2e-6
"""
import os
import numpy as np
import pytest

import cases
from oracle import adorym_oracle as O      # checker only

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')


def load(name):
    return np.load(os.path.join(G, name + '.npz'))


def rel(a, b):
    return np.linalg.norm(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)) / np.linalg.norm(np.asarray(b, dtype=np.float64))


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


# --------------------------------------------------------------------------- F2/F3 tile level
SINGLE_MODE = [n for n in cases.TILE_CASES if cases.TILE_CASES[n][5] == 1]


@pytest.mark.parametrize('name', SINGLE_MODE)
def test_tiles_forward_adjoint_vs_reference(A, ctx, name):
    c = cases.tile_case_inputs(name)
    g = load('F23_' + name)
    P, S, B = c['P'], c['S'], cases.TILE_B
    # the B tiles stacked along y form an object [B*P, P, S, 2]; no rotation; positions (b*P, 0)
    obj = c['guess'].reshape(B * P, P, S, 2)
    pos = np.array([(b * P, 0) for b in range(B)])
    eng = A.MultisliceEngine(ctx, (B * P, P, S), (P, P), pos, cases.ENERGY_EV, cases.PSIZE_CM,
                             free_prop_cm=c['free_prop_cm'], binning=c['binning'], fresnel_approx=c['fresnel_approx'],
                             sign_convention=c['sigma'], normalize_fft=c['normalize_fft'])
    d_obj = ctx.array(obj, np.float32)
    d_grad = ctx.zeros(obj.shape)
    d_probe = ctx.array(c2(c['probes'][0]))
    d_gp = ctx.zeros((P, P, 2))
    eng.set_batch(pos, g['meas'])
    eng.rotate(d_obj, None)
    eng.multislice(d_probe, grad_probe=d_gp, want_pred=True)
    eng.rotate_adjoint(d_grad, None)
    loss = eng.loss()
    pred = eng.pred()
    grad = d_grad.get().reshape(B, P, P, S, 2)
    gp = d_gp.get()

    tol_fwd = 2e-6 if S < 32 else 5e-6
    assert rel(pred, g['pred_64']) < tol_fwd
    assert abs(loss - g['loss_64']) <= 1e-5 * abs(g['loss_64'])
    if 'grad_tiles_32' in g.files:
        e_ref = rel(g['grad_tiles_32'], g['grad_tiles_64'])
    else:
        e_ref = float(g['grad_tiles_relerr_32'])
    e = rel(grad, g['grad_tiles_64'])
    assert e < 1e-4, e
    assert e <= 3 * e_ref + 1e-5, (e, e_ref)
    gp64 = np.stack([g['grad_probe_real_64'][0], g['grad_probe_imag_64'][0]], -1)
    assert rel(gp, gp64) < 1e-4


@pytest.mark.regression
def test_forward_only_matches_full(A, ctx):
    """grad_rot == NULL (predict-only) gives the same pred / loss as the full call."""
    name = 'p16_s32_far_bin4'
    c = cases.tile_case_inputs(name)
    g = load('F23_' + name)
    P, S, B = c['P'], c['S'], cases.TILE_B
    obj = c['guess'].reshape(B * P, P, S, 2)
    pos = np.array([(b * P, 0) for b in range(B)])
    eng = A.MultisliceEngine(ctx, (B * P, P, S), (P, P), pos, cases.ENERGY_EV, cases.PSIZE_CM, binning=c['binning'])
    d_obj = ctx.array(obj, np.float32)
    d_probe = ctx.array(c2(c['probes'][0]))
    eng.set_batch(pos, g['meas'])
    eng.rotate(d_obj, None)
    eng.multislice(d_probe, want_grad=False, want_pred=True)
    assert rel(eng.pred(), g['pred_64']) < 5e-6
    assert abs(eng.loss() - g['loss_64']) <= 1e-5 * abs(g['loss_64'])


def test_unsupported_configs_raise(A, ctx):
    """Unlisted and non-square probe sizes now run the generic kernel (tests/test_gpu_engine.py); what still raises is a
    probe whose field does not fit one workgroup, and per-position probes on a size without a tuned kernel."""
    pos = np.array([(0, 0)])
    A.MultisliceEngine(ctx, (20, 20, 4), (20, 20), pos, 5000., 1e-7)                # unlisted size: accepted
    A.MultisliceEngine(ctx, (16, 12, 4), (16, 12), pos, 5000., 1e-7)                # non-square: accepted
    with pytest.raises(NotImplementedError):
        A.MultisliceEngine(ctx, (200, 200, 2), (160, 160), pos, 5000., 1e-7)        # 25 600 pixels: more than one workgroup holds
    eng = A.MultisliceEngine(ctx, (30, 30, 2), (20, 20), pos, 5000., 1e-7, max_batch=1)
    eng.set_batch(pos, np.ones((1, 20, 20), np.float32))
    with pytest.raises(NotImplementedError):
        eng.multislice(ctx.zeros((1, 20, 20, 2)), shifts=ctx.zeros((1, 2)))          # sub-pixel probe shifts need a tuned size


# --------------------------------------------------------------------------- F4 rotation
@pytest.mark.parametrize('name', list(cases.ROT_CASES))
def test_rotation_vs_reference(A, ctx, name):
    g = load('F4_rotation')
    size, theta, obj, cot = cases.rot_case_inputs(name)
    Y, X, Z = size
    from adorym_amd.util import rotation_lookup
    coords = rotation_lookup(size, theta)
    assert np.array_equal(coords, g[name + '_coords'])
    eng = A.MultisliceEngine(ctx, size, (X, X), np.array([(0, 0)]), 5000., 1e-7)   # no pads (Y may be < P: tile unused)
    d_obj = ctx.array(obj, np.float32)
    d_coords = ctx.array(coords.view(np.uint16))
    eng.rotate(d_obj, d_coords)
    rot = eng.obj_rot.get()                                   # [Z][Yp][Xp][2]
    (py0, _), (px0, _) = eng.pads
    rot = rot[:, py0:py0 + Y, px0:px0 + X, :].transpose(1, 2, 0, 3)       # -> [Y,X,Z,2]
    assert np.abs(rot - g[name + '_rot_64']).max() < 5e-6
    # adjoint: grad_rot := cot, grad_obj += R^T cot
    full = np.zeros(eng.plan.rot_shape, np.float32)
    full[:, py0:py0 + Y, px0:px0 + X, :] = cot.transpose(2, 0, 1, 3)
    eng.grad_rot.set(full)
    d_g = ctx.zeros(obj.shape)
    eng.rotate_adjoint(d_g, d_coords)
    assert np.abs(d_g.get() - g[name + '_adj_64']).max() < 3e-5
    # the deterministic CSR-gather form of the same operator
    tab = A.RotationTable(ctx, size, theta)
    assert np.array_equal(tab.host, g[name + '_coords'])
    # ... and the table the product actually uses, formed on the DEVICE (adm_rotation_table_build): the reference's bits
    assert np.array_equal(tab.coords.get().view(np.float16), g[name + '_coords'])
    d_g2 = ctx.zeros(obj.shape)
    eng.rotate_adjoint(d_g2, tab)
    assert np.abs(d_g2.get() - g[name + '_adj_64']).max() < 3e-5
    d_g3 = ctx.zeros(obj.shape)
    eng.rotate_adjoint(d_g3, tab)
    assert np.array_equal(d_g2.get(), d_g3.get())          # bitwise reproducible


# --------------------------------------------------------------------------- F5 optimisers
def test_adam_gd_vs_reference(A, ctx):
    from adorym_amd._lib import check
    g = load('F5_optimizers')
    x = ctx.array(g['x0'], np.float32)
    m = ctx.zeros(g['x0'].shape)
    v = ctx.zeros(g['x0'].shape)
    n = x.size
    for k, t in enumerate((0, 0, 1, 1)):
        gk = ctx.array(g['gseq'][k], np.float32)
        check(ctx.lib.adm_adam_step(ctx.handle, x.ptr, gk.ptr, m.ptr, v.ptr, 0, n, t, 1e-4, 0.9, 0.999, 1e-7, 0, None))
        assert rel(x.get(), g['adam_x_32'][k]) < 2e-6
        assert rel(m.get(), g['adam_m_32'][k]) < 2e-6
        assert rel(v.get(), g['adam_v_32'][k]) < 2e-6
        assert rel(x.get(), g['adam_x_64'][k]) < 2e-6
    x = ctx.array(g['x0'], np.float32)
    for k, t in enumerate((0, 25, 70, 200)):
        gk = ctx.array(g['gseq'][k], np.float32)
        step = O.gd_step_size(t, 1e-2, True, 20)
        check(ctx.lib.adm_gd_step(ctx.handle, x.ptr, gk.ptr, 0, n, step, 0, None))
        assert rel(x.get(), g['gd_x_32'][k]) < 2e-6


def test_adam_shard_range_and_constraints(A, ctx):
    """[lo,hi) sharding (multi-GPU ZeRO-style update), non-negativity, channel zeroing and mask."""
    from adorym_amd._lib import check, FLAG_NONNEG, FLAG_ZERO_CH1
    r = cases.rng(21)
    shape = (4, 5, 6, 2)
    x0 = r.standard_normal(shape).astype(np.float32) * 1e-3
    gr = r.standard_normal(shape).astype(np.float32)
    mask = (r.uniform(size=shape[:-1]) > 0.3).astype(np.float32)
    xe, me, ve = O.adam_step(x0.copy(), gr, np.zeros_like(x0), np.zeros_like(x0), 3, step_size=1e-3)
    xe = O.apply_constraints(xe, non_negativity=True, object_type='phase_only', mask=mask)
    x = ctx.array(x0); m = ctx.zeros(shape); v = ctx.zeros(shape); gd = ctx.array(gr); dm = ctx.array(mask)
    n = x.size
    cut = 2 * 37          # shard boundary on a voxel boundary
    for lo, hi in ((0, cut), (cut, n)):
        check(ctx.lib.adm_adam_step(ctx.handle, x.ptr, gd.ptr, m.ptr, v.ptr, lo, hi, 3, 1e-3, 0.9, 0.999, 1e-7,
                                    FLAG_NONNEG | FLAG_ZERO_CH1, dm.ptr))
    assert np.allclose(x.get(), xe, rtol=2e-6, atol=1e-12)
    assert np.allclose(m.get(), me, rtol=2e-6, atol=0)


# --------------------------------------------------------------------------- F7 regularisers
def test_reg_grad_vs_reference(A, ctx):
    from adorym_amd._lib import check
    g = load('F7_regularizers')
    obj = g['obj']
    eng = A.MultisliceEngine(ctx, obj.shape[:3], (12, 12), np.array([(0, 0)]), 5000., 1e-7)
    d_obj = ctx.array(obj, np.float32)
    for (ad, ab, gm, gref, vref) in ((1.5, 0.7, 0.0, g['l1_grad'], g['l1_val']), (0.0, 0.0, 2.0, g['tv_grad'], g['tv_val'])):
        d_g = ctx.zeros(obj.shape)
        d_v = ctx.zeros((1,))
        check(ctx.lib.adm_reg_grad(eng.plan.handle, d_obj.ptr, ad, ab, gm, d_g.ptr, d_v.ptr))
        assert rel(d_g.get(), gref) < 1e-6
        assert abs(d_v.get()[0] - vref) <= 1e-5 * abs(vref)


# --------------------------------------------------------------------------- whole step vs oracle
@pytest.mark.parametrize('free_prop_cm', ['inf', 0])
def test_full_step_rotation_overlap_padding(A, ctx, free_prop_cm):
    """rotate -> overlapping, overhanging tiles -> multislice -> adjoint -> rotate^T, vs the fp64 oracle."""
    r = cases.rng(31)
    N, P, S = 32, 16, 32
    obj = np.stack([1e-3 * cases.smooth_field((N, N, S), 5), 1e-4 * cases.smooth_field((N, N, S), 6)], -1)
    pos = np.array([(-4, -4), (-4, 4), (4, 4), (12, 12), (20, 20), (20, 12), (4, 20)])
    probe = (0.5 + r.uniform(0, 1, (P, P))) * np.exp(1j * r.uniform(-np.pi, np.pi, (P, P)))
    theta = np.float32(1.234)
    coords = O.rotation_coords((N, N, S), theta)
    phys = O.Physics((P, P), cases.ENERGY_EV, cases.PSIZE_CM, free_prop_cm=free_prop_cm)
    # truth independent of the guess => residual of the order of the signal (well-conditioned gradient)
    truth = np.stack([1e-3 * cases.smooth_field((N, N, S), 15), 1e-4 * cases.smooth_field((N, N, S), 16)], -1)
    tiles, _ = O.extract_tiles(O.rotate_fwd(truth, coords, 'float64'), pos, (P, P))
    target = np.abs(O.multislice_forward(tiles, probe, phys, 'float64'))
    loss_o, pred_o, g_o, gp_o = O.forward_adjoint_object(obj, coords, probe, pos, target, phys, 'float64')
    loss32, _, g32, gp32 = O.forward_adjoint_object(obj.astype(np.float32), coords, probe, pos, target, phys, 'float32')

    eng = A.MultisliceEngine(ctx, (N, N, S), (P, P), pos, cases.ENERGY_EV, cases.PSIZE_CM, free_prop_cm=free_prop_cm)
    d_obj = ctx.array(obj, np.float32)
    d_coords = ctx.array(coords.view(np.uint16))
    d_probe = ctx.array(c2(probe))
    tab = A.RotationTable(ctx, (N, N, S), theta)
    for footprint, rot in ((True, d_coords), (False, d_coords), (True, tab)):
        d_grad = ctx.zeros(obj.shape)
        d_gp = ctx.zeros((P, P, 2))
        loss = eng.loss_and_grad(d_obj, d_grad, rot, d_probe, pos, target, grad_probe=d_gp, footprint=footprint)
        # guess and truth are close here, so the loss is a small difference of large magnitudes:
        # judge against the fp32 restatement of the reference on the same inputs (3x rule)
        assert abs(loss - loss_o) <= 3 * abs(loss32 - loss_o) + 1e-5 * abs(loss_o), (loss, loss_o, loss32)
        e = rel(d_grad.get(), g_o)
        e32 = rel(g32, g_o)
        assert e < 3e-4 and e <= 3 * e32 + 1e-5, (e, e32)
        ep, ep32 = rel(d_gp.get(), c2(gp_o[0])), rel(c2(gp32[0]), c2(gp_o[0]))
        assert ep < 5e-4 and ep <= 3 * ep32 + 1e-5, (ep, ep32)


def test_full_depth_256_slices_vs_oracle(A, ctx):
    """Config-3 depth (P=72, S=256, far field) on a small lateral object: forward and gradient vs the fp64 oracle."""
    r = cases.rng(41)
    P, S, Y, X = 72, 256, 84, 84
    obj = np.stack([3e-4 * cases.smooth_field((Y, X, S), 7, cutoff=0.15), 1.5e-5 * cases.smooth_field((Y, X, S), 8, cutoff=0.15)], -1)
    pos = np.array([(0, 0), (12, 12), (-6, 5)])
    probe = (0.5 + r.uniform(0, 1, (P, P))) * np.exp(1j * r.uniform(-np.pi, np.pi, (P, P)))
    phys = O.Physics((P, P), cases.ENERGY_EV, cases.PSIZE_CM, free_prop_cm='inf')
    truth = np.stack([3e-4 * cases.smooth_field((Y, X, S), 17, cutoff=0.15), 1.5e-5 * cases.smooth_field((Y, X, S), 18, cutoff=0.15)], -1)
    tiles, _ = O.extract_tiles(truth, pos, (P, P))
    target = np.abs(O.multislice_forward(tiles, probe, phys, 'float64'))
    loss_o, pred_o, g_o, _ = O.forward_adjoint_object(obj, None, probe, pos, target, phys, 'float64')
    loss32, pred32, g32, _ = O.forward_adjoint_object(obj.astype(np.float32), None, probe, pos, target, phys, 'float32')
    eng = A.MultisliceEngine(ctx, (Y, X, S), (P, P), pos, cases.ENERGY_EV, cases.PSIZE_CM)
    d_obj = ctx.array(obj, np.float32)
    d_grad = ctx.zeros(obj.shape)
    d_probe = ctx.array(c2(probe))
    eng.set_batch(pos, target)
    eng.rotate(d_obj, None)
    eng.multislice(d_probe, want_pred=True)
    eng.rotate_adjoint(d_grad, None)
    # 255 propagations deep, fp32 twiddle / transfer-function rounding errors are coherent from slice
    # to slice and grow ~linearly with depth in ANY fp32 FFT (the fp32 restatement of the reference
    # shows the same); the bar is the 3x rule against that restatement, plus an absolute cap.
    e_pred, e_pred32 = rel(eng.pred(), pred_o), rel(pred32, pred_o)
    assert e_pred <= 3 * e_pred32 + 2e-6 and e_pred < 5e-5, (e_pred, e_pred32)
    assert abs(eng.loss() - loss_o) <= 3 * abs(loss32 - loss_o) + 1e-5 * abs(loss_o), (eng.loss(), loss_o, loss32)
    e_g, e_g32 = rel(d_grad.get(), g_o), rel(g32, g_o)
    assert e_g <= 3 * e_g32 + 1e-5 and e_g < 2e-3, (e_g, e_g32)
    print('depth-256 errors vs fp64: pred %.2e (cpu fp32 %.2e), grad %.2e (cpu fp32 %.2e)' % (e_pred, e_pred32, e_g, e_g32))


def test_c3_shape_energy_conservation(A, ctx):
    """Full BASELINE size (256^3 object, 72x72 probe, 256 slices, minibatch 32): with beta = 0 the
    multislice operator is unitary (|H| = 1, |c| = 1), so by Parseval sum(pred^2) = Py*Px*sum|probe|^2
    for every position -- a size-independent property of the forward path at full scale."""
    N, P, B = 256, 72, 32
    r = cases.rng(51)
    ys = np.arange(23) * 12 - 36
    allpos = np.array([(y, x) for y in ys for x in ys])
    pos = allpos[200:200 + B]
    eng = A.MultisliceEngine(ctx, (N, N, N), (P, P), allpos, cases.ENERGY_EV, cases.PSIZE_CM)
    obj = np.zeros((N, N, N, 2), np.float32)
    obj[..., 0] = (3e-4 * r.uniform(size=(N, N, N))).astype(np.float32)
    d_obj = ctx.array(obj)
    probe = (0.5 + r.uniform(0, 1, (P, P))) * np.exp(1j * r.uniform(-np.pi, np.pi, (P, P)))
    d_probe = ctx.array(c2(probe))
    target = np.zeros((B, P, P), np.float32)
    from adorym_amd.util import rotation_lookup
    d_coords = ctx.array(rotation_lookup((N, N, N), np.float32(0.4)).view(np.uint16))
    d_grad = ctx.zeros(obj.shape)
    loss = eng.loss_and_grad(d_obj, d_grad, d_coords, d_probe, pos, target)
    eng.multislice(d_probe, want_grad=False, want_pred=True)
    pred = eng.pred().astype(np.float64)
    e_in = P * P * np.sum(np.abs(probe) ** 2)
    e_out = (pred ** 2).sum(axis=(1, 2))
    assert np.all(np.abs(e_out / e_in - 1) < 5e-5), np.abs(e_out / e_in - 1).max()   # fp32, 255 propagations
    assert np.isfinite(loss) and abs(loss - (pred ** 2).mean()) <= 1e-5 * loss
    gg = d_grad.get()
    assert np.all(np.isfinite(gg)) and np.abs(gg).max() > 0
    # beta-gradient identity for target = 0 : dL/dbeta_s summed over a tile = -2*k1*mean-energy ... (sign check)
    assert gg[..., 1].sum() < 0


# --------------------------------------------------------------------------- F9: Poisson, Momentum, reweighted L1
@pytest.mark.parametrize('rdt', ['magnitude', 'intensity'])
@pytest.mark.parametrize('pm', [1.0, 50.0])
def test_poisson_loss_vs_reference(A, ctx, rdt, pm):
    g = load('F9_variants')
    name = 'p12_s9_far_pos'
    c = cases.tile_case_inputs(name)
    meas = load('F23_' + name)['meas']
    P, S, B = c['P'], c['S'], cases.TILE_B
    obj = c['guess'].reshape(B * P, P, S, 2)
    pos = np.array([(b * P, 0) for b in range(B)])
    eng = A.MultisliceEngine(ctx, (B * P, P, S), (P, P), pos, cases.ENERGY_EV, cases.PSIZE_CM, loss_function_type='poisson',
                             poisson_multiplier=pm)
    d_grad = ctx.zeros(obj.shape)
    d_gp = ctx.zeros((P, P, 2))
    target = meas ** 2            # Poisson target is the measured INTENSITY for both raw data types
    eng.set_batch(pos, target)
    eng.rotate(ctx.array(obj, np.float32), None)
    eng.multislice(ctx.array(c2(c['probes'][0])), grad_probe=d_gp)
    eng.rotate_adjoint(d_grad, None)
    tag = '%s_pm%d_' % (rdt, int(pm))
    assert abs(eng.loss() - g['poisson_loss_' + tag + '64']) <= 2e-5 * abs(g['poisson_loss_' + tag + '64'])
    e = rel(d_grad.get().reshape(B, P, P, S, 2), g['poisson_grad_tiles_' + tag + '64'])
    e_ref = rel(g['poisson_grad_tiles_' + tag + '32'], g['poisson_grad_tiles_' + tag + '64'])
    assert e < 1e-4 and e <= 3 * e_ref + 1e-5, (e, e_ref)
    gp64 = np.stack([g['poisson_grad_probe_real_' + tag + '64'], g['poisson_grad_probe_imag_' + tag + '64']], -1)
    assert rel(d_gp.get(), gp64) < 1e-4


def test_momentum_vs_reference(A, ctx):
    g = load('F9_variants')
    opt = A.MomentumOptimizer('obj', options_dict={})
    opt.create_container(g['mom_x0'].shape, False, ctx)
    x = ctx.array(g['mom_x0'], np.float32)
    for k in range(3):
        x = opt.apply_gradient(x, ctx.array(g['mom_gseq'][k], np.float32), k, step_size=1e-3, gamma=0.9)
        assert rel(x.get(), g['mom_x_32'][k]) < 2e-6
    assert rel(opt.params_whole_array_dict['v'].get(), g['mom_v_32']) < 2e-6


def test_reweighted_l1_vs_reference(A, ctx):
    from adorym_amd._lib import check
    g = load('F9_variants')
    obj = g['rwl1_obj']
    eng = A.MultisliceEngine(ctx, obj.shape[:3], (12, 12), np.array([(0, 0)]), 5000., 1e-7)
    d_obj = ctx.array(obj, np.float32)
    d_w = ctx.empty(obj.shape)
    d_s = ctx.empty((2 * 1024 + 2,))
    check(ctx.lib.adm_rwl1_update(eng.plan.handle, d_obj.ptr, d_w.ptr, d_s.ptr))
    assert rel(d_w.get(), g['rwl1_weight']) < 2e-6
    d_g = ctx.zeros(obj.shape)
    d_v = ctx.zeros((1,))
    check(ctx.lib.adm_reg_grad_weighted(eng.plan.handle, d_obj.ptr, d_w.ptr, 0.8, 0.3, d_g.ptr, d_v.ptr))
    assert rel(d_g.get(), g['rwl1_grad']) < 2e-6
    assert abs(d_v.get()[0] - g['rwl1_val']) <= 1e-5 * abs(g['rwl1_val'])


# --------------------------------------------------------------------------- incoherent probe modes (next row f2)
@pytest.mark.parametrize('free_prop_cm', ['inf', 0])
def test_probe_modes_vs_reference_and_oracle(A, ctx, free_prop_cm):
    """pred = sqrt(sum_m |Psi_m|^2) (adorym/forward_model.py:354-375): far field against the reference golden
    (3 modes), near field against the pinned oracle."""
    name = 'p12_s9_far_modes3'
    c = cases.tile_case_inputs(name)
    g = load('F23_' + name)
    P, S, B, M = c['P'], c['S'], cases.TILE_B, c['n_modes']
    obj = c['guess'].reshape(B * P, P, S, 2)
    pos = np.array([(b * P, 0) for b in range(B)])
    if free_prop_cm == 'inf':
        meas, loss_ref, pred_ref = g['meas'], float(g['loss_64']), g['pred_64']
        gt_ref, e_ref = g['grad_tiles_64'], rel(g['grad_tiles_32'], g['grad_tiles_64'])
        gp_ref = np.stack([g['grad_probe_real_64'], g['grad_probe_imag_64']], -1)
    else:
        phys = O.Physics((P, P), cases.ENERGY_EV, cases.PSIZE_CM, free_prop_cm=0)
        meas = np.abs(O.predict(c['truth'], c['probes'], phys, 'float64')[0])
        loss_ref, pred_ref, gt_ref, gp = O.forward_adjoint_tiles(c['guess'], c['probes'], meas, phys, 'float64')
        _, _, gt32, _ = O.forward_adjoint_tiles(c['guess'].astype(np.float32), c['probes'], meas, phys, 'float32')
        e_ref = rel(gt32, gt_ref)
        gp_ref = np.stack([gp.real, gp.imag], -1)
    eng = A.MultisliceEngine(ctx, (B * P, P, S), (P, P), pos, cases.ENERGY_EV, cases.PSIZE_CM, free_prop_cm=free_prop_cm,
                             n_probe_modes=M)
    d_grad = ctx.zeros(obj.shape)
    d_probe = ctx.array(np.stack([c2(pm) for pm in c['probes']]))
    d_gp = ctx.zeros((M, P, P, 2))
    eng.set_batch(pos, meas)
    eng.rotate(ctx.array(obj, np.float32), None)
    eng.multislice(d_probe, grad_probe=d_gp, want_pred=True)
    eng.rotate_adjoint(d_grad, None)
    assert rel(eng.pred(), pred_ref) < 2e-6
    assert abs(eng.loss() - loss_ref) <= 1e-5 * abs(loss_ref)
    e = rel(d_grad.get().reshape(B, P, P, S, 2), gt_ref)
    assert e < 1e-4 and e <= 3 * e_ref + 1e-5, (e, e_ref)
    assert rel(d_gp.get(), gp_ref) < 1e-4
    # forward-only (predict) path with modes
    eng.multislice(d_probe, want_grad=False, want_pred=True)
    assert rel(eng.pred(), pred_ref) < 2e-6


# --------------------------------------------------------------------------- unknown_type = 'real_imag'
@pytest.mark.parametrize('fp', ['inf', 0])
def test_real_imag_vs_reference(A, ctx, fp):
    """Slices hold the complex transmission (adorym/propagate.py:243-249); pads are 1 + 0i (util.py:1338-1350)."""
    g = load('F10_real_imag')
    c = cases.tile_case_inputs('p12_s9_far_pos')
    P, S, B = 12, 9, cases.TILE_B
    obj = g['tiles'].reshape(B * P, P, S, 2)
    pos = np.array([(b * P, 0) for b in range(B)])
    eng = A.MultisliceEngine(ctx, (B * P, P, S), (P, P), pos, cases.ENERGY_EV, cases.PSIZE_CM, free_prop_cm=fp, unknown_type='real_imag')
    d_grad = ctx.zeros(obj.shape)
    d_gp = ctx.zeros((P, P, 2))
    eng.set_batch(pos, g['meas_%s' % fp])
    eng.rotate(ctx.array(obj, np.float32), None)
    eng.multislice(ctx.array(c2(c['probes'][0])), grad_probe=d_gp, want_pred=True)
    eng.rotate_adjoint(d_grad, None)
    tag = '%s_' % fp
    assert rel(eng.pred(), g['pred_' + tag + '64']) < 2e-6
    assert abs(eng.loss() - g['loss_' + tag + '64']) <= 1e-5 * abs(g['loss_' + tag + '64'])
    e, e_ref = rel(d_grad.get().reshape(B, P, P, S, 2), g['grad_tiles_' + tag + '64']), rel(g['grad_tiles_' + tag + '32'], g['grad_tiles_' + tag + '64'])
    assert e < 1e-4 and e <= 3 * e_ref + 1e-5, (e, e_ref)
    gp64 = np.stack([g['grad_probe_real_' + tag + '64'], g['grad_probe_imag_' + tag + '64']], -1)
    assert rel(d_gp.get(), gp64) < 1e-4


def test_real_imag_overhanging_tiles_vs_oracle(A, ctx):
    r = cases.rng(95)
    N, P, S = 20, 12, 4
    mag, ph = 1 - 0.2 * r.uniform(size=(N, N, S)), 0.5 * r.uniform(-1, 1, (N, N, S))
    obj = np.stack([mag * np.cos(ph), mag * np.sin(ph)], -1)
    pos = np.array([(-5, -5), (12, 14), (3, 2)])
    probe = (0.5 + r.uniform(0, 1, (P, P))) * np.exp(1j * r.uniform(-np.pi, np.pi, (P, P)))
    phys = O.Physics((P, P), 5000., 1e-7, unknown_type='real_imag')
    target = np.abs(r.standard_normal((3, P, P))) * 6
    loss_o, _, g_o, _ = O.forward_adjoint_object(obj, None, probe, pos, target, phys, 'float64')
    eng = A.MultisliceEngine(ctx, (N, N, S), (P, P), pos, 5000., 1e-7, unknown_type='real_imag')
    d_grad = ctx.zeros(obj.shape)
    loss = eng.loss_and_grad(ctx.array(obj, np.float32), d_grad, None, ctx.array(c2(probe)), pos, target)
    assert abs(loss - loss_o) <= 1e-5 * abs(loss_o)
    assert rel(d_grad.get(), g_o) < 1e-4


@pytest.mark.parametrize('theta,full', [(0.4, True), (0.7953982, False)])
def test_c3_full_size_minibatch_vs_oracle(A, ctx, theta, full):
    """BASELINE's config 3 at FULL size on the GPU -- 256^3 object, 72x72 probe, 256 slices, rotation by theta (0.4 rad
    and a 45-degree-class angle), L1 + TV -- against the fp64 oracle, gradient judged on the footprint planes under the
    3x rule measured against the REFERENCE-STRUCTURED fp32 restatement (oracle/torch_structured.py, bit-identical to
    the reference's PyTorch-CPU path).  Rotation about axis 0 acts on every y plane separately, so the CPU checkers work
    on the slab of planes the minibatch touches (+1 plane each side for the TV stencil).  At 0.4 rad the minibatch is a full
    one: 32 positions of the scan, two rows of the 23x23 grid, 16 of each (measured gradient error 7.3e-4 against 2.6e-3 for
    the reference-structured fp32 arithmetic and 1.4e-3 for the oracle's fp32 run, which is the yardstick here); at the 45-degree-class angle 6 positions of
    the same rows (the batch size only enters through the 2/(B*Py*Px) factor).  Three consecutive minibatches and a 'per angle'
    update through the driver, object against object: tests/test_gpu_fullsize.py."""
    from oracle import torch_child as T          # the torch-based checker runs in a child process (see oracle/torch_child.py)
    from adorym_amd.util import rotation_lookup
    N, P = 256, 72
    theta = np.float32(theta)
    ys = np.arange(23) * 12 - 36
    allpos = np.array([(y, x) for y in ys for x in ys])
    if full:
        sel = [11 * 23 + i for i in (0, 1, 2, 4, 5, 7, 8, 10, 11, 13, 14, 16, 17, 19, 21, 22)] + \
              [12 * 23 + i for i in (0, 1, 3, 4, 6, 7, 9, 10, 12, 13, 15, 16, 18, 19, 20, 22)]  # rows y = 96 and 108, x from -36 to 228
    else:
        sel = [11 * 23 + 0, 11 * 23 + 7, 11 * 23 + 22, 12 * 23 + 3, 12 * 23 + 12, 12 * 23 + 19]
    pos = allpos[sel]
    y_lo, y_hi = int(pos[:, 0].min()), int(pos[:, 0].max()) + P          # footprint planes [96, 180)
    s0, s1 = y_lo - 1, y_hi + 1                                           # slab for the CPU checkers
    r = cases.rng(61)
    shp = (s1 - s0, N, N)
    truth = np.stack([3e-4 * cases.smooth_field(shp, 71, cutoff=0.12), 1.5e-5 * cases.smooth_field(shp, 72, cutoff=0.12)], -1)
    guess = np.stack([3e-4 * cases.smooth_field(shp, 73, cutoff=0.12), 1.5e-5 * cases.smooth_field(shp, 74, cutoff=0.12)], -1)
    guess = 0.6 * truth + 0.4 * guess
    coords = rotation_lookup((N, N, N), theta)
    cfgp = dict(probe_type='gaussian', probe_mag_sigma=6, probe_phase_sigma=6, probe_phase_max=0.5)
    from adorym_amd.util import initialize_probe
    pr, pi = initialize_probe((P, P), **cfgp)
    probe = np.squeeze(pr) + 1j * np.squeeze(pi)
    phys = O.Physics((P, P), cases.ENERGY_EV, cases.PSIZE_CM, free_prop_cm='inf')
    pos_s = pos - np.array([s0, 0])
    tt, _ = O.extract_tiles(O.rotate_fwd(truth, coords, np.float64), pos_s, (P, P))
    meas = O.predict(tt, probe, phys, 'float64')[0]
    del tt
    a_d, a_b, gam = 1e-9 * 1.7e7, 1e-10 * 1.7e7, 1e-9 * 1.7e7
    V = float(N) ** 3

    def reg_grad(x):
        # regulariser gradient of the FULL object restricted to the slab's interior planes: the oracle normalises by the
        # slab's own size, the object's is N^3
        sc = x.shape[0] * x.shape[1] * x.shape[2] / V
        return (O.l1_value_grad(x, a_d, a_b)[1] + O.tv_value_grad(x, gam)[1]) * sc

    loss64, _, g64, _ = O.forward_adjoint_object(guess, coords, probe, pos_s, meas, phys, 'float64')
    g64 = (g64 + reg_grad(guess))[1:-1]
    # the yardstick of the 3x rule: fp32 arithmetic of the same path.  6 positions: the reference's own op structure on
    # PyTorch-CPU autograd, fp32 rotation either side; the full minibatch: the oracle's fp32 run (the autograd graph of 32
    # positions x 256 slices takes another minute of CPU for the same kind of number)
    g32in = guess.astype(np.float32)
    if full:
        loss32, _, g32, _ = O.forward_adjoint_object(g32in, coords, probe, pos_s, meas.astype(np.float32), phys, 'float32')
        g32 = (g32 + reg_grad(g32in).astype(np.float32))[1:-1]
    else:
        rot32 = O.rotate_fwd(g32in, coords, np.float32)
        # (PyTorch-CPU is fastest at ~16 threads on this path: bench.py's thread sweep)
        loss32, grot32 = T.loss_and_grad_subprocess(rot32, pos_s, probe, phys.h, phys.k1, meas.astype(np.float32), threads=min(16, os.cpu_count() or 1))
        g32 = (O.rotate_adj(np.asarray(grot32), coords, np.float32) + reg_grad(g32in).astype(np.float32))[1:-1]
        del rot32, grot32
    # ---- GPU: the full 256^3 object (zero outside the slab; the TV term sees that edge only on the two slab-edge planes,
    # which are not compared) ----
    obj = np.zeros((N, N, N, 2), np.float32)
    obj[s0:s1] = guess
    eng = A.MultisliceEngine(ctx, (N, N, N), (P, P), allpos, cases.ENERGY_EV, cases.PSIZE_CM, max_batch=len(pos))
    d_obj = ctx.array(obj)
    d_grad = ctx.zeros(obj.shape)
    tab = A.RotationTable(ctx, (N, N, N), theta)
    d_probe = ctx.array(c2(probe))
    loss = eng.loss_and_grad(d_obj, d_grad, tab, d_probe, pos, meas.astype(np.float32))
    from adorym_amd._lib import check
    check(ctx.lib.adm_reg_grad(eng.plan.handle, d_obj.ptr, a_d, a_b, gam, d_grad.ptr, None))
    g = d_grad.get()[y_lo:y_hi]
    e, e32 = rel(g, g64), rel(g32, g64)
    print('   loss rel. error %.2e (fp32 yardstick: %.2e)' % (abs(loss - loss64) / abs(loss64), abs(loss32 - loss64) / abs(loss64)))
    if full:
        # NumPy's fp32 transforms drift less with depth than the reference's (torch / pocketfft): the oracle's fp32 loss is
        # within 5e-6 here, the reference's own fp32 loss error at this depth is 9.8e-5 (golden F17) -- that is the bar
        assert abs(loss - loss64) <= 1.5e-4 * abs(loss64), (loss, loss64, loss32)
    else:
        assert abs(loss - loss64) <= 3 * abs(loss32 - loss64) + 1e-5 * abs(loss64), (loss, loss64, loss32)
    print('full-size C3 minibatch, theta %.4f, %d positions: gradient rel-L2 vs fp64 %.2e (%s fp32: %.2e)'
          % (theta, len(pos), e, 'oracle' if full else 'reference-structured', e32))
    assert e <= 3 * e32 + 1e-5 and e < 5e-3, (e, e32)
    # outside the footprint the data term is exactly zero: only the regulariser (of the zero object: sign(0) = 0) remains
    assert not d_grad.get()[:max(0, s0 - 1)].any()
