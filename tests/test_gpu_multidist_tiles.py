"""SURVEY section 8 row f1, the branch config 5 does not take: multi-distance holograms DIVIDED INTO SUB-TILES and propagated with a
safe zone around every tile (adorym/forward_model.py:884-1034: n_blocks > 1, safe_zone_width >= 0).  The product runs a minibatch
of tiles as n_dists launches of the multislice kernel (one near-field plan per distance, one probe window per tile, detector mask =
the sub-hologram's window).  Checked against golden F18 -- the reference DRIVER run here in fp64 / fp32 on 3 x 3 tiles of 16 x 16
pixels, safe zone 4 and 0, plane and field-dependent probe, both unknown types -- and against the oracle pinned to it."""
import os

import numpy as np
import pytest

import cases
from oracle import adorym_oracle as O      # checker only

pytestmark = pytest.mark.gpu
F18 = os.path.join(os.path.dirname(__file__), 'golden', 'F18_multidist_tiles.npz')
RUNS = sorted(cases.C5TILES['runs'])


@pytest.fixture(scope='module')
def A():
    import adorym_amd
    return adorym_amd


@pytest.fixture(scope='module')
def ctx(A):
    c = A.Context(0)
    yield c
    c.close()


def _init(inp):
    g = inp['guess']
    return [g[0] * np.cos(g[1]), g[0] * np.sin(g[1])] if inp['unknown_type'] == 'real_imag' else [g[0], g[1]]


def _model(A, ctx, rn, prj):
    """MultiDistModel over the tile engine, put together the way reconstruct_ptychography does."""
    from adorym_amd.forward_model import MultiDistModel
    C = cases.C5TILES
    inp = cases.c5tiles_inputs(rn)
    N, SUB, szw = C['N'], C['SUB'], inp['szw']
    T = SUB + 2 * szw
    nd = len(C['dists_cm'])
    window = np.zeros((T, T), np.float32)
    window[szw:T - szw, szw:T - szw] = 1
    pos = np.round(inp['pos']).astype(int)
    eng = A.MultisliceEngine(ctx, (N, N, 1), (T, T), np.repeat(pos - szw, nd, axis=0), C['energy_ev'], C['psize_cm'], free_prop_cm=C['dists_cm'],
                             max_batch=C['minibatch_size'] * nd, unknown_type=inp['unknown_type'], beamstop=window)
    cv = dict(unknown_type=inp['unknown_type'], prj=prj, engine=eng, tile_engine=eng, holo_engine=None, two_d_mode=True,
              safe_zone_width=szw, n_dp_batch=20, sign_convention=1, scale_ri_by_k=True)
    return MultiDistModel(device=ctx, common_vars_dict=cv, raw_data_type='magnitude'), inp


@pytest.mark.parametrize('rn', RUNS)
def test_first_minibatch_vs_reference_and_oracle(A, ctx, rn):
    """Predicted magnitudes (safe zone cut off), loss and object gradient of the reference's first minibatch (tiles 0-3, three
    distances): against the reference's fp64 run under the 3x rule (error <= 3 x the reference's own fp32 error), and against
    the fp64 oracle."""
    f = np.load(F18)
    C = cases.C5TILES
    prj = f[rn + '_prj']
    fm, inp = _model(A, ctx, rn, prj)
    ind = f[rn + '_batches'][0]
    assert list(ind) == [0, 1, 2, 3]
    init = np.stack(_init(inp), -1).astype(np.float32)
    obj = ctx.array(init)
    pc = inp['probe_mag'] * np.exp(1j * inp['probe_phase'])
    pr, pi = pc.real[None].astype(np.float32), pc.imag[None].astype(np.float32)
    args = dict(probe_defocus_mm=0., probe_pos_offset=None, this_i_theta=0, this_pos_batch=inp['pos'][ind], prj=prj, probe_pos_correction=None,
                this_ind_batch=ind, free_prop_cm=np.array(C['dists_cm']), safe_zone_width=inp['szw'], prj_affine_ls=None, ctf_lg_kappa=None,
                prj_pos_offset=None)
    pred = fm.predict(obj, pr, pi, **args)
    p64, p32 = f['first_pred_%s_64' % rn], f['first_pred_%s_32' % rn]
    assert pred.shape == p64.shape
    e, e_ref = np.linalg.norm(pred - p64) / np.linalg.norm(p64), np.linalg.norm(p32 - p64) / np.linalg.norm(p64)
    assert e < max(2e-6, 3 * e_ref), (e, e_ref)
    g = ctx.empty(obj.shape)
    out = fm.loss_and_gradients([0], g, obj, pr, pi, _init_grad=True, **args)
    assert out[0] is g
    loss = fm.current_loss
    l64 = f['losses_%s_64' % rn][0]
    assert abs(loss - l64) < max(2e-5 * abs(l64), 3 * abs(f['losses_%s_32' % rn][0] - l64)), (loss, l64)
    g64, g32 = f['first_grad_%s_64' % rn], f['first_grad_%s_32' % rn]
    e, e_ref = np.linalg.norm(g.get() - g64) / np.linalg.norm(g64), np.linalg.norm(g32 - g64) / np.linalg.norm(g64)
    print('%s: object gradient vs reference fp64 %.2e (reference fp32: %.2e)' % (rn, e, e_ref))
    assert e < max(1e-5, 3 * e_ref), (e, e_ref)
    # the oracle on the same inputs
    full = np.concatenate([ind + i * len(inp['pos']) for i in range(len(C['dists_cm']))])
    ol, op, og = O.multidist_tiles_forward_adjoint(init.astype(np.float64), pc, inp['pos'][ind], (C['SUB'], C['SUB']), inp['szw'], C['dists_cm'],
                                                   prj[0, full].astype(np.float64), C['energy_ev'], C['psize_cm'], unknown_type=inp['unknown_type'])
    assert abs(loss - ol) < 2e-5 * abs(ol)
    assert np.linalg.norm(pred - op) < 2e-6 * np.linalg.norm(op)
    assert np.linalg.norm(g.get() - og) < 2e-5 * np.linalg.norm(og)
    # the loss function alone gives the same value
    val = fm.get_loss_function()(obj, pr, pi, **args)
    assert abs(val - loss) <= 1e-6 * abs(loss)


@pytest.mark.parametrize('rn', RUNS)
def test_driver_vs_reference_driver(A, ctx, rn, tmp_path):
    """reconstruct_ptychography on the tiled data (two epochs of three minibatches of four tiles; the short last minibatch topped
    up as the reference does) against the reference driver's own run: every loss, the final object."""
    f = np.load(F18)
    C = cases.C5TILES
    inp = cases.c5tiles_inputs(rn)
    N = C['N']
    ri = inp['unknown_type'] == 'real_imag'
    pk = dict(probe_type='plane') if inp['probe_type'] == 'plane' else dict(probe_type='supplied', probe_initial=[inp['probe_mag'], inp['probe_phase']])
    st = A.reconstruct_ptychography(
        fname=f[rn + '_prj'], obj_size=(N, N, 1), probe_pos=inp['pos'], theta_st=0, theta_end=0, n_theta=1, two_d_mode=True,
        energy_ev=C['energy_ev'], psize_cm=C['psize_cm'], free_prop_cm=np.array(C['dists_cm']), minibatch_size=C['minibatch_size'],
        n_epochs=C['n_epochs'], initial_guess=[inp['guess'][0], inp['guess'][1]], raw_data_type='magnitude', unknown_type=inp['unknown_type'],
        gamma=0, alpha_d=0, alpha_b=0, optimizer='adam', learning_rate=C['learning_rate'] if ri else 1e-7, n_dp_batch=20,
        randomize_probe_pos=False, safe_zone_width=inp['szw'], save_path=str(tmp_path), output_folder='tiles', store_checkpoint=False,
        use_checkpoint=False, return_state=True, **pk)
    l64, l32 = f['losses_%s_64' % rn], f['losses_%s_32' % rn]
    assert len(st['losses']) == len(l64) == 6
    assert np.all(np.abs(np.array(st['losses']) - l64) <= np.maximum(2e-4 * np.abs(l64), 3 * np.abs(l32 - l64)))
    x = np.stack([st['delta'], st['beta']], -1)
    o64, o32 = f['obj_%s_64' % rn], f['obj_%s_32' % rn]
    upd = np.linalg.norm(o64 - np.stack(_init(inp), -1))
    e, e_ref = np.linalg.norm(x - o64) / upd, np.linalg.norm(o32 - o64) / upd
    print('%s: final object vs reference fp64, relative to the update: %.2e (reference fp32: %.2e)' % (rn, e, e_ref))
    assert e < max(5e-3, 3 * e_ref), (e, e_ref)


def test_undivided_field_with_a_safe_zone_vs_oracle(A, ctx, tmp_path):
    """n_blocks == 1 with safe_zone_width > 0 (forward_model.py:975-978: the whole padded object is one tile): the same engines
    with a single 40 x 40 tile at (-4, -4).  Against the fp64 oracle through the driver."""
    r = cases.rng(1801)
    N, szw = 32, 4
    dists = np.array([30., 55.])
    mag = 1 - 0.2 * cases.smooth_field((N, N, 1), 351)
    ph = 0.4 * cases.smooth_field((N, N, 1), 352)
    truth = np.stack([mag * np.cos(ph), mag * np.sin(ph)], -1)
    _, data, _ = O.multidist_tiles_forward_adjoint(truth, np.ones((N, N), complex), np.zeros((1, 2)), (N, N), szw, dists, np.zeros((2, N, N)),
                                                   17050., 1e-4)
    prj = data[None].astype(np.float32)
    g0 = [np.full((N, N, 1), 0.95), 0.02 * r.standard_normal((N, N, 1))]
    st = A.reconstruct_ptychography(
        fname=prj, obj_size=(N, N, 1), probe_pos=np.array([[0., 0.]]), theta_st=0, theta_end=0, n_theta=1, two_d_mode=True, energy_ev=17050.,
        psize_cm=1e-4, free_prop_cm=dists, minibatch_size=1, n_epochs=3, initial_guess=g0, probe_type='plane', raw_data_type='magnitude',
        unknown_type='real_imag', gamma=0, alpha_d=0, alpha_b=0, optimizer='adam', learning_rate=1e-2, safe_zone_width=szw,
        save_path=str(tmp_path), output_folder='u', store_checkpoint=False, use_checkpoint=False, return_state=True)
    out = O.reconstruct_multidist_tiles(prj.astype(np.float64), [g0[0] * np.cos(g0[1]), g0[0] * np.sin(g0[1])], np.ones((N, N), complex),
                                        np.zeros((1, 2)), (N, N), szw, dists, 17050., 1e-4, n_epochs=3, minibatch_size=1, learning_rate=1e-2)
    assert np.allclose(st['losses'], out['losses'], rtol=2e-4)
    x = np.stack([st['delta'], st['beta']], -1)
    upd = np.linalg.norm(out['obj'] - np.stack([g0[0] * np.cos(g0[1]), g0[0] * np.sin(g0[1])], -1))
    assert np.linalg.norm(x - out['obj']) < 5e-3 * upd


def test_combinations_the_reference_cannot_run_are_refused(A, tmp_path):
    """A minibatch whose last n_dp_batch chunk is a single tile (forward_model.py:931-943 cuts object and probe to different sizes),
    tiles over the edge without a safe zone (:921-925), refinements that have no tiled adjoint here: NotImplementedError before any launch."""
    f = np.load(F18)
    C = cases.C5TILES
    inp = cases.c5tiles_inputs('ri_szw4')
    N = C['N']
    base = dict(fname=f['ri_szw4_prj'], obj_size=(N, N, 1), probe_pos=inp['pos'], theta_st=0, theta_end=0, n_theta=1, two_d_mode=True,
                energy_ev=C['energy_ev'], psize_cm=C['psize_cm'], free_prop_cm=np.array(C['dists_cm']), minibatch_size=4, n_epochs=1,
                unknown_type='real_imag', safe_zone_width=4, save_path=str(tmp_path), output_folder='x', store_checkpoint=False, use_checkpoint=False)
    for extra, what in ((dict(minibatch_size=5, n_dp_batch=4), 'single tile'), (dict(optimize_free_prop=True), 'optimize_free_prop'),
                        (dict(safe_zone_width=0, probe_pos=inp['pos'] - 2.0), 'hanging over'), (dict(safe_zone_width=60), 'larger than 128')):
        kw = dict(base); kw.update(extra)
        with pytest.raises(NotImplementedError, match=what):
            A.reconstruct_ptychography(**kw)


@pytest.mark.regression
def test_probe_windows_through_the_any_size_kernel(A, ctx):
    """One probe per position through ms_generic_kernel (tile sizes outside the tuned set, e.g. 40 = 32 + 2 x 4) against the
    tuned kernel on a size both serve (24): same losses, predictions and tile-gradient sums to rounding."""
    r = cases.rng(1802)
    N, T, B = 40, 24, 5
    pos = np.stack([r.integers(-5, N - 18, B), r.integers(-5, N - 18, B)], 1)
    obj = ctx.array(np.stack([1 + 0.1 * r.standard_normal((N, N, 1)), 0.1 * r.standard_normal((N, N, 1))], -1).astype(np.float32))
    probes = ctx.array(r.standard_normal((B, 1, T, T, 2)).astype(np.float32))
    meas = np.abs(r.standard_normal((B, T, T))).astype(np.float32)
    out = []
    for generic in (False, True):
        eng = A.MultisliceEngine(ctx, (N, N, 1), (T, T), pos, 17050., 1e-4, free_prop_cm=50., max_batch=B, unknown_type='real_imag', generic=generic)
        eng.set_batch(pos, meas)
        eng.rotate(obj, None, None)
        eng.multislice(None, want_pred=True, probes_b=probes)
        g = ctx.zeros(obj.shape)
        eng.rotate_adjoint(g, None, None)
        out.append((eng.loss(), eng.pred(), g.get()))
    assert abs(out[0][0] - out[1][0]) < 1e-5 * abs(out[0][0])
    assert np.linalg.norm(out[0][1] - out[1][1]) < 1e-5 * np.linalg.norm(out[0][1])
    assert np.linalg.norm(out[0][2] - out[1][2]) < 1e-5 * np.linalg.norm(out[0][2]) and np.abs(out[0][2]).max() > 0


@pytest.mark.regression
@pytest.mark.parametrize('T', [24, 40])
def test_one_launch_over_all_distances_equals_one_engine_per_distance(A, ctx, T):
    """adm_plan_set_detector_kernels: a plan with n detector-plane kernels propagates entry b of a launch with kernel b % n.  Against
    n engines with one kernel each on the same tiles: per-entry loss sums and predictions bit for bit, the summed gradient to
    rounding (the overlap-add adds the n contributions of a pixel in another order).  T = 24: tuned kernel; 40: the any-size one."""
    r = cases.rng(1803 + T)
    N, B, dists = 56, 6, (25., 45., 80.)
    nd = len(dists)
    pos = np.stack([r.integers(-4, N - T + 4, B), r.integers(-4, N - T + 4, B)], 1)
    obj = ctx.array(np.stack([1 + 0.1 * r.standard_normal((N, N, 1)), 0.1 * r.standard_normal((N, N, 1))], -1).astype(np.float32))
    probes = r.standard_normal((B, 1, T, T, 2)).astype(np.float32)
    meas = np.abs(r.standard_normal((B, nd, T, T))).astype(np.float32)              # [tile][distance]
    kw = dict(max_batch=B * nd, unknown_type='real_imag')
    stack = A.MultisliceEngine(ctx, (N, N, 1), (T, T), np.repeat(pos, nd, 0), 17050., 1e-4, free_prop_cm=dists, **kw)
    assert stack.n_dists == nd
    stack.set_batch(np.repeat(pos, nd, 0), meas.reshape(B * nd, T, T))
    stack.rotate(obj, None, None)
    stack.multislice(None, want_pred=True, probes_b=ctx.array(np.repeat(probes, nd, 0)))
    g1 = ctx.zeros(obj.shape)
    stack.rotate_adjoint(g1, None, None)
    sums1, pred1 = stack.loss_sums(B * nd).reshape(B, nd), stack.pred().reshape(B, nd, T, T)
    g2 = ctx.zeros(obj.shape)
    pb = ctx.array(probes)
    for i, d in enumerate(dists):
        e = A.MultisliceEngine(ctx, (N, N, 1), (T, T), pos, 17050., 1e-4, free_prop_cm=d, **kw)
        e.set_batch(pos, meas[:, i])
        e.rotate(obj, None, None)
        e.multislice(None, want_pred=True, probes_b=pb, grad_scale=2.0 / (B * nd * e.n_det))
        e.rotate_adjoint(g2, None, None)
        assert np.array_equal(e.loss_sums(B), sums1[:, i])
        assert np.array_equal(e.pred(), pred1[:, i])
    a, b = g1.get(), g2.get()
    assert np.abs(a).max() > 0 and np.abs(a - b).max() <= 2e-6 * np.abs(a).max()
    # a batch that is not a multiple of the number of kernels is refused
    stack.set_batch(np.repeat(pos, nd, 0)[:B * nd - 1], meas.reshape(B * nd, T, T)[:B * nd - 1])
    with pytest.raises(Exception, match='multiple of n'):
        stack.multislice(None, probes_b=ctx.array(np.repeat(probes, nd, 0)[:B * nd - 1]))


# ------------------------------------------------------------------------------------ two ranks (tiles shard like probe positions)
def _w2_worker(rank, world, port, tmp, q, transport):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, 'tests', 'golden'))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), ADM_COMM=transport)
    try:
        import adorym_amd as A
        from adorym_amd import comm as Cm
        import cases as cs
        C = cs.C5TILES
        inp = cs.c5tiles_inputs('ri_szw4')
        N = C['N']
        comm = Cm.from_env()
        assert comm.size == world
        st = A.reconstruct_ptychography(
            comm=comm, fname=np.load(F18)['ri_szw4_prj'], obj_size=(N, N, 1), probe_pos=inp['pos'], theta_st=0, theta_end=0, n_theta=1,
            two_d_mode=True, energy_ev=C['energy_ev'], psize_cm=C['psize_cm'], free_prop_cm=np.array(C['dists_cm']), minibatch_size=2,
            n_epochs=C['n_epochs'], initial_guess=[inp['guess'][0], inp['guess'][1]], probe_type='plane', raw_data_type='magnitude',
            unknown_type='real_imag', gamma=0, alpha_d=0, alpha_b=0, optimizer='adam', learning_rate=C['learning_rate'], n_dp_batch=20,
            randomize_probe_pos=False, safe_zone_width=inp['szw'], save_path=tmp, output_folder='w2', store_checkpoint=False,
            use_checkpoint=False, return_state=True)
        comm.close()
        q.put(dict(rank=rank, obj=np.stack([st['delta'], st['beta']], -1), losses=np.array(st['losses'])))
    except Exception as e:
        import traceback
        q.put(dict(rank=rank, error='%r\n%s' % (e, traceback.format_exc())))


@pytest.mark.parametrize('transport', ['host', 'p2p'])
def test_two_ranks_vs_reference_driver_as_two_processes(tmp_path, transport):
    """The tiled run at world size 2 -- two fresh processes sharing GPU 0, minibatch 2 tiles per rank, gradients summed by the
    host-staged transport or by the direct exchange -- against golden F18_world2 (the reference driver as two processes): both
    ranks' losses, the final object (3x rule), identical replicas."""
    import multiprocessing as mp
    import socket
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    os.environ['ADM_RDV_TOKEN'] = __import__('secrets').token_hex(16)      # this job's secret: other jobs on the machine are not admitted
    mpc = mp.get_context('spawn')
    q = mpc.Queue()
    procs = [mpc.Process(target=_w2_worker, args=(r, 2, port, str(tmp_path), q, transport)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda r: r['rank'])
    [p.join(60) for p in procs]
    for r in res:
        assert 'error' not in r, r['error']
    w2 = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'F18_world2.npz'))
    assert np.array_equal(res[0]['obj'], res[1]['obj'])
    for r in res:
        l64, l32 = w2['r%d_losses_64' % r['rank']], w2['r%d_losses_32' % r['rank']]
        assert np.all(np.abs(r['losses'] - l64) <= np.maximum(2e-4 * np.abs(l64), 3 * np.abs(l32 - l64))), (r['losses'], l64)
    inp = cases.c5tiles_inputs('ri_szw4')
    o64, o32 = w2['obj_64'], w2['obj_32']
    upd = np.linalg.norm(o64 - np.stack(_init(inp), -1))
    e, e_ref = np.linalg.norm(res[0]['obj'] - o64) / upd, np.linalg.norm(o32 - o64) / upd
    print('world 2 (%s): final object vs the reference two-process run, relative to the update: %.2e (reference fp32: %.2e)' % (transport, e, e_ref))
    assert e < max(5e-3, 3 * e_ref), (e, e_ref)


def test_per_angle_update_scheme_adds_the_minibatch_gradients(A, ctx, tmp_path):
    """update_scheme='per angle' with tiled data: the three minibatches of the (one) angle are evaluated at the same object, each
    loss a mean over ITS tiles x distances, their gradients are ADDED and one Adam step follows (adorym/ptychography.py:1063-1066,
    1095-1099) -- restated with the oracle's single-minibatch function; the driver must not fuse them into one averaged call."""
    f = np.load(F18)
    C = cases.C5TILES
    rn = 'ri_szw4'
    inp = cases.c5tiles_inputs(rn)
    N = C['N']
    prj = f[rn + '_prj']
    st = A.reconstruct_ptychography(
        fname=prj, obj_size=(N, N, 1), probe_pos=inp['pos'], theta_st=0, theta_end=0, n_theta=1, two_d_mode=True, energy_ev=C['energy_ev'],
        psize_cm=C['psize_cm'], free_prop_cm=np.array(C['dists_cm']), minibatch_size=C['minibatch_size'], n_epochs=1,
        initial_guess=[inp['guess'][0], inp['guess'][1]], probe_type='plane', raw_data_type='magnitude', unknown_type='real_imag', gamma=0,
        alpha_d=0, alpha_b=0, optimizer='adam', learning_rate=C['learning_rate'], n_dp_batch=20, randomize_probe_pos=False,
        safe_zone_width=inp['szw'], update_scheme='per angle', save_path=str(tmp_path), output_folder='pa', store_checkpoint=False,
        use_checkpoint=False, return_state=True)
    obj0 = np.stack(_init(inp), -1).astype(np.float64)
    batches = O.epoch_task_list(0, 1, len(inp['pos']), C['minibatch_size'], 1, 'per angle', two_d_mode=True)
    g, losses = np.zeros_like(obj0), []
    nd, nb = len(C['dists_cm']), len(inp['pos'])
    for i in range(len(batches)):
        _, ind = O.rank_batch(batches, i, 0, C['minibatch_size'], 1)
        full = np.concatenate([ind + k * nb for k in range(nd)])
        l, _, gi = O.multidist_tiles_forward_adjoint(obj0, np.ones((N, N), complex), inp['pos'][ind], (C['SUB'], C['SUB']), inp['szw'], C['dists_cm'],
                                                     prj[0, full].astype(np.float64), C['energy_ev'], C['psize_cm'])
        g += gi
        losses.append(l)
    want, _, _ = O.adam_step(obj0, g, np.zeros_like(obj0), np.zeros_like(obj0), 0, step_size=C['learning_rate'])
    assert abs(st['losses'][-1] - losses[-1]) < 2e-4 * losses[-1]          # (the log carries the angle's last minibatch)
    x = np.stack([st['delta'], st['beta']], -1)
    upd = np.linalg.norm(want - obj0)
    assert upd > 0 and np.linalg.norm(x - want) < 5e-3 * upd
