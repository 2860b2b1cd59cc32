#!/usr/bin/env python3
"""
Golden-vector generator.  Runs ONLY in the development container: it imports the reference
(/root/reference, PyTorch-CPU backend) behind two I/O shims (h5py, dxchange -- absent here, pure
I/O, no arithmetic) and records the reference's outputs for the inputs defined in cases.py.
Nothing here travels to the GPU box except the resulting *.npz files (data only).

    python tests/golden/gen_goldens.py            # rewrites tests/golden/*.npz
"""
import os
import sys
import types
import tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cases  # noqa: E402

# ----------------------------------------------------------------------------- I/O shims
STORE, TIFFS = {}, {}


class _DS:
    def __init__(s, a): s.a = a
    shape = property(lambda s: s.a.shape)
    def __getitem__(s, k): return s.a[k]
    def __setitem__(s, k, v): s.a[k] = v
    def __len__(s): return len(s.a)
    def __array__(s, dtype=None, copy=None): return np.asarray(s.a, dtype=dtype)


class _File:
    def __init__(s, path, mode='r', **kw):
        if kw:
            raise OSError('no mpio')
        s.d = STORE[os.path.basename(path)]
    def __getitem__(s, k): return _DS(s.d[k])
    def close(s): pass
    def flush(s): pass


h5 = types.ModuleType('h5py'); h5.File = _File
dx = types.ModuleType('dxchange')
dx.write_tiff = lambda data, fname='', dtype=None, overwrite=True: TIFFS.__setitem__(fname, np.array(data))
dx.read_tiff = lambda fname, *a, **k: TIFFS[fname]
sys.modules['h5py'], sys.modules['dxchange'] = h5, dx
sys.path.insert(0, '/root/reference')

import torch  # noqa: E402
import adorym  # noqa: E402
import adorym.global_settings as gs  # noqa: E402
import adorym.wrappers as w  # noqa: E402
from adorym.propagate import get_kernel, multislice_propagate_batch  # noqa: E402
from adorym import util as U  # noqa: E402

gs.backend = 'pytorch'
torch.set_num_threads(4)


def save(name, **arrs):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **arrs)
    print('wrote', name, '%.1f KB' % (os.path.getsize(path) / 1024))


# ----------------------------------------------------------------------------- F1
def gen_f1():
    out = {}
    for P in (12, 64, 72):
        for d in (1., 8.):
            for sg in (1, -1):
                for fa in (True, False):
                    lm = 1240. / cases.ENERGY_EV
                    vox = np.array([cases.PSIZE_CM] * 3) * 1e7
                    H = get_kernel(d, lm, vox, (P, P), fresnel_approx=fa, sign_convention=sg)
                    out['P%d_d%d_s%d_f%d' % (P, int(d), sg, int(fa))] = H
    save('F1_kernel', **out)


# ----------------------------------------------------------------------------- F2 / F3
def ref_forward(tiles, probe, c, fp64):
    gs.run_fp64 = fp64
    dt = torch.float64 if fp64 else torch.float32
    t = torch.tensor(tiles, dtype=dt, requires_grad=True)
    pr = torch.tensor(probe.real.copy(), dtype=dt, requires_grad=True)
    pi = torch.tensor(probe.imag.copy(), dtype=dt, requires_grad=True)
    ex_r, ex_i = multislice_propagate_batch(
        t, pr, pi, cases.ENERGY_EV, cases.PSIZE_CM, kernel=None, free_prop_cm=c['free_prop_cm'],
        binning=c['binning'], fresnel_approx=c['fresnel_approx'], normalize_fft=c['normalize_fft'],
        sign_convention=c['sigma'], type='delta_beta')
    return t, pr, pi, ex_r, ex_i


def ref_case(c, tiles, fp64):
    """pred/loss/gradients exactly as PtychographyModel does it (forward_model.py:337-375, 88-93)."""
    ts, prs, pis, exs = [], [], [], []
    # one leaf tile tensor shared by all modes
    gs.run_fp64 = fp64
    dt = torch.float64 if fp64 else torch.float32
    t = torch.tensor(tiles, dtype=dt, requires_grad=True)
    ex_int = None
    fields = []
    for m in range(c['n_modes']):
        pr = torch.tensor(c['probes'][m].real.copy(), dtype=dt, requires_grad=True)
        pi = torch.tensor(c['probes'][m].imag.copy(), dtype=dt, requires_grad=True)
        ex_r, ex_i = multislice_propagate_batch(
            t, pr, pi, cases.ENERGY_EV, cases.PSIZE_CM, kernel=None, free_prop_cm=c['free_prop_cm'],
            binning=c['binning'], fresnel_approx=c['fresnel_approx'], normalize_fft=c['normalize_fft'],
            sign_convention=c['sigma'], type='delta_beta')
        prs.append(pr); pis.append(pi); fields.append((ex_r, ex_i))
        if c['n_modes'] > 1:
            ex_int = ex_r ** 2 + ex_i ** 2 if ex_int is None else ex_int + ex_r ** 2 + ex_i ** 2
    if c['n_modes'] == 1:
        pred = w.norm(fields[0][0], fields[0][1])
    else:
        pred = w.sqrt(ex_int)
    return t, prs, pis, fields, pred


def gen_f2_f3():
    for name in cases.TILE_CASES:
        c = cases.tile_case_inputs(name)
        # measured data = reference forward of the truth tiles (fp64)
        _, _, _, _, pred_truth = ref_case(c, c['truth'], True)
        meas = pred_truth.detach().numpy().copy()
        out = {'meas': meas}
        for fp64 in (True, False):
            t, prs, pis, fields, pred = ref_case(c, c['guess'], fp64)
            meas_t = torch.tensor(meas, dtype=pred.dtype)
            loss = w.mean((pred - w.abs(meas_t)) ** 2)
            grads = torch.autograd.grad(loss, [t] + prs + pis)
            tag = '64' if fp64 else '32'
            out['ex_real_' + tag] = np.stack([f[0].detach().numpy() for f in fields])
            out['ex_imag_' + tag] = np.stack([f[1].detach().numpy() for f in fields])
            out['pred_' + tag] = pred.detach().numpy()
            out['loss_' + tag] = np.array(loss.item())
            M = c['n_modes']
            gt = grads[0].numpy()
            gpr = np.stack([g.numpy() for g in grads[1:1 + M]])
            gpi = np.stack([g.numpy() for g in grads[1 + M:1 + 2 * M]])
            if c['P'] >= 64:
                # keep the big fixtures small: fp32 storage of the fp64 gradient; for the reference's
                # own fp32 run only its rel-L2 error against its fp64 run is kept (that is all the
                # "<= 3x the reference-fp32 error" criterion needs).
                if fp64:
                    gt64 = gt
                    out['grad_tiles_64'] = gt.astype(np.float32)
                else:
                    out['grad_tiles_relerr_32'] = np.array(np.linalg.norm(gt - gt64) / np.linalg.norm(gt64))
            else:
                out['grad_tiles_' + tag] = gt
            out['grad_probe_real_' + tag] = gpr
            out['grad_probe_imag_' + tag] = gpi
        save('F23_' + name, **out)
    gs.run_fp64 = False


# ----------------------------------------------------------------------------- F4
def gen_f4():
    out = {}
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)
        try:
            for name in cases.ROT_CASES:
                size, theta, obj, cot = cases.rot_case_inputs(name)
                gs.run_fp64 = False
                U.save_rotation_lookup(list(size), np.array([theta], dtype='float32'), dest_folder='rot_' + name)
                coords = U.read_origin_coords('rot_' + name, theta)
                coords_inv = U.read_origin_coords('rot_' + name, theta, reverse=True)
                out[name + '_coords'] = coords
                out[name + '_coords_inv'] = coords_inv
                for fp64 in (True, False):
                    gs.run_fp64 = fp64
                    dt = torch.float64 if fp64 else torch.float32
                    o = torch.tensor(obj, dtype=dt, requires_grad=True)
                    rot = U.apply_rotation(o, coords, device=None)
                    g, = torch.autograd.grad((rot * torch.tensor(cot, dtype=dt)).sum(), [o])
                    tag = '64' if fp64 else '32'
                    out[name + '_rot_' + tag] = rot.detach().numpy()
                    out[name + '_adj_' + tag] = g.numpy()
        finally:
            os.chdir(cwd)
    gs.run_fp64 = False
    # pad lengths for C3-style negative positions (util.py:1374-1406)
    pos = np.array([(y, x) for y in np.arange(23) * 12 - 36 for x in np.arange(23) * 12 - 36])
    out['pad_c3_full'] = U.calculate_pad_len([256, 256, 256], pos, [72, 72])
    out['pad_c3_first32'] = U.calculate_pad_len([256, 256, 256], pos[:32], [72, 72])
    out['pad_c3_last32'] = U.calculate_pad_len([256, 256, 256], pos[-32:], [72, 72])
    save('F4_rotation', **out)


# ----------------------------------------------------------------------------- F5
def gen_f5():
    r = cases.rng(5)
    shape = (6, 6, 6, 2)
    x0 = r.standard_normal(shape) * 1e-3
    gseq = r.standard_normal((4,) + shape)
    out = {'x0': x0, 'gseq': gseq}
    for fp64 in (True, False):
        gs.run_fp64 = fp64
        dt = torch.float64 if fp64 else torch.float32
        tag = '64' if fp64 else '32'
        opt = adorym.AdamOptimizer('obj', options_dict={'step_size': 1e-4})
        opt.create_container(list(shape), False, None)
        x = torch.tensor(x0, dtype=dt)
        xs, ms, vs = [], [], []
        for k, t in enumerate((0, 0, 1, 1)):
            x = opt.apply_gradient(x, torch.tensor(gseq[k], dtype=dt), t, step_size=1e-4)
            xs.append(x.numpy().copy())
            ms.append(opt.params_whole_array_dict['m'].numpy().copy())
            vs.append(opt.params_whole_array_dict['v'].numpy().copy())
        out['adam_x_' + tag] = np.stack(xs); out['adam_m_' + tag] = np.stack(ms); out['adam_v_' + tag] = np.stack(vs)
        gd = adorym.GDOptimizer('obj', options_dict={})
        x = torch.tensor(x0, dtype=dt)
        xs = []
        for k, t in enumerate((0, 25, 70, 200)):
            x = gd.apply_gradient(x, torch.tensor(gseq[k], dtype=dt), t, step_size=1e-2, dynamic_rate=True,
                                  first_downrate_iteration=20)
            xs.append(x.numpy().copy())
        out['gd_x_' + tag] = np.stack(xs)
    gs.run_fp64 = False
    save('F5_optimizers', **out)


# ----------------------------------------------------------------------------- F7
def gen_f7():
    r = cases.rng(7)
    obj = r.standard_normal((8, 8, 8, 2)) * 1e-3
    out = {'obj': obj}
    gs.run_fp64 = True
    o = torch.tensor(obj, dtype=torch.float64, requires_grad=True)
    l1 = adorym.L1Regularizer(alpha_d=1.5, alpha_b=0.7).get_value(o)
    out['l1_val'] = np.array(l1.item()); out['l1_grad'] = torch.autograd.grad(l1, [o])[0].numpy()
    tv = adorym.TVRegularizer(gamma=2.0).get_value(o)
    out['tv_val'] = np.array(tv.item()); out['tv_grad'] = torch.autograd.grad(tv, [o])[0].numpy()
    gs.run_fp64 = False
    save('F7_regularizers', **out)


# ----------------------------------------------------------------------------- F6 / F8 (driver)
def run_driver(prj, obj_size, probe_pos, theta_end, n_theta, extra, record, ri=False):
    """Run the reference driver in a scratch cwd; record per-minibatch (i_theta, ind) / loss / first grad."""
    import adorym.ptychography as PT
    import adorym.differentiator as DF
    STORE['data.h5'] = {'exchange/data': prj}
    orig_get = DF.Differentiator.get_gradients
    orig_split = PT.split_tasks

    def rec_split(arr, n):
        res = orig_split(arr, n)
        record.setdefault('task_lists', []).append([np.array(x) for x in res])
        return res

    def rec_get(self, **kw):
        g = orig_get(self, **kw)
        record.setdefault('batches', []).append((int(kw['this_i_theta']), np.array(kw['this_ind_batch'])))
        if 'first_grad' not in record:
            record['first_grad'] = g[0].detach().numpy().copy()
        return g

    DF.Differentiator.get_gradients = rec_get
    PT.split_tasks = rec_split
    cwd = os.getcwd()
    TIFFS.clear()
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)
        try:
            params = dict(fname='data.h5', obj_size=obj_size, probe_pos=probe_pos, theta_st=0, theta_end=theta_end,
                          n_theta=n_theta, energy_ev=cases.ENERGY_EV, psize_cm=cases.PSIZE_CM, free_prop_cm='inf',
                          save_path='.', output_folder='out', use_checkpoint=False, store_checkpoint=False,
                          save_intermediate=False, cpu_only=True, backend='pytorch', gamma=0, alpha_d=0, alpha_b=0,
                          n_dp_batch=20, shared_probe_among_angles=True)
            params.update(extra)
            PT.reconstruct_ptychography(**params)
            with open(os.path.join('out', 'convergence', 'loss_rank_0.txt')) as f:
                lines = f.read().strip().split('\n')[1:]
            record['losses'] = np.array([float(l.split(',')[2]) for l in lines])
            if ri:
                record['mag'] = TIFFS[[k for k in TIFFS if k.endswith('obj_mag_ds_1')][0]].copy()
                record['phase'] = TIFFS[[k for k in TIFFS if k.endswith('obj_phase_ds_1')][0]].copy()
            else:
                key_d = [k for k in TIFFS if k.endswith('delta_ds_1')]
                key_b = [k for k in TIFFS if k.endswith('beta_ds_1')]
                record['delta'] = TIFFS[key_d[0]].copy()
                record['beta'] = TIFFS[key_b[0]].copy()
        finally:
            os.chdir(cwd)
            DF.Differentiator.get_gradients = orig_get
            PT.split_tasks = orig_split
    return record


def gen_f6():
    sys.path.insert(0, os.path.join(HERE, '..', '..'))
    from oracle import adorym_oracle as O
    inp = cases.e2e_inputs()
    E = cases.E2E
    N, P = E['N'], E['P']
    phys = O.Physics((P, P), E['energy_ev'], E['psize_cm'], free_prop_cm='inf')
    probe = inp['probe_mag'] * np.exp(1j * inp['probe_phase'])
    truth = np.stack(inp['truth'], -1)
    # synthetic measurement from the build's own fp64 oracle forward (no reference data files exist)
    prj = np.zeros((E['n_theta'], len(inp['probe_pos']), P, P))
    for it, th in enumerate(inp['theta_ls']):
        coords = O.rotation_coords((N, N, N), th)
        rot = O.rotate_fwd(truth, coords, 'float64')
        tiles, _ = O.extract_tiles(rot, inp['probe_pos'], (P, P))
        prj[it] = np.abs(O.multislice_forward(tiles, probe, phys, 'float64'))
    out = {'prj': prj.astype(np.float32)}
    prj = out['prj'].astype(np.float64)
    common = dict(minibatch_size=E['minibatch_size'], initial_guess=[inp['guess'][0], inp['guess'][1]],
                  probe_type='supplied', probe_initial=[inp['probe_mag'], inp['probe_phase']])
    runs = {
        'adam_e1':      dict(n_epochs=1, optimizer='adam', learning_rate=1e-6),
        'adam_e2':      dict(n_epochs=2, optimizer='adam', learning_rate=1e-6),
        'gd_e1':        dict(n_epochs=1, optimizer='gd', learning_rate=1e-9),
        'adam_e1_reg':  dict(n_epochs=1, optimizer='adam', learning_rate=1e-6, gamma=1e-6, alpha_d=1e-4, alpha_b=1e-5),
        'adam_e1_perangle': dict(n_epochs=1, optimizer='adam', learning_rate=1e-6, update_scheme='per angle'),
        'adam_e1_nonneg': dict(n_epochs=1, optimizer='adam', learning_rate=1e-5, non_negativity=True),
    }
    for rn, extra in runs.items():
        for fp64 in (True, False):
            rec = {}
            ex = dict(common); ex.update(extra); ex['run_float64'] = fp64
            run_driver(prj, [N, N, N], inp['probe_pos'], 2 * np.pi, E['n_theta'], ex, rec)
            tag = rn + ('_64' if fp64 else '_32')
            keep64 = fp64 and rn == 'adam_e1'     # one run kept in full fp64 for a tight oracle check
            out['delta_' + tag] = rec['delta'].astype(np.float64 if keep64 else np.float32)
            out['beta_' + tag] = rec['beta'].astype(np.float64 if keep64 else np.float32)
            out['losses_' + tag] = rec['losses']
            if rn in ('adam_e1', 'adam_e1_reg'):
                out['first_grad_' + tag] = rec['first_grad'].astype(np.float64 if keep64 else np.float32)
            if rn == 'adam_e2' and fp64:
                out['batches_theta'] = np.array([b[0] for b in rec['batches']])
                out['batches_ind'] = np.stack([b[1] for b in rec['batches']])
    gs.run_fp64 = False
    save('F6_e2e', **out)


def gen_f8():
    import adorym.pseudo as PS
    out = {}
    n_theta, n_pos, mb, P, N = 5, 7, 3, 4, 8
    prj = np.abs(cases.rng(8).standard_normal((n_theta, n_pos, P, P))) + 0.5
    pos = np.array([(i % 3, i // 3) for i in range(n_pos)], dtype=float)
    g0 = [np.full((N, N, N), 1e-6), np.full((N, N, N), 1e-7)]
    orig_size, orig_rank = PS.Comm.Get_size, PS.Comm.Get_rank
    try:
        for n_ranks in (1, 2):
            for rank in (0,):   # the fake comm's identity bcast only supports rank 0; the global task list is rank-independent
                PS.Comm.Get_size = lambda self, n=n_ranks: n
                PS.Comm.Get_rank = lambda self, r=rank: r
                rec = {}
                run_driver(prj, [N, N, N], pos, 2 * np.pi, n_theta,
                           dict(minibatch_size=mb, n_epochs=2, initial_guess=g0, probe_type='gaussian',
                                probe_mag_sigma=2., probe_phase_sigma=2., probe_phase_max=0.5,
                                optimizer='adam', learning_rate=1e-9), rec)
                for e, tl in enumerate(rec['task_lists']):
                    for k, b in enumerate(tl):
                        out['r%d_e%d_task_%d' % (n_ranks, e, k)] = b
                    out['r%d_e%d_ntask' % (n_ranks, e)] = np.array(len(tl))
                out['r%d_rank%d_theta' % (n_ranks, rank)] = np.array([b[0] for b in rec['batches']])
                out['r%d_rank%d_ind' % (n_ranks, rank)] = np.stack([b[1] for b in rec['batches']])
    finally:
        PS.Comm.Get_size, PS.Comm.Get_rank = orig_size, orig_rank
    save('F8_tasks', **out)


# ----------------------------------------------------------------------------- F9 (next-row f4 variants)
def gen_f9():
    """Poisson loss (forward_model.py:94-102), Momentum optimizer (optimizers.py:366-411), reweighted L1
    (regularizers.py:49-84 + weight update ptychography.py:995-1000)."""
    out = {}
    name = 'p12_s9_far_pos'
    c = cases.tile_case_inputs(name)
    meas = np.load(os.path.join(HERE, 'F23_' + name + '.npz'))['meas']
    for rdt in ('magnitude', 'intensity'):
        for pm in (1.0, 50.0):
            for fp64 in (True, False):
                t, prs, pis, fields, pred = ref_case(c, c['guess'], fp64)
                fm = adorym.ForwardModel(loss_function_type='poisson', raw_data_type=rdt)
                fm.poisson_multiplier = pm
                m_t = torch.tensor(meas if rdt == 'magnitude' else meas ** 2, dtype=pred.dtype)
                loss = fm.get_mismatch_loss(pred, m_t)
                g = torch.autograd.grad(loss, [t] + prs + pis)
                tag = '%s_pm%d_%s' % (rdt, int(pm), '64' if fp64 else '32')
                out['poisson_loss_' + tag] = np.array(loss.item())
                out['poisson_grad_tiles_' + tag] = g[0].numpy()
                out['poisson_grad_probe_real_' + tag] = g[1].numpy()
                out['poisson_grad_probe_imag_' + tag] = g[2].numpy()
    gs.run_fp64 = False
    # momentum
    r = cases.rng(9)
    shape = (5, 4, 3, 2)
    x0 = r.standard_normal(shape) * 1e-3
    gseq = r.standard_normal((3,) + shape)
    out['mom_x0'] = x0; out['mom_gseq'] = gseq
    for fp64 in (True, False):
        gs.run_fp64 = fp64
        dt = torch.float64 if fp64 else torch.float32
        opt = adorym.MomentumOptimizer('obj', options_dict={})
        opt.create_container(list(shape), False, None)
        x = torch.tensor(x0, dtype=dt)
        xs = []
        for k in range(3):
            x = opt.apply_gradient(x, torch.tensor(gseq[k], dtype=dt), k, step_size=1e-3, gamma=0.9)
            xs.append(x.numpy().copy())
        out['mom_x_' + ('64' if fp64 else '32')] = np.stack(xs)
        out['mom_v_' + ('64' if fp64 else '32')] = opt.params_whole_array_dict['v'].numpy().copy()
    # reweighted L1
    gs.run_fp64 = True
    obj = np.abs(r.standard_normal((6, 6, 6, 2))) * 1e-3 + 1e-5
    o = torch.tensor(obj, dtype=torch.float64, requires_grad=True)
    with torch.no_grad():
        wgt = w.max(o) / (w.abs(o) + 1e-4 * w.mean(o))
    reg = adorym.ReweightedL1Regularizer(alpha_d=0.8, alpha_b=0.3)
    reg.update_l1_weight(wgt)
    val = reg.get_value(o)
    out['rwl1_obj'] = obj
    out['rwl1_weight'] = wgt.numpy()
    out['rwl1_val'] = np.array(val.item())
    out['rwl1_grad'] = torch.autograd.grad(val, [o])[0].numpy()
    gs.run_fp64 = False
    save('F9_variants', **out)


# ----------------------------------------------------------------------------- F10 (unknown_type='real_imag')
def gen_f10():
    """multislice_propagate_batch(type='real_imag') (propagate.py:243-249): slices ARE the complex transmission."""
    out = {}
    name = 'p12_s9_far_pos'
    c = cases.tile_case_inputs(name)
    r = cases.rng(10)
    shape = c['guess'].shape[:-1]
    mag = 1.0 - 0.2 * r.uniform(size=shape)
    ph = 0.8 * r.uniform(-1, 1, size=shape)
    tiles = np.stack([mag * np.cos(ph), mag * np.sin(ph)], -1)
    out['tiles'] = tiles
    for fp in ('inf', 0):
        meas = None
        for fp64 in (True, False):
            gs.run_fp64 = fp64
            dt = torch.float64 if fp64 else torch.float32
            t = torch.tensor(tiles, dtype=dt, requires_grad=True)
            pr = torch.tensor(c['probes'][0].real.copy(), dtype=dt, requires_grad=True)
            pi = torch.tensor(c['probes'][0].imag.copy(), dtype=dt, requires_grad=True)
            ex_r, ex_i = multislice_propagate_batch(t, pr, pi, cases.ENERGY_EV, cases.PSIZE_CM, kernel=None, free_prop_cm=fp,
                                                    binning=1, type='real_imag')
            pred = w.norm(ex_r, ex_i)
            if meas is None:
                meas = (pred.detach().numpy() * (1 + 0.3 * cases.rng(11).uniform(-1, 1, size=pred.shape))).copy()
                out['meas_%s' % fp] = meas
            loss = w.mean((pred - torch.tensor(meas, dtype=dt)) ** 2)
            g = torch.autograd.grad(loss, [t, pr, pi])
            tag = '%s_%s' % (fp, '64' if fp64 else '32')
            out['pred_' + tag] = pred.detach().numpy()
            out['loss_' + tag] = np.array(loss.item())
            out['grad_tiles_' + tag] = g[0].numpy()
            out['grad_probe_real_' + tag] = g[1].numpy()
            out['grad_probe_imag_' + tag] = g[2].numpy()
    gs.run_fp64 = False
    # padding semantics of pad_object for real_imag (util.py:1338-1350): real part padded with 1, imaginary with 0
    o = torch.tensor(cases.rng(12).standard_normal((5, 6, 2, 2)))
    padded, pad_arr = U.pad_object(o, [5, 6, 2], np.array([[-2, -1], [3, 4]]), [4, 4], unknown_type='real_imag')
    out['pad_in'] = o.numpy(); out['pad_out'] = padded.numpy(); out['pad_arr'] = pad_arr
    save('F10_real_imag', **out)

# ----------------------------------------------------------------------------- F11 (f2 row: sub-pixel positions, real_imag regularisers, probe init)
def _shifted_pred(tiles, pr, pi, shifts, kind, free_prop_cm, energy, psize):
    """The probe-shift branch of PtychographyModel.predict (forward_model.py:296-375) on explicit tiles."""
    B = tiles.shape[0]
    prl, pil = [], []
    for j in range(B):
        a, b = U.realign_image_fourier(pr, pi, shifts[j], axes=(1, 2), device=None)
        prl.append(a); pil.append(b)
    prl, pil = w.stack(prl), w.stack(pil)                       # [B, M, y, x]
    ex_int = None
    for m in range(pr.shape[0]):
        er, ei = multislice_propagate_batch(tiles, prl[:, m], pil[:, m], energy, psize, kernel=None, free_prop_cm=free_prop_cm,
                                            binning=1, type=kind)
        ex_int = er ** 2 + ei ** 2 if ex_int is None else ex_int + er ** 2 + ei ** 2
    return w.sqrt(ex_int), prl, pil


def gen_f11():
    out = {}
    C = cases.C1MINI
    r = cases.rng(1100)
    # (a) realign_image_fourier (util.py:380-397) on a stack of modes
    P, M = 16, 2
    pr0 = r.standard_normal((M, P, P)); pi0 = r.standard_normal((M, P, P))
    out['shift_probe'] = pr0 + 1j * pi0
    for k, sh in enumerate([(0.37, -1.21), (-2.5, 0.0), (0.0, 0.0)]):
        for fp64 in (True, False):
            gs.run_fp64 = fp64
            dt = torch.float64 if fp64 else torch.float32
            a, b = U.realign_image_fourier(torch.tensor(pr0, dtype=dt), torch.tensor(pi0, dtype=dt), torch.tensor(sh, dtype=dt), axes=(1, 2))
            out['shift%d_%s' % (k, '64' if fp64 else '32')] = a.numpy() + 1j * b.numpy()
        out['shift%d_s' % k] = np.array(sh)
    # (b) gradients w.r.t. the per-position shifts, the probe modes and the tiles
    for name, kind, S, fp in (('db_s5_far', 'delta_beta', 5, 'inf'), ('ri_s1_far', 'real_imag', 1, 'inf'), ('ri_s3_near', 'real_imag', 3, 0)):
        B, P = 3, 12
        rr = cases.rng(cases.hash_name('f11' + name))
        if kind == 'delta_beta':
            tiles = np.stack([rr.uniform(0, 2e-3, (B, P, P, S)), rr.uniform(0, 2e-4, (B, P, P, S))], -1)
        else:
            mag = 1 - 0.3 * rr.uniform(size=(B, P, P, S)); ph = 0.7 * rr.uniform(-1, 1, (B, P, P, S))
            tiles = np.stack([mag * np.cos(ph), mag * np.sin(ph)], -1)
        probe = rr.standard_normal((M, P, P)) + 1j * rr.standard_normal((M, P, P))
        probe[1] *= 0.4
        shifts = rr.uniform(-0.8, 0.8, (B, 2))
        out[name + '_tiles'] = tiles; out[name + '_probe'] = probe; out[name + '_shifts'] = shifts
        meas = None
        for fp64 in (True, False):
            gs.run_fp64 = fp64
            dt = torch.float64 if fp64 else torch.float32
            t = torch.tensor(tiles, dtype=dt, requires_grad=True)
            pr = torch.tensor(probe.real.copy(), dtype=dt, requires_grad=True)
            pi = torch.tensor(probe.imag.copy(), dtype=dt, requires_grad=True)
            sh = torch.tensor(shifts, dtype=dt, requires_grad=True)
            pred, _, _ = _shifted_pred(t, pr, pi, sh, kind, fp, cases.ENERGY_EV, cases.PSIZE_CM)
            if meas is None:
                meas = (pred.detach().numpy() * (1 + 0.3 * cases.rng(1101).uniform(-1, 1, size=pred.shape))).copy()
                out[name + '_meas'] = meas
            loss = w.mean((pred - torch.tensor(meas, dtype=dt)) ** 2)
            g = torch.autograd.grad(loss, [t, pr, pi, sh])
            tag = name + ('_64' if fp64 else '_32')
            out[tag + '_pred'] = pred.detach().numpy(); out[tag + '_loss'] = np.array(loss.item())
            out[tag + '_grad_tiles'] = g[0].numpy(); out[tag + '_grad_probe'] = g[1].numpy() + 1j * g[2].numpy()
            out[tag + '_grad_shifts'] = g[3].numpy()
    # (c) real_imag regularisers (regularizers.py:38-45, 105-110)
    gs.run_fp64 = True
    mag = 1 - 0.3 * r.uniform(size=(6, 7, 3)); ph = 2.5 * r.uniform(-1, 1, (6, 7, 3))
    obj = np.stack([mag * np.cos(ph), mag * np.sin(ph)], -1)
    out['reg_obj'] = obj
    o = torch.tensor(obj, dtype=torch.float64, requires_grad=True)
    for nm, reg in (('tv', adorym.TVRegularizer(0.7, unknown_type='real_imag')), ('l1', adorym.L1Regularizer(0.8, 0.3, unknown_type='real_imag'))):
        val = reg.get_value(o)
        out['reg_%s_val' % nm] = np.array(val.item())
        out['reg_%s_grad' % nm] = torch.autograd.grad(val, [o])[0].numpy()
    gs.run_fp64 = False
    # (d) initialize_probe: aperture_defocus (+ beamstop), rescale_intensity against a data file (util.py:205-219, 254-281)
    Pp = 16
    dat = np.abs(cases.rng(1102).standard_normal((1, 6, Pp, Pp))) * 30
    STORE['probe_data.h5'] = {'exchange/data': dat}
    out['pinit_data'] = dat
    lm = 1240. / C['energy_ev']
    # The reference evaluates the aperture_defocus branch with override_backend='autograd' (NumPy fp64); that package is
    # absent here, so the same reference pieces (generate_disk, get_kernel, convolve_with_transfer_function) are driven
    # through the PyTorch backend in fp64, and the rescaling is pinned through the 'supplied' branch of initialize_probe.
    gs.run_fp64 = True
    for k, (ar, br, dcm, sg) in enumerate(((5, 2, 0.0069, 1), (6, 0, 0.004, 1), (5, 2, 0.0069, -1))):
        mag = U.generate_disk([Pp, Pp], ar)
        if br > 0:
            mag = mag * (1 - U.generate_disk([Pp, Pp], br))
        hk = get_kernel(dcm * 1e7, lm, [C['psize_cm'] * 1e7] * 3, [Pp, Pp], sign_convention=sg)
        a, b = w.convolve_with_transfer_function(torch.tensor(mag, dtype=torch.float64), torch.tensor(np.zeros_like(mag), dtype=torch.float64),
                                                 torch.tensor(np.real(hk)), torch.tensor(np.imag(hk)))
        out['pinit_ad%d' % k] = a.numpy() + 1j * b.numpy()
        out['pinit_ad%d_args' % k] = np.array([ar, br, dcm, sg], dtype=float)
    gs.run_fp64 = False
    pm_ = np.abs(cases.rng(1103).standard_normal((3, Pp, Pp))); pp_ = cases.rng(1104).uniform(-3, 3, (3, Pp, Pp))
    out['pinit_sup_mag'] = pm_; out['pinit_sup_phase'] = pp_
    for k, (rdt, nf, sg, nm) in enumerate((('intensity', False, 1, 3), ('magnitude', True, 1, 2), ('intensity', False, -1, 1))):
        a, b = U.initialize_probe([Pp, Pp], 'supplied', probe_initial=[pm_[:nm] if nm > 1 else pm_[0], pp_[:nm] if nm > 1 else pp_[0]],
                                  rescale_intensity=True, save_path='.', fname='probe_data.h5',
                                  raw_data_type=rdt, stdout_options={'save_stdout': False, 'output_folder': '.', 'timestamp': ''},
                                  sign_convention=sg, normalize_fft=nf, n_probe_modes=nm)
        out['pinit_rescale%d' % k] = a + 1j * b
    # (e) end-to-end driver, config-1 shape
    import adorym.ptychography as PT
    inp = cases.c1mini_inputs()
    Y, X, P, M = C['Y'], C['X'], C['P'], C['M']
    gs.run_fp64 = True
    truth = np.stack([inp['truth'][0] * np.cos(inp['truth'][1]), inp['truth'][0] * np.sin(inp['truth'][1])], -1)
    pos_t = inp['pos_true']; pos_i = np.round(pos_t).astype(int)
    o = torch.tensor(truth, dtype=torch.float64)
    op, pad = U.pad_object(o, [Y, X, 1], pos_i, [P, P], unknown_type='real_imag')
    tiles = torch.stack([op[y + pad[0, 0]:y + pad[0, 0] + P, x + pad[1, 0]:x + pad[1, 0] + P] for y, x in pos_i])
    ptrue = inp['probe_true'][0] * np.exp(1j * inp['probe_true'][1])
    pred, _, _ = _shifted_pred(tiles, torch.tensor(ptrue.real.copy()), torch.tensor(ptrue.imag.copy()),
                               torch.tensor(pos_t - pos_i), 'real_imag', 'inf', C['energy_ev'], C['psize_cm'])
    prj = (pred.numpy() ** 2)[None].astype(np.float32)           # intensity data
    out['e2e_prj'] = prj
    gs.run_fp64 = False
    orig_up = PT.update_parameters
    for fp64 in (True, False):
        rec = {}
        def rec_up(opt_ls, optimizable_params, kw, _rec=rec):
            res = orig_up(opt_ls, optimizable_params, kw)
            _rec['pos_corr'] = res['probe_pos_correction'].detach().numpy().copy()
            _rec['probe'] = res['probe_real'].detach().numpy().copy() + 1j * res['probe_imag'].detach().numpy().copy()
            return res
        PT.update_parameters = rec_up
        try:
            run_driver(prj.astype(np.float64), [Y, X, 1], inp['pos_nominal'], 0, 1,
                       dict(minibatch_size=C['minibatch_size'], n_epochs=2, two_d_mode=True, energy_ev=C['energy_ev'], psize_cm=C['psize_cm'],
                            initial_guess=[inp['guess'][0], inp['guess'][1]], probe_type='supplied',
                            probe_initial=[inp['probe_guess'][0], inp['probe_guess'][1]], n_probe_modes=M, rescale_probe_intensity=True,
                            raw_data_type='intensity', optimize_probe=True, probe_learning_rate=1e-3, optimize_all_probe_pos=True,
                            all_probe_pos_learning_rate=1e-2, unknown_type='real_imag', gamma=1e-6, alpha_d=None, alpha_b=None,
                            optimizer='adam', learning_rate=1e-3, n_dp_batch=C['n_dp_batch'], run_float64=fp64,
                            random_guess_means_sigmas=(1., 0., 0.001, 0.002)), rec, ri=True)
        finally:
            PT.update_parameters = orig_up
        tag = '_64' if fp64 else '_32'
        out['e2e_obj' + tag] = np.stack([rec['mag'] * np.cos(rec['phase']), rec['mag'] * np.sin(rec['phase'])], -1)
        out['e2e_losses' + tag] = rec['losses']
        out['e2e_pos_corr' + tag] = rec['pos_corr']
        out['e2e_probe' + tag] = rec['probe']
        out['e2e_first_grad' + tag] = rec['first_grad']
    gs.run_fp64 = False
    save('F11_c1', **out)

# ----------------------------------------------------------------------------- F12 (f1 row: multi-distance holography)
def _multidist_chain(o, pr, pi, dists_cm, affine, data, energy, psize_cm, raw='intensity'):
    """MultiDistModel.predict + get_loss_function (forward_model.py:819-1092) for one undivided tile (n_blocks = 1):
    S = 1 modulation, fresnel_propagate_wrapped to every distance, loss against the affine-registered data."""
    N = o.shape[0]
    lm = 1240. / energy
    vox = np.array([psize_cm * 1e7] * 3)
    from adorym.propagate import gen_freq_mesh, fresnel_propagate_wrapped
    u, v = gen_freq_mesh(vox, [N, N])
    u = torch.tensor(u, dtype=o.dtype); v = torch.tensor(v, dtype=o.dtype)
    preds = []
    for i in range(len(dists_cm)):
        er, ei = multislice_propagate_batch(o[None], pr, pi, energy, psize_cm, kernel=None, free_prop_cm=dists_cm[i],
                                            obj_batch_shape=[1, N, N, 1], type='real_imag', optimize_free_prop=True, u_free=u, v_free=v)
        preds.append(w.sqrt(er ** 2 + ei ** 2))
    pred = w.concatenate(preds, 0)
    tgt = w.concatenate([w.affine_transform(data[i:i + 1], affine[i]) for i in range(len(dists_cm))])
    fm = adorym.ForwardModel(loss_function_type='lsq', raw_data_type=raw)
    return fm.get_mismatch_loss(pred, tgt), pred, tgt


def gen_f12():
    out = {}
    C = cases.C5MINI
    inp = cases.c5mini_inputs()
    N = C['N']
    gs.run_fp64 = True
    truth = np.stack([inp['truth'][0] * np.cos(inp['truth'][1]), inp['truth'][0] * np.sin(inp['truth'][1])], -1)
    ident = np.tile(np.array([[1., 0, 0], [0, 1., 0]]), [3, 1, 1])
    one = torch.ones((N, N), dtype=torch.float64); zero = torch.zeros((N, N), dtype=torch.float64)
    # data: |forward(truth)|^2 at the true distances, then de-registered with the inverse of affine_true
    _, pred_t, _ = _multidist_chain(torch.tensor(truth), one, zero, torch.tensor(C['dists_cm'], dtype=torch.float64),
                                    torch.tensor(ident), torch.zeros((3, N, N), dtype=torch.float64), C['energy_ev'], C['psize_cm'])
    inten = pred_t.detach() ** 2
    inv = []
    for a in inp['affine_true']:
        m = np.vstack([a, [0, 0, 1]]); inv.append(np.linalg.inv(m)[:2])
    data = torch.cat([w.affine_transform(inten[i:i + 1], torch.tensor(inv[i])) for i in range(3)]).numpy()
    out['data'] = data.astype(np.float32)
    data = out['data'].astype(np.float64)
    # (a) affine_transform (wrappers.py:1158-1174) forward
    out['affine_in'] = data[1]
    out['affine_theta'] = inp['affine_true'][1]
    out['affine_out'] = w.affine_transform(torch.tensor(data[1:2]), torch.tensor(inp['affine_true'][1])).numpy()[0]
    # (b) loss + gradients w.r.t. object, probe, distances, affine matrices
    g0 = inp['guess'][0] * np.exp(1j * inp['guess'][1])
    guess = np.stack([g0.real, g0.imag], -1)
    out['guess'] = guess
    aff_guess = inp['affine_true'].copy(); aff_guess[1:] += 0.01 * cases.rng(1200).uniform(-1, 1, (2, 2, 3))
    out['aff_guess'] = aff_guess
    pr0 = 1 + 0.1 * cases.rng(1201).uniform(-1, 1, (N, N)); pi0 = 0.1 * cases.rng(1202).uniform(-1, 1, (N, N))
    out['probe'] = pr0 + 1j * pi0
    for fp64 in (True, False):
        gs.run_fp64 = fp64
        dt = torch.float64 if fp64 else torch.float32
        o = torch.tensor(guess, dtype=dt, requires_grad=True)
        pr = torch.tensor(pr0, dtype=dt, requires_grad=True); pi = torch.tensor(pi0, dtype=dt, requires_grad=True)
        d = torch.tensor(inp['dists_guess'], dtype=dt, requires_grad=True)
        a = torch.tensor(aff_guess, dtype=dt, requires_grad=True)
        loss, pred, tgt = _multidist_chain(o, pr, pi, d, a, torch.tensor(data, dtype=dt), C['energy_ev'], C['psize_cm'])
        g = torch.autograd.grad(loss, [o, pr, pi, d, a])
        tag = '_64' if fp64 else '_32'
        out['loss' + tag] = np.array(loss.item()); out['pred' + tag] = pred.detach().numpy(); out['target' + tag] = tgt.detach().numpy()
        out['grad_obj' + tag] = g[0].numpy(); out['grad_probe' + tag] = g[1].numpy() + 1j * g[2].numpy()
        out['grad_dists' + tag] = g[3].numpy(); out['grad_affine' + tag] = g[4].numpy()
    gs.run_fp64 = False
    # (c) the reference driver end to end (config-5 shape): distances and affine registration optimised with the object
    import adorym.ptychography as PT
    orig_up = PT.update_parameters
    prj = out['data'][None].astype(np.float64)                 # [1, n_dists, N, N]

    class MultiDistPlugin(adorym.MultiDistModel):
        # the reference's 'auto' selection passes run_bfloat16/run_float64 to MultiDistModel.__init__, which does not accept
        # them (TypeError at ptychography.py:535); a forward_model= plugin that drops the two keywords is the way to run it
        def __init__(self, *a, run_bfloat16=False, run_float64=False, **k):
            super().__init__(*a, **k)

    for fp64 in (True, False):
        rec = {}
        def rec_up(opt_ls, optimizable_params, kw, _rec=rec):
            res = orig_up(opt_ls, optimizable_params, kw)
            _rec['dists'] = res['free_prop_cm'].detach().numpy().copy()
            _rec['affine'] = res['prj_affine_ls'].detach().numpy().copy()
            return res
        PT.update_parameters = rec_up
        try:
            run_driver(prj, [N, N, 1], np.array([[0., 0.]]), 0, 1,
                       dict(minibatch_size=1, n_epochs=4, two_d_mode=True, energy_ev=C['energy_ev'], psize_cm=C['psize_cm'],
                            free_prop_cm=np.array(inp['dists_guess']), initial_guess=[inp['guess'][0], inp['guess'][1]],
                            probe_type='plane', raw_data_type='intensity', unknown_type='real_imag', gamma=0, alpha_d=0, alpha_b=0,
                            optimizer='adam', learning_rate=1e-2, optimize_free_prop=True, free_prop_learning_rate=1e-1,
                            optimize_prj_affine=True, prj_affine_learning_rate=1e-3, n_dp_batch=1, run_float64=fp64,
                            randomize_probe_pos=True, safe_zone_width=0, forward_model=MultiDistPlugin), rec, ri=True)
        finally:
            PT.update_parameters = orig_up
        tag = '_64' if fp64 else '_32'
        out['e2e_obj' + tag] = np.stack([rec['mag'] * np.cos(rec['phase']), rec['mag'] * np.sin(rec['phase'])], -1)
        out['e2e_losses' + tag] = rec['losses']
        out['e2e_dists' + tag] = rec['dists']
        out['e2e_affine' + tag] = rec['affine']
        out['e2e_first_grad' + tag] = rec['first_grad']
    gs.run_fp64 = False
    save('F12_multidist', **out)

# ----------------------------------------------------------------------------- F13 (beamstop mask, forward_model.py:128-136)
def gen_f13():
    out = {}
    name = 'p12_s9_far_pos'
    c = cases.tile_case_inputs(name)
    meas = np.load(os.path.join(HERE, 'F23_' + name + '.npz'))['meas']
    P = meas.shape[-1]
    r = cases.rng(1300)
    bs = r.uniform(0, 1, (P, P))
    bs[bs < 0.35] = 0.0                      # dropped pixels
    bs[3, 4] = 5e-6                          # below the 1e-5 threshold: dropped as well
    out['beamstop'] = bs
    for fp64 in (True, False):
        t, prs, pis, fields, pred = ref_case(c, c['guess'], fp64)
        fm = adorym.ForwardModel(loss_function_type='lsq', raw_data_type='magnitude')
        fm.common_vars = {'beamstop': torch.tensor(bs.copy(), dtype=pred.dtype)}
        loss = fm.loss(pred, torch.tensor(meas, dtype=pred.dtype), t)
        g = torch.autograd.grad(loss, [t] + prs + pis)
        tag = '64' if fp64 else '32'
        out['loss_' + tag] = np.array(loss.item())
        out['grad_tiles_' + tag] = g[0].numpy()
        out['grad_probe_' + tag] = g[1].numpy() + 1j * g[2].numpy()
    gs.run_fp64 = False
    save('F13_beamstop', **out)


# ----------------------------------------------------------------------------- F17 (config-3 depth: the reference's own fp32 error)
def gen_f17():
    """P = 72, 256 slices, far field (config 3's depth) on a small lateral object: the REFERENCE in fp64 and in fp32 on the same
    inputs (cases.depth256_inputs).  Stored: the fp64 prediction and loss, a strided sample of the fp64 object gradient, and the
    reference's own fp32-vs-fp64 errors (prediction, loss, gradient) -- the yardstick of the 3x rule at this depth."""
    d = cases.depth256_inputs()
    P, S = d['P'], d['S']
    Y, X = d['obj'].shape[:2]
    out = {}
    res = {}
    for fp64 in (True, False):
        gs.run_fp64 = fp64
        dt = torch.float64 if fp64 else torch.float32
        o = torch.tensor(d['obj'], dtype=dt, requires_grad=True)
        op, pad = U.pad_object(o, [Y, X, S], d['pos'], [P, P], unknown_type='delta_beta')
        tiles = torch.stack([op[y + pad[0, 0]:y + pad[0, 0] + P, x + pad[1, 0]:x + pad[1, 0] + P] for y, x in d['pos']])
        er, ei = multislice_propagate_batch(tiles, torch.tensor(d['probe'].real.copy(), dtype=dt), torch.tensor(d['probe'].imag.copy(), dtype=dt),
                                            cases.ENERGY_EV, cases.PSIZE_CM, kernel=None, free_prop_cm='inf', obj_batch_shape=[len(d['pos']), P, P, S])
        pred = w.norm(er, ei)
        if fp64:
            truth = torch.tensor(d['truth'], dtype=dt)
            tp, tpad = U.pad_object(truth, [Y, X, S], d['pos'], [P, P], unknown_type='delta_beta')
            tt = torch.stack([tp[y + tpad[0, 0]:y + tpad[0, 0] + P, x + tpad[1, 0]:x + tpad[1, 0] + P] for y, x in d['pos']])
            tr, ti = multislice_propagate_batch(tt, torch.tensor(d['probe'].real.copy(), dtype=dt), torch.tensor(d['probe'].imag.copy(), dtype=dt),
                                                cases.ENERGY_EV, cases.PSIZE_CM, kernel=None, free_prop_cm='inf', obj_batch_shape=[len(d['pos']), P, P, S])
            out['target'] = w.norm(tr, ti).detach().numpy()
        loss = w.mean((pred - torch.tensor(out['target'], dtype=dt)) ** 2)
        g, = torch.autograd.grad(loss, [o])
        res[fp64] = (pred.detach().numpy().astype(np.float64), float(loss.item()), g.numpy().astype(np.float64))
    gs.run_fp64 = False
    p64, l64, g64 = res[True]
    p32, l32, g32 = res[False]
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
    out.update(pred_64=p64, loss_64=np.array(l64), grad_64_sample=g64[::4, ::4, ::4].copy(), grad_64_norm=np.array(np.linalg.norm(g64)),
               ref32_pred_err=np.array(rel(p32, p64)), ref32_loss_err=np.array(abs(l32 - l64) / abs(l64)), ref32_grad_err=np.array(rel(g32, g64)),
               ref32_grad_sample_err=np.array(rel(g32[::4, ::4, ::4], g64[::4, ::4, ::4])))
    print('reference fp32 vs fp64 at depth 256: pred %.2e, loss %.2e, grad %.2e' % (out['ref32_pred_err'], out['ref32_loss_err'], out['ref32_grad_err']))
    save('F17_depth256', **out)


# ----------------------------------------------------------------------------- F16 (reweighted L1, unknown_type='real_imag')
def gen_f16():
    """ReweightedL1Regularizer with unknown_type='real_imag' (adorym/regularizers.py:73-82) and the weight update of the DP
    branch (adorym/ptychography.py:995-1000)."""
    out = {}
    r = cases.rng(16)
    shape = (6, 7, 5)
    mag = 1 - 0.3 * r.uniform(size=shape)
    ph = 0.6 * r.uniform(size=shape) - 0.25
    obj = np.stack([mag * np.cos(ph), mag * np.sin(ph)], -1)
    out['obj'] = obj
    for fp64 in (True, False):
        gs.run_fp64 = fp64
        dt = torch.float64 if fp64 else torch.float32
        o = torch.tensor(obj, dtype=dt, requires_grad=True)
        with torch.no_grad():
            wgt = w.max(o) / (w.abs(o) + 1e-4 * w.mean(o))
        reg = adorym.ReweightedL1Regularizer(alpha_d=0.8, alpha_b=0.3, unknown_type='real_imag')
        reg.update_l1_weight(wgt)
        val = reg.get_value(o)
        tag = '64' if fp64 else '32'
        out['weight_' + tag] = wgt.numpy()
        out['val_' + tag] = np.array(val.item())
        out['grad_' + tag] = torch.autograd.grad(val, [o])[0].numpy()
    gs.run_fp64 = False
    save('F16_rwl1_real_imag', **out)


# ----------------------------------------------------------------------------- F15 (rotate_out_of_loop, DP mode)
def gen_f15():
    """rotate_out_of_loop=True through the reference driver (adorym/ptychography.py:917-947, 1011, 1063-1078): the object is
    rotated outside the differentiated block once per angle, the gradient buffer is resampled back with the -theta table."""
    g6 = np.load(os.path.join(HERE, 'F6_e2e.npz'))
    prj = g6['prj'].astype(np.float64)
    inp = cases.e2e_inputs()
    E = cases.E2E
    N = E['N']
    common = dict(minibatch_size=E['minibatch_size'], initial_guess=[inp['guess'][0], inp['guess'][1]],
                  probe_type='supplied', probe_initial=[inp['probe_mag'], inp['probe_phase']], rotate_out_of_loop=True)
    out = {}
    for rn, extra in cases.ROOL_RUNS.items():
        for fp64 in (True, False):
            rec = {}
            ex = dict(common); ex.update(extra); ex['run_float64'] = fp64
            run_driver(prj, [N, N, N], inp['probe_pos'], 2 * np.pi, E['n_theta'], ex, rec)
            tag = rn + ('_64' if fp64 else '_32')
            out['delta_' + tag] = rec['delta'].astype(np.float64 if fp64 else np.float32)
            out['beta_' + tag] = rec['beta'].astype(np.float64 if fp64 else np.float32)
            out['losses_' + tag] = rec['losses']
    gs.run_fp64 = False
    save('F15_rotate_out_of_loop', **out)


# ----------------------------------------------------------------------------- F18 (f1 row, data divided into sub-tiles + safe zone)
def gen_f18():
    """MultiDistModel with n_blocks > 1 and safe_zone_width >= 0 (forward_model.py:884-1034) through the reference DRIVER: the
    holograms of a full-field propagation are cut into 3 x 3 tiles; recorded per run: the task list, the first minibatch's
    predicted magnitudes and object gradient, every loss, the final object."""
    import adorym.ptychography as PT  # noqa: F401
    C = cases.C5TILES
    N, SUB = C['N'], C['SUB']
    dists = np.array(C['dists_cm'])
    out = {}

    class MultiDistPlugin(adorym.MultiDistModel):
        # (see gen_f12: the 'auto' selection passes two keywords MultiDistModel.__init__ does not accept)
        first_pred = None

        def __init__(self, *a, run_bfloat16=False, run_float64=False, **k):
            super().__init__(*a, **k)          # (reads predict's argument names: the recorder goes on afterwards)
            inner = self.predict

            def recording_predict(*pa, **pk):
                res = inner(*pa, **pk)
                if MultiDistPlugin.first_pred is None:
                    MultiDistPlugin.first_pred = res.detach().numpy().copy()
                return res
            self.predict = recording_predict

    for rn in C['runs']:
        inp = cases.c5tiles_inputs(rn)
        ut, szw = inp['unknown_type'], inp['szw']
        # data: the truth lit by the run's probe, propagated as ONE field to every distance (fp64), cut into tiles
        gs.run_fp64 = True
        pc = inp['probe_mag'] * np.exp(1j * inp['probe_phase'])
        holo = []
        for d in dists:
            er, ei = multislice_propagate_batch(torch.tensor(inp['truth'])[None], torch.tensor(pc.real), torch.tensor(pc.imag),
                                                C['energy_ev'], C['psize_cm'], kernel=None, free_prop_cm=d,
                                                obj_batch_shape=[1, N, N, 1], type=ut)
            holo.append(np.sqrt(er.numpy()[0] ** 2 + ei.numpy()[0] ** 2))
        pos = inp['pos']
        prj = np.zeros((1, len(dists) * len(pos), SUB, SUB))
        for i in range(len(dists)):
            for j, (y, x) in enumerate(pos.astype(int)):
                prj[0, i * len(pos) + j] = holo[i][y:y + SUB, x:x + SUB]
        out[rn + '_prj'] = prj.astype(np.float32)
        prj = out[rn + '_prj'].astype(np.float64)
        pk = dict(probe_type='plane') if inp['probe_type'] == 'plane' else \
            dict(probe_type='supplied', probe_initial=[inp['probe_mag'], inp['probe_phase']])
        for fp64 in (True, False):
            rec = {}
            MultiDistPlugin.first_pred = None
            run_driver(prj, [N, N, 1], pos, 0, 1,
                       dict(minibatch_size=C['minibatch_size'], n_epochs=C['n_epochs'], two_d_mode=True, energy_ev=C['energy_ev'],
                            psize_cm=C['psize_cm'], free_prop_cm=dists, initial_guess=[inp['guess'][0], inp['guess'][1]],
                            raw_data_type='magnitude', unknown_type=ut, gamma=0, alpha_d=0, alpha_b=0, optimizer='adam',
                            learning_rate=C['learning_rate'] if ut == 'real_imag' else 1e-7, n_dp_batch=20, run_float64=fp64,
                            randomize_probe_pos=False, safe_zone_width=szw, forward_model=MultiDistPlugin, **pk), rec,
                       ri=(ut == 'real_imag'))
            tag = rn + ('_64' if fp64 else '_32')
            if ut == 'real_imag':
                out['obj_' + tag] = np.stack([rec['mag'] * np.cos(rec['phase']), rec['mag'] * np.sin(rec['phase'])], -1)
            else:
                out['obj_' + tag] = np.stack([rec['delta'], rec['beta']], -1)
            out['losses_' + tag] = rec['losses']
            out['first_grad_' + tag] = rec['first_grad']
            out['first_pred_' + tag] = MultiDistPlugin.first_pred
            if fp64:
                out[rn + '_batches'] = np.stack([b[1] for b in rec['batches']])
    gs.run_fp64 = False
    save('F18_multidist_tiles', **out)


# ----------------------------------------------------------------------------- F19 (f1 row: per-distance shift refinement)
def _multidist_chain_shift(o, dists_cm, shifts, data, energy, psize_cm, raw='intensity'):
    """MultiDistModel.predict + get_loss_function with optimize_all_probe_pos (forward_model.py:1075-1085): plane probe, every
    measured hologram Fourier-shifted by its own (sy, sx) (realign_image_fourier, real part kept) before the loss."""
    N = o.shape[0]
    one = torch.ones((N, N), dtype=o.dtype); zero = torch.zeros((N, N), dtype=o.dtype)
    preds = []
    for i in range(len(dists_cm)):
        er, ei = multislice_propagate_batch(o[None], one, zero, energy, psize_cm, kernel=None, free_prop_cm=float(dists_cm[i]),
                                            obj_batch_shape=[1, N, N, 1], type='real_imag')
        preds.append(w.sqrt(er ** 2 + ei ** 2))
    pred = w.concatenate(preds, 0)
    tg = []
    for i in range(len(dists_cm)):
        t, _ = U.realign_image_fourier(data[i:i + 1], w.zeros_like(data[i:i + 1]), shifts[i], axes=(1, 2))
        tg.append(t)
    tgt = w.concatenate(tg)
    fm = adorym.ForwardModel(loss_function_type='lsq', raw_data_type=raw)
    return fm.get_mismatch_loss(pred, tgt), pred, tgt


def gen_f19():
    """demos/2d_multidist_holography_w_position_correction.py at a small size: the holograms of distances 1 and 2 are recorded
    shifted by a fraction of a pixel; `optimize_all_probe_pos` refines one (sy, sx) per distance with the object."""
    out = {}
    C = cases.C5MINI
    inp = cases.c5mini_inputs()
    N = C['N']
    nd = len(C['dists_cm'])
    gs.run_fp64 = True
    truth = np.stack([inp['truth'][0] * np.cos(inp['truth'][1]), inp['truth'][0] * np.sin(inp['truth'][1])], -1)
    shift_true = np.array([[0., 0.], [0.6, -0.45], [-0.35, 0.8]])
    d64 = torch.tensor(C['dists_cm'], dtype=torch.float64)
    _, pred_t, _ = _multidist_chain_shift(torch.tensor(truth), d64, torch.zeros((nd, 2), dtype=torch.float64),
                                          torch.zeros((nd, N, N), dtype=torch.float64), C['energy_ev'], C['psize_cm'])
    inten = pred_t.detach() ** 2
    # recorded holograms = the true ones displaced by -shift_true (so that +shift_true registers them)
    data = torch.cat([U.realign_image_fourier(inten[i:i + 1], torch.zeros_like(inten[i:i + 1]), torch.tensor(-shift_true[i]), axes=(1, 2))[0]
                      for i in range(nd)]).numpy()
    out['data'] = data.astype(np.float32)
    out['shift_true'] = shift_true
    data = out['data'].astype(np.float64)
    g0 = inp['guess'][0] * np.exp(1j * inp['guess'][1])
    guess = np.stack([g0.real, g0.imag], -1)
    out['guess'] = guess
    shift_guess = shift_true + 0.2 * cases.rng(1900).uniform(-1, 1, (nd, 2))
    out['shift_guess'] = shift_guess
    for fp64 in (True, False):
        gs.run_fp64 = fp64
        dt = torch.float64 if fp64 else torch.float32
        o = torch.tensor(guess, dtype=dt, requires_grad=True)
        sh = torch.tensor(shift_guess, dtype=dt, requires_grad=True)
        loss, pred, tgt = _multidist_chain_shift(o, torch.tensor(C['dists_cm'], dtype=dt), sh, torch.tensor(data, dtype=dt), C['energy_ev'], C['psize_cm'])
        g = torch.autograd.grad(loss, [o, sh])
        tag = '_64' if fp64 else '_32'
        out['loss' + tag] = np.array(loss.item()); out['pred' + tag] = pred.detach().numpy(); out['target' + tag] = tgt.detach().numpy()
        out['grad_obj' + tag] = g[0].numpy(); out['grad_shifts' + tag] = g[1].numpy()
    gs.run_fp64 = False
    # the reference driver end to end
    import adorym.ptychography as PT
    orig_up = PT.update_parameters
    prj = out['data'][None].astype(np.float64)

    class MultiDistPlugin(adorym.MultiDistModel):
        def __init__(self, *a, run_bfloat16=False, run_float64=False, **k):
            super().__init__(*a, **k)

    for fp64 in (True, False):
        rec = {}
        def rec_up(opt_ls, optimizable_params, kw, _rec=rec):
            res = orig_up(opt_ls, optimizable_params, kw)
            _rec['shifts'] = res['probe_pos_correction'].detach().numpy().copy()
            _rec.setdefault('shift_trace', []).append(_rec['shifts'])
            return res
        PT.update_parameters = rec_up
        try:
            run_driver(prj, [N, N, 1], np.array([[0., 0.]]), 0, 1,
                       dict(minibatch_size=1, n_epochs=5, two_d_mode=True, energy_ev=C['energy_ev'], psize_cm=C['psize_cm'],
                            free_prop_cm=np.array(C['dists_cm']), initial_guess=[inp['guess'][0], inp['guess'][1]],
                            probe_type='plane', raw_data_type='intensity', unknown_type='real_imag', gamma=0, alpha_d=0, alpha_b=0,
                            optimizer='adam', learning_rate=1e-2, optimize_all_probe_pos=True, all_probe_pos_learning_rate=1e-1,
                            n_dp_batch=1, run_float64=fp64, randomize_probe_pos=True, safe_zone_width=0, forward_model=MultiDistPlugin),
                       rec, ri=True)
        finally:
            PT.update_parameters = orig_up
        tag = '_64' if fp64 else '_32'
        out['e2e_obj' + tag] = np.stack([rec['mag'] * np.cos(rec['phase']), rec['mag'] * np.sin(rec['phase'])], -1)
        out['e2e_losses' + tag] = rec['losses']
        out['e2e_shifts' + tag] = rec['shifts']
        out['e2e_shift_trace' + tag] = np.stack(rec['shift_trace'])
        out['e2e_first_grad' + tag] = rec['first_grad']
    gs.run_fp64 = False
    save('F19_multidist_shifts', **out)


# ----------------------------------------------------------------------------- F20 (probe_type='ifft', util.py:225-236, 300-333)
def gen_f20():
    """The probe estimated from the measured data (demos/2d_ptychography_w_probe_optimization.py: probe_type='ifft'):
    create_probe_initial_guess_ptycho for both raw data types and both sign conventions, and through initialize_probe with
    the intensity rescaling on."""
    r = cases.rng(2000)
    data = (np.abs(r.standard_normal((2, 5, 16, 12))) * (1 + 4 * np.exp(-((np.arange(16)[:, None] - 8) ** 2 + (np.arange(12)[None, :] - 6) ** 2) / 8.))).astype(np.float32)
    STORE['ifft.h5'] = {'exchange/data': data}
    out = {'data': data}
    for raw in ('intensity', 'magnitude'):
        for sc in (1, -1):
            out['guess_%s_%d' % (raw, sc)] = U.create_probe_initial_guess_ptycho('ifft.h5', raw_data_type=raw, sign_convention=sc)
    pr, pi = U.initialize_probe([16, 12], 'ifft', save_path='.', fname='ifft.h5', raw_data_type='intensity', rescale_intensity=True,
                                n_probe_modes=1, normalize_fft=False, sign_convention=1, stdout_options={})
    out['init_rescaled'] = pr + 1j * pi
    save('F20_probe_ifft', **out)


if __name__ == '__main__':
    which = sys.argv[1:] or ['f15', 'f16', 'f17', 'f1', 'f23', 'f4', 'f5', 'f6', 'f7', 'f8', 'f9', 'f10', 'f11', 'f12', 'f13', 'f18', 'f19', 'f20']
    if 'f1' in which: gen_f1()
    if 'f23' in which: gen_f2_f3()
    if 'f4' in which: gen_f4()
    if 'f5' in which: gen_f5()
    if 'f7' in which: gen_f7()
    if 'f8' in which: gen_f8()
    if 'f6' in which: gen_f6()
    if 'f9' in which: gen_f9()
    if 'f10' in which: gen_f10()
    if 'f11' in which: gen_f11()
    if 'f12' in which: gen_f12()
    if 'f13' in which: gen_f13()
    if 'f15' in which: gen_f15()
    if 'f16' in which: gen_f16()
    if 'f17' in which: gen_f17()
    if 'f18' in which: gen_f18()
    if 'f19' in which: gen_f19()
    if 'f20' in which: gen_f20()
