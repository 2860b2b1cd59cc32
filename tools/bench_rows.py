#!/usr/bin/env python3
"""
Measurement of the "next" rows at the shapes BASELINE.json names for them, beside the
pinned NumPy oracle on the host.  Not the headline metric (bench.py is); prints one JSON object per row.

    python tools/bench_rows.py            # on a GPU box
C2 shape: full-field multislice tomography, 64^3 object, plane probe 64 x 64, 64 slices, near field (free_prop_cm = 0), L1,
          Adam; BASELINE's "minibatch 16" = 16 virtual ranks x minibatch 1: 16 angles, gradients summed, one update
          (tests/test_multislice_tomography_64.py:20-65 of the reference)
C1 shape: 2-D ptychography, 618 x 606 x 1 real_imag object, 5 incoherent probe modes (P = 64), minibatch 35, intensity
          data, Adam on object + probe + sub-pixel probe positions, TV regulariser  (demos/2d_ptychography_experimental_data.py)
C5 shape: multi-distance holography, 512 x 512 x 1 real_imag object, plane probe, 4 distances, Adam on object +
          distances + affine registration  (demos/2d_multidist_holography_w_affine.py)
tiles:    the same kind of data divided into sub-tiles with a safe zone (adorym/forward_model.py:884-1034), 480 x 480, 4 distances,
          100 holograms of 48 x 48 per distance, safe zone 12, minibatch 50 tiles
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import adorym_amd as A                      # noqa: E402
from adorym_amd._lib import check           # noqa: E402
from adorym_amd.optimizers import AdamOptimizer, apply_small_params      # noqa: E402
import bench                                # noqa: E402  (its cpu_baseline legs own the oracle timing)


def bench_c2(ctx, steps=30):
    from adorym_amd import workloads as W
    from adorym_amd.dp import DataParallelObject, HipOps
    from adorym_amd.comm import LocalComm
    cfg = W.c2_config()
    N, R = cfg['obj_size'][0], 16
    r = np.random.default_rng(2)
    thetas = np.linspace(cfg['theta_st'], cfg['theta_end'], cfg['n_theta'], dtype='float32')[3:3 + R]
    guess = W.random_guess((N, N, N), seed=0)
    probe_h = np.ones((N, N), complex)
    probe = ctx.array(np.stack([probe_h.real, probe_h.imag], -1)[None].astype(np.float32))
    data_h = (1 + 0.05 * np.abs(r.standard_normal((R, N, N)))).astype(np.float32)       # throughput is data independent
    data = ctx.array(data_h)
    tables = [A.RotationTable(ctx, (N, N, N), th) for th in thetas]
    a_d, a_b = cfg['alpha_d'], cfg['alpha_b']
    lr = cfg['learning_rate']
    alg = R * 4 * (3 * N * N * N * 2 + N * N + 2 * 2 * N ** 3)         # 10.5 MB per angle
    out = {}
    # ---- (a) the 16 ranks' evaluations one after the other (what one GPU does with the reference's control flow) ----
    eng = A.MultisliceEngine(ctx, (N, N, N), (N, N), cfg['probe_pos'], cfg['energy_ev'], cfg['psize_cm'], free_prop_cm=0, max_batch=1,
                             transmissions_only=True)
    st = DataParallelObject(HipOps(ctx), LocalComm(), (N, N, N, 2))
    obj, grad = st.obj.view(0, (N, N, N, 2)), st.grad.view(0, (N, N, N, 2))
    obj.set(guess)
    for t in tables:
        t.csr(eng.plan)
    ev = [ctx.event() for _ in range(2)]

    def step_seq(k, timed):
        ms = 0.0
        check(ctx.lib.adm_reg_grad_set(eng.plan.handle, obj.ptr, a_d * R, a_b * R, 0.0, grad.ptr, None))      # every rank adds the L1 term
        for i, t in enumerate(tables):
            eng.set_batch(cfg['probe_pos'], data.view(i * N * N, (1, N, N)))
            eng.rotate(obj, t, None)
            if timed:
                ev[0].record()
            eng.multislice(probe, accumulate=False)
            if timed:
                ev[1].record()
            eng.accumulate_tiles()
            eng.rotate_adjoint(grad, t, None)
            if timed:
                ms += ev[0].elapsed_ms(ev[1])           # (blocks: the sequential leg is timed per launch, separately from the loop below)
        st.exchange_and_update('adam', k, {'step_size': lr})
        return ms

    for k in range(2):
        step_seq(k, False)
    kern_seq = np.mean([step_seq(k, True) for k in range(3)]) / R
    ctx.sync()
    t0 = time.perf_counter()
    for k in range(steps):
        step_seq(k, False)
    ctx.sync()
    dt_seq = (time.perf_counter() - t0) / steps
    g_seq = None
    obj.set(guess)
    check(ctx.lib.adm_reg_grad_set(eng.plan.handle, obj.ptr, a_d * R, a_b * R, 0.0, grad.ptr, None))
    for i, t in enumerate(tables):
        eng.set_batch(cfg['probe_pos'], data.view(i * N * N, (1, N, N)))
        eng.rotate(obj, t, None)
        eng.multislice(probe)
        eng.rotate_adjoint(grad, t, None)
    g_seq = grad.get()
    # ---- (b) the 16 angles in ONE launch (adorym_amd.AngleBatch): 16 workgroups instead of 16 launches of one ----
    ab = A.AngleBatch(ctx, (N, N, N), (N, N), R, cfg['energy_ev'], cfg['psize_cm'], free_prop_cm=0, transmissions_only=True)
    e2 = ab.engine
    for t in tables:
        t.csr(e2.plan)
    Yb = N

    def step_stack(k, timed):
        check(ctx.lib.adm_reg_grad_set(eng.plan.handle, obj.ptr, a_d * R, a_b * R, 0.0, grad.ptr, None))
        e2.set_batch(ab.pos, data)
        ab.rotate_all(obj, tables)
        if timed:
            ev[0].record()
        e2.multislice(probe, grad_scale=2.0 / e2.n_det, accumulate=False)
        if timed:
            ev[1].record()
        e2.accumulate_tiles()
        ab.rotate_adjoint_all(grad, tables)
        st.exchange_and_update('adam', k, {'step_size': lr})
        return ev[0].elapsed_ms(ev[1]) if timed else 0.0

    obj.set(guess)
    check(ctx.lib.adm_reg_grad_set(eng.plan.handle, obj.ptr, a_d * R, a_b * R, 0.0, grad.ptr, None))
    g_loss = ab.loss_and_grad(obj, grad, tables, probe, data)
    g_stack = grad.get()
    same = float(np.linalg.norm(g_stack - g_seq) / np.linalg.norm(g_seq))
    for k in range(2):
        step_stack(k, False)
    kern_stack = float(np.mean([step_stack(k, True) for k in range(5)]))
    ctx.sync()
    t0 = time.perf_counter()
    for k in range(steps):
        step_stack(k, False)
    ctx.sync()
    dt_stack = (time.perf_counter() - t0) / steps
    done, t_cpu = bench.cpu_baseline_c2(guess, thetas, probe_h, data_h, cfg['energy_ev'], cfg['psize_cm'])
    roof = lambda ms, n_ang: {'bound': 'hbm', 'kernel': 'ms_fwd_adj_kernel<64,8,8>', 'achieved': n_ang * alg / R / (ms * 1e-3) / 1e9,
                              'peak': bench.PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': n_ang * alg / R / (ms * 1e-3) / 1e9 / bench.PEAK_HBM_GBS,
                              'kernel_ms': ms, 'algorithmic_bytes_per_launch': n_ang * alg // R, 'traffic': None}
    return {'row': 'config-2 shape', 'workload': cfg['name'] + ', 16 virtual ranks x minibatch 1 (16 angles per update, gradients summed), L1, Adam',
            'value': R / dt_stack, 'unit': 'angles/s (= probe-positions/s: one 64x64 full-field position per angle)', 'ms_per_step': 1e3 * dt_stack,
            'dtype': 'f32', 'roofline': dict(roof(kern_stack, R), whole_step_frac=alg / dt_stack / 1e9 / bench.PEAK_HBM_GBS),
            'one_launch_per_angle': {'value': R / dt_seq, 'ms_per_step': 1e3 * dt_seq, 'roofline': dict(roof(kern_seq, 1), whole_step_frac=alg / dt_seq / 1e9 / bench.PEAK_HBM_GBS)},
            'stacked_vs_sequential_gradient_rel_l2': same, 'loss_first_angle': float(g_loss[0]),
            'cpu_baseline': {'value': done / t_cpu, 'unit': 'angles/s', 'cores': 1, 'kind': 'port',
                             'sample': '%d whole angles (rotation + 64-slice chain fwd + hand adjoint + back-rotation + L1) in %.1f s, oracle fp32, one core' % (done, t_cpu)}}


def bench_c1(ctx, steps=50):
    Y, X, P, M, B = 618, 606, 64, 5, 35
    r = np.random.default_rng(0)
    energy, psize = 8801.121930115722, 1.32789376566526e-06
    pos = np.array([(y, x) for y in range(-20, Y - 40, 16) for x in range(-20, X - 40, 16)], dtype=float)
    pos += r.uniform(-0.4, 0.4, pos.shape)
    pos_int = np.round(pos).astype(int)
    eng = A.MultisliceEngine(ctx, (Y, X, 1), (P, P), pos_int, energy, psize, free_prop_cm='inf', n_probe_modes=M, max_batch=B,
                             unknown_type='real_imag')
    obj_h = np.stack([r.normal(1, 1e-3, (Y, X, 1)), r.normal(0, 2e-3, (Y, X, 1))], -1).astype(np.float32)
    obj = ctx.array(obj_h)
    m, v, g = ctx.zeros(obj.shape), ctx.zeros(obj.shape), ctx.zeros(obj.shape)
    probe_h = (r.standard_normal((M, P, P)) + 1j * r.standard_normal((M, P, P)))
    probe = ctx.array(np.stack([probe_h.real, probe_h.imag], -1).astype(np.float32))
    gp = ctx.zeros(probe.shape)
    corr = ctx.array((pos - pos_int)[None].astype(np.float32))
    gc = ctx.zeros(corr.shape)
    meas = ctx.array((np.abs(r.standard_normal((B, P, P))) * 50).astype(np.float32))
    idx_all = ctx.array(np.arange(len(pos_int), dtype=np.int32))
    lib = ctx.lib
    # the driver's optimiser objects and its ONE launch for the small parameters (adorym_amd/ptychography.py)
    o_probe = AdamOptimizer('probe', options_dict={'step_size': 1e-3}); o_probe.create_param_arrays(list(probe.shape), device=ctx)
    o_pos = AdamOptimizer('probe_pos_correction', options_dict={'step_size': 1e-2}); o_pos.create_param_arrays(list(corr.shape), device=ctx)
    o_obj = AdamOptimizer('obj', options_dict={'step_size': 1e-3}); o_obj.params_whole_array_dict = {'m': m, 'v': v}
    obj_flat, g_flat = obj.view(0, (obj.size,)), g.view(0, (g.size,))
    ring = ctx.uploader()
    pending = [None]
    ev = [ctx.event(), ctx.event()]
    timed = [False]

    def step(k):
        s = (k * B) % (len(pos_int) - B)
        eng.flush_loss_copy()               # the previous minibatch's loss read-back goes first: its event precedes this minibatch's work
        eng.set_batch(pos_int[s:s + B], meas)
        idx = idx_all.view(s, (B,))         # a run of consecutive positions: a view of the resident index array, like the driver
        eng.rotate(obj, None, None)
        check(lib.adm_reg_grad_set(eng.plan.handle, obj.ptr, 0., 0., 1e-6, g.ptr, None))      # initialises the gradient buffer (no zero fill)
        if timed[0]:
            ev[0].record()
        eng.multislice(probe, grad_probe=gp, shifts=corr, shift_index=idx, grad_shifts=gc)
        if timed[0]:
            ev[1].record()
        eng.rotate_adjoint(g, None, None)
        # (as the driver does for a small unconstrained object on one rank: the object is one more array of the one launch)
        apply_small_params(ctx, [dict(opt=o_obj, x=obj_flat, g=g_flat), dict(opt=o_probe, x=probe, g=gp, zero_grad=True),
                                 dict(opt=o_pos, x=corr, g=gc, center_cols=2, zero_grad=True)], 0)
        tok = eng.loss_async()
        out = eng.loss_result(pending[0]) if pending[0] is not None else None     # the PREVIOUS minibatch's loss: no pipeline drain
        pending[0] = tok
        return out

    for k in range(3):
        step(k)
    ctx.sync()
    t0 = time.perf_counter()
    for k in range(steps):
        step(3 + k)
    t_issue = (time.perf_counter() - t0) / steps
    ctx.sync()
    dt = (time.perf_counter() - t0) / steps
    timed[0] = True
    ks = []
    for k in range(10):
        step(3 + steps + k)
        ks.append(ev[0].elapsed_ms(ev[1]))
    timed[0] = False
    kern = float(np.median(ks))
    # SURVEY 8(d)'s byte count at this shape: tile data read fwd + read bwd + gradient written (S = 1), measured intensities, the
    # object read once and its gradient written once; plus the M probe modes read and their gradient written
    V = Y * X
    alg = 4 * (3 * B * P * P * 1 * 2 + B * P * P + 2 * (2 * V)) + 2 * 4 * (M * P * P * 2)
    tc = bench.cpu_baseline_c1(obj_h, pos, pos_int, probe_h, B, P, energy, psize)
    return {'row': 'f2 / config-1 shape', 'dtype': 'f32',
            'roofline': {'bound': 'latency (a chain of 10 launches of 4-60 us; 35 workgroups on 256 CUs in the dominant one) -- priced against hbm',
                         'kernel': 'probe_shift + ms_fwd_adj_kernel<64,8,8,...,MULTI> + probe_shift_adj (the forward+adjoint launch group, HIP events)',
                         'achieved': alg / (kern * 1e-3) / 1e9, 'peak': bench.PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': alg / (kern * 1e-3) / 1e9 / bench.PEAK_HBM_GBS,
                         'kernel_ms': kern, 'algorithmic_bytes_per_launch': alg, 'whole_step_frac': alg / dt / 1e9 / bench.PEAK_HBM_GBS, 'traffic': None}, 'workload': '2-D ptychography 618x606x1 real_imag, P=64, 5 modes, minibatch 35, object+probe+position Adam, TV',
            'value': B / dt, 'unit': 'probe-positions/s', 'ms_per_step': 1e3 * dt, 'host_issue_ms_per_step': 1e3 * t_issue,
            'cpu_baseline': {'value': B / tc, 'unit': 'probe-positions/s', 'cores': 1, 'kind': 'port',
                             'sample': 'one minibatch fwd+adjoint incl. probe shifts, oracle fp32 (optimiser and regulariser excluded)'}}


def bench_c5(ctx, steps=50):
    N, nd = 512, 4
    r = np.random.default_rng(1)
    energy, psize = 17050., 1e-4
    eng = A.HolographyEngine(ctx, (N, N), nd, energy, psize)
    obj_h = np.stack([r.normal(1, 0., (N, N, 1)), r.normal(0, 0.01, (N, N, 1))], -1).astype(np.float32)
    obj = ctx.array(obj_h)
    m, v, g = ctx.zeros(obj.shape), ctx.zeros(obj.shape), ctx.zeros(obj.shape)
    probe = ctx.array(np.stack([np.ones((1, N, N)), np.zeros((1, N, N))], -1).astype(np.float32))
    d_h = np.array([40., 60., 90., 140.])
    dists = ctx.array(d_h.astype(np.float32))
    gd = ctx.zeros((nd,))
    a_h = np.tile(np.array([[1., 0, 0], [0, 1., 0]]), [nd, 1, 1])
    aff = ctx.array(a_h.astype(np.float32))
    ga = ctx.zeros(aff.shape)
    data_h = (1 + 0.1 * r.standard_normal((nd, N, N))) ** 2
    data = ctx.array(data_h.astype(np.float32))
    ident = ctx.array(np.array([[1., 0, 0], [0, 1., 0]], np.float32))
    lib = ctx.lib
    o_d = AdamOptimizer('free_prop_cm', options_dict={'step_size': 1e-1}); o_d.create_param_arrays([nd], device=ctx)
    o_a = AdamOptimizer('prj_affine_ls', options_dict={'step_size': 1e-3}); o_a.create_param_arrays(list(aff.shape), device=ctx)
    o_obj = AdamOptimizer('obj', options_dict={'step_size': 1e-2}); o_obj.params_whole_array_dict = {'m': m, 'v': v}
    obj_flat, g_flat = obj.view(0, (obj.size,)), g.view(0, (g.size,))
    pending = [None]
    ev = [ctx.event(), ctx.event()]
    timed = [False]

    mv = lambda o: (o.params_whole_array_dict['m'], o.params_whole_array_dict['v'])
    fused = [True]          # the driver's path for this configuration (one rank, no regulariser, plain Adam): adm_holo_fwd_adj_adam

    def step():
        if timed[0]:
            ev[0].record()
        if fused[0]:
            eng.forward_adjoint_adam(obj, probe, dists, data, mv(o_obj), 1e-2, 0, affine=aff, dists_mv=mv(o_d), step_dists=1e-1,
                                     affine_mv=mv(o_a), step_affine=1e-3, affine_pin=ident)
        else:
            eng.forward_adjoint(obj, probe, dists, data, affine=aff, grad_obj=g, grad_dists=gd, grad_affine=ga, overwrite=True)
        if timed[0]:
            ev[1].record()
        if not fused[0]:
            apply_small_params(ctx, [dict(opt=o_obj, x=obj_flat, g=g_flat), dict(opt=o_d, x=dists, g=gd), dict(opt=o_a, x=aff, g=ga, pin=ident)], 0)
        tok = eng.loss_async()
        out = pending[0]() if pending[0] is not None else None           # the PREVIOUS minibatch's loss
        pending[0] = tok
        return out

    def loop():
        for _ in range(3):
            step()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        ti = (time.perf_counter() - t0) / steps          # host time to queue a minibatch (the loss read-back keeps it within one minibatch of dt)
        ctx.sync()
        return (time.perf_counter() - t0) / steps, ti

    fused[0] = False
    dt_sep, _ = loop()
    fused[0] = True
    dt, t_issue = loop()
    # the other demo of this row (demos/2d_multidist_holography_w_position_correction.py): one (sy, sx) per distance refined with the
    # object -- registered targets in front of the launch group, shift gradient behind it, one small-parameter launch
    spec = eng.data_spectrum(data)
    sh, gs = ctx.zeros((nd, 2)), ctx.zeros((nd, 2))
    o_s = AdamOptimizer('probe_pos_correction', options_dict={'step_size': 1e-1}); o_s.create_param_arrays([nd, 2], device=ctx)

    def step_shift():
        eng.forward_adjoint_shifted(obj, probe, dists, spec, sh, grad_obj=g, grad_shifts=gs, overwrite=True)
        apply_small_params(ctx, [dict(opt=o_obj, x=obj_flat, g=g_flat), dict(opt=o_s, x=sh, g=gs, center_cols=2, zero_grad=True)], 0)
        tok = eng.loss_async()
        out = pending[0]() if pending[0] is not None else None
        pending[0] = tok
        return out
    for _ in range(3):
        step_shift()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_shift()
    ctx.sync()
    dt_shift = (time.perf_counter() - t0) / steps
    timed[0] = True
    ks = []
    for _ in range(10):
        step()
        ks.append(ev[0].elapsed_ms(ev[1]))
    timed[0] = False
    kern = float(np.median(ks))
    tc = bench.cpu_baseline_c5(obj_h, d_h, a_h, data_h, N, energy, psize)
    # algorithmic bytes of one minibatch (SURVEY 8d's rule: compulsory traffic only, intermediates are the design's): object and
    # probe read, the nd holograms read, object gradient written
    alg = 4 * (2 * N * N + 2 * N * N + nd * N * N + 2 * N * N)
    # the transforms of one minibatch: FFT2(psi), nd inverse, nd forward, one inverse = 2 nd + 2 transforms of 5 N^2 log2(N^2) flop
    flop = (2 * nd + 2) * 5.0 * N * N * np.log2(N * N)
    return {'row': 'f1 / config-5 shape', 'dtype': 'f32',
            'roofline': {'bound': 'latency (five dependent line-transform kernels of 128-512 workgroups; all intermediates stay in L2 / Infinity Cache) -- priced against hbm',
                         'kernel': 'holo_k1..k5<512> (the forward+adjoint(+Adam) launch group, HIP events)', 'achieved': alg / (kern * 1e-3) / 1e9,
                         'peak': bench.PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': alg / (kern * 1e-3) / 1e9 / bench.PEAK_HBM_GBS, 'kernel_ms': kern,
                         'algorithmic_bytes_per_launch': alg, 'whole_step_frac': alg / dt / 1e9 / bench.PEAK_HBM_GBS, 'traffic': None,
                         'transform_tflops': flop / (kern * 1e-3) / 1e12}, 'workload': 'multi-distance holography 512x512x1 real_imag, 4 distances, object+distance+affine Adam',
            'value': 1.0 / dt, 'unit': 'minibatches/s (4 holograms each)', 'ms_per_step': 1e3 * dt, 'host_issue_ms_per_step': 1e3 * t_issue,
            'update': 'Adam of object, distances and affine matrices inside the last kernel of the launch group (adm_holo_fwd_adj_adam)',
            'ms_per_step_with_a_separate_adam_launch': 1e3 * dt_sep,
            'ms_per_step_with_shift_refinement': 1e3 * dt_shift,
            'cpu_baseline': {'value': 1.0 / tc, 'unit': 'minibatches/s', 'cores': 1, 'kind': 'port',
                             'sample': 'one fwd+adjoint of the 4-distance chain, oracle fp32 (optimiser excluded)'}}


def bench_tiles(ctx, steps=50):
    """f1 with the data divided into sub-tiles (adorym/forward_model.py:884-1034): a 480 x 480 object recorded at 4 distances as
    10 x 10 holograms of 48 x 48 pixels, safe zone 12 (tile 72 x 72: the tuned kernel), minibatch 50 tiles, Adam on the object.
    Timed through MultiDistModel.loss_and_gradients + the optimiser kernel, as the driver runs it."""
    from adorym_amd.forward_model import MultiDistModel
    from adorym_amd.dp import HipOps
    N, SUB, szw, nd, mb = 480, 48, 12, 4, 50
    T = SUB + 2 * szw
    r = np.random.default_rng(2)
    energy, psize = 17050., 1e-4
    d_h = np.array([40., 60., 90., 140.])
    pos = np.array([[y, x] for y in range(0, N, SUB) for x in range(0, N, SUB)])
    window = np.zeros((T, T), np.float32)
    window[szw:T - szw, szw:T - szw] = 1
    eng = A.MultisliceEngine(ctx, (N, N, 1), (T, T), np.repeat(pos - szw, nd, axis=0), energy, psize, free_prop_cm=d_h, max_batch=mb * nd,
                             unknown_type='real_imag', beamstop=window)
    prj = (1 + 0.1 * r.standard_normal((1, nd * len(pos), SUB, SUB))).astype(np.float32)
    cv = dict(unknown_type='real_imag', prj=prj, engine=eng, tile_engine=eng, holo_engine=None, two_d_mode=True, safe_zone_width=szw,
              n_dp_batch=20, sign_convention=1, scale_ri_by_k=True)
    fm = MultiDistModel(device=ctx, common_vars_dict=cv, raw_data_type='magnitude')
    obj_h = np.stack([r.normal(1, 0.01, (N, N, 1)), r.normal(0, 0.01, (N, N, 1))], -1).astype(np.float32)
    obj = ctx.array(obj_h)
    g, m, v = ctx.empty(obj.shape), ctx.zeros((obj.size,)), ctx.zeros((obj.size,))
    probe = ctx.array(np.stack([np.ones((1, N, N)), np.zeros((1, N, N))], -1).astype(np.float32))
    ops = HipOps(ctx)
    batches = [np.arange(0, mb), np.arange(mb, 2 * mb)]
    ev = [ctx.event(), ctx.event()]
    pending = [None]

    def step(k, timed=False):
        ind = batches[k % 2]
        if timed:
            ev[0].record()
        fm.loss_and_gradients([0], g, obj, probe, None, 0., None, 0, pos[ind], prj, None, ind, d_h, szw, None, None, None, _init_grad=True)
        if timed:
            ev[1].record()
        ops.adam(obj, g, 0, m, v, 0, 0, obj.size, k, 1e-3, 0.9, 0.999, 1e-7, 0, None)
        tok = fm.take_loss_thunk()
        out = pending[0]() if pending[0] is not None else None
        pending[0] = tok
        return out

    for k in range(4):
        step(k)
    ctx.sync()
    t0 = time.perf_counter()
    for k in range(steps):
        step(k)
    ti = (time.perf_counter() - t0) / steps
    ctx.sync()
    dt = (time.perf_counter() - t0) / steps
    ks = []
    for k in range(10):
        step(k, True)
        ctx.sync()
        ks.append(ev[0].elapsed_ms(ev[1]))
    kern = float(np.median(ks))
    full = np.concatenate([batches[0] + i * len(pos) for i in range(nd)])
    tc = bench.cpu_baseline_tiles(obj_h.astype(np.float64), pos[batches[0]], SUB, szw, d_h, prj[0, full].astype(np.float64), energy, psize)
    # compulsory traffic of one minibatch: per distance and tile the object window in (T^2 x 8 B), its probe window in (T^2 x 8), the
    # sub-hologram in (SUB^2 x 4), the window's gradient out (T^2 x 8); object gradient written once (N^2 x 8)
    alg = nd * mb * (3 * T * T * 8 + SUB * SUB * 4) + N * N * 8
    return {'row': 'f1 / sub-tiles + safe zone', 'dtype': 'f32',
            'workload': 'multi-distance holography 480x480x1 real_imag, 4 distances, 100 holograms of 48x48 per distance, safe zone 12 (tiles 72x72), minibatch 50 tiles, Adam',
            'value': mb / dt, 'unit': 'tiles/s (each at 4 distances)', 'ms_per_step': 1e3 * dt, 'host_issue_ms_per_step': 1e3 * ti,
            'roofline': {'bound': 'latency (rotation copy, ONE launch of 200 workgroups = 50 tiles x 4 distances, overlap-add, back-copy) -- priced against hbm',
                         'kernel': 'the forward+adjoint launch group of one minibatch (rotate_fwd, ms_fwd_adj_kernel<72,8,9,...,PP>, cover_build, tile_accumulate, rotate_adj), HIP events',
                         'achieved': alg / (kern * 1e-3) / 1e9, 'peak': bench.PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': alg / (kern * 1e-3) / 1e9 / bench.PEAK_HBM_GBS,
                         'kernel_ms': kern, 'algorithmic_bytes_per_launch': alg, 'whole_step_frac': alg / dt / 1e9 / bench.PEAK_HBM_GBS, 'traffic': None},
            'cpu_baseline': {'value': mb / tc, 'unit': 'tiles/s', 'cores': 1, 'kind': 'port',
                             'sample': 'one minibatch (50 tiles x 4 distances) fwd+adjoint, oracle fp32 (optimiser excluded)'}}


if __name__ == '__main__':
    # python tools/bench_rows.py [c2] [c1] [c5] [tiles]   (default: all; one name = one row, e.g. under rocprofv3 --stats)
    rows = {'c2': bench_c2, 'c1': bench_c1, 'c5': bench_c5, 'tiles': bench_tiles}
    want = [a for a in sys.argv[1:] if a in rows] or ['c2', 'c1', 'c5', 'tiles']
    ctx = A.Context(0)
    for name in want:
        print(json.dumps(rows[name](ctx)))
