#!/usr/bin/env python3
"""
Benchmark of the hot path on BASELINE.json's metric: probe-positions/s (forward + gradient),
256^3 multislice ptychotomography (config 3), minibatch 32 per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus N --steps K --warmup W          (starts its own N ranks as child processes: adorym_amd/launch.py)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W          (or any launcher that sets RANK / WORLD_SIZE / MASTER_*)

One "step" = one minibatch iteration of reconstruct_ptychography (update_scheme='immediate'):
rotate object to theta (footprint planes) -> multislice forward + far-field LSQ loss + adjoint for the
32 positions -> rotate gradient back -> L1+TV regulariser gradient -> [reduce-scatter over ranks] ->
fused Adam (+ all-gather).  No part of that computation is skipped inside the timed region.  As the bench contract
allows, the INPUTS are resident in HBM when the timed region starts: the measured magnitudes of every timed minibatch and
the per-angle rotation tables (fp16 lookup + adjoint CSR) are staged before the clock starts.  What the product's driver
does on top of that -- host-resident data handed over per minibatch through the pinned ring, rotation tables built the
first time an angle is met -- is timed separately, on reconstruct_ptychography itself, in the `driver` keys (mean per
minibatch over 16 first-touch angles; 1.04x the engine loop).  With N ranks the global batch is N x 32 positions (the
reference's `mpirun -n N` semantics): weak scaling.  Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
EXIT_LEG_ABANDONED = 5      # a secondary leg of an N > 1 run hung: the headline line was printed, with `legs_abandoned`


def algorithmic_bytes_fwd_grad(B, Py, Px, S, V):
    """SURVEY.md 8(d) / BASELINE.md 3: tile data read fwd + read bwd + gradient written once,
    measured magnitudes once, object read once + object gradient written once."""
    return 4 * (3 * B * Py * Px * S * 2 + B * Py * Px + 2 * (2 * V))


KERNEL_SOURCES = ('adm_multislice.hip', 'adm_fft.h', 'adm_ms_math.h', 'adm_common.h')


def kernel_sources_sha():
    """Identity of the multislice kernel's sources (what a PMC pass was taken on)."""
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, 'adorym_amd', 'csrc', f), 'rb').read())
    return h.hexdigest()[:16]


def read_traffic(B):
    """roofline.traffic = HBM bytes of one multislice launch from the PMC passes (tools/pmc_sq.sh + tools/traffic_update.py ->
    profiles/traffic_latest.json).  Counters cannot be collected inside this run (rocprofv3 wraps the process), so the file is
    used only if it was taken on THESE kernel sources and this batch size; otherwise traffic is null and the reason is stated."""
    tpath = os.path.join(ROOT, 'profiles', 'traffic_latest.json')
    src = {'file': 'profiles/traffic_latest.json', 'kernel_sources_sha': kernel_sources_sha()}
    try:
        j = json.load(open(tpath))
    except Exception as e:
        src['unused_because'] = 'unreadable: %r' % (e,)
        return None, src
    src.update({k: j.get(k) for k in ('pmc_kernel_sources_sha', 'pmc_batch', 'command', 'source')})
    if j.get('pmc_kernel_sources_sha') != src['kernel_sources_sha']:
        src['unused_because'] = 'the PMC passes were taken on other kernel sources (stale)'
        return None, src
    if j.get('pmc_batch') != B:
        src['unused_because'] = 'the PMC passes were taken at another batch size'
        return None, src
    return j.get('ms_fwd_adj_kernel_hbm_bytes_per_launch'), src


def cpu_baseline(cfg, seconds_budget=12.0):
    """The pinned NumPy oracle (fp32, the reference's dtype) timed on this host's cores: forward + hand adjoint of the
    multislice chain for a bounded sample of probe positions (rotation excluded => favours the CPU).
      value / cores            the port on ALL host cores: a pool of single-threaded worker processes over whole positions
                               (oracle/cpu_pool_bench.py, run as a child process: a fresh interpreter without the GPU);
      one_core                 the same on one core;
      reference_structured     the reference's own op structure on PyTorch-CPU autograd (oracle/torch_structured.py), best of a
                               thread sweep -- the "what the reference does on a CPU" figure."""
    import subprocess
    from oracle import adorym_oracle as O
    P = cfg['probe_size'][0]
    S = cfg['obj_size'][2]
    phys = O.Physics(cfg['probe_size'], cfg['energy_ev'], cfg['psize_cm'], free_prop_cm=cfg['free_prop_cm'])
    r = np.random.default_rng(0)
    from adorym_amd.workloads import probe_array
    pa = probe_array(cfg)
    probe = pa[..., 0] + 1j * pa[..., 1]
    nb, done, t_used = 2, 0, 0.0
    while True:
        tiles = np.stack([r.normal(8.7e-7, 1e-7, (nb, P, P, S)), r.normal(5.1e-8, 1e-8, (nb, P, P, S))], -1).astype(np.float32)
        meas = np.abs(r.standard_normal((nb, P, P))).astype(np.float32)
        t0 = time.perf_counter()
        O.forward_adjoint_tiles(tiles, probe, meas, phys, 'float32')
        t_used += time.perf_counter() - t0
        done += nb
        if t_used > 0.4 * seconds_budget or done >= 512:
            break
    one = {'value': done / t_used, 'unit': 'probe-positions/s', 'cores': 1, 'positions': done, 'seconds': t_used}
    n_cpu = os.cpu_count() or 1
    out = None
    try:
        txt = subprocess.run([sys.executable, os.path.join(ROOT, 'oracle', 'cpu_pool_bench.py'), str(n_cpu), str(seconds_budget)],
                             capture_output=True, text=True, timeout=300).stdout.strip().split('\n')[-1]
        pool = json.loads(txt)
        out = {'value': pool['value'], 'unit': 'probe-positions/s', 'cores': pool['cores'], 'kind': 'port',
               'sample': '%d positions, P=%d, S=%d slices, fwd + hand adjoint of the multislice chain in fp32 NumPy/pocketfft '
                         '(oracle/adorym_oracle.py) on a pool of %d single-threaded worker processes, rotation and optimiser excluded, '
                         '%.1f s wall; host has %d cores' % (pool['positions'], P, S, pool['cores'], pool['seconds'], n_cpu)}
    except Exception as e:
        out = {'value': one['value'], 'unit': 'probe-positions/s', 'cores': 1, 'kind': 'port',
               'sample': '%d positions on one core (the all-core pool failed: %r)' % (done, e)}
    out['one_core'] = one
    try:
        out['reference_structured'] = cpu_baseline_torch()
    except Exception as e:      # a reported extra, never fatal for the bench line
        out['reference_structured'] = {'error': repr(e)}
    return out


def cpu_baseline_torch(budget_s=75.0):
    """Second flavour (SURVEY.md 8d ii): the reference's own op structure on PyTorch-CPU autograd, timed by
    oracle/torch_structured_bench.py in a CHILD process -- this process, which holds the GPU context, never imports torch (a
    PyTorch-ROCm wheel loads its own copy of the HIP runtime beside the one libadm is linked to; the two do not shut down
    cleanly together, and the benchmark must exit 0)."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'oracle', 'torch_structured_bench.py'), str(budget_s)],
                       capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        raise RuntimeError('torch_structured_bench.py exited %d: %s' % (r.returncode, r.stderr[-400:]))
    return json.loads(r.stdout.strip().split('\n')[-1])


def cpu_baseline_c1(obj_h, pos, pos_int, probe_h, B, P, energy, psize):
    """CPU leg of tools/bench_rows.py (config-1 shape): seconds for the oracle's fwd + adjoint of one minibatch, fp32, 1 core."""
    from oracle import adorym_oracle as O
    phys = O.Physics((P, P), energy, psize, free_prop_cm='inf', unknown_type='real_imag')
    tiles, _ = O.extract_tiles(obj_h, pos_int[:B], (P, P), 'real_imag')
    meas = np.abs(np.random.default_rng(2).standard_normal((B, P, P)))
    t1 = time.perf_counter()
    O.forward_adjoint_tiles(tiles, probe_h, meas, phys, 'float32', shifts=(pos - pos_int)[:B])
    return time.perf_counter() - t1


def cpu_baseline_c2(guess, thetas, probe_h, data, energy, psize, budget_s=20.0):
    """CPU leg of tools/bench_rows.py (config-2 shape): the oracle's fp32 forward + hand adjoint of whole angles (rotation,
    multislice chain, back-rotation, L1 term) on one core, as many of the 16 angles as fit the budget."""
    from oracle import adorym_oracle as O
    N = guess.shape[0]
    phys = O.Physics((N, N), energy, psize, free_prop_cm=0)
    pos = np.array([(0, 0)])
    g32 = guess.astype(np.float32)
    done, t_used = 0, 0.0
    for th, meas in zip(thetas, data):
        t1 = time.perf_counter()
        c = O.rotation_coords((N, N, N), th)
        O.forward_adjoint_object(g32, c, probe_h, pos, meas, phys, 'float32')
        O.l1_value_grad(g32, 1e-9 * N ** 3, 1e-10 * N ** 3)
        t_used += time.perf_counter() - t1
        done += 1
        if t_used > budget_s:
            break
    return done, t_used


def cpu_baseline_c5(obj_h, d_h, a_h, data_h, N, energy, psize):
    """CPU leg of tools/bench_rows.py (config-5 shape): seconds for the oracle's fwd + adjoint of the 4-distance chain."""
    from oracle import adorym_oracle as O
    t1 = time.perf_counter()
    O.holo_forward_adjoint(obj_h.astype(np.float64), np.ones((N, N), complex), d_h, a_h, data_h, energy, psize, dtype='float32')
    return time.perf_counter() - t1


def cpu_baseline_tiles(obj_h, pos, sub, szw, d_h, meas, energy, psize):
    """CPU leg of tools/bench_rows.py (multi-distance data divided into sub-tiles): seconds for the oracle's fwd + adjoint of one
    minibatch of tiles at every distance."""
    from oracle import adorym_oracle as O
    t1 = time.perf_counter()
    O.multidist_tiles_forward_adjoint(obj_h, np.ones(obj_h.shape[:2], complex), pos, (sub, sub), szw, d_h, meas, energy, psize, dtype='float32')
    return time.perf_counter() - t1


def fused_batch_measure(ctx, eng, state, probe, tables, cfg, targets, check, n_groups, label, reps=3, comm=None, rank=0, world=1):
    """Secondary figures (not `value`): steps whose global batch holds `n_groups` reference minibatches of ONE angle, fused
    into one launch so that every CU has work.  Two reference semantics give such a step:
      * update_scheme='per angle' (adorym/ptychography.py:1095-1099): the 17 minibatches of an angle (529 positions padded
        to 544, :816-823) see the same object and are accumulated before the update;
      * `mpirun -n R` (:786,905-909,1113-1114): the global batch is R x minibatch_size positions, every rank's loss is the
        mean over its own minibatch, the gradients are SUMMED and every rank adds the regulariser term -- `virtual_ranks`
        R on one GPU is exactly that sum.
    One step = whole-object rotation + n_groups*32 positions fwd/adjoint + overlap-add + back-rotation + regulariser
    gradient (x n_groups) + Adam.
    With `world` > 1 ranks every rank runs n_groups minibatches of ITS slice of each global batch (the reference's rank split,
    :905-909), the gradients are reduce-scattered ONCE per step and the object gathered once: the regime with the fewest
    exchanges per position.  Timed between barriers, maximum over ranks; `value` counts the positions of all ranks."""
    import time as _t
    mb = cfg['minibatch_size']
    n_pos = len(cfg['probe_pos'])
    B = n_groups * mb
    # global batch j of the step = positions [j * mb * world, (j + 1) * mb * world) of the angle; rank r takes the r-th slice
    ind = (np.arange(n_groups)[:, None] * mb * world + rank * mb + np.arange(mb)[None, :]).reshape(-1) % n_pos
    pos = cfg['probe_pos'][ind]
    Py, Px = cfg['probe_size']
    tgt = ctx.empty((B, Py, Px))
    any_t = next(iter(targets.values()))
    for j in range(n_groups):                # synthetic magnitudes: tile the ones generated for the main run
        tgt.view(j * mb * Py * Px, (mb, Py, Px)).copy_from(any_t)
    eng._reserve(B)
    e_pairs = [(ctx.event(), ctx.event()) for _ in range(2)]
    best = None
    for attempt in range(2):        # secondary figure: the better of two measurements (one host hiccup in three steps is +30 %)
        m_ = _fused_once(ctx, eng, state, probe, tables, cfg, check, n_groups, reps, comm, pos, tgt, e_pairs, mb, Py, Px)
        if best is None or m_[0] < best[0]:
            best = m_
    dt, kern, loss = best
    Y, X, Z = cfg['obj_size']
    alg = algorithmic_bytes_fwd_grad(B, Py, Px, Z, Y * X * Z)
    out = {'positions_per_step': B * world, 'positions_per_step_per_gpu': B, 'n_gpus': world, 'value': B * world / dt, 'unit': 'probe-positions/s',
           'ms_per_step': 1e3 * dt, 'fwd_adj_overlap_add_ms': float(np.mean(kern)),
           'fwd_adj_overlap_add_frac_of_hbm_peak': alg / (np.mean(kern) * 1e-3) / 1e9 / PEAK_HBM_GBS,
           'whole_step_frac_of_hbm_peak': alg / dt / 1e9 / PEAK_HBM_GBS, 'loss_last': loss, 'measured': 'better of two runs of %d steps' % reps}
    out.update(label)
    return out


def _fused_once(ctx, eng, state, probe, tables, cfg, check, n_groups, reps, comm, pos, tgt, e_pairs, mb, Py, Px):
    import time as _t
    kern = []
    prev = None
    ctx.sync()
    t0 = None
    for r in range(reps + 1):
        e0, e1 = e_pairs[r & 1]
        if r == 1:
            if prev is not None:
                loss = eng.loss_result(prev[0])
                prev = None
            ctx.sync()
            if comm is not None:
                comm.barrier()
            t0 = _t.perf_counter()
        it = r % len(tables)
        eng.set_batch(pos, tgt)
        eng.rotate(state.obj, tables[it], None)
        ctx.fork()
        eng.flush_loss_copy()
        state.finish_update()
        check(ctx.lib.adm_reg_grad_set(eng.plan.handle, state.obj.ptr, cfg['alpha_d'] * n_groups, cfg['alpha_b'] * n_groups,
                                       cfg['gamma'] * n_groups, state.grad.ptr, None))
        ctx.end_fork()              # no join: the overlapped launch forks again and the side stream is in order
        e0.record()
        eng.multislice_overlapped(probe, grad_scale=2.0 / (mb * Py * Px))     # full rounds | overlap-add beside the last round
        e1.record()
        eng.rotate_adjoint(state.grad, tables[it], None)
        state.exchange_and_update('adam', r, {'step_size': cfg['learning_rate']}, first=(0, state.n))
        # as in the main loop and the driver: the loss of step r is read back after step r+1 has been queued
        token = eng.loss_async(last=mb)
        if prev is not None:
            loss = eng.loss_result(prev[0])
            if prev[1] >= 1:
                kern.append(e_pairs[prev[1] & 1][0].elapsed_ms(e_pairs[prev[1] & 1][1]))
        prev = (token, r)
    loss = eng.loss_result(prev[0])
    kern.append(e_pairs[prev[1] & 1][0].elapsed_ms(e_pairs[prev[1] & 1][1]))
    state.finish_update()
    ctx.sync()
    if comm is not None:
        comm.barrier()
    dt = (_t.perf_counter() - t0) / reps
    if comm is not None:
        dt = comm.max_over_ranks(dt)
    return dt, kern, loss


def kernel_sweep(ctx, eng, probe, cfg, targets, batches=(1, 8, 32, 64, 128, 256, 512, 544), reps=5):
    """Duration of the multislice forward+adjoint launch alone (HIP events on its stream) over the number of positions in
    flight: ms per launch, positions/s and fraction of the HBM roofline with SURVEY 8(d)'s algorithmic bytes."""
    mb = cfg['minibatch_size']
    Py, Px = cfg['probe_size']
    Y, X, Z = cfg['obj_size']
    n_pos = len(cfg['probe_pos'])
    any_t = next(iter(targets.values()))
    e0, e1 = ctx.event(), ctx.event()
    rows = []
    for B in batches:
        eng._reserve(B)
        tgt = ctx.empty((B, Py, Px))
        for j in range(0, B, mb):
            n = min(mb, B - j)
            tgt.view(j * Py * Px, (n, Py, Px)).copy_from(any_t.view(0, (n, Py, Px)))
        eng.set_batch(cfg['probe_pos'][(np.arange(B) + 200) % n_pos], tgt)
        ts = []
        for r in range(reps + 1):
            e0.record()
            if B > eng.N_CU:
                eng.multislice_overlapped(probe, grad_scale=2.0 / (mb * Py * Px))
            else:
                eng.multislice(probe, accumulate=False)
            e1.record()
            ts.append(e0.elapsed_ms(e1))
        ms = float(np.median(ts[1:]))          # the median of the repetitions (the best of three flattered 256 positions)
        alg = algorithmic_bytes_fwd_grad(B, Py, Px, Z, Y * X * Z)
        rows.append({'positions': B, 'ms': ms, 'ms_min': float(np.min(ts[1:])), 'ms_max': float(np.max(ts[1:])), 'repetitions': reps,
                     'positions_per_s': B / (ms * 1e-3), 'frac_of_hbm_peak': alg / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                     'includes_overlap_add_of_all_but_last_round': B > eng.N_CU})
    return rows


def driver_measure(cfg, n_theta=16, update_scheme='immediate', one_block_of_data=False):
    """The PRODUCT's driver, not the engine loop: one epoch of adorym_amd.reconstruct_ptychography over `n_theta` angles of
    config 3 that it has never seen (rotation tables and their adjoint CSR are built inside the timed run), host-resident
    measured data handed over minibatch by minibatch (adorym/forward_model.py:113-119).  Step time = MEAN spacing of the
    per-minibatch time stamps of convergence/loss_rank_0.txt (adorym/ptychography.py:1261), first-touch angles included."""
    import tempfile
    import adorym_amd as A
    from adorym_amd import workloads as W
    r = np.random.default_rng(0)
    n_pos = len(cfg['probe_pos'])
    Py, Px = cfg['probe_size']
    if one_block_of_data:
        # a whole epoch (500 angles = 5.5 GB of magnitudes): one angle's worth of synthetic data seen through a stride-0 view, so
        # that nothing is generated for minutes on the host; the driver still hands over every minibatch from host memory
        prj = np.broadcast_to(np.abs(r.standard_normal((1, n_pos, Py, Px), dtype=np.float32)) * 30, (n_theta, n_pos, Py, Px))
    else:
        prj = (np.abs(r.standard_normal((n_theta, n_pos, Py, Px), dtype=np.float32)) * 30)
    g = W.random_guess(cfg['obj_size'], seed=1)
    import contextlib
    with tempfile.TemporaryDirectory() as td, open(os.devnull, 'w') as sink, contextlib.redirect_stdout(sink):
        t0 = time.perf_counter()        # (the driver's progress lines go to the sink: stdout carries the one JSON line)
        st = A.reconstruct_ptychography(
            fname=prj, obj_size=cfg['obj_size'], probe_pos=cfg['probe_pos'], theta_st=0, theta_end=2 * np.pi, n_theta=n_theta,
            energy_ev=cfg['energy_ev'], psize_cm=cfg['psize_cm'], free_prop_cm='inf', minibatch_size=cfg['minibatch_size'], n_epochs=1,
            alpha_d=cfg['alpha_d'], alpha_b=cfg['alpha_b'], gamma=cfg['gamma'], learning_rate=cfg['learning_rate'], optimizer='adam',
            initial_guess=[g[..., 0], g[..., 1]], save_path=td, output_folder='drv', store_checkpoint=False, use_checkpoint=False,
            update_scheme=update_scheme, return_state=True, **cfg['probe'])
        wall = time.perf_counter() - t0
        lines = open(os.path.join(st['output_folder'], 'convergence', 'loss_rank_0.txt')).read().strip().split('\n')[1:]
    ts = np.array([float(l.split(',')[3]) for l in lines])
    n = len(ts)
    mean_ms = 1e3 * (ts[-1] - ts[0]) / (n - 1)
    per_step = cfg['minibatch_size'] if update_scheme == 'immediate' else -(-n_pos // cfg['minibatch_size']) * cfg['minibatch_size']
    return {'update_scheme': update_scheme, 'first_touch_angles': n_theta, 'logged_steps': n, 'positions_per_step': per_step,
            'ms_per_step_mean': mean_ms, 'ms_per_step_max': float(1e3 * np.diff(ts).max()), 'value': per_step / (mean_ms * 1e-3),
            'unit': 'probe-positions/s', 'wall_s_incl_setup_and_output': wall,
            'how': 'mean spacing of the time stamps of convergence/loss_rank_0.txt over the whole epoch'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--minibatch', type=int, default=32)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-per-angle', action='store_true', help="skip the secondary full-chip legs (per angle, virtual ranks, sweep)")
    ap.add_argument('--no-driver', action='store_true', help='skip timing reconstruct_ptychography itself')
    ap.add_argument('--no-driver-epoch', action='store_true', help='skip the whole-epoch (500 angles) run of reconstruct_ptychography (~25 s)')
    ap.add_argument('--legs', default='per_angle,vr8,vr16,sweep', help='which secondary full-chip legs to run (profiling aid)')
    ap.add_argument('--force-dist', action='store_true', help='use the multi-GPU (RCCL) code path even with one rank')
    ap.add_argument('--comm', choices=('rccl', 'p2p', 'host'), default=os.environ.get('ADM_COMM', 'rccl'),
                    help="rccl (default): RCCL reduce-scatter / all-gather behind libadm's C ABI; p2p: direct all-pairs exchange through "
                         "IPC-mapped peer buffers, fused with the optimiser (one kernel per update; the ranks may share one GPU); host: "
                         "staged through host memory (validation only; the ranks may share one GPU)")
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='weak',
                    help="weak: every rank runs --minibatch positions per step (global batch N x 32, the reference's `mpirun -n N`); "
                         "strong: ONE minibatch of --minibatch positions split over the ranks (N x 32/N)")
    ap.add_argument('--overlap-gather', choices=('auto', '0', '1'), default='0',
                    help='N > 1: two-part object gather (planes the next minibatches read first, rest beside the next kernel on a '
                         'second communicator).  Default 0: the headline of a multi-GPU run uses the plain all-gather on ONE '
                         'communicator -- nothing that has never run on real multi-GPU hardware sits in front of the measurement; '
                         'auto = check the two-part gather bitwise against the plain one on the running job, time both, keep the '
                         'faster (an experiment: ask for it explicitly)')
    ap.add_argument('--leg-timeout', type=int, default=int(os.environ.get('ADM_BENCH_LEG_TIMEOUT', '240')),
                    help='N > 1: seconds a secondary leg may take before it is abandoned and the line is printed without it (0 = no watchdog)')
    ap.add_argument('--no-p2p-leg', action='store_true', help='N > 1: skip the extra leg that repeats the headline loop through the peer-to-peer transport')
    ap.add_argument('--test-hang-leg', action='store_true', help=argparse.SUPPRESS)     # tests/: the last secondary leg never returns
    ap.add_argument('--test-crash-leg', action='store_true', help=argparse.SUPPRESS)    # tests/: rank 0 abort()s inside the last secondary leg
    ap.add_argument('--restricted-exchange', action='store_true',
                    help='N > 1 experiment: sum only the y-planes the global batch touches, each part onto its owner; the owners add '
                         'the regulariser term N-fold (DESIGN.md section 6; tests/test_gpu_world2.py).  Off by default.')
    args = ap.parse_args()

    # `bench.py --gpus N` without a launcher around it: THIS process -- which has not loaded libadm or touched the GPU --
    # starts the N ranks as child processes, relays rank 0's JSON line and exits with their code (adorym_amd/launch.py)
    if args.gpus > 1 and 'RANK' not in os.environ and int(os.environ.get('WORLD_SIZE', '1')) == 1:
        from adorym_amd import launch
        rc, _ = launch.run(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:])
        raise SystemExit(rc)

    import adorym_amd as A
    from adorym_amd import comm as C, workloads as W
    from adorym_amd.dp import DataParallelObject, HipOps
    from adorym_amd.util import rotation_lookup
    from adorym_amd._lib import check

    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.comm in ('host', 'p2p'):
        local_rank %= max(1, C.device_count())     # these transports let several ranks share a GPU
    use_dist = world > 1 or args.force_dist
    if not use_dist:
        comm = C.LocalComm()
    elif args.comm == 'host':
        comm = C.HostStagedComm(device_index=local_rank)
    elif args.comm == 'p2p':
        comm = C.P2PComm(device_index=local_rank)
    else:
        comm = C.RcclComm(device_index=local_rank)
    rank = comm.rank
    if use_dist and args.overlap_gather == '0':
        os.environ['ADM_COMM_AUX'] = '0'        # no side-stream communicator unless the two-part gather is asked for
    # libadm kernels and the collectives share the context's own stream
    ctx = A.Context(local_rank)
    comm_note = None
    if hasattr(comm, 'attach'):
        # attach() either succeeds on every rank or raises on every rank (agreement over the control plane inside).  There is
        # NO fall-back to another transport: a line measured on a different system would not be this system's number.  Every
        # rank leaves with the same non-zero code; the launcher (or the caller) decides what to start instead.
        try:
            comm.attach(ctx)
        except Exception as e:
            sys.stderr.write("bench.py: rank %d: the '%s' transport could not be brought up on every rank (%r); no line is printed. "
                             "Other transports: --comm rccl | p2p | host\n" % (rank, args.comm, e))
            try:
                comm.ctx = None
                comm.close()
            except Exception:
                pass
            raise SystemExit(4)
    # the communicator really spans N ranks (a silent 1-rank communicator would make every collective the identity)
    abi_size = int(ctx.lib.adm_p2p_size(ctx.handle) if getattr(comm, 'backend', '') == 'p2p' else ctx.lib.adm_comm_size(ctx.handle))
    if use_dist and comm.size != world:
        raise SystemExit('bench.py: communicator has %d ranks, expected %d' % (comm.size, world))
    if getattr(comm, 'backend', '') in ('rccl', 'p2p') and abi_size != world:
        raise SystemExit('bench.py: adm_comm_size() / adm_p2p_size() = %d, expected %d' % (abi_size, world))

    cfg = W.c3_config()
    B_global = args.minibatch * world if args.scaling == 'weak' else args.minibatch
    if args.scaling == 'strong' and args.minibatch % world:
        raise SystemExit('bench.py: --scaling strong needs --minibatch divisible by the number of ranks')
    B = B_global // world                      # positions per rank and step
    Y, X, Z = cfg['obj_size']
    Py, Px = cfg['probe_size']
    eng = A.MultisliceEngine(ctx, cfg['obj_size'], cfg['probe_size'], cfg['probe_pos'], cfg['energy_ev'], cfg['psize_cm'],
                             free_prop_cm=cfg['free_prop_cm'], binning=cfg['binning'], max_batch=B, transmissions_only=True)
    ops = HipOps(ctx)
    state = DataParallelObject(ops, comm, (Y, X, Z, 2))
    # reference-default Gaussian random initial guess (throughput is data independent)
    obj0 = ctx.array(W.random_guess((Y, X, Z), seed=1).reshape(-1))

    def reset_state():
        state.finish_update()
        state.obj.view(0, (obj0.size,)).copy_from(obj0)
        for m_ in state.moments:
            m_.zero_()

    reset_state()
    probe = ctx.array(W.probe_array(cfg))
    n_theta_used = max(2, min(8, args.steps + args.warmup))
    thetas = np.linspace(cfg['theta_st'], cfg['theta_end'], cfg['n_theta'], dtype='float32')[:: cfg['n_theta'] // n_theta_used][:n_theta_used]
    tables = [A.RotationTable(ctx, cfg['obj_size'], th) for th in thetas]
    for t in tables:
        t.csr(eng.plan)          # per-angle setup (cached for the whole reconstruction), outside the timed region
    pos_all = cfg['probe_pos']
    n_pos = len(pos_all)

    # synthetic measurements: the forward model itself applied to a foam phantom ("data": "synthetic")
    truth = ctx.array(W.foam_object(cfg['obj_size'], seed=0))
    total = args.steps + args.warmup
    plan_batches, plan_all = [], []
    for k in range(total):
        it = k % n_theta_used
        # the global batch of step k = B_global consecutive scan positions of one angle, rank r takes the r-th slice of B
        # (adorym/ptychography.py:905-909)
        g0 = ((k // n_theta_used) * B_global) % (n_pos - B_global + 1)
        plan_batches.append((it, np.arange(g0 + rank * B, g0 + (rank + 1) * B)))
        plan_all.append(np.arange(g0, g0 + B_global))      # what ALL ranks process in step k
    targets = {}
    # (synthesised with the in-loop modulator, i.e. another template instance of the kernel, so that a rocprofv3 --stats
    # summary of this command lists the forward-only synthesis launches separately from the measured forward+adjoint ones)
    eng.plan.set_transmission_cache(False)
    for it, ind in plan_batches:
        key = (it, int(ind[0]))
        if key in targets:
            continue
        eng.set_batch(pos_all[ind], np.zeros((B, Py, Px), np.float32))
        eng.rotate(truth, tables[it], eng.y_footprint(pos_all[ind]))
        eng.multislice(probe, want_grad=False, want_pred=True)
        t = ctx.empty((B, Py, Px))
        t.copy_from(eng._pred.view(0, (B, Py, Px)))
        targets[key] = t
    truth.free()
    ctx.sync()
    eng.plan.set_transmission_cache(2 if eng.transmissions_only else eng.transmission_cache)

    opt_options = {'step_size': cfg['learning_rate']}
    # two sets of timing events / loss read-backs: step k's results are looked at after step k+1 has been queued, so the
    # GPU never waits for the host (the loss still comes back for EVERY minibatch, as in the reference's log)
    ev_ms = [(ctx.event(), ctx.event()) for _ in range(2)]
    ms_kernel_total = [0.0]
    loss_box = [0.0]
    pending = [None]

    def resolve():
        if pending[0] is not None:
            token, evs = pending[0]
            if evs is not None:
                ms_kernel_total[0] += evs[0].elapsed_ms(evs[1])
            loss_box[0] = eng.loss_result(token)
            pending[0] = None

    # --restricted-exchange makes the footprint-restricted exchange the HEADLINE's; with several ranks it is timed as a second
    # leg of the same line either way (`immediate_restricted`)
    leg = {'restricted': bool(args.restricted_exchange and use_dist)}

    def reg_shard(lo_, hi_, alo_, ahi_):      # the regulariser term of all `world` ranks, on this rank's shard [lo_, hi_)
        check(ctx.lib.adm_reg_grad_range(eng.plan.handle, state.obj.ptr, cfg['alpha_d'] * world, cfg['alpha_b'] * world,
                                         cfg['gamma'] * world, state.grad.ptr, lo_, hi_, alo_, ahi_))

    def step(k, timed):
        it, ind = plan_batches[k]
        pos = pos_all[ind]
        # side stream: zero the gradient buffer and add the regulariser gradient (they only read the object) while
        # the multislice chain, which occupies `minibatch` of the 256 CUs, runs on the main stream
        eng.set_batch(pos, targets[(it, int(ind[0]))])
        yr = eng.y_footprint(pos)
        eng.rotate(state.obj, tables[it], yr)
        ctx.fork()
        eng.flush_loss_copy()       # the previous step's loss read-back, off the main stream
        state.finish_update()       # the part of the previous Adam pass that was deferred (planes this minibatch does not read)
        restricted = leg['restricted']
        if restricted:
            # only the planes the GLOBAL batch touches are initialised (zero) and exchanged; the regulariser comes in after the sum
            ty0, ty1 = eng.y_footprint(pos_all[plan_all[k]])
            touched = (ty0 * X * Z * 2, ty1 * X * Z * 2)
            state.grad.view(touched[0], (touched[1] - touched[0],)).zero_()
        else:
            check(ctx.lib.adm_reg_grad_set(eng.plan.handle, state.obj.ptr, cfg['alpha_d'], cfg['alpha_b'], cfg['gamma'],
                                           state.grad.ptr, None))      # initialises the gradient buffer: no separate zero fill
        eng.build_cover()           # cover lists of the overlap-add: positions only, built beside the kernel
        ctx.end_fork()
        evs = ev_ms[k & 1] if timed else None
        if timed:
            evs[0].record()
        eng.multislice(probe, accumulate=False)
        if timed:
            evs[1].record()
        ctx.join()
        eng.accumulate_tiles()
        eng.rotate_adjoint(state.grad, tables[it], yr)
        # update the y-planes the next minibatch reads first; the rest of the Adam pass overlaps the next kernel
        # (several ranks: the planes the next minibatches of ALL ranks read are gathered first, the rest of the all-gather
        # runs on the side stream beside the next kernel -- the range shapes a collective, so it is the same on every rank)
        first = None
        if k + 1 < len(plan_batches):
            ny0, ny1 = eng.y_footprint(pos_all[plan_all[k + 1]])
            first = (ny0 * X * Z * 2, ny1 * X * Z * 2)
        if restricted:
            state.exchange_and_update('adam', k, opt_options, first=first, touched=touched, reg_shard=reg_shard)
        else:
            state.exchange_and_update('adam', k, opt_options, first=first)
        token = eng.loss_async()
        resolve()                       # the PREVIOUS step's kernel time and loss
        pending[0] = (token, evs)

    # ---- N > 1: the two-part gather is checked against the plain all-gather ON THIS JOB before it is used -------------
    gather = {'overlap_gather': bool(state.overlap_gather), 'checked': False}
    if use_dist and state.inplace and hasattr(comm, 'broadcast') and args.overlap_gather != 'auto':
        state.overlap_gather = args.overlap_gather == '1'
        gather['overlap_gather'] = state.overlap_gather
    if use_dist and state.inplace and hasattr(comm, 'broadcast') and args.overlap_gather == 'auto' and len(plan_batches) >= 3:
        n_chk = min(3, len(plan_batches))
        res, ms = {}, {}
        for mode in (False, True):
            reset_state()
            state.overlap_gather, state.poison = mode, mode      # poison: stale planes are NaN until finish_update()
            for k in range(n_chk):
                step(k, False)
            resolve()
            state.finish_update()
            ctx.sync()
            res[mode] = state.obj.view(0, (state.n,)).get()
            state.poison = False
            comm.barrier()
            t_ = time.perf_counter()
            for k in range(n_chk):
                step(k, False)
            resolve()
            state.finish_update()
            comm.barrier()
            ms[mode] = comm.max_over_ranks(1e3 * (time.perf_counter() - t_) / n_chk)
        same = bool(np.array_equal(res[False], res[True])) and bool(np.all(np.isfinite(res[True])))
        same_all = comm.sum_over_ranks(1.0 if same else 0.0) >= world
        del res
        state.overlap_gather = bool(same_all and ms[True] < 0.99 * ms[False])
        gather = {'overlap_gather': state.overlap_gather, 'checked': True, 'bitwise_equal_to_plain_on_all_ranks': bool(same_all),
                  'plain_ms_per_step': ms[False], 'two_part_ms_per_step': ms[True], 'check_steps': n_chk}
        reset_state()

    from adorym_amd.device import PhaseClock

    def timed_loop():
        """W untimed + exactly K timed steps between barriers (device idle on both sides), maximum over ranks."""
        ms_kernel_total[0] = 0.0
        for k in range(args.warmup):
            step(k, False)
        resolve()
        if use_dist:        # (one GPU: nothing to explain, and the event records would sit between the kernels of the step's tail)
            state.clock = PhaseClock(ctx)
        ctx.sync()
        comm.barrier()
        t0 = time.perf_counter()
        for k in range(args.warmup, total):
            step(k, True)
        resolve()
        ctx.sync()
        comm.barrier()
        dt_ = comm.max_over_ranks(time.perf_counter() - t0)
        state.finish_update()
        ph = {n_: t_ / args.steps for n_, (t_, c_) in state.clock.totals().items()} if state.clock is not None else {}
        state.clock = None
        return dt_, ms_kernel_total[0] / args.steps, ph

    def phases_dict(kern_ms_, ph):
        d_ = {'kernel_ms': kern_ms_, 'reduce_scatter_ms': ph.get('reduce_scatter', 0.0), 'update_ms': ph.get('update', 0.0),
              'first_gather_ms': ph.get('first_gather', 0.0), 'deferred_gather_ms': ph.get('deferred_gather', 0.0)}
        if 'fused_exchange' in ph:      # --comm p2p: reduce-scatter + optimiser + all-gather are one kernel between two stream barriers
            d_['fused_exchange_ms'] = ph['fused_exchange']
        return d_

    dt, kern_ms, phases = timed_loop()
    loss_headline = loss_box[0]

    restricted = leg['restricted']

    # ---- the headline is complete: assemble its line NOW, so that nothing a secondary leg does can lose it ----
    out = None
    if rank == 0:
        ms_per_step = 1e3 * dt / args.steps
        value = B_global * args.steps / dt
        alg = algorithmic_bytes_fwd_grad(B, Py, Px, Z, Y * X * Z)
        achieved = alg / (kern_ms * 1e-3) / 1e9
        traffic, traffic_source = read_traffic(B)
        out = {
            'metric': 'probe-positions/sec (fwd+grad), 256^3 multislice ptycho', 'value': value, 'unit': 'probe-positions/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': cfg['name'], 'object': [Y, X, Z], 'probe': [Py, Px], 'slices': Z,
                       'minibatch_per_gpu': B, 'global_batch': B_global, 'update_scheme': 'immediate', 'optimizer': 'adam',
                       'regularizers': 'L1+TV', 'far_field': True, 'parallelism': 'dp%d' % world,
                       'collectives': getattr(comm, 'backend', 'local'),
                       'step': 'rotate_fwd + multislice fwd/loss/adjoint + rotate_adj + reg_grad + (reduce_scatter) + adam (+all_gather)'},
            'roofline': {'bound': 'hbm', 'kernel': 'ms_fwd_adj_kernel<72,8,9>', 'achieved': achieved, 'peak': PEAK_HBM_GBS,
                         'unit': 'GB/s', 'frac': achieved / PEAK_HBM_GBS, 'traffic': traffic, 'traffic_source': traffic_source,
                         'algorithmic_bytes_per_launch': alg, 'kernel_ms': kern_ms,
                         'whole_step_frac': alg / (ms_per_step * 1e-3) / 1e9 / PEAK_HBM_GBS},
            'loss_last': loss_headline,
            # device time per step of the exchange's phases (HIP events on the stream each phase is queued on; rank 0):
            # where an N-GPU step's time goes beyond the multislice kernel
            'phases_ms': phases_dict(kern_ms, phases),
            'comm': {'backend': getattr(comm, 'backend', 'local'), 'size': comm.size, 'adm_comm_size': abi_size, 'expected': world,
                     'side_stream_communicator': getattr(comm, 'backend', '') == 'rccl' and os.environ.get('ADM_COMM_AUX', '1') == '1',
                     'gather': gather, 'restricted_exchange': restricted, 'note': comm_note},
        }

    def emit(line):
        # RCCL prints a banner through C stdio, which would otherwise be flushed at exit, AFTER the JSON line
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(line), flush=True)

    # Secondary legs of a multi-rank run use collectives (grouped reductions onto the owners) or shapes (one exchange per fused
    # angle) that the headline loop did not: each runs under a watchdog on EVERY rank -- all ranks arm it at the same barrier --
    # so that a leg which does not come back within --leg-timeout seconds costs that leg, not the line: rank 0 prints what it has
    # and every rank leaves without tearing the communicator down.
    import threading

    class Watchdog(object):
        def __init__(self, what):
            self.what = what
            self.t = None

        def __enter__(self):
            if use_dist and world > 1 and out is not None:
                # a hard fault inside the leg (the runtime abort()s on a GPU memory fault) must not take the measured line with it
                crashed = dict(out)
                crashed['legs_abandoned'] = out.get('legs_abandoned', []) + [self.what]
                crashed['comm'] = dict(out['comm'], note=((out['comm']['note'] + '; ') if out['comm']['note'] else '') +
                                       "the process was killed by a signal inside leg '%s'" % self.what)
                ctx.lib.adm_crash_line_set(json.dumps(crashed).encode(), EXIT_LEG_ABANDONED)
            if use_dist and world > 1 and args.leg_timeout > 0:
                comm.barrier()

                def fire():
                    if out is not None:
                        out['legs_abandoned'] = out.get('legs_abandoned', []) + [self.what]
                        out['comm']['note'] = ((out['comm']['note'] + '; ') if out['comm']['note'] else '') + \
                            "leg '%s' did not finish within %d s and was abandoned" % (self.what, args.leg_timeout)
                        emit(out)
                    sys.stderr.write('bench.py: rank %d: leg %s timed out\n' % (rank, self.what))
                    os._exit(EXIT_LEG_ABANDONED)      # the headline line is out, but the run is NOT clean: automation can tell
                self.t = threading.Timer(args.leg_timeout, fire)
                self.t.daemon = True
                self.t.start()
            return self

        def __exit__(self, *exc):
            if self.t is not None:
                self.t.cancel()
            if use_dist and world > 1 and out is not None:
                ctx.lib.adm_crash_line_set(None, 0)
            return False

    # secondary leg on EVERY rank count: update_scheme='per angle' -- all minibatches of an angle fused, ONE exchange per step.
    # (N > 1: the same two collectives on the same communicator as the headline loop that has just run.)
    per_angle = None
    if not args.no_per_angle and 'per_angle' in args.legs.split(',') and args.scaling == 'weak':
        k_angle = -(-n_pos // (B * world))
        with Watchdog('per_angle'):
            per_angle = fused_batch_measure(ctx, eng, state, probe, tables, cfg, targets, check, k_angle, {'update_scheme': 'per angle'},
                                            comm=comm if use_dist else None, rank=rank, world=world)
        if out is not None:
            out['per_angle'] = per_angle

    # last leg with several ranks: the headline loop again with the footprint-restricted exchange (or, if that was asked for as
    # the headline, with the full exchange), from the same initial state
    if use_dist and world > 1 and args.scaling == 'weak':
        with Watchdog('immediate_restricted' if not leg['restricted'] else 'immediate_full_exchange'):
            if args.test_hang_leg:
                time.sleep(10 ** 6)
            if args.test_crash_leg and rank == 0:
                os.abort()
            leg['restricted'] = not leg['restricted']
            reset_state()
            dt2, kern2, ph2 = timed_loop()
            other_leg = {'restricted_exchange': leg['restricted'], 'value': B_global * args.steps / dt2, 'unit': 'probe-positions/s',
                         'ms_per_step': 1e3 * dt2 / args.steps, 'phases_ms': phases_dict(kern2, ph2), 'loss_last': loss_box[0]}
            leg['restricted'] = not leg['restricted']
        if out is not None:
            out['immediate_restricted' if other_leg['restricted_exchange'] else 'immediate_full_exchange'] = other_leg

    # ... and the headline loop once more through the OTHER device transport, from the same initial state: a line measured with
    # RCCL also carries the direct exchange (`immediate_p2p`: IPC-mapped peer buffers, one fused kernel per update), so that one
    # run on a multi-GPU node decides between them.  The peer-to-peer group shares the control plane; its buffers are its own.
    # A transport that cannot come up (no peer access between two GPUs, ...) raises on every rank together: the leg is
    # recorded as unavailable, the line is unaffected.
    if use_dist and world > 1 and args.scaling == 'weak' and args.comm != 'p2p' and not args.no_p2p_leg:
        with Watchdog('immediate_p2p'):
            p2p_leg = {'transport': 'p2p'}
            p2p = C.P2PComm(device_index=local_rank, group=comm.group_)
            try:
                p2p.attach(ctx)
            except Exception as e:
                p2p_leg['error'] = 'could not be brought up on every rank: %r' % (e,)
                p2p = None
            if p2p is not None:
                state_main, restricted_main = state, leg['restricted']
                try:
                    state = DataParallelObject(ops, p2p, (Y, X, Z, 2))
                    leg['restricted'] = False
                    reset_state()
                    dt3, kern3, ph3 = timed_loop()
                    ctx.sync()
                    p2p.check_status()
                    p2p_leg.update({'value': B_global * args.steps / dt3, 'unit': 'probe-positions/s', 'ms_per_step': 1e3 * dt3 / args.steps,
                                    'phases_ms': phases_dict(kern3, ph3), 'loss_last': loss_box[0], 'loss_headline': loss_headline})
                except Exception as e:
                    p2p_leg['error'] = repr(e)
                finally:
                    state, leg['restricted'] = state_main, restricted_main
                    try:
                        p2p.close(keep_group=True)
                    except Exception as e:
                        p2p_leg.setdefault('error', repr(e))
        if out is not None:
            out['immediate_p2p'] = p2p_leg

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(cfg)
        if world == 1 and not args.no_per_angle:
            legs = args.legs.split(',')
            for R in (8, 16):
                if 'vr%d' % R in legs:
                    out['virtual_ranks_%d' % R] = fused_batch_measure(ctx, eng, state, probe, tables, cfg, targets, check, R,
                                                                      {'update_scheme': 'immediate', 'virtual_ranks': R,
                                                                       'semantics': 'global batch R x 32 of one angle, gradients summed (mpirun -n R)'})
            if 'sweep' in legs:
                out['kernel_sweep'] = kernel_sweep(ctx, eng, probe, cfg, targets)
        if world == 1 and not args.no_driver:
            out['driver'] = driver_measure(cfg, 16, 'immediate')
            out['driver']['engine_loop_ms_per_step'] = ms_per_step
            out['driver']['ratio_to_engine_loop'] = out['driver']['ms_per_step_mean'] / ms_per_step
            out['driver_per_angle'] = driver_measure(cfg, 16, 'per angle')
            if not args.no_driver_epoch:
                # config 3 AS WRITTEN: one whole epoch, all 500 angles x 17 minibatches, rotation tables / adjoint CSR of every
                # angle built on first touch inside the timed run (tools/driver_c3_epoch.py: the same with synthesised data, a
                # second epoch, device memory and reconstruction quality; profiles/r06/r06b_c3_epoch.json)
                out['driver_epoch'] = driver_measure(cfg, cfg['n_theta'], 'immediate', one_block_of_data=True)
                out['driver_epoch']['ratio_to_16_angle_driver_leg'] = out['driver_epoch']['ms_per_step_mean'] / out['driver']['ms_per_step_mean']
    if use_dist:
        comm.close()            # RCCL may print its banner here; the JSON line goes last
    if out is not None:
        emit(out)


if __name__ == '__main__':
    main()
