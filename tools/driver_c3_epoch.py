#!/usr/bin/env python3
"""Config 3 AS WRITTEN through the product's driver: 500 angles x 529 positions (23 x 23 scan, padded to 17 minibatches of 32
per angle), 256^3 object, 72 x 72 probe, 256 slices, far field, L1 + TV, Adam lr 5e-5, update_scheme='immediate'
(reference: demos/multislice_ptycho_256_theta.py:52-93, tools/create_ptycho_data.py:139-163; BASELINE.json config 3).

    python tools/driver_c3_epoch.py [--n-theta 500] [--epochs 2] [--out gpurun_out/r06_c3_epoch.json]
    python tools/driver_c3_epoch.py --reduced [--out ...]      32^3 object, same script, with the fp64 / fp32 oracle beside it

What it does: (1) synthesises the measured magnitudes of every angle with the product's own forward kernel from a foam
phantom (host array [n_theta, 529, 72, 72], 5.5 GB at 500 angles: "data": "synthetic"); (2) runs
adorym_amd.reconstruct_ptychography on them, host-resident data handed over minibatch by minibatch, rotation tables and
their adjoint CSR built the first time an angle is met (epoch 0) and reused afterwards (epoch 1); (3) reports, from the
driver's own convergence log (adorym/ptychography.py:1261): positions/s over each whole epoch, the slowest and the median
minibatch, mean loss per angle, device memory in use (sampled every 20 ms through adm_mem_info), and the reconstruction
quality against the phantom (RMSE and Pearson correlation of delta inside the cone's bounding region) after every epoch.
With --reduced the same inputs go through the pinned NumPy oracle in fp64 and fp32 (test infrastructure, imported by this
tool only) and its quality numbers are printed next to the product's.

Not part of the product; nothing here is imported by adorym_amd."""
import argparse
import contextlib
import json
import os
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def quality(x, truth):
    """RMSE and Pearson correlation of delta (and beta) against the phantom."""
    out = {}
    for c, name in ((0, 'delta'), (1, 'beta')):
        a, b = x[..., c].astype(np.float64).ravel(), truth[..., c].astype(np.float64).ravel()
        out[name + '_rmse'] = float(np.sqrt(np.mean((a - b) ** 2)))
        a0, b0 = a - a.mean(), b - b.mean()
        den = np.sqrt((a0 ** 2).sum() * (b0 ** 2).sum())
        out[name + '_corr'] = float((a0 * b0).sum() / den) if den > 0 else 0.0
    return out


def synthesise(A, W, cfg, thetas, truth_h, mb):
    """|far field| of the phantom at every angle and scan position with the product's forward kernel (its own context, closed
    afterwards so that nothing of it counts as the reconstruction's memory)."""
    ctx = A.Context(0)
    Py, Px = cfg['probe_size']
    pos = cfg['probe_pos']
    n_pos = len(pos)
    eng = A.MultisliceEngine(ctx, cfg['obj_size'], cfg['probe_size'], pos, cfg['energy_ev'], cfg['psize_cm'], free_prop_cm=cfg['free_prop_cm'],
                             binning=cfg['binning'], max_batch=mb)
    truth = ctx.array(truth_h)
    probe = ctx.array(W.probe_array(cfg))
    prj = np.empty((len(thetas), n_pos, Py, Px), np.float32)
    zeros = np.zeros((mb, Py, Px), np.float32)
    t0 = time.perf_counter()
    for it, th in enumerate(thetas):
        table = A.RotationTable(ctx, cfg['obj_size'], th)
        first = True
        for j in range(0, n_pos, mb):
            ind = np.arange(j, min(j + mb, n_pos))
            if len(ind) < mb:
                ind = np.concatenate([ind, np.arange(mb - len(ind))])       # (a full launch; the surplus is dropped)
            eng.set_batch(pos[ind], zeros)
            if first:
                eng.rotate(truth, table, None)
                first = False
            eng.multislice(probe, want_grad=False, want_pred=True)
            n_keep = min(mb, n_pos - j)
            prj[it, j:j + n_keep] = eng._pred.view(0, (mb, Py, Px)).get()[:n_keep]
        del table
    dt = time.perf_counter() - t0
    ctx.sync()
    del eng, truth, probe
    ctx.close()
    return prj, dt


class MemSampler(object):
    def __init__(self, A, period=0.02):
        self.ctx = A.Context(0)
        self.period = period
        self.peak = 0
        self.base = None
        self.samples = []
        self._stop = False
        self.t = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        t0 = time.perf_counter()
        while not self._stop:
            f, tot = self.ctx.mem_info()
            used = tot - f
            self.peak = max(self.peak, used)
            self.samples.append((time.perf_counter() - t0, used))
            time.sleep(self.period)

    def start(self):
        f, tot = self.ctx.mem_info()
        self.base = tot - f
        self.t.start()
        return self

    def stop(self):
        self._stop = True
        self.t.join()
        return self


def parse_log(folder):
    lines = open(os.path.join(folder, 'convergence', 'loss_rank_0.txt')).read().strip().split('\n')[1:]
    rows = [l.split(',') for l in lines]
    return np.array([(int(r[0]), int(r[1]), float(r[2]), float(r[3])) for r in rows])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n-theta', type=int, default=500)
    ap.add_argument('--epochs', type=int, default=2)
    ap.add_argument('--reduced', action='store_true', help='32^3 object / 16 x 16 probe / 3 x 3 scan / 16 angles, with the oracle beside it')
    ap.add_argument('--update-scheme', default='immediate')
    ap.add_argument('--learning-rate', type=float, default=None, help="Adam step (default: the reference demo's 5e-5)")
    ap.add_argument('--out', default=None)
    args = ap.parse_args()

    import adorym_amd as A
    from adorym_amd import workloads as W, ptychography as PT
    from adorym_amd.util import epoch_task_list

    cfg = W.c3_config()
    if args.reduced:
        N, P = 32, 16
        ys = np.arange(3) * 8 - 4
        cfg.update(obj_size=(N, N, N), probe_size=(P, P), probe_pos=np.array([(y, x) for y in ys for x in ys], dtype=np.int64), minibatch_size=3,
                   n_theta=16, alpha_d=1e-9 * N ** 3, alpha_b=1e-10 * N ** 3, gamma=1e-9 * N ** 3,
                   probe=dict(probe_type='gaussian', probe_mag_sigma=3, probe_phase_sigma=3, probe_phase_max=0.5))
        args.n_theta = 16
    if args.learning_rate is not None:
        cfg['learning_rate'] = args.learning_rate
    mb = cfg['minibatch_size']
    n_theta = args.n_theta
    thetas = np.linspace(cfg['theta_st'], cfg['theta_end'], n_theta, dtype='float32')
    truth_h = W.foam_object(cfg['obj_size'], seed=0, n_bubbles=120 if not args.reduced else 12)
    prj, t_syn = synthesise(A, W, cfg, thetas, truth_h, mb)
    guess = W.random_guess(cfg['obj_size'], seed=1)
    n_pos = len(cfg['probe_pos'])
    Py, Px = cfg['probe_size']

    # the object after every epoch, captured where the driver writes delta_ds_1 / beta_ds_1 (adorym/util.py:1958-2028)
    per_epoch, cur = [], {}
    orig_write = PT.write_tiff

    def capture(arr, path, **kw):
        name = os.path.basename(path)
        if name.startswith('delta_ds_'):
            cur['delta'] = np.array(arr)
        if name.startswith('beta_ds_'):
            cur['beta'] = np.array(arr)
            per_epoch.append(quality(np.stack([cur['delta'], cur['beta']], -1), truth_h))
        return orig_write(arr, path, **kw)

    PT.write_tiff = capture
    sampler = MemSampler(A).start()
    with tempfile.TemporaryDirectory() as td, open(os.devnull, 'w') as sink, contextlib.redirect_stdout(sink):
        t0 = time.perf_counter()
        st = A.reconstruct_ptychography(
            fname=prj, obj_size=cfg['obj_size'], probe_pos=cfg['probe_pos'], theta_st=cfg['theta_st'], theta_end=cfg['theta_end'], n_theta=n_theta,
            energy_ev=cfg['energy_ev'], psize_cm=cfg['psize_cm'], free_prop_cm='inf', minibatch_size=mb, n_epochs=args.epochs,
            alpha_d=cfg['alpha_d'], alpha_b=cfg['alpha_b'], gamma=cfg['gamma'], learning_rate=cfg['learning_rate'], optimizer='adam',
            initial_guess=[guess[..., 0], guess[..., 1]], save_path=td, output_folder='c3', store_checkpoint=False, use_checkpoint=False,
            update_scheme=args.update_scheme, return_state=True, **cfg['probe'])
        wall = time.perf_counter() - t0
        log = parse_log(st['output_folder'])
    sampler.stop()
    PT.write_tiff = orig_write

    per_mb = mb if args.update_scheme == 'immediate' else -(-n_pos // mb) * mb
    epochs = []
    for e in range(args.epochs):
        rows = log[log[:, 0] == e]
        ts = rows[:, 3]
        gaps = np.diff(ts)
        batches = epoch_task_list(e, n_theta, n_pos, mb, 1, update_scheme=args.update_scheme)
        theta_of = np.array([b[0, 0] for b in batches])
        if args.update_scheme != 'immediate':
            theta_of = theta_of[np.r_[True, np.diff(theta_of) != 0]] if len(theta_of) != len(rows) else theta_of
        losses = rows[:, 2]
        per_angle = {}
        if len(theta_of) == len(losses):
            for t_, l_ in zip(theta_of, losses):
                per_angle.setdefault(int(t_), []).append(float(l_))
        order = list(dict.fromkeys(int(t_) for t_ in theta_of))
        mean_loss = [float(np.mean(per_angle[t_])) for t_ in order] if per_angle else []
        n_steps = len(rows)
        span = ts[-1] - ts[0]
        epochs.append({
            'epoch': e, 'logged_minibatches': int(n_steps), 'positions': int(n_steps * per_mb),
            'seconds_first_to_last_stamp': float(span),
            'positions_per_s': float((n_steps - 1) * per_mb / span) if span > 0 else None,
            'ms_per_minibatch_mean': float(1e3 * span / (n_steps - 1)), 'ms_per_minibatch_median': float(1e3 * np.median(gaps)),
            'ms_per_minibatch_p99': float(1e3 * np.percentile(gaps, 99)), 'ms_per_minibatch_max': float(1e3 * gaps.max()),
            'loss_first_angle': mean_loss[0] if mean_loss else None, 'loss_last_angle': mean_loss[-1] if mean_loss else None,
            'loss_mean_over_epoch': float(losses.mean()),
            'loss_per_angle_in_processing_order': [round(v, 6) for v in mean_loss],
            'quality_after_epoch': per_epoch[e] if e < len(per_epoch) else None})
    out = {
        'workload': cfg['name'] + (' [REDUCED 32^3]' if args.reduced else ''), 'n_theta': n_theta, 'positions_per_angle': n_pos,
        'minibatch': mb, 'update_scheme': args.update_scheme, 'learning_rate': cfg['learning_rate'], 'epochs': epochs,
        'quality_initial_guess': quality(guess, truth_h),
        'quality_metric': 'RMSE and Pearson correlation of delta / beta against the foam phantom, whole volume',
        'synthesis_seconds': t_syn, 'data_bytes_host': int(prj.nbytes),
        'wall_s_driver_incl_setup_and_outputs': wall,
        'device_memory': {'before_driver_bytes': int(sampler.base), 'peak_bytes': int(sampler.peak),
                          'peak_minus_before_bytes': int(sampler.peak - sampler.base), 'samples': len(sampler.samples),
                          'how': 'hipMemGetInfo (adm_mem_info) every 20 ms from a helper thread; device-wide, includes the runtime itself'},
    }
    if args.reduced:
        from oracle import adorym_oracle as O       # the checker, for the quality comparison only
        from adorym_amd.util import initialize_probe
        pr, pi = initialize_probe(cfg['probe_size'], **cfg['probe'])
        probe = np.squeeze(pr) + 1j * np.squeeze(pi)
        phys = O.Physics(cfg['probe_size'], cfg['energy_ev'], cfg['psize_cm'], free_prop_cm='inf')
        kw = dict(n_epochs=args.epochs, minibatch_size=mb, optimizer='adam', learning_rate=cfg['learning_rate'], alpha_d=cfg['alpha_d'],
                  alpha_b=cfg['alpha_b'], gamma=cfg['gamma'], update_scheme=args.update_scheme)
        g64 = (guess[..., 0].astype(np.float64), guess[..., 1].astype(np.float64))
        x64 = O.reconstruct(prj.astype(np.float64), g64, probe, cfg['probe_pos'].astype(float), thetas, phys, dtype='float64', **kw)
        x32 = O.reconstruct(prj, (guess[..., 0], guess[..., 1]), probe, cfg['probe_pos'].astype(float), thetas, phys, dtype='float32', **kw)
        xs = np.stack([st['delta'], st['beta']], -1)
        out['oracle'] = {'fp64_quality': quality(x64, truth_h), 'fp32_quality': quality(x32, truth_h), 'product_quality': quality(xs, truth_h),
                         'product_vs_fp64_delta_rmse': float(np.sqrt(np.mean((xs[..., 0] - x64[..., 0]) ** 2))),
                         'oracle_fp32_vs_fp64_delta_rmse': float(np.sqrt(np.mean((x32[..., 0] - x64[..., 0]) ** 2)))}
    txt = json.dumps(out)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        open(args.out, 'w').write(txt + '\n')
    brief = {k: v for k, v in out.items() if k != 'epochs'}
    brief['epochs'] = [{k: v for k, v in e.items() if k != 'loss_per_angle_in_processing_order'} for e in epochs]
    print(json.dumps(brief, indent=1))


if __name__ == '__main__':
    main()
