#!/usr/bin/env python3
"""
Golden F18w: the REFERENCE driver at world size 2 on multi-distance data divided into sub-tiles (MultiDistModel with n_blocks > 1
and a safe zone, adorym/forward_model.py:884-1034; rank split and summed gradients adorym/ptychography.py:786,846,905-909,
1113-1125).  Same machinery as gen_f14_world2.py -- two processes, the reference behind the I/O shims, a stand-in mpi4py that
only transports and adds -- on run 'ri_szw4' of cases.C5TILES with minibatch 2 per rank (global batch 4 tiles).  Runs ONLY in the
development container; only F18_world2.npz travels.

    python tests/golden/gen_f18_world2.py
"""
import os
import pickle
import subprocess
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_f14_world2 as W  # noqa: E402

RUN, MB = 'ri_szw4', 2


def worker(rank, port, workdir):
    mpi = types.ModuleType('mpi4py')
    comm = W.SocketComm(rank, port)
    mpi.MPI = types.SimpleNamespace(COMM_WORLD=comm)
    sys.modules['mpi4py'] = mpi
    sys.modules['mpi4py.MPI'] = mpi.MPI
    import gen_goldens as GG
    import cases
    import adorym
    import adorym.ptychography as PT
    assert PT.MPI.COMM_WORLD is comm
    C = cases.C5TILES
    inp = cases.c5tiles_inputs(RUN)
    N = C['N']
    prj = np.load(os.path.join(HERE, 'F18_multidist_tiles.npz'))[RUN + '_prj'].astype(np.float64)

    class MultiDistPlugin(adorym.MultiDistModel):
        def __init__(self, *a, run_bfloat16=False, run_float64=False, **k):
            super().__init__(*a, **k)

    out = {}
    os.chdir(workdir)
    for fp64 in (True, False):
        rec = {}
        comm.record = {}
        extra = dict(minibatch_size=MB, n_epochs=C['n_epochs'], two_d_mode=True, energy_ev=C['energy_ev'], psize_cm=C['psize_cm'],
                     free_prop_cm=np.array(C['dists_cm']), initial_guess=[inp['guess'][0], inp['guess'][1]], probe_type='plane',
                     raw_data_type='magnitude', unknown_type='real_imag', optimizer='adam', learning_rate=C['learning_rate'],
                     randomize_probe_pos=False, safe_zone_width=inp['szw'], forward_model=MultiDistPlugin, run_float64=fp64,
                     output_folder='out_%d' % int(fp64))
        run_driver(GG, PT, prj, [N, N, 1], inp['pos'], extra, rec, rank)
        tag = '64' if fp64 else '32'
        out['r%d_ind_%s' % (rank, tag)] = np.stack([b[1] for b in rec['batches']])
        out['r%d_losses_%s' % (rank, tag)] = rec['losses']
        if rank == 0:
            out['obj_' + tag] = np.stack([rec['mag'] * np.cos(rec['phase']), rec['mag'] * np.sin(rec['phase'])], -1)
            out['first_grad_sum_' + tag] = comm.record['first_allreduce'].astype(np.float64 if fp64 else np.float32)
    with open(os.path.join(workdir, 'rank%d.pkl' % rank), 'wb') as f:
        pickle.dump(out, f)
    comm.Barrier()


def run_driver(GG, PT, prj, obj_size, probe_pos, extra, record, rank):
    import adorym.differentiator as DF
    import cases
    GG.STORE['data.h5'] = {'exchange/data': prj}
    orig_get = DF.Differentiator.get_gradients

    def rec_get(self, **kw):
        g = orig_get(self, **kw)
        record.setdefault('batches', []).append((int(kw['this_i_theta']), np.array(kw['this_ind_batch'])))
        return g

    DF.Differentiator.get_gradients = rec_get
    GG.TIFFS.clear()
    try:
        params = dict(fname='data.h5', obj_size=obj_size, probe_pos=probe_pos, theta_st=0, theta_end=0, n_theta=1, save_path='.',
                      use_checkpoint=False, store_checkpoint=False, save_intermediate=False, cpu_only=True, backend='pytorch', gamma=0,
                      alpha_d=0, alpha_b=0, n_dp_batch=20, shared_probe_among_angles=True)
        params.update(extra)
        PT.reconstruct_ptychography(**params)
        with open(os.path.join(params['output_folder'], 'convergence', 'loss_rank_%d.txt' % rank)) as f:
            lines = f.read().strip().split('\n')[1:]
        record['losses'] = np.array([float(l.split(',')[2]) for l in lines])
        if rank == 0:
            T = GG.TIFFS
            record['mag'] = T[[k for k in T if k.endswith('obj_mag_ds_1')][0]].copy()
            record['phase'] = T[[k for k in T if k.endswith('obj_phase_ds_1')][0]].copy()
    finally:
        DF.Differentiator.get_gradients = orig_get


def main():
    import socket
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    with tempfile.TemporaryDirectory() as td:
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--rank', str(r), '--port', str(port), '--dir', td])
                 for r in range(W.WORLD)]
        rcs = [p.wait() for p in procs]
        assert rcs == [0] * W.WORLD, rcs
        out = {}
        for r in range(W.WORLD):
            with open(os.path.join(td, 'rank%d.pkl' % r), 'rb') as f:
                out.update(pickle.load(f))
    path = os.path.join(HERE, 'F18_world2.npz')
    np.savez_compressed(path, **out)
    print('wrote F18_world2 %.1f KB, %d arrays' % (os.path.getsize(path) / 1024, len(out)))


if __name__ == '__main__':
    if '--rank' in sys.argv:
        a = sys.argv
        worker(int(a[a.index('--rank') + 1]), int(a[a.index('--port') + 1]), a[a.index('--dir') + 1])
    else:
        main()
