#!/usr/bin/env python3
"""
Golden F14: the REFERENCE driver run at world size 2 (`mpirun -n 2` semantics, adorym/ptychography.py:786,846,905-909,
1113-1125; adorym/optimizers.py:1022-1032), for BASELINE config 4's exchange step.  Runs ONLY in the development
container (needs /root/reference); only the resulting F14_world2.npz travels.

mpi4py is absent here, and the reference's own fallback (adorym/pseudo.py) is a 1-rank identity.  So two PROCESSES are
started, each importing the reference behind the same I/O shims as gen_goldens.py, with a stand-in `mpi4py` module whose
COMM_WORLD moves pickled objects over a multiprocessing.connection socket: Get_rank / Get_size / bcast / Bcast /
allreduce (sum in rank order, which is what MPI's object allreduce with the default op does) / Barrier.  No arithmetic
of the reference is replaced: the stand-in only transports and adds what the reference hands it.

    python tests/golden/gen_f14_world2.py            # writes tests/golden/F14_world2.npz
"""
import os
import subprocess
import sys
import types
import tempfile
import pickle
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
WORLD = 2

sys.path.insert(0, HERE)
import cases  # noqa: E402

# With all 9 positions (minibatch 3, 2 ranks: global batch 6) a global batch can STRADDLE two angles; the reference then
# evaluates `is_last_batch_of_this_theta` with each rank's own angle (adorym/ptychography.py:910), the ranks' optimiser
# counters i_opt_batch drift apart (:1266-1271) and so do their object replicas: run 'immediate' records that behaviour.
# With 6 positions every global batch is one angle and the replicas stay identical: the runs the product is compared to.
RUNS = cases.W2_RUNS


# ------------------------------------------------------------------------------------------------ the stand-in comm
class SocketComm(object):
    """COMM_WORLD of two ranks over one connection (rank 0 listens)."""

    def __init__(self, rank, port):
        from multiprocessing.connection import Listener, Client
        self.rank = rank
        if rank == 0:
            self._l = Listener(('127.0.0.1', port), authkey=b'f14')
            self.c = self._l.accept()
        else:
            import time
            for _ in range(600):
                try:
                    self.c = Client(('127.0.0.1', port), authkey=b'f14')
                    break
                except (ConnectionRefusedError, OSError):
                    time.sleep(0.1)
        self.record = {}

    # plain pickle, as mpi4py's object collectives do (multiprocessing's own pickler would try to share tensor storage
    # through file descriptors)
    def _send(self, a):
        self.c.send_bytes(pickle.dumps(a, protocol=pickle.HIGHEST_PROTOCOL))

    def _recv(self):
        return pickle.loads(self.c.recv_bytes())

    def Get_rank(self):
        return self.rank

    def Get_size(self):
        return WORLD

    def Barrier(self):
        self._send('b')
        assert self._recv() == 'b'

    def bcast(self, a, root=0):
        assert root == 0
        if self.rank == 0:
            self._send(a)
            return a
        return self._recv()

    def Bcast(self, a, root=0):
        assert root == 0
        if self.rank == 0:
            self._send(np.array(a))
        else:
            a[...] = self._recv()
        return a

    def allreduce(self, a):
        # rank 0 forms a_0 + a_1 (rank order) and returns the same object to both
        if self.rank == 0:
            other = self._recv()
            tot = a + other
            self._send(tot)
        else:
            self._send(a)
            tot = self._recv()
        if 'first_allreduce' not in self.record and hasattr(tot, 'shape') and len(tot.shape) == 4:
            t = tot.detach().numpy() if hasattr(tot, 'detach') else np.asarray(tot)
            self.record['first_allreduce'] = t.copy()
        return tot


def worker(rank, port, workdir):
    mpi = types.ModuleType('mpi4py')
    comm = SocketComm(rank, port)
    mpi.MPI = types.SimpleNamespace(COMM_WORLD=comm)
    sys.modules['mpi4py'] = mpi
    sys.modules['mpi4py.MPI'] = mpi.MPI
    sys.path.insert(0, HERE)
    import gen_goldens as GG            # I/O shims + reference import (now with the stand-in mpi4py)
    import cases
    import adorym.ptychography as PT
    assert PT.MPI.COMM_WORLD is comm
    g6 = np.load(os.path.join(HERE, 'F6_e2e.npz'))
    prj = g6['prj'].astype(np.float64)
    inp = cases.e2e_inputs()
    E = cases.E2E
    N = E['N']
    common = dict(minibatch_size=E['minibatch_size'], initial_guess=[inp['guess'][0], inp['guess'][1]],
                  probe_type='supplied', probe_initial=[inp['probe_mag'], inp['probe_phase']])
    out = {}
    os.chdir(workdir)
    for rn, (n_use, extra) in RUNS.items():
        for fp64 in (True, False):
            rec = {}
            comm.record = {}
            ex = dict(common); ex.update(extra); ex['run_float64'] = fp64
            ex['output_folder'] = 'out_%s_%d' % (rn, int(fp64))
            run_driver(GG, PT, prj[:, :n_use], [N, N, N], inp['probe_pos'][:n_use], 2 * np.pi, E['n_theta'], ex, rec, rank)
            tag = '%s_%s' % (rn, '64' if fp64 else '32')
            out['r%d_theta_%s' % (rank, tag)] = np.array([b[0] for b in rec['batches']])
            out['r%d_ind_%s' % (rank, tag)] = np.stack([b[1] for b in rec['batches']])
            out['r%d_losses_%s' % (rank, tag)] = rec['losses']
            if rank == 0:
                out['delta_' + tag] = rec['delta'].astype(np.float64 if fp64 else np.float32)
                out['beta_' + tag] = rec['beta'].astype(np.float64 if fp64 else np.float32)
                if 'first_allreduce' in comm.record and rn in ('immediate', 'immediate6_reg'):
                    out['first_grad_sum_' + tag] = comm.record['first_allreduce'].astype(np.float64 if fp64 else np.float32)
                if rn == 'probe6':
                    out['probe_mag_' + tag] = rec['probe_mag']
                    out['probe_phase_' + tag] = rec['probe_phase']
    with open(os.path.join(workdir, 'rank%d.pkl' % rank), 'wb') as f:
        pickle.dump(out, f)
    comm.Barrier()


def run_driver(GG, PT, prj, obj_size, probe_pos, theta_end, n_theta, extra, record, rank):
    """gen_goldens.run_driver for one rank of two: a SHARED output folder (the reference creates it on rank 0 and every
    rank writes its own convergence/loss_rank_{r}.txt into it)."""
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
        params = dict(fname='data.h5', obj_size=obj_size, probe_pos=probe_pos, theta_st=0, theta_end=theta_end, n_theta=n_theta,
                      energy_ev=cases.ENERGY_EV, psize_cm=cases.PSIZE_CM, free_prop_cm='inf', save_path='.', use_checkpoint=False,
                      store_checkpoint=False, save_intermediate=False, cpu_only=True, backend='pytorch', gamma=0, alpha_d=0,
                      alpha_b=0, n_dp_batch=20, shared_probe_among_angles=True)
        params.update(extra)
        PT.reconstruct_ptychography(**params)
        with open(os.path.join(params['output_folder'], 'convergence', 'loss_rank_%d.txt' % rank)) as f:
            lines = f.read().strip().split('\n')[1:]
        record['losses'] = np.array([float(l.split(',')[2]) for l in lines])
        if rank == 0:
            T = GG.TIFFS
            record['delta'] = T[[k for k in T if k.endswith('delta_ds_1')][0]].copy()
            record['beta'] = T[[k for k in T if k.endswith('beta_ds_1')][0]].copy()
            record['probe_mag'] = T[[k for k in T if k.endswith('probe_mag_ds_1')][0]].copy()
            record['probe_phase'] = T[[k for k in T if k.endswith('probe_phase_ds_1')][0]].copy()
    finally:
        DF.Differentiator.get_gradients = orig_get


def main():
    import socket
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    with tempfile.TemporaryDirectory() as td:
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--rank', str(r), '--port', str(port), '--dir', td])
                 for r in range(WORLD)]
        rcs = [p.wait() for p in procs]
        assert rcs == [0] * WORLD, rcs
        out = {}
        for r in range(WORLD):
            with open(os.path.join(td, 'rank%d.pkl' % r), 'rb') as f:
                out.update(pickle.load(f))
    path = os.path.join(HERE, 'F14_world2.npz')
    np.savez_compressed(path, **out)
    print('wrote F14_world2 %.1f KB, %d arrays' % (os.path.getsize(path) / 1024, len(out)))


if __name__ == '__main__':
    if '--rank' in sys.argv:
        a = sys.argv
        worker(int(a[a.index('--rank') + 1]), int(a[a.index('--port') + 1]), a[a.index('--dir') + 1])
    else:
        main()
