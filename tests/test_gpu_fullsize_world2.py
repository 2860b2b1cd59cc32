"""BASELINE config 4's exchange at config 4's size: the PRODUCT at world size 2 on config 3's shape -- 256^3 object (16.8 M-element
shards at 8 ranks, 67 M here), 72 x 72 probe, 256 slices, L1 + TV, Adam, minibatch 32 per rank -- two 'immediate' updates of a
global batch of 64 positions each (reference semantics: adorym/ptychography.py:786, 905-912 rank split, :1113-1129 summed
gradients + update, forward_model.py:138-139 one regulariser term per rank).  Two fresh processes share GPU 0 through the
host-staged transport (RCCL refuses two ranks on one device); kernels, sharded optimiser and driver are the product's.

Checked: (a) the two replicas hold the same bits; (b) the result equals, bit for bit, the same two global batches run on ONE
rank as a serial sum of the ranks' gradient buffers (the one-rank run on the same global batch: same kernels, rank-order
sum, one full-range Adam step per update); (c) against the fp64 oracle's 2-rank run under the RMSE < 1e-5 / 3x / sign-flip
rules of tests/test_gpu_fullsize.py."""
import multiprocessing as mp
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import cases
from oracle import adorym_oracle as O      # checker only

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _driver_kwargs(cfg, inp, prj, tmp):
    N = cases.FULLSIZE['N']
    g0 = inp['guess']
    return dict(fname=prj, obj_size=[N] * 3, probe_pos=inp['pos'], theta_st=float(inp['theta']), theta_end=float(inp['theta']), n_theta=1,
                energy_ev=cfg['energy_ev'], psize_cm=cfg['psize_cm'], free_prop_cm='inf', minibatch_size=cfg['minibatch_size'], n_epochs=1,
                initial_guess=[g0[..., 0], g0[..., 1]], optimizer='adam', learning_rate=cfg['learning_rate'], alpha_d=cfg['alpha_d'],
                alpha_b=cfg['alpha_b'], gamma=cfg['gamma'], update_scheme='immediate', save_path=tmp, output_folder='out',
                store_checkpoint=False, use_checkpoint=False, return_state=True, **cfg['probe'])


def _worker(rank, world, port, prj_path, tmp, q, transport):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    sys.path.insert(0, HERE)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), ADM_COMM=transport)
    try:
        import adorym_amd as A
        from adorym_amd import comm as C
        import fullsize_oracle as F
        cfg, inp, _, _ = F.setup(1)
        comm = C.from_env()
        assert type(comm) is {'host': C.HostStagedComm, 'p2p': C.P2PComm}[transport] and comm.size == world
        st = A.reconstruct_ptychography(comm=comm, **_driver_kwargs(cfg, inp, np.load(prj_path).astype(np.float32), tmp))
        np.save(os.path.join(tmp, 'rank%d.npy' % rank), np.stack([st['delta'], st['beta']], -1))
        comm.close()
        q.put(dict(rank=rank, losses=st['losses']))
    except Exception as e:
        import traceback
        q.put(dict(rank=rank, error='%r\n%s' % (e, traceback.format_exc())))


def _serial_one_rank(A, ctx, cfg, inp, prj, world):
    """The same global batches on ONE context: per update, every rank's buffer = regulariser term ('set' mode) + its minibatch's
    back-rotated gradient; buffers added in rank order in fp32 on the host (what the transport's reduce-scatter does); one
    full-range Adam step.  adorym/ptychography.py:1113-1129 for `mpirun -n 2`, executed by one process."""
    from adorym_amd.dp import HipOps
    from adorym_amd._lib import check
    from adorym_amd.util import epoch_task_list, rank_batch
    N, P = cases.FULLSIZE['N'], cases.FULLSIZE['P']
    size = (N, N, N)
    pos = np.round(inp['pos']).astype(int)
    mb = cfg['minibatch_size']
    eng = A.MultisliceEngine(ctx, size, (P, P), pos, cfg['energy_ev'], cfg['psize_cm'], free_prop_cm='inf', max_batch=mb)
    obj = ctx.array(inp['guess'].astype(np.float32))
    from adorym_amd.workloads import probe_array
    probe = ctx.array(probe_array(cfg)[None])
    table = A.RotationTable(ctx, size, inp['theta'])
    batches = epoch_task_list(0, 1, len(pos), mb, world)
    n = obj.size
    m, v = ctx.zeros((n,)), ctx.zeros((n,))
    g = ctx.empty(obj.shape)
    i_opt = 0
    for i_batch in range(len(batches)):
        total = None
        for r in range(world):
            _, ind = rank_batch(batches, i_batch, r, mb, world)
            check(ctx.lib.adm_reg_grad_set(eng.plan.handle, obj.ptr, cfg['alpha_d'], cfg['alpha_b'], cfg['gamma'], g.ptr, None))
            eng.loss_and_grad(obj, g, table, probe, pos[ind], np.abs(prj[0, ind]).astype(np.float32))
            total = g.get() if total is None else total + g.get()
        gsum = ctx.array(total)
        HipOps(ctx).adam(obj, gsum, 0, m, v, 0, 0, n, i_opt, cfg['learning_rate'], 0.9, 0.999, 1e-7, 0, None)
        gsum.free()
        if i_batch == len(batches) - 1:
            i_opt += 1
    return obj.get(), len(batches)


@pytest.mark.parametrize('transport,world', [('host', 2), ('p2p', 2), ('p2p', 4)])
def test_world2_exchange_at_config4_size(tmp_path, transport, world):
    """transport 'host': every collective staged through host memory; 'p2p': the direct exchange (adm_p2p.hip) -- each rank's fused
    kernel reads the other ranks' 134 MB gradient buffers and writes their objects through IPC mappings, at 2 and at 4 ranks
    (4 ranks: one global batch of 128 positions, the 69 of the scan topped up from its start as the reference does)."""
    sys.path.insert(0, HERE)
    import fullsize_oracle as F
    import adorym_amd as A
    cfg, inp, probe, phys = F.setup(1)
    prj = F.measured(inp, probe, phys)
    prj_path = str(tmp_path / 'prj.npy')
    np.save(prj_path, prj)
    # the fp64 / fp32 oracle runs of `mpirun -n world` beside everything else
    orc = {dt: subprocess.Popen([sys.executable, os.path.join(HERE, 'fullsize_oracle.py'), str(tmp_path / ('o_%s.npy' % dt)), 'immediate', dt,
                                 prj_path, '1', str(world)]) for dt in ('float64', 'float32')}
    try:
        os.environ['ADM_RDV_TOKEN'] = __import__('secrets').token_hex(16)      # this job's secret: other jobs on the machine are not admitted
        mpc = mp.get_context('spawn')
        q = mpc.Queue()
        port = _free_port()
        procs = [mpc.Process(target=_worker, args=(r, world, port, prj_path, str(tmp_path), q, transport)) for r in range(world)]
        [p.start() for p in procs]
        res = [q.get(timeout=900) for _ in procs]
        [p.join(120) for p in procs]
        for r in res:
            assert 'error' not in r, r['error']
        x0r = np.load(tmp_path / 'rank0.npy')
        # (a) one sharded object: after the all-gather every replica holds the same bits
        for r_ in range(1, world):
            assert np.array_equal(x0r, np.load(tmp_path / ('rank%d.npy' % r_)))
        assert np.all(np.isfinite(x0r))
        # (b) the one-rank run on the same global batches, same kernels, rank-order sum: bit for bit
        ctx = A.Context(0)
        serial, n_updates = _serial_one_rank(A, ctx, cfg, inp, prj, world)
        ctx.close()
        assert n_updates == (2 if world == 2 else 1)       # 69 positions -> two global batches of 64, or one of 128
        nd = int((serial != x0r).sum())
        print('world %d (%s) at 256^3: %d of %d voxels differ from the serial one-rank sum' % (world, transport, nd, serial.size))
        assert nd == 0
        # (c) the fp64 oracle's 2-rank run
        for dt, p in orc.items():
            assert p.wait(timeout=1500) == 0, 'oracle %s failed' % dt
        k = n_updates
        s0, s1 = inp['s0'], inp['s1']
        x64 = np.load(tmp_path / 'o_float64.npy').astype(np.float64)[k:-k]
        x32 = np.load(tmp_path / 'o_float32.npy').astype(np.float64)[k:-k]
        xs, g0 = x0r[s0 + k:s1 - k].astype(np.float64), inp['guess'][s0 + k:s1 - k]
        lr = cfg['learning_rate']
        upd = np.linalg.norm(x64 - g0)
        d, d32 = np.abs(xs - x64), np.abs(x32 - x64)
        rmse = np.sqrt(np.mean((xs[..., 0] - x64[..., 0]) ** 2))
        fl, fl32 = d > 0.5 * lr, d32 > 0.5 * lr
        e_us, e_ref = np.linalg.norm((xs - x64)[~fl]), np.linalg.norm((x32 - x64)[~fl32])
        print('   vs fp64 oracle (same number of ranks): delta RMSE %.2e; |x-x64|/|update| %.2e (oracle fp32 %.2e); voxels off by > lr/2: %d (oracle fp32: %d) of %d'
              % (rmse, e_us / upd, e_ref / upd, fl.sum(), fl32.sum(), d.size))
        assert upd > 50 * lr and rmse < 1e-5
        assert fl.sum() <= 3 * fl32.sum() + 1e-4 * d.size
        assert e_us <= 3 * e_ref + 1e-4 * upd, (e_us, e_ref, upd)
    finally:
        for p in orc.values():
            if p.poll() is None:
                p.kill()
