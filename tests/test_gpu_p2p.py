"""The peer-to-peer transport (adorym_amd/csrc/adm_p2p.hip, comm.P2PComm) by itself, through the C ABI: R fresh processes share
GPU 0, map each other's buffers (IPC handles over the TCP star) and run the fused exchange

    gradient.arr = comm.allreduce(gradient.arr); obj.arr = opt.apply_gradient(...); constraints; mask
    (adorym/ptychography.py:1113-1158, optimizers.py:309-318 / 376-411 / 440-464, array_ops.py:239-251)

as ONE kernel per rank.  The check is the same arithmetic on ONE context: the ranks' buffers added in rank order in fp32
(NumPy), then the one-rank optimiser kernel (adm_adam_step / adm_gd_step / adm_momentum_step) over the whole array -- bit for
bit, on every replica, for shard boundaries that are and are not multiples of four elements, with constraints and mask, and
with the sum restricted to a sub-range (footprint-restricted exchange)."""
import multiprocessing as mp
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _inputs(n, world, seed):
    r = np.random.default_rng(seed)
    x0 = r.standard_normal(n).astype(np.float32) * 1e-3
    gs = [r.standard_normal(n).astype(np.float32) for _ in range(world)]
    mask = (r.uniform(size=n // 2 + 1) > 0.25).astype(np.float32)
    return x0, gs, mask


def _rank(rank, world, port, case, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), ADM_COMM='p2p')
    os.environ.update(case.get('env', {}))
    try:
        import adorym_amd as A
        from adorym_amd import comm as C, _lib
        from adorym_amd.dp import DataParallelObject, HipOps
        comm = C.from_env()
        assert type(comm) is C.P2PComm
        ctx = A.Context(comm.device_index)
        comm.attach(ctx)
        n = case['n']
        st = DataParallelObject(HipOps(ctx), comm, (n,))
        x0, gs, mask_h = _inputs(n, world, case['seed'])
        mask = ctx.array(mask_h) if case.get('mask') else None
        st.obj.view(0, (n,)).set(x0)
        out = {'rank': rank, 'lo': st.lo, 'hi': st.hi, 'steps': []}
        small = ctx.array(np.arange(case.get('small', 5), dtype=np.float32) * (rank + 1) + np.float32(0.1) * rank)
        for k in range(case['steps']):
            if case.get('skip_rank') == rank and k == case.get('skip_step', 0):
                break                   # this rank leaves the sequence: its peers must time out cleanly
            g = gs[rank] * np.float32(1 + k)
            st.grad.view(0, (n,)).set(g)
            kw = {}
            if case.get('touched'):
                t_lo, t_hi = case['touched']

                def reg_shard(lo, hi, a_lo, a_hi, k=k):
                    # the owner completes ITS buffer on its shard: data term outside the touched range is absent (written), inside
                    # the range a rank-independent term is added
                    cur = st.grad.view(0, (n,)).get()
                    reg = (np.arange(n, dtype=np.float32) % 7 - 3) * np.float32(0.01 * (k + 1))
                    i = np.arange(n)
                    own = (i >= lo) & (i < hi)
                    ins = (i >= a_lo) & (i < a_hi)
                    cur[own & ins] += reg[own & ins]
                    cur[own & ~ins] = reg[own & ~ins]
                    st.grad.view(0, (n,)).set(cur)
                kw = dict(touched=(t_lo, t_hi), reg_shard=reg_shard)
            st.exchange_and_update(case['opt'], k, dict(case['options']), flags=case.get('flags', 0), mask=mask, **kw)
            comm.all_reduce_device(small)
            ctx.sync()
            comm.check_status()
            out['steps'].append(st.obj.view(0, (n,)).get())
        out['small'] = small.get()
        out['moments'] = [m_.get() for m_ in st.moments]
        err = None
        try:
            comm.barrier()
            comm.close()
        except Exception as e:
            err = repr(e)
        out['close_error'] = err
        q.put(out)
    except Exception as e:
        import traceback
        q.put(dict(rank=rank, error='%r\n%s' % (e, traceback.format_exc())))


def _run(world, case, timeout=300):
    os.environ['ADM_RDV_TOKEN'] = __import__('secrets').token_hex(16)      # this job's secret: other jobs on the machine are not admitted
    mpc = mp.get_context('spawn')
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_rank, args=(r, world, port, case, q)) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=timeout) for _ in procs]
    [p.join(60) for p in procs]
    return sorted(res, key=lambda r: r['rank'])


def _serial(world, case):
    """The same updates on ONE context: rank-order fp32 sum on the host, one-rank optimiser kernel over the whole array."""
    import adorym_amd as A
    from adorym_amd.dp import HipOps
    ctx = A.Context(0)
    ops = HipOps(ctx)
    n = case['n']
    x0, gs, mask_h = _inputs(n, world, case['seed'])
    mask = ctx.array(mask_h) if case.get('mask') else None
    x = ctx.array(x0)
    m, v = ctx.zeros((n,)), ctx.zeros((n,))
    o = case['options']
    outs = []
    per = -(-n // (2 * world)) * 2
    for k in range(case['steps']):
        bufs = [g_ * np.float32(1 + k) for g_ in gs]
        if case.get('touched'):
            t_lo, t_hi = case['touched']
            reg = (np.arange(n, dtype=np.float32) % 7 - 3) * np.float32(0.01 * (k + 1))
            i = np.arange(n)
            ins = (i >= t_lo) & (i < t_hi)
            total = np.zeros(n, np.float32)
            for r in range(world):              # element i belongs to rank i // per: inside the range all ranks' data terms, rank order,
                own = (i >= r * per) & (i < (r + 1) * per)      # with the owner's buffer carrying the extra term; outside only the term
                b = [bq.copy() for bq in bufs]
                b[r][own & ins] += reg[own & ins]
                acc = b[0].copy()
                for q_ in range(1, world):
                    acc = acc + b[q_]
                total[own & ins] = acc[own & ins]
                total[own & ~ins] = reg[own & ~ins]
        else:
            total = bufs[0].copy()
            for q_ in range(1, world):
                total = total + bufs[q_]
        g = ctx.array(total)
        if case['opt'] == 'adam':
            ops.adam(x, g, 0, m, v, 0, 0, n, k, o.get('step_size', 0.001), o.get('b1', 0.9), o.get('b2', 0.999), o.get('eps', 1e-7),
                     case.get('flags', 0), mask)
        elif case['opt'] == 'gd':
            ops.gd(x, g, 0, 0, n, o['step_size'], case.get('flags', 0), mask)
        else:
            ops.momentum(x, g, 0, m, 0, 0, n, o.get('step_size', 0.001), o.get('gamma', 0.9), case.get('flags', 0), mask)
        outs.append(x.get())
        g.free()
    mom = [m.get(), v.get()]
    ctx.close()
    return outs, mom


CASES = {
    # n chosen so that the shard boundaries are / are not multiples of 4 elements (vector and scalar paths of the kernel)
    'adam_w2_aligned': (2, dict(n=2 * 4096, opt='adam', options={'step_size': 1e-3}, steps=3, seed=1)),
    'adam_w2_odd_shards': (2, dict(n=2 * 4098, opt='adam', options={'step_size': 1e-3}, steps=3, seed=2)),
    'adam_w4_constraints_mask': (4, dict(n=2 * 30011, opt='adam', options={'step_size': 1e-3}, steps=2, seed=3, flags=1, mask=True)),
    'adam_w3_tail': (3, dict(n=2 * 5003, opt='adam', options={'step_size': 1e-3}, steps=2, seed=4, flags=4)),
    'gd_w4': (4, dict(n=2 * 8192, opt='gd', options={'step_size': 1e-2}, steps=2, seed=5, flags=2, mask=True)),
    'momentum_w2': (2, dict(n=2 * 6002, opt='momentum', options={'step_size': 1e-2, 'gamma': 0.8}, steps=3, seed=6)),
    'adam_w4_touched': (4, dict(n=2 * 16384, opt='adam', options={'step_size': 1e-3}, steps=2, seed=7, touched=(5000, 23002))),
    'adam_w8_large': (8, dict(n=2 * (1 << 20), opt='adam', options={'step_size': 1e-3}, steps=2, seed=8)),
}


@pytest.mark.parametrize('name', sorted(CASES))
def test_fused_exchange_equals_serial_rank_order_sum(name):
    world, case = CASES[name]
    res = _run(world, case)
    for r in res:
        assert 'error' not in r, r['error']
        assert r['close_error'] is None, r['close_error']
    want, mom = _serial(world, case)
    n = case['n']
    for r in res:
        for k in range(case['steps']):
            nd = int((r['steps'][k] != want[k]).sum())
            assert nd == 0, '%s: rank %d step %d: %d of %d elements differ from the serial rank-order sum' % (name, r['rank'], k, nd, n)
        # the moments are the rank's shard of the serial run's
        if case['opt'] != 'gd':
            lo, hi = r['lo'], max(r['lo'], r['hi'])
            assert np.array_equal(r['moments'][0][:hi - lo], mom[0][lo:hi])
            if case['opt'] == 'adam':
                assert np.array_equal(r['moments'][1][:hi - lo], mom[1][lo:hi])
        # small all-reduce through the mailboxes: rank-order sum, applied once per step (each step sums the previous result R-fold)
        ns = case.get('small', 5)
        s = [np.arange(ns, dtype=np.float32) * (q + 1) + np.float32(0.1) * q for q in range(world)]
        for _ in range(case['steps']):
            acc = s[0].copy()
            for q in range(1, world):
                acc = acc + s[q]
            s = [acc.copy() for _ in range(world)]
        assert np.array_equal(r['small'], s[0])
    assert np.abs(want[-1] - want[0]).max() > 0


def test_a_peer_that_leaves_the_sequence_is_a_clean_error():
    """Rank 1 stops before the second update.  Rank 0's wait kernel gives up after ADM_P2P_TIMEOUT_S, the fused kernel behind it
    touches nothing, and the next status check raises naming the rank that did not arrive -- no hang, the GPU stays usable."""
    case = dict(n=2 * 4096, opt='adam', options={'step_size': 1e-3}, steps=2, seed=9, skip_rank=1, skip_step=1, env={'ADM_P2P_TIMEOUT_S': '1.5'})
    os.environ['ADM_RDV_TOKEN'] = __import__('secrets').token_hex(16)      # this job's secret: other jobs on the machine are not admitted
    mpc = mp.get_context('spawn')
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_rank, args=(r, 2, port, case, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r['rank'])
    [p.join(60) for p in procs]
    r0, r1 = res
    assert 'error' in r0 and "rank 1 did not signal 'gradients ready' to rank 0" in r0['error'], r0
    assert 'error' not in r1 and len(r1['steps']) == 1
    # the GPU is fine afterwards
    world, c2 = CASES['adam_w2_aligned']
    again = _run(world, c2)
    assert all('error' not in r for r in again)
