"""The framework-free control plane of the multi-GPU path (adorym_amd/rendezvous.py), three processes on localhost; replaces
the reference's mpi4py seam (adorym/ptychography.py:39-50: rank / size / bcast / Barrier) and, for the host-staged
validation transport, carries whole arrays."""
import multiprocessing as mp
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, env, q, scenario):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1')
    os.environ.update(env)
    try:
        from adorym_amd.rendezvous import TcpGroup
        g = TcpGroup.from_env()
        out = {'rank': rank, 'size': g.size}
        if scenario == 'mismatch':
            try:
                if rank == 0:
                    g.sum_over_ranks(1.0)
                else:
                    g.max_over_ranks(1.0)
                out['raised'] = False
            except RuntimeError as e:
                out['raised'] = 'mismatch' in str(e)
            except Exception:
                out['raised'] = True          # the peer closed first: also not a silent hang
            g.close()
            q.put(out)
            return
        g.barrier()
        out['sum'] = g.sum_over_ranks(rank + 1.0)
        out['max'] = g.max_over_ranks(10.0 - rank)
        out['bc0'] = g.bcast_object({'ids': [b'\x01' * 128, b'\x02' * 128]} if rank == 0 else None, root=0)
        out['bc2'] = g.bcast_object('from two' if rank == 2 else None, root=2)
        a = (np.arange(12, dtype=np.float32).reshape(3, 4) + 100 * rank)
        out['allreduce'] = g.all_reduce_sum(a.copy())
        out['reduce1'] = g.reduce_sum(a.copy(), 1)
        b = np.full(5, rank, np.float32)
        out['bcast_arr'] = g.broadcast(b.copy() if rank == 1 else np.zeros(5, np.float32), 1)
        out['gather'] = g.all_gather(np.full(4, rank, np.float32))
        big = np.random.default_rng(rank).standard_normal(1 << 20).astype(np.float32)      # 4 MB: many TCP segments
        out['big'] = float(np.abs(g.all_reduce_sum(big.copy())).sum())
        g.barrier()
        g.close()
        q.put(out)
    except Exception as e:
        import traceback
        q.put({'rank': rank, 'error': '%r\n%s' % (e, traceback.format_exc())})


def _run(world, env, scenario='all'):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, env, q, scenario)) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=120) for _ in procs]
    [p.join(60) for p in procs]
    for r in res:
        assert 'error' not in r, r['error']
    return sorted(res, key=lambda r: r['rank'])


@pytest.mark.parametrize('mode', ['master_port', 'torchrun_like', 'exact_port'])
def test_tcp_group_collectives_world3(mode):
    port = _free_port()
    env = {'MASTER_PORT': str(port)}
    holder = None
    if mode == 'torchrun_like':
        # torch.distributed.run keeps MASTER_PORT for its own store: the star must come up on a port above it
        holder = socket.socket(); holder.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1); holder.bind(('127.0.0.1', port)); holder.listen(8)
        env['TORCHELASTIC_RUN_ID'] = 'none'
    if mode == 'exact_port':
        env = {'ADM_RDV_PORT': str(port), 'ADM_RDV_JOB': 'job-7', 'MASTER_PORT': '1'}
    try:
        res = _run(3, env)
    finally:
        if holder is not None:
            holder.close()
    ref = sum(np.arange(12, dtype=np.float32).reshape(3, 4) + 100 * r for r in range(3))
    bigs = [np.random.default_rng(r).standard_normal(1 << 20).astype(np.float32) for r in range(3)]
    big_ref = float(np.abs((bigs[0] + bigs[1]) + bigs[2]).sum())         # rank order, like the star
    for r in res:
        k = r['rank']
        assert r['size'] == 3 and r['sum'] == 6.0 and r['max'] == 10.0
        assert r['bc0'] == {'ids': [b'\x01' * 128, b'\x02' * 128]} and r['bc2'] == 'from two'
        assert np.array_equal(r['allreduce'], ref)
        mine = np.arange(12, dtype=np.float32).reshape(3, 4) + 100 * k
        assert np.array_equal(r['reduce1'], ref if k == 1 else mine)       # only the root's array changes
        assert np.array_equal(r['bcast_arr'], np.full(5, 1, np.float32))
        assert np.array_equal(r['gather'], np.repeat(np.arange(3, dtype=np.float32), 4))
        assert r['big'] == big_ref                                         # bit-identical on every rank


def test_tcp_group_detects_mismatched_collectives():
    res = _run(2, {'MASTER_PORT': str(_free_port()), 'ADM_RDV_TIMEOUT': '20'}, scenario='mismatch')
    assert res[0]['raised']         # rank 0 reads the peer's frame and sees the wrong operation name


def test_single_rank_group_needs_no_socket():
    sys.path.insert(0, ROOT)
    from adorym_amd.rendezvous import TcpGroup
    g = TcpGroup(0, 1)
    g.barrier()
    assert g.sum_over_ranks(3.5) == 3.5 and g.max_over_ranks(2) == 2.0 and g.bcast_object('x') == 'x'
    a = np.arange(4, dtype=np.float32)
    assert np.array_equal(g.all_reduce_sum(a.copy()), a) and np.array_equal(g.all_gather(a), a)
    g.close()


def test_product_imports_no_torch():
    """`import torch` appears nowhere under adorym_amd/ nor in bench.py (north_star: no PyTorch backend): the
    torch.distributed stand-in of the CPU sharding tests lives in tests/torch_comm.py."""
    import re
    pkg = os.path.join(ROOT, 'adorym_amd')
    hits = []
    files = [os.path.join(pkg, n) for n in sorted(os.listdir(pkg)) if n.endswith('.py')] + [os.path.join(ROOT, 'bench.py')]
    for path in files:
        src = open(path).read()
        for m in re.finditer(r'^\s*(import torch|from torch)', src, re.M):
            hits.append((os.path.basename(path), src.count('\n', 0, m.start()) + 1))
    assert not hits, hits
    assert 'TorchComm' not in open(os.path.join(pkg, 'comm.py')).read()


def test_missing_rank_is_a_clean_error_not_a_hang():
    """Rank 0 of a two-rank job whose peer never shows up: a RuntimeError after ADM_RDV_TIMEOUT, naming what is missing."""
    sys.path.insert(0, ROOT)
    from adorym_amd.rendezvous import TcpGroup
    with pytest.raises(RuntimeError, match='only 1 of 2 ranks connected'):
        TcpGroup(0, 2, '127.0.0.1', _free_port(), job='lonely', exact_port=True, timeout=1.5)
    with pytest.raises(RuntimeError, match='could not reach rank 0'):
        TcpGroup(1, 2, '127.0.0.1', _free_port(), job='lonely', exact_port=True, timeout=1.5)


def test_control_plane_objects_travel_without_pickle():
    """What the driver broadcasts (RCCL ids as bytes, time stamps, seeds, counters, initial guesses and probes as arrays) survives
    the JSON + blob codec unchanged; anything else is refused on the sending side (no pickle on the wire)."""
    sys.path.insert(0, ROOT)
    from adorym_amd.rendezvous import _pack_obj, _unpack_obj
    r = np.random.default_rng(0)
    obj = [b'\x00\x01' * 64, ('2026_10_03', 7, 2.5, None, True), {'a': r.standard_normal((3, 4)).astype(np.float32), 'b': np.arange(5)},
           [np.float32(1.5), np.int64(3)], r.standard_normal((2, 2)) + 1j * r.standard_normal((2, 2))]
    back = _unpack_obj(_pack_obj(obj))
    assert back[0] == obj[0] and back[1] == obj[1] and isinstance(back[1], tuple)
    assert np.array_equal(back[2]['a'], obj[2]['a']) and back[2]['a'].dtype == np.float32 and np.array_equal(back[2]['b'], obj[2]['b'])
    assert back[3] == [1.5, 3] and np.array_equal(back[4], obj[4])
    with pytest.raises(TypeError):
        _pack_obj({'f': open})
    assert b'pickle' not in open(os.path.join(ROOT, 'adorym_amd', 'rendezvous.py'), 'rb').read().replace(b'as pickles', b'').replace(b'no pickle', b'')


def test_a_peer_without_the_job_token_is_not_admitted():
    """Rank 0 admits only peers whose hello carries the HMAC of the job's token; a connection with the right job id but another
    token is dropped like any foreign connection, and rank 0 reports the missing rank after its time-out."""
    import threading
    sys.path.insert(0, ROOT)
    from adorym_amd.rendezvous import TcpGroup
    port = _free_port()
    errs = {}

    def rank0():
        try:
            TcpGroup(0, 2, '127.0.0.1', port, job='j', exact_port=True, timeout=2.0, token='right')
        except Exception as e:
            errs[0] = e

    t = threading.Thread(target=rank0)
    t.start()
    with pytest.raises(RuntimeError, match='could not reach rank 0'):
        TcpGroup(1, 2, '127.0.0.1', port, job='j', exact_port=True, timeout=1.5, token='wrong')
    t.join(10)
    assert 'only 1 of 2 ranks connected' in str(errs.get(0))


def test_a_rank_that_reconnects_replaces_its_abandoned_connection():
    """A rank that gave a connection up (rank 0 was slow to answer on a loaded machine) and connected again: rank 0 has the hello of
    BOTH connections in its queue.  It must end up talking to the live one -- registering the first left it with a closed socket
    and the rank locked out until the time-out (seen once as a spurious failure of a 4-rank GPU test on a slow box)."""
    import hashlib
    import hmac
    import struct
    import threading
    import time
    sys.path.insert(0, ROOT)
    from adorym_amd import rendezvous as R
    port = _free_port()
    out = {}
    srv_ready = threading.Event()

    def rank0():
        try:
            srv_ready.set()
            g = R.TcpGroup(0, 3, '127.0.0.1', port, job='j', exact_port=True, timeout=20.0, token='t')
            out[0] = g.bcast_object({'x': 7})
            g.barrier()
        except Exception as e:
            out[0] = e

    def rank(r):
        try:
            g = R.TcpGroup(r, 3, '127.0.0.1', port, job='j', exact_port=True, timeout=20.0, token='t')
            out[r] = g.bcast_object(None)
            g.barrier()
        except Exception as e:
            out[r] = e

    t0 = threading.Thread(target=rank0, daemon=True)
    t0.start()
    srv_ready.wait(5)
    # rank 1's abandoned attempt: a complete hello, then the socket is closed without waiting for the answer
    proof = hmac.new(b't', b'adm-rendezvous:j', hashlib.sha256).digest()
    hello = R._MAGIC + struct.pack('!I', 1) + b'j' + proof + struct.pack('!I', 1)
    for _ in range(200):
        try:
            s = socket.create_connection(('127.0.0.1', port), timeout=1.0)
            break
        except OSError:
            time.sleep(0.02)
    s.sendall(hello)
    s.close()
    time.sleep(0.2)
    ts = [threading.Thread(target=rank, args=(r,), daemon=True) for r in (2, 1)]       # (rank 2 first: rank 0 must not count the dead connection as rank 1 and stop listening)
    [t.start() for t in ts]
    [t.join(30) for t in ts + [t0]]
    assert out.get(0) == {'x': 7} and out.get(1) == {'x': 7} and out.get(2) == {'x': 7}, out


def test_a_rank_never_takes_a_tcp_self_connection_for_rank_0():
    """Nobody listens yet and the kernel hands the connecting socket the target port as its own: the socket is connected to ITSELF and
    reads its own hello back, which begins with the greeting.  Twice a GPU test failed that way on a loaded box ("collective mismatch
    ... the peer sent ''").  Forced here: the socket is bound to the port it then connects to; the rank must see through it (the answer of
    rank 0 carries rank 0's own proof, and a socket whose two ends are the same address is dropped) and keep looking until its time-out."""
    import threading
    import time
    sys.path.insert(0, ROOT)
    from adorym_amd import rendezvous as R
    port = _free_port()
    real_create = socket.create_connection

    def self_connecting(addr, timeout=None, **kw):
        s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        s.settimeout(timeout)
        s.bind(('127.0.0.1', addr[1]))
        s.connect(addr)                      # simultaneous open with itself
        return s

    socket.create_connection = self_connecting
    try:
        t0 = time.time()
        with pytest.raises(RuntimeError, match='could not reach rank 0'):
            R.TcpGroup(1, 2, '127.0.0.1', port, job='j', exact_port=True, timeout=1.0, token='t')
        assert time.time() - t0 < 20
    finally:
        socket.create_connection = real_create
