"""
Control plane of the data-parallel mode without any framework: a TCP star on MASTER_ADDR / MASTER_PORT.

The reference's control plane is mpi4py's COMM_WORLD (adorym/ptychography.py:39-50: ``comm.Get_rank()``, ``comm.bcast``,
``comm.Barrier``); its replacement here carries only what the multi-GPU driver needs besides the RCCL data plane of
libadm: the hand-over of the 128-byte RCCL unique ids, agreement flags, seeds / time stamps, barriers, and -- for the
host-staged validation transport only -- whole arrays.  Rank 0 listens, every other rank connects once and keeps its
socket; every operation is a collective issued in the same order on every rank, carried as length-prefixed frames that name
the operation, so a mismatched sequence raises instead of deadlocking silently.

Environment (the variables torch.distributed.run, mpirun wrappers and bench.py's own launcher set): RANK, WORLD_SIZE,
MASTER_ADDR, MASTER_PORT; optional ADM_RDV_PORT (exact port of the star), ADM_RDV_TIMEOUT (seconds for the ranks to find each
other, default 300), ADM_RDV_OP_TIMEOUT (seconds a collective may wait for a late rank, default 3600, 0 = for ever),
ADM_RDV_TOKEN (the job's shared secret, set by adorym_amd/launch.py: only peers that know it pass the handshake).  Rank 0
listens on MASTER_ADDR's interface only; control-plane objects travel as a JSON tree plus raw blobs, never as pickles.
Under torch.distributed.run MASTER_PORT itself belongs to the launcher's own store, so the star takes the first free
port ABOVE it; peers find it by a handshake that carries the job id (TORCHELASTIC_RUN_ID or the port number).
"""
import hashlib
import hmac
import json
import os
import socket
import struct
import time

import numpy as np

_MAGIC = b'ADMRDV2\0'
_ACK = b'\x06'
_SCAN = 32            # ports tried above the base port


# ---- control-plane objects on the wire: a JSON tree + raw blobs (no pickle: a peer's frame is data, never code) --------------------
def _pack_obj(obj):
    """None / bool / int / float / str / bytes / NumPy arrays and scalars / lists, tuples and str-keyed dicts of those."""
    blobs = []

    def enc(o):
        if o is None or isinstance(o, (bool, int, float, str)):
            return o
        if isinstance(o, (np.integer,)):
            return int(o)
        if isinstance(o, (np.floating,)):
            return float(o)
        if isinstance(o, (bytes, bytearray, memoryview)):
            blobs.append(bytes(o))
            return {'__b__': len(blobs) - 1}
        if isinstance(o, np.ndarray):
            a = np.ascontiguousarray(o)
            if a.dtype.hasobject:
                raise TypeError('rendezvous: object arrays cannot travel over the control plane')
            blobs.append(a.tobytes())
            return {'__nd__': len(blobs) - 1, 'dtype': a.dtype.str, 'shape': list(a.shape)}
        if isinstance(o, tuple):
            return {'__t__': [enc(v) for v in o]}
        if isinstance(o, list):
            return [enc(v) for v in o]
        if isinstance(o, dict) and all(isinstance(k, str) for k in o):
            return {'__d__': {k: enc(v) for k, v in o.items()}}
        raise TypeError('rendezvous: cannot send an object of type %s over the control plane' % type(o).__name__)

    head = json.dumps({'tree': enc(obj), 'sizes': [len(b) for b in blobs]}).encode('utf-8')
    return struct.pack('!I', len(head)) + head + b''.join(blobs)


def _unpack_obj(buf):
    nh = struct.unpack('!I', buf[:4])[0]
    head = json.loads(bytes(buf[4:4 + nh]).decode('utf-8'))
    offs, pos = [], 4 + nh
    for n in head['sizes']:
        offs.append((pos, n))
        pos += n
    if pos != len(buf):
        raise RuntimeError('rendezvous: malformed control-plane frame')

    def dec(o):
        if isinstance(o, list):
            return [dec(v) for v in o]
        if isinstance(o, dict):
            if '__b__' in o:
                a, n = offs[o['__b__']]
                return bytes(buf[a:a + n])
            if '__nd__' in o:
                a, n = offs[o['__nd__']]
                return np.frombuffer(bytes(buf[a:a + n]), dtype=np.dtype(o['dtype'])).reshape(o['shape']).copy()
            if '__t__' in o:
                return tuple(dec(v) for v in o['__t__'])
            if '__d__' in o:
                return {k: dec(v) for k, v in o['__d__'].items()}
        return o

    return dec(head['tree'])


def _send_frame(sock, tag, payload):
    head = tag.encode('ascii')
    sock.sendall(struct.pack('!HQ', len(head), len(payload)) + head)
    if len(payload):
        sock.sendall(payload)


def _recv_exact(sock, n, into=None):
    buf = into if into is not None else bytearray(n)
    view = memoryview(buf).cast('B')
    got = 0
    while got < n:
        k = sock.recv_into(view[got:], n - got)
        if k == 0:
            raise ConnectionError('rendezvous: peer closed the connection')
        got += k
    return buf


def _recv_frame(sock, tag, into=None):
    nh, nb = struct.unpack('!HQ', bytes(_recv_exact(sock, 10)))
    head = bytes(_recv_exact(sock, nh)).decode('ascii')
    if head != tag:
        raise RuntimeError("rendezvous: collective mismatch -- this rank is in '%s', the peer sent '%s'" % (tag, head))
    if into is not None:
        if nb != memoryview(into).nbytes:
            raise RuntimeError("rendezvous: '%s' carries %d bytes, expected %d" % (tag, nb, memoryview(into).nbytes))
        return _recv_exact(sock, nb, into)
    return bytes(_recv_exact(sock, nb)) if nb else b''


class TcpGroup(object):
    """rank / size + the collectives of the control plane.  ``TcpGroup.from_env()`` reads the launcher's variables."""

    def __init__(self, rank, size, addr='127.0.0.1', port=29511, job='', exact_port=False, timeout=None, token=None):
        self.rank, self.size = int(rank), int(size)
        # ADM_RDV_TIMEOUT bounds the connection set-up only; a collective may legitimately wait much longer for a rank that is
        # loading data or writing a checkpoint (ADM_RDV_OP_TIMEOUT, default one hour; 0 = no limit)
        self.timeout = float(timeout if timeout is not None else os.environ.get('ADM_RDV_TIMEOUT', '300'))
        op = float(os.environ.get('ADM_RDV_OP_TIMEOUT', '3600'))
        self.op_timeout = op if op > 0 else None
        # the job's shared secret (launch.py puts a random one into every rank's environment): only a peer that knows it gets past
        # the handshake.  Absent under foreign launchers: the handshake then rests on the job id alone, as before.
        token = os.environ.get('ADM_RDV_TOKEN', '') if token is None else token
        self._peers = {}          # rank 0: {rank: socket}
        self._root = None         # other ranks: socket to rank 0
        self._closed = False
        if self.size == 1:
            return
        proof = hmac.new(token.encode('utf-8'), b'adm-rendezvous:' + job.encode('utf-8'), hashlib.sha256).digest()
        hello = _MAGIC + struct.pack('!I', len(job)) + job.encode('utf-8') + proof
        # ... and rank 0 proves itself to the rank that knocks (the ranks scan a few ports: on a machine shared with other jobs of
        # this package -- several users' test runs on one host -- the first listener found need not be ours)
        answer = _MAGIC + hmac.new(token.encode('utf-8'), b'adm-rendezvous-root:' + job.encode('utf-8'), hashlib.sha256).digest()
        ports = [int(port)] if exact_port else [int(port) + k for k in range(_SCAN)]
        deadline = time.time() + self.timeout
        if self.rank == 0:
            srv, err = None, None
            for p in ports:
                try:
                    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                    # listen on MASTER_ADDR's own interface (loopback for a one-node job), never on all of them
                    srv.bind((socket.gethostbyname(addr), p))
                    break
                except OSError as e:
                    err = e
                    srv.close()
                    srv = None
            if srv is None:
                raise RuntimeError('rendezvous: rank 0 found no free port in %s (%r)' % (ports, err))
            srv.listen(self.size)
            self.port = srv.getsockname()[1]
            while len(self._peers) < self.size - 1:
                srv.settimeout(max(0.1, deadline - time.time()))
                try:
                    c, _ = srv.accept()
                except socket.timeout:
                    raise RuntimeError('rendezvous: only %d of %d ranks connected within %.0f s'
                                       % (len(self._peers) + 1, self.size, self.timeout))
                try:
                    c.settimeout(5.0)
                    got = bytes(_recv_exact(c, len(hello)))
                    r = struct.unpack('!I', bytes(_recv_exact(c, 4)))[0]
                    if not hmac.compare_digest(got, hello) or not (0 < r < self.size):
                        raise ConnectionError('foreign connection')
                    c.sendall(answer)
                    # ... and the rank confirms that it is still there: a connection it gave up meanwhile (it waited too long for
                    # this answer and connected again) is closed at its end and fails here instead of being counted
                    if bytes(_recv_exact(c, 1)) != _ACK:
                        raise ConnectionError('no confirmation')
                except Exception:
                    c.close()           # not one of this job's ranks (a port scanner, another job): ignore it
                    continue
                if r in self._peers:
                    # the rank connected again: it gave its earlier connection up (this process was slow to accept it).  The last
                    # connection is the live one -- registering the dead one left rank 0 talking to a closed socket.
                    self._peers.pop(r).close()
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                c.settimeout(self.op_timeout)
                self._peers[r] = c
            srv.close()
        else:
            sock = None
            while sock is None:
                for p in ports:
                    try:
                        s = socket.create_connection((addr, p), timeout=2.0)
                    except OSError:
                        continue
                    if s.getsockname() == s.getpeername():
                        # TCP self-connection: nobody listens on p yet and the kernel gave this socket p as its OWN port (the ports
                        # of one-node jobs come from the ephemeral range).  The rank would read its own hello back -- which begins
                        # with the greeting -- and take the rest of it for frames ("collective mismatch ... the peer sent ''":
                        # seen twice on loaded boxes, where rank 0 is slow to bind).
                        s.close()
                        continue
                    try:
                        # rank 0 answers when it gets to this connection -- on a loaded machine that can take a while: wait for it
                        # as long as the rendezvous may take (a connection rank 0 refuses is CLOSED by it, which ends the wait)
                        s.settimeout(max(5.0, deadline - time.time()))
                        s.sendall(hello + struct.pack('!I', self.rank))
                        if hmac.compare_digest(bytes(_recv_exact(s, len(answer))), answer):
                            s.sendall(_ACK)
                            sock, self.port = s, p
                            break
                    except Exception:
                        pass
                    s.close()
                if sock is None:
                    if time.time() > deadline:
                        raise RuntimeError('rendezvous: rank %d could not reach rank 0 at %s:%s within %.0f s'
                                           % (self.rank, addr, ports, self.timeout))
                    time.sleep(0.05)
            sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            sock.settimeout(self.op_timeout)
            self._root = sock

    @classmethod
    def from_env(cls):
        rank = int(os.environ.get('RANK', '0'))
        size = int(os.environ.get('WORLD_SIZE', '1'))
        addr = os.environ.get('MASTER_ADDR', '127.0.0.1')
        if 'ADM_RDV_PORT' in os.environ:
            return cls(rank, size, addr, int(os.environ['ADM_RDV_PORT']), job=os.environ.get('ADM_RDV_JOB', ''), exact_port=True)
        port = int(os.environ.get('MASTER_PORT', '29511'))
        job = os.environ.get('TORCHELASTIC_RUN_ID', '') + ':' + str(port)
        if 'TORCHELASTIC_RUN_ID' in os.environ or 'TORCHELASTIC_RESTART_COUNT' in os.environ:
            port += 1           # MASTER_PORT is the launcher's own store
        return cls(rank, size, addr, port, job=job)

    # ---- generic pattern: everything goes through rank 0 ---------------------------------------------------------------
    def _gather(self, tag, payload):
        """rank 0: [payload of rank 0, 1, ...]; others: None."""
        if self.rank == 0:
            return [payload] + [_recv_frame(self._peers[r], tag) for r in range(1, self.size)]
        _send_frame(self._root, tag, payload)
        return None

    def _scatter_same(self, tag, payload):
        if self.rank == 0:
            for r in range(1, self.size):
                _send_frame(self._peers[r], tag, payload)
            return payload
        return _recv_frame(self._root, tag)

    def barrier(self):
        if self.size > 1:
            self._gather('barrier', b'')
            self._scatter_same('barrier.', b'')

    def bcast_object(self, obj, root=0):
        if self.size == 1:
            return obj
        if root != 0:           # the owner hands it to rank 0 first
            if self.rank == root:
                _send_frame(self._root, 'bcast>', _pack_obj(obj))
            elif self.rank == 0:
                obj = _unpack_obj(_recv_frame(self._peers[root], 'bcast>'))
        return _unpack_obj(self._scatter_same('bcast', _pack_obj(obj) if self.rank == 0 else b''))

    def _reduce_scalar(self, value, fn, tag):
        if self.size == 1:
            return float(value)
        parts = self._gather(tag, struct.pack('!d', float(value)))
        out = struct.pack('!d', fn(struct.unpack('!d', p_)[0] for p_ in parts)) if self.rank == 0 else b''
        return struct.unpack('!d', self._scatter_same(tag + '.', out))[0]

    def sum_over_ranks(self, value):
        return self._reduce_scalar(value, lambda it: float(sum(it)), 'sum')

    def max_over_ranks(self, value):
        return self._reduce_scalar(value, max, 'max')

    # ---- whole arrays (host-staged validation transport only; slow by construction) ------------------------------------
    def all_reduce_sum(self, arr):
        """In-place sum over ranks of a contiguous NumPy array, added in rank order (0 + 1 + ...) on rank 0: deterministic."""
        if self.size == 1:
            return arr
        if self.rank == 0:
            tmp = np.empty_like(arr)
            for r in range(1, self.size):
                _recv_frame(self._peers[r], 'allreduce', into=tmp)
                arr += tmp
            for r in range(1, self.size):
                _send_frame(self._peers[r], 'allreduce.', memoryview(arr).cast('B'))
        else:
            _send_frame(self._root, 'allreduce', memoryview(arr).cast('B'))
            _recv_frame(self._root, 'allreduce.', into=arr)
        return arr

    def reduce_sum(self, arr, root):
        """arr of rank ``root`` = sum over ranks (rank order); the others' arrays are left alone."""
        if self.size == 1:
            return arr
        if self.rank == 0:
            acc = arr.copy()
            tmp = np.empty_like(arr)
            for r in range(1, self.size):
                _recv_frame(self._peers[r], 'reduce', into=tmp)
                acc += tmp
            if root == 0:
                arr[...] = acc
            else:
                _send_frame(self._peers[root], 'reduce.', memoryview(acc).cast('B'))
        else:
            _send_frame(self._root, 'reduce', memoryview(arr).cast('B'))
            if self.rank == root:
                _recv_frame(self._root, 'reduce.', into=arr)
        return arr

    def broadcast(self, arr, root):
        if self.size == 1:
            return arr
        if root != 0:
            if self.rank == root:
                _send_frame(self._root, 'bc>', memoryview(arr).cast('B'))
            elif self.rank == 0:
                _recv_frame(self._peers[root], 'bc>', into=arr)
        if self.rank == 0:
            for r in range(1, self.size):
                if r != root:
                    _send_frame(self._peers[r], 'bc', memoryview(arr).cast('B'))
        elif self.rank != root:
            _recv_frame(self._root, 'bc', into=arr)
        return arr

    def all_gather(self, mine):
        """Concatenation over ranks of equally sized contiguous arrays."""
        if self.size == 1:
            return mine.copy()
        full = np.empty((self.size,) + mine.shape, mine.dtype)
        if self.rank == 0:
            full[0] = mine
            for r in range(1, self.size):
                _recv_frame(self._peers[r], 'allgather', into=full[r])
            for r in range(1, self.size):
                _send_frame(self._peers[r], 'allgather.', memoryview(full).cast('B'))
        else:
            _send_frame(self._root, 'allgather', memoryview(mine).cast('B'))
            _recv_frame(self._root, 'allgather.', into=full)
        return full.reshape((self.size * mine.shape[0],) + mine.shape[1:]) if mine.ndim else full

    def close(self):
        if self._closed:
            return
        self._closed = True
        for s in list(self._peers.values()) + ([self._root] if self._root is not None else []):
            try:
                s.close()
            except Exception:
                pass
        self._peers, self._root = {}, None
