"""
One process per GPU without a framework launcher:  ``python -m adorym_amd.launch -n 8 script.py [args ...]``.

The reference is started as ``mpirun -n R python script.py`` (adorym/ptychography.py:39-50 reads rank and size from
mpi4py).  Here the ranks are plain child processes of this one, each with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR and
the exact port of the control plane (ADM_RDV_PORT, adorym_amd/rendezvous.py) in its environment; adorym_amd.comm.from_env()
picks them up.  The parent never touches the GPU (no HIP call, no library load), so starting children from it is safe on
hosts where an exec from a GPU-initialised process is not.  bench.py uses run() when it is called as ``bench.py --gpus N``
without a launcher around it.

Rank 0's stdout is relayed to this process's stdout line by line (a benchmark's one JSON line arrives unchanged); the other
ranks' stdout goes to stderr.  The exit code is the first non-zero child code, else 0; when one rank fails the others are
given a grace period, then terminated, then -- if SIGTERM does not end them -- killed (they would otherwise wait in a
collective for ever); SIGTERM to the launcher itself takes the same path, and every child is reaped before run() returns.
"""
import os
import secrets
import signal
import socket
import subprocess
import sys
import threading
import time


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def rank_envs(n, base=None, port=None, job=None):
    """The environment of each of the n ranks (dicts), derived from ``base`` (default os.environ)."""
    base = dict(os.environ if base is None else base)
    port = int(port) if port is not None else free_port()
    job = job or 'adm-%d-%d' % (os.getpid(), port)
    token = base.get('ADM_RDV_TOKEN') or secrets.token_hex(16)      # the job's shared secret for the control plane's handshake
    envs = []
    for r in range(int(n)):
        e = dict(base)
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                 MASTER_PORT=str(port), ADM_RDV_PORT=str(port), ADM_RDV_JOB=job, ADM_RDV_TOKEN=token)
        e.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: RCCL's intra-node transport needs it on this driver
        e.pop('TORCHELASTIC_RUN_ID', None)
        envs.append(e)
    return envs


def run(n, argv, grace_s=20.0, out=None, err=None, kill_after_s=5.0):
    """Start ``argv`` n times (one rank each), wait for all, return (exit code, rank 0's stdout lines)."""
    out = out or sys.stdout
    err = err or sys.stderr
    envs = rank_envs(n)
    procs = []
    for r, e in enumerate(envs):
        procs.append(subprocess.Popen(list(argv), env=e, stdout=subprocess.PIPE, stderr=None, text=True, bufsize=1))
    lines0 = []

    def pump(r, p):
        for line in p.stdout:
            if r == 0:
                lines0.append(line.rstrip('\n'))
                out.write(line)
                out.flush()
            else:
                err.write('[rank %d] %s' % (r, line))
                err.flush()

    threads = [threading.Thread(target=pump, args=(r, p), daemon=True) for r, p in enumerate(procs)]
    [t.start() for t in threads]
    rc, t_fail, t_term = 0, None, None

    def on_term(signum, frame):             # SIGTERM to the launcher: unwind through the finally block below
        raise KeyboardInterrupt('launcher received signal %d' % signum)

    old_handler = None
    if threading.current_thread() is threading.main_thread():
        old_handler = signal.signal(signal.SIGTERM, on_term)
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [c for c in codes if c not in (None, 0)]
            if bad and rc == 0:
                rc, t_fail = bad[0], time.time()
            if all(c is not None for c in codes):
                break
            if t_fail is not None and t_term is None and time.time() - t_fail > grace_s:
                for p in procs:             # exactly the processes started above
                    if p.poll() is None:
                        p.terminate()
                t_term = time.time()
            if t_term is not None and time.time() - t_term > kill_after_s:
                for p in procs:             # a rank stuck in a driver call does not die of SIGTERM
                    if p.poll() is None:
                        p.kill()
                t_term = time.time() + 1e9
            time.sleep(0.05)
    finally:
        # the launcher is interrupted, terminated or failing: its ranks must not outlive it.  terminate -> wait -> kill -> reap
        alive = [p for p in procs if p.poll() is None]
        for p in alive:
            p.terminate()
        deadline = time.time() + kill_after_s
        for p in alive:
            try:
                p.wait(max(0.0, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
        for p in procs:
            try:
                p.wait(5)
            except Exception:
                pass
        if old_handler is not None:
            signal.signal(signal.SIGTERM, old_handler)
    [t.join(5) for t in threads]
    return rc, lines0


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if len(argv) < 3 or argv[0] not in ('-n', '--nproc'):
        sys.stderr.write('usage: python -m adorym_amd.launch -n N script.py [args ...]\n')
        return 2
    n = int(argv[1])
    rc, _ = run(n, [sys.executable] + argv[2:])
    return rc


if __name__ == '__main__':
    sys.exit(main())
