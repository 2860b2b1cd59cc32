"""`python bench.py --gpus N` must work by itself: the parent starts the N ranks as child processes before anything touches
the GPU (adorym_amd/launch.py), relays rank 0's JSON line and returns the ranks' exit code.  The reference is started with
`mpirun -n N` (adorym/ptychography.py:39-50, 786, 905-909)."""
import io
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

_RANK_SCRIPT = r'''
import json, os, sys
sys.path.insert(0, %r)
from adorym_amd.rendezvous import TcpGroup
g = TcpGroup.from_env()
total = g.sum_over_ranks(float(os.environ['LOCAL_RANK']) + 1)
names = g.bcast_object(os.environ['ADM_RDV_JOB'] if g.rank == 0 else None)
assert names == os.environ['ADM_RDV_JOB'] and 'TORCHELASTIC_RUN_ID' not in os.environ
print('noise from rank %%d' %% g.rank)
g.barrier()
if g.rank == 0:
    print(json.dumps({'n_gpus': g.size, 'sum': total, 'argv': sys.argv[1:]}))
g.close()
sys.exit(int(os.environ.get('FAIL_RANK', '-1')) == g.rank and 7 or 0)
''' % ROOT


def test_launch_runs_ranks_relays_rank0_and_returns_code(tmp_path, monkeypatch):
    from adorym_amd import launch
    script = tmp_path / 'rank.py'
    script.write_text(_RANK_SCRIPT)
    out, err = io.StringIO(), io.StringIO()
    rc, lines = launch.run(3, [sys.executable, str(script), '--steps', '3'], out=out, err=err)
    assert rc == 0
    j = json.loads(lines[-1])
    assert j == {'n_gpus': 3, 'sum': 6.0, 'argv': ['--steps', '3']}
    assert out.getvalue().strip().split('\n')[-1] == lines[-1]                       # rank 0's line arrives unchanged
    assert 'noise from rank 1' in err.getvalue() and 'noise from rank 1' not in out.getvalue()
    monkeypatch.setenv('FAIL_RANK', '2')
    rc, _ = launch.run(3, [sys.executable, str(script)], out=io.StringIO(), err=io.StringIO())
    assert rc == 7


def test_rank_envs_are_complete_and_distinct():
    from adorym_amd import launch
    envs = launch.rank_envs(4, base={'PATH': '/bin', 'TORCHELASTIC_RUN_ID': 'x'}, port=4321)
    assert [e['RANK'] for e in envs] == ['0', '1', '2', '3'] == [e['LOCAL_RANK'] for e in envs]
    for e in envs:
        assert e['WORLD_SIZE'] == '4' and e['MASTER_ADDR'] == '127.0.0.1' and e['MASTER_PORT'] == '4321' == e['ADM_RDV_PORT']
        assert e['HSA_ENABLE_IPC_MODE_LEGACY'] == '0' and 'TORCHELASTIC_RUN_ID' not in e and e['PATH'] == '/bin'
    assert len({e['ADM_RDV_JOB'] for e in envs}) == 1


def test_bench_gpus_n_without_launcher_starts_its_own_ranks(monkeypatch):
    """bench.py --gpus 2 with no RANK / WORLD_SIZE in the environment routes to launch.run with its own argv and exits with
    its code -- before libadm is loaded (the import of the device layer would fail the assertion below on a GPU-less box
    only at Context creation, so the check is on the call itself)."""
    import importlib
    from adorym_amd import launch
    bench = importlib.import_module('bench')
    seen = {}

    def fake_run(n, argv, **kw):
        seen['n'], seen['argv'] = n, list(argv)
        return 5, []

    monkeypatch.setattr(launch, 'run', fake_run)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '2', '--steps', '3', '--warmup', '1', '--comm', 'host'])
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == 5
    assert seen['n'] == 2
    assert seen['argv'][0] == sys.executable and os.path.samefile(seen['argv'][1], os.path.join(ROOT, 'bench.py'))
    assert seen['argv'][2:] == ['--gpus', '2', '--steps', '3', '--warmup', '1', '--comm', 'host']


def test_bench_under_a_launcher_does_not_relaunch(monkeypatch):
    """With RANK / WORLD_SIZE set (torch.distributed.run, mpirun wrapper) bench.py is a rank, not a launcher; a mismatch of
    --gpus and WORLD_SIZE is refused."""
    import importlib
    from adorym_amd import launch
    bench = importlib.import_module('bench')
    monkeypatch.setattr(launch, 'run', lambda *a, **k: (_ for _ in ()).throw(AssertionError('must not launch')))
    monkeypatch.setenv('RANK', '0')
    monkeypatch.setenv('WORLD_SIZE', '4')
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '2'])
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert 'WORLD_SIZE=4' in str(ex.value.code)


def test_a_failing_rank_does_not_leave_the_others_waiting_for_ever(tmp_path):
    """Rank 1 dies before the rendezvous; rank 0 would wait in it for minutes.  The launcher gives the survivors a grace period,
    terminates exactly the processes it started and returns the failing rank's code."""
    import time
    from adorym_amd import launch
    script = tmp_path / 'rank.py'
    script.write_text("import os, sys, time\n"
                      "if os.environ['RANK'] == '1':\n    sys.exit(9)\n"
                      "time.sleep(600)\n")
    t0 = time.time()
    rc, _ = launch.run(2, [sys.executable, str(script)], grace_s=1.0, out=io.StringIO(), err=io.StringIO())
    assert rc == 9 and time.time() - t0 < 30


def test_sigterm_to_the_launcher_takes_its_ranks_down_even_if_they_ignore_sigterm(tmp_path):
    """ranks must not outlive a terminated launcher, and a rank that does not die of SIGTERM (stuck in a driver call;
    here: it ignores the signal) is killed after the escalation period; every child is reaped."""
    import signal
    import subprocess
    import time
    rank = tmp_path / 'rank.py'
    rank.write_text("import os, signal, sys, time\n"
                    "if os.environ['RANK'] == '1':\n    signal.signal(signal.SIGTERM, signal.SIG_IGN)\n"
                    "open(os.path.join(%r, 'pid%%s' %% os.environ['RANK']), 'w').write(str(os.getpid()))\n"
                    "time.sleep(600)\n" % str(tmp_path))
    drv = tmp_path / 'drv.py'
    drv.write_text("import sys\nsys.path.insert(0, %r)\nfrom adorym_amd import launch\n"
                   "rc, _ = launch.run(2, [sys.executable, %r], kill_after_s=1.0)\nsys.exit(rc)\n" % (ROOT, str(rank)))
    p = subprocess.Popen([sys.executable, str(drv)])
    deadline = time.time() + 30
    while time.time() < deadline and not ((tmp_path / 'pid0').exists() and (tmp_path / 'pid1').exists()):
        time.sleep(0.05)
    pids = [int((tmp_path / ('pid%d' % r)).read_text()) for r in range(2)]
    p.send_signal(signal.SIGTERM)
    p.wait(30)
    time.sleep(0.2)
    for pid in pids:
        alive = True
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            alive = False
        assert not alive, 'rank process %d outlived its launcher' % pid
