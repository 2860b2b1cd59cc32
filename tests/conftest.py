import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'regression: self-comparison of two paths of THIS build (bit-equality guards) -- not parity evidence; '
                                       'the parity tests proper are `-m "gpu and not regression"`')


import pytest

def pytest_sessionfinish(session, exitstatus):
    """The product never imports torch, and the GPU tests must not either: a PyTorch-ROCm wheel carries its own copy of the HIP
    runtime and asks for it by an unversioned name, so when libadm.so (linked to /opt/rocm's libamdhip64.so.7) is loaded
    first and torch later, the process holds two HIP runtimes and aborts at exit ("double free or corruption") after every
    test has passed.  Torch-based checkers run in child processes (oracle/torch_child.py).  Say so if it happens anyway."""
    lib = sys.modules.get('adorym_amd._lib')
    if lib is not None and getattr(lib, '_lib', None) is not None and 'torch' in sys.modules and getattr(lib, 'CREATED_CONTEXT', False):
        sys.stderr.write('\nWARNING (tests/conftest.py): torch was imported into a process that holds a libadm GPU context; '
                         'the interpreter may abort at exit.  Move the torch-based check into a child process.\n')


@pytest.fixture(scope='session')
def rccl_world1():
    """ONE RcclComm (TCP control plane + RCCL behind the C ABI) at world size 1 for the whole session, shared by the GPU tests
    that walk the multi-GPU code path.  attach(ctx) gives every context its own RCCL communicator."""
    import socket
    from adorym_amd import comm as C
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    rc = C.RcclComm(device_index=0)
    yield rc
    rc.close()
