"""The boundary is a C ABI, not a Python extension: include/adm.h must compile as plain C (C99, -Wall -Werror) and a C program
must be able to link libadm.so and use it with no Python in between (examples/abi_consumer.c).  CPU part: version, device count,
error contract.  GPU part (`-m gpu`): a device round trip from C."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.build()
    exe = str(tmp_path / 'abi_consumer')
    subprocess.check_call(['gcc', '-std=c99', '-Wall', '-Werror', '-pedantic', '-I' + os.path.join(ROOT, 'include'),
                           os.path.join(ROOT, 'examples', 'abi_consumer.c'), '-o', exe, '-L' + os.path.join(ROOT, 'adorym_amd'), '-ladm',
                           '-Wl,-rpath,' + os.path.join(ROOT, 'adorym_amd')])
    return exe


def test_header_is_plain_c_and_a_c_program_links_the_library(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'adm_version 100' in r.stdout and 'OK (no GPU needed)' in r.stdout
    assert 'adm_ctx_create(device' in r.stdout and '-> -' in r.stdout          # a negative status with its message


@pytest.mark.gpu
def test_c_program_device_round_trip(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe, 'gpu'], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'bit-exact' in r.stdout and 'OK (gpu)' in r.stdout
