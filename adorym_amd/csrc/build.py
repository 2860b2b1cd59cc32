#!/usr/bin/env python3
"""Build libadm.so (gfx950) in-tree with hipcc.  Usage: python adorym_amd/csrc/build.py [--force] [--safe-sync]"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT = os.path.join(PKG, 'libadm.so')
SRCS = ['adm_api.hip', 'adm_object.hip', 'adm_multislice.hip', 'adm_ms_generic.hip', 'adm_rotcsr.hip', 'adm_comm.hip', 'adm_p2p.hip', 'adm_holo.hip']
HDRS = ['adm_common.h', 'adm_fft.h', 'adm_ms_math.h', 'adm_optim.h', os.path.join('..', '..', 'include', 'adm.h')]


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError('hipcc not found')


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(HERE, f)) > t for f in SRCS + HDRS + ['build.py'])


def build(force=False, extra=(), out=None, tag=''):
    out = out or OUT
    if not force and out == OUT and not needs_build():
        return OUT
    objs, cmds = [], []
    hdr_t = max(os.path.getmtime(os.path.join(HERE, f)) for f in HDRS + ['build.py'])
    for s in SRCS:
        o = os.path.join(HERE, s.replace('.hip', tag + '.o'))
        objs.append(o)
        if not force and not extra and os.path.exists(o) and os.path.getmtime(o) > max(hdr_t, os.path.getmtime(os.path.join(HERE, s))):
            continue          # object is newer than its source and every header
        cmds.append([_hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-munsafe-fp-atomics',
                     '-Wno-unused-result', '-fno-slp-vectorize', '-c', os.path.join(HERE, s), '-o', o] + list(extra))
    # the translation units are independent: compile them side by side (adm_multislice.hip alone takes ~75 s)
    procs = [subprocess.Popen(c) for c in cmds]
    failed = [c for c, p_ in zip(cmds, procs) if p_.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, failed[0])
    subprocess.check_call([_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs + ['-ldl'])
    return out


if __name__ == '__main__':
    # experiment builds: python build.py --variant NAME -DFLAG ...  -> adorym_amd/libadm_NAME.so
    if '--variant' in sys.argv:
        name = sys.argv[sys.argv.index('--variant') + 1]
        flags = [a for i_, a in enumerate(sys.argv[1:]) if a.startswith('-D') or a.startswith('-m') or a.startswith('-f') or a.startswith('-amdgpu')
                 or sys.argv[i_] == '-mllvm']          # (the value that follows -mllvm)
        print(build(force=True, extra=flags, out=os.path.join(PKG, 'libadm_%s.so' % name), tag='_' + name))
    else:
        extra = ['-DADM_SAFE_SYNC'] if '--safe-sync' in sys.argv else []
        print(build(force='--force' in sys.argv or bool(extra), extra=extra))
