"""CPU-only: the C-ABI library builds, loads, and exports every symbol include/adm.h declares."""
import os
import re
import ctypes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, 'include', 'adm.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(adm_[a-z0-9_]+)\s*\(', src)))


def test_build_and_symbols():
    import __graft_entry__ as g
    g.build()
    from adorym_amd import _lib
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 25
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), 'libadm.so does not export %s' % n
        assert n in _lib.SIGNATURES, 'ctypes binding missing for %s' % n
    assert set(_lib.SIGNATURES) == set(names)
    assert lib.adm_version() == 100


def test_host_helpers_match_oracle():
    """Host-side setup code of the product (rotation table, pads, transfer function) vs the pinned oracle."""
    import numpy as np
    from adorym_amd.util import rotation_lookup, calculate_pad_len
    from adorym_amd.propagate import get_kernel
    from oracle import adorym_oracle as O
    for th in (0.3, 1.0, 3.44159, 6.2831855):
        assert np.array_equal(rotation_lookup((4, 16, 16), np.float32(th)), O.rotation_coords((4, 16, 16), np.float32(th)))
    pos = np.array([(y, x) for y in np.arange(23) * 12 - 36 for x in np.arange(23) * 12 - 36])
    assert np.array_equal(calculate_pad_len([256] * 3, pos, [72, 72]), O.calculate_pad_len([256] * 3, pos, [72, 72]))
    a = get_kernel(1., 0.248, np.array([1., 1., 1.]), (72, 72))
    assert np.array_equal(a, O.get_kernel(1., 0.248, np.array([1., 1., 1.]), (72, 72)))
