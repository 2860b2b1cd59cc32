"""CPU-only: the C-ABI library builds, loads, and exports every symbol include/adm.h declares."""
import os
import re
import ctypes
import numpy as np

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


def test_probe_initialisers_match_reference_goldens():
    """Product host code adorym_amd.util.initialize_probe (aperture_defocus, intensity rescaling) against the values captured
    from the reference (golden F11), and the CSR / staged form of the rotation adjoint against the oracle's scatter."""
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    import cases
    from adorym_amd.util import initialize_probe, build_rotation_adjoint_csr, rotation_lookup
    from oracle import adorym_oracle as O
    f = np.load(os.path.join(ROOT, 'tests', 'golden', 'F11_c1.npz'))
    C = cases.C1MINI
    lm = 1240. / C['energy_ev']
    for k in range(3):
        ar, br, dcm, sg = f['pinit_ad%d_args' % k]
        pr, pi = initialize_probe([16, 16], 'aperture_defocus', aperture_radius=ar, beamstop_radius=br, probe_defocus_cm=dcm,
                                  lmbda_nm=lm, psize_cm=C['psize_cm'], sign_convention=int(sg))
        assert np.abs(pr + 1j * pi - f['pinit_ad%d' % k]).max() < 1e-12
    pm, pp = f['pinit_sup_mag'], f['pinit_sup_phase']
    for k, (rdt, nf, sg, nm) in enumerate((('intensity', False, 1, 3), ('magnitude', True, 1, 2), ('intensity', False, -1, 1))):
        init = [pm[:nm], pp[:nm]] if nm > 1 else [pm[0], pp[0]]
        pr, pi = initialize_probe([16, 16], 'supplied', probe_initial=init, rescale_intensity=True, data_first_angle=f['pinit_data'],
                                  raw_data_type=rdt, normalize_fft=nf, sign_convention=sg, n_probe_modes=nm)
        ref = f['pinit_rescale%d' % k]
        assert np.abs(pr + 1j * pi - ref).max() < 1e-10 * np.abs(ref).max()
    # rotation adjoint: CSR (+ staged boxes / local offsets) reproduces the oracle's adjoint
    Y, X, Z = 3, 20, 24
    coords = rotation_lookup((Y, X, Z), np.float32(0.7))
    Yp, Xp, pad_x0 = Y + 4, X + 6, 2
    ptr, src, lsrc, w, boxes = build_rotation_adjoint_csr(coords, (Y, X, Z), Yp, Xp, pad_x0, staged=True)
    r = np.random.default_rng(0)
    g_rot = r.standard_normal((Y, X, Z, 2))
    ref = O.rotate_adj(g_rot, coords, np.float64)
    flat = np.zeros((Z, Yp, Xp, 2))
    flat[:, 1:1 + Y, pad_x0:pad_x0 + X] = np.transpose(g_rot, (2, 0, 1, 3))          # [Z][Yp][Xp] with pad_y0 = 1
    out = np.zeros((Y, X * Z, 2))
    nbx = (X + 15) // 16
    for t in range(X * Z):
        bx, bz = (t // Z) // 16, (t % Z) // 16
        x0, z0, bw, bh = boxes[bz * nbx + bx]
        for j in range(ptr[t], ptr[t + 1]):
            zz, xx = src[j] // (Yp * Xp), src[j] % (Yp * Xp) - pad_x0
            if bw > 0:
                assert lsrc[j] == (zz - z0) * bw + (xx - x0) and 0 <= xx - x0 < bw and 0 <= zz - z0 < bh
            out[:, t] += w[j] * flat[zz, 1:1 + Y, pad_x0 + xx]
    assert np.abs(out.reshape(Y, X, Z, 2) - ref).max() < 1e-5


def test_sharded_checkpoint_files_round_trip(tmp_path):
    """Two ranks: rank 0 writes the object, every rank its moment shard (opt_obj_params_checkpoint_rank_{r}.npy) and its
    pickled parameters; restore_checkpoint gives each rank its own shard back and refuses a shard of the wrong size
    (adorym/misc.py:179-211, adorym/optimizers.py:170-188)."""
    import pytest
    from adorym_amd.ptychography import save_checkpoint, restore_checkpoint
    r = np.random.default_rng(0)
    obj = r.standard_normal((4, 5, 6, 2)).astype(np.float32)
    shards = [[r.standard_normal(120).astype(np.float32) for _ in range(2)] for _ in range(2)]
    for rank in range(2):
        save_checkpoint(3, 8, str(tmp_path), obj if rank == 0 else None, shards[rank], rank=rank, n_ranks=2,
                        params={'probe_real': np.full((1, 2, 2), rank, np.float32), 'probe_imag': np.zeros((1, 2, 2), np.float32)})
    names = sorted(os.listdir(os.path.join(str(tmp_path), 'checkpoint')))
    assert names == ['checkpoint.txt', 'obj_checkpoint.npy', 'opt_obj_params_checkpoint_rank_0.npy',
                     'opt_obj_params_checkpoint_rank_1.npy', 'params_0', 'params_1', 'stamp_rank_0.txt', 'stamp_rank_1.txt']
    for rank in range(2):
        e, b, o, mom, params = restore_checkpoint(str(tmp_path), 2, rank=rank, n_ranks=2, obj_shape=obj.shape, shard_size=120)
        assert (e, b) == (3, 8) and np.array_equal(o, obj)
        assert np.array_equal(mom[0], shards[rank][0]) and np.array_equal(mom[1], shards[rank][1])
        assert params['probe_real'][0, 0, 0] == rank
    with pytest.raises(ValueError, match='another rank count'):
        restore_checkpoint(str(tmp_path), 2, rank=0, n_ranks=2, obj_shape=obj.shape, shard_size=60)
    with pytest.raises(ValueError, match='shape'):
        restore_checkpoint(str(tmp_path), 2, rank=0, n_ranks=2, obj_shape=(4, 5, 7, 2), shard_size=120)


def test_crash_line_is_left_behind_when_the_process_is_killed_by_a_signal():
    """adm_crash_line_set: after abort() -- what the ROCm runtime does on a GPU memory fault -- the armed line is on stdout and the
    exit code is the one given; disarmed, the signal takes its default course.  (bench.py arms it around the secondary legs of a
    multi-rank run.)  No GPU involved."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, os; sys.path.insert(0, %r)\n"
            "from adorym_amd import _lib\n"
            "lib = _lib.load()\n"
            "assert lib.adm_crash_line_set(b'{\"value\": 1.5, \"legs_abandoned\": [\"x\"]}', 5) == 0\n"
            "%s"
            "os.abort()\n") % (root, '%s')
    r = subprocess.run([sys.executable, '-c', code % ''], capture_output=True, timeout=120)
    assert r.returncode == 5 and r.stdout.decode().strip().splitlines()[-1] == '{"value": 1.5, "legs_abandoned": ["x"]}'
    r = subprocess.run([sys.executable, '-c', code % 'lib.adm_crash_line_set(None, 0)\n'], capture_output=True, timeout=120)
    assert r.returncode < 0 and r.stdout == b''
