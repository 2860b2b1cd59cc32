"""Checkpoint files (adorym/misc.py:179-211, adorym/optimizers.py:170-188 formats) on the host side: round trip, atomic
replacement, and refusal of a checkpoint whose per-rank stamp disagrees with checkpoint.txt (a save torn by a crash).  CPU."""
import os
import numpy as np
import pytest

from adorym_amd.ptychography import save_checkpoint, restore_checkpoint


def test_round_trip_and_no_temporaries(tmp_path):
    obj = np.random.default_rng(0).random((3, 4, 5, 2)).astype(np.float32)
    mom = [np.random.default_rng(k).random(obj.size).astype(np.float32) for k in (1, 2)]
    save_checkpoint(1, 20, str(tmp_path), obj, mom, params={'probe_real': np.ones((1, 2, 2))})
    files = sorted(os.listdir(tmp_path / 'checkpoint'))
    assert files == ['checkpoint.txt', 'obj_checkpoint.npy', 'opt_obj_params_checkpoint.npy', 'params_0', 'stamp_rank_0.txt']
    e, b, o, m, p = restore_checkpoint(str(tmp_path), 2, obj_shape=obj.shape)
    assert (e, b) == (1, 20) and np.array_equal(o, obj) and m.shape == (2,) + obj.shape
    assert np.array_equal(m[1].reshape(-1), mom[1]) and np.array_equal(p['probe_real'], np.ones((1, 2, 2)))


def test_torn_checkpoint_is_refused(tmp_path):
    """Rank 1 wrote its shard for batch 20, rank 0 never got that far (checkpoint.txt still says 10): every rank compares
    EVERY rank's stamp with checkpoint.txt, so both refuse (and the driver's agreement step drops the checkpoint on all ranks)."""
    obj = np.zeros((2, 2, 2, 2), np.float32)
    shard = [np.zeros(8, np.float32), np.zeros(8, np.float32)]
    save_checkpoint(0, 10, str(tmp_path), obj, shard, rank=0, n_ranks=2)
    save_checkpoint(0, 10, str(tmp_path), None, shard, rank=1, n_ranks=2)
    restore_checkpoint(str(tmp_path), 2, rank=1, n_ranks=2, obj_shape=obj.shape, shard_size=8)       # consistent: accepted
    save_checkpoint(0, 20, str(tmp_path), None, shard, rank=1, n_ranks=2)                              # rank 0 "crashed" before its save
    for r in (0, 1):                  # every rank checks every rank's stamp: both refuse, not only the one whose own stamp is off
        with pytest.raises(ValueError, match='torn checkpoint'):
            restore_checkpoint(str(tmp_path), 2, rank=r, n_ranks=2, obj_shape=obj.shape, shard_size=8)


def test_rank0_dying_inside_its_save_is_refused_by_the_other_ranks_too(tmp_path, monkeypatch):
    """rank 0 dies after replacing obj_checkpoint.npy and before checkpoint.txt; rank 1 has not started its save, so
    ITS stamp still equals checkpoint.txt -- yet it would load rank 0's NEW object with its own OLD moments.  Rank 1 reads rank
    0's invalidated stamp and refuses as well: a clean refusal on all ranks instead of an asymmetric failure."""
    import adorym_amd.ptychography as PT
    obj = np.zeros((2, 2, 2, 2), np.float32)
    shard = [np.zeros(8, np.float32), np.zeros(8, np.float32)]
    save_checkpoint(0, 10, str(tmp_path), obj, shard, rank=0, n_ranks=2)
    save_checkpoint(0, 10, str(tmp_path), None, shard, rank=1, n_ranks=2)
    real = PT._atomic_write

    def dying(path, writer):
        if 'opt_obj_params_checkpoint_rank_0' in path:
            raise KeyboardInterrupt('rank 0 killed after the object, before its moments')
        real(path, writer)

    monkeypatch.setattr(PT, '_atomic_write', dying)
    with pytest.raises(KeyboardInterrupt):
        save_checkpoint(0, 20, str(tmp_path), obj + 1, shard, rank=0, n_ranks=2)
    monkeypatch.setattr(PT, '_atomic_write', real)
    for r in (0, 1):
        with pytest.raises(ValueError, match='rank 0 was interrupted in the middle of a save'):
            restore_checkpoint(str(tmp_path), 2, rank=r, n_ranks=2, obj_shape=obj.shape, shard_size=8)


def test_reference_written_checkpoint_without_stamps_is_accepted(tmp_path):
    d = tmp_path / 'checkpoint'
    os.makedirs(d)
    obj = np.ones((2, 2, 2, 2), np.float32)
    np.savetxt(d / 'checkpoint.txt', np.array([2, 5]), fmt='%d')
    np.save(d / 'obj_checkpoint.npy', obj)
    np.save(d / 'opt_obj_params_checkpoint.npy', np.zeros((2,) + obj.shape, np.float32))
    e, b, o, m, p = restore_checkpoint(str(tmp_path), 2, obj_shape=obj.shape)
    assert (e, b, p) == (2, 5, None) and np.array_equal(o, obj)


def test_crash_inside_one_ranks_own_save_is_refused(tmp_path, monkeypatch):
    """rank 0 dies after obj_checkpoint.npy has been replaced but before the moments, its stamp and checkpoint.txt are:
    the old stamp and the old checkpoint.txt would still agree.  The stamp is invalidated BEFORE the first data file is
    replaced, so the half-new checkpoint is refused."""
    import adorym_amd.ptychography as PT
    obj = np.zeros((2, 2, 2, 2), np.float32)
    mom = [np.zeros(obj.size, np.float32), np.zeros(obj.size, np.float32)]
    save_checkpoint(0, 10, str(tmp_path), obj, mom)
    real = PT._atomic_write

    def dying(path, writer):
        if path.endswith('opt_obj_params_checkpoint.npy'):
            raise KeyboardInterrupt('killed between two files of one save')
        real(path, writer)

    monkeypatch.setattr(PT, '_atomic_write', dying)
    with pytest.raises(KeyboardInterrupt):
        save_checkpoint(0, 20, str(tmp_path), obj + 1, mom)
    monkeypatch.setattr(PT, '_atomic_write', real)
    assert np.load(tmp_path / 'checkpoint' / 'obj_checkpoint.npy').max() == 1          # the object IS the new one
    with pytest.raises(ValueError, match='middle of a save'):
        restore_checkpoint(str(tmp_path), 2, obj_shape=obj.shape)
    save_checkpoint(0, 30, str(tmp_path), obj + 2, mom)                                # the next complete save heals it
    assert restore_checkpoint(str(tmp_path), 2, obj_shape=obj.shape)[:2] == (0, 30)
