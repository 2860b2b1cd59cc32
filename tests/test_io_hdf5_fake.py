"""The HDF5 branch of adorym_amd/_io.py (the reference reads 'exchange/data' and 'metadata/*' through h5py,
adorym/ptychography.py:237-283) with a stand-in h5py module -- h5py is not installed in this image; gen_goldens.py feeds
the reference the same kind of stand-in."""
import sys
import types
import numpy as np


class _FakeDataset(object):
    def __init__(self, a):
        self.a = np.asarray(a)
        self.shape = self.a.shape
        self.dtype = self.a.dtype

    def __getitem__(self, k):
        return self.a[k]

    def __len__(self):
        return len(self.a)


class _FakeFile(object):
    store = {}
    opened, closed = [], []

    def __init__(self, path, mode='r'):
        assert mode == 'r'
        _FakeFile.opened.append(path)
        self.path = path

    def __getitem__(self, key):
        return _FakeDataset(_FakeFile.store[self.path][key])

    def close(self):
        _FakeFile.closed.append(self.path)


def test_datafile_reads_hdf5_through_h5py(monkeypatch):
    fake = types.ModuleType('h5py')
    fake.File = _FakeFile
    monkeypatch.setitem(sys.modules, 'h5py', fake)
    from adorym_amd._io import DataFile
    r = np.random.default_rng(0)
    data = r.standard_normal((3, 5, 4, 4)).astype(np.float32)
    _FakeFile.store['/x/run.h5'] = {'exchange/data': data, 'metadata/probe_pos_px': np.arange(10.).reshape(5, 2),
                                    'metadata/energy_ev': np.array(5000.), 'metadata/psize_cm': np.array(1e-7)}
    f = DataFile('/x/run.h5')
    assert f.data.shape == (3, 5, 4, 4)
    assert np.array_equal(f.data[1, np.array([0, 3])], data[1, [0, 3]])          # the driver's fancy-indexed minibatch read
    assert np.array_equal(f.get('metadata/probe_pos_px'), np.arange(10.).reshape(5, 2))
    assert float(f.get('metadata/energy_ev')) == 5000.
    f.close()
    assert _FakeFile.closed == ['/x/run.h5']


def test_datafile_without_h5py_says_what_to_do(monkeypatch):
    monkeypatch.setitem(sys.modules, 'h5py', None)        # import h5py -> ImportError
    from adorym_amd._io import DataFile
    import pytest
    with pytest.raises(ImportError, match='convert the file to .npz'):
        DataFile('/x/other.h5')
