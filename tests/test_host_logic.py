"""Host-side logic that needs no GPU: the resident-data decision of PtychographyModel._target (adorym/forward_model.py:113-119 is what
it replaces), with a stand-in for the device.  CPU."""
import numpy as np

from adorym_amd.forward_model import PtychographyModel


class _FakeArray(object):
    def __init__(self, host, off=0, shape=None):
        self.host, self.off = host, off
        self.shape = tuple(shape if shape is not None else host.shape)

    def view(self, offset_elems, shape):
        return _FakeArray(self.host, self.off + offset_elems, shape)

    def numpy(self):
        n = int(np.prod(self.shape))
        return self.host.reshape(-1)[self.off:self.off + n].reshape(self.shape)


class _FakeDevice(object):
    def __init__(self):
        self.uploads = 0

    def array(self, host, dtype=None):
        self.uploads += 1
        return _FakeArray(np.ascontiguousarray(host, dtype=dtype))


def _model(prj, raw='magnitude', loss='lsq'):
    m = PtychographyModel.__new__(PtychographyModel)
    m.prj, m.raw_data_type, m.loss_function_type = prj, raw, loss
    m.common_vars = {'theta_downsample': None, 'ds_level': 1}
    m.device = _FakeDevice()
    return m


def test_consecutive_minibatch_is_a_view_of_the_resident_angle(monkeypatch):
    r = np.random.default_rng(0)
    prj = r.standard_normal((3, 10, 4, 5)).astype(np.float32)
    m = _model(prj, raw='intensity')
    t = m._target(1, np.array([2, 3, 4]))
    assert isinstance(t, _FakeArray) and m.device.uploads == 1
    assert np.array_equal(t.numpy(), np.sqrt(np.abs(prj[1, 2:5])))          # get_data's processing, once per angle
    t2 = m._target(1, np.array([7, 8]))
    assert m.device.uploads == 1 and np.array_equal(t2.numpy(), np.sqrt(np.abs(prj[1, 7:9])))
    m._target(2, np.array([0, 1]))
    assert m.device.uploads == 2                                            # one upload per angle
    # a minibatch that is not a run of consecutive positions (the padded last one) goes the host way
    host = m._target(1, np.array([0, 1, 5]))
    assert isinstance(host, np.ndarray) and np.array_equal(host, np.sqrt(np.abs(prj[1, [0, 1, 5]]))) and m.device.uploads == 2


def test_large_datasets_keep_streaming(monkeypatch):
    monkeypatch.setenv('ADM_RESIDENT_DATA_MB', '0')
    prj = np.ones((2, 6, 4, 4), np.float32)
    m = _model(prj)
    t = m._target(0, np.array([1, 2, 3]))
    assert isinstance(t, np.ndarray) and m.device.uploads == 0
