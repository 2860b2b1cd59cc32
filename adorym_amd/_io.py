"""Minimal data I/O for the driver.  The reference reads HDF5 ('exchange/data' + 'metadata/*',
README.rst:179-225) through h5py and writes 32-bit TIFFs through dxchange; neither package is in this
image, so: HDF5 is used when h5py is importable, ``.npz`` / ``.npy`` files with the same keys are always
accepted, and TIFF output uses the small baseline-TIFF writer below (uncompressed float32, one page per
leading-axis slice -- what dxchange.write_tiff(..., dtype='float32') produces for ImageJ)."""
import os
import struct
import numpy as np


class DataFile(object):
    """Uniform view of the measurement file: ``self.data`` ([n_theta, n_pos, py, px]) + ``get(key)``."""

    def __init__(self, path_or_array):
        self._h5 = None
        self._meta = {}
        if isinstance(path_or_array, np.ndarray):
            self.data = path_or_array
        elif isinstance(path_or_array, dict):
            self.data = path_or_array['exchange/data']
            self._meta = path_or_array
        else:
            path = str(path_or_array)
            ext = os.path.splitext(path)[1].lower()
            if ext == '.npy':
                self.data = np.load(path, mmap_mode='r')
            elif ext == '.npz':
                z = np.load(path)
                self._meta = {k: z[k] for k in z.files}
                self.data = self._meta['exchange/data'] if 'exchange/data' in self._meta else self._meta['data']
            else:
                try:
                    import h5py
                except ImportError:
                    raise ImportError('reading %s needs h5py, which is not installed; convert the file to .npz '
                                      "(keys 'exchange/data', 'metadata/...') or pass the array itself" % path)
                self._h5 = h5py.File(path, 'r')
                self.data = self._h5['exchange/data']

    def get(self, key):
        if self._h5 is not None:
            return self._h5[key][...]
        if key in self._meta:
            return np.asarray(self._meta[key])
        raise KeyError(key)

    def close(self):
        if self._h5 is not None:
            self._h5.close()


def write_tiff(data, fname, dtype='float32', overwrite=True):
    """Baseline TIFF, little endian, uncompressed, one strip per page, float32/uint8/uint16/int32."""
    arr = np.asarray(data)
    arr = arr.astype(dtype if dtype is not None else arr.dtype, copy=False)
    if arr.ndim == 2:
        arr = arr[None]
    if arr.ndim != 3:
        raise ValueError('write_tiff: 2-D or 3-D arrays only')
    arr = np.ascontiguousarray(arr).astype(arr.dtype.newbyteorder('<'), copy=False)
    if not fname.lower().endswith(('.tif', '.tiff')):
        fname = fname + '.tiff'
    d = os.path.dirname(fname)
    if d and not os.path.exists(d):
        os.makedirs(d, exist_ok=True)
    fmt = {'f': 3, 'u': 1, 'i': 2}[arr.dtype.kind]
    bits = arr.dtype.itemsize * 8
    n, h, w = arr.shape
    page_bytes = h * w * arr.dtype.itemsize
    n_tags = 10
    ifd_size = 2 + n_tags * 12 + 4
    with open(fname, 'wb') as f:
        f.write(struct.pack('<2sHI', b'II', 42, 8))
        offset = 8
        for i in range(n):
            data_off = offset + ifd_size
            tags = [(256, 4, 1, w), (257, 4, 1, h), (258, 3, 1, bits), (259, 3, 1, 1), (262, 3, 1, 1),
                    (273, 4, 1, data_off), (277, 3, 1, 1), (278, 4, 1, h), (279, 4, 1, page_bytes), (339, 3, 1, fmt)]
            next_off = data_off + page_bytes if i < n - 1 else 0
            f.write(struct.pack('<H', n_tags))
            for tag, typ, cnt, val in tags:
                f.write(struct.pack('<HHI', tag, typ, cnt))
                f.write(struct.pack('<HH', val, 0) if typ == 3 else struct.pack('<I', val))
            f.write(struct.pack('<I', next_off))
            f.write(arr[i].tobytes())
            offset = data_off + page_bytes
    return fname


def read_tiff(fname):
    """Reads what write_tiff writes (uncompressed, single strip per page); also .npy."""
    if fname.lower().endswith('.npy'):
        return np.load(fname)
    with open(fname, 'rb') as f:
        buf = f.read()
    bo = '<' if buf[:2] == b'II' else '>'
    (off,) = struct.unpack(bo + 'I', buf[4:8])
    pages = []
    while off:
        (nt,) = struct.unpack(bo + 'H', buf[off:off + 2])
        t = {}
        for k in range(nt):
            tag, typ, cnt = struct.unpack(bo + 'HHI', buf[off + 2 + 12 * k: off + 10 + 12 * k])
            raw = buf[off + 10 + 12 * k: off + 14 + 12 * k]
            t[tag] = struct.unpack(bo + 'H', raw[:2])[0] if typ == 3 else struct.unpack(bo + 'I', raw)[0]
        if t.get(259, 1) != 1:
            raise NotImplementedError('read_tiff: compressed TIFFs are not supported; convert the file to .npy')
        w, h, bits, fmt = t[256], t[257], t.get(258, 8), t.get(339, 1)
        kind = {1: 'u', 2: 'i', 3: 'f'}[fmt]
        dt = np.dtype('%s%s%d' % (bo, kind, bits // 8))
        if t.get(278, h) < h:
            raise NotImplementedError('read_tiff: multi-strip TIFFs are not supported; convert the file to .npy')
        pages.append(np.frombuffer(buf, dt, h * w, t[273]).reshape(h, w))
        (off,) = struct.unpack(bo + 'I', buf[off + 2 + 12 * nt: off + 6 + 12 * nt])
    a = np.stack(pages)
    return a[0] if len(pages) == 1 else a
