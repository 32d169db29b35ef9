"""ann3depth_amd/matv5.py against files written by scipy.io.savemat (an independent writer of the same format): what the
reference's Make3D processors get from scipy.io.loadmat (tools/data_preprocessor.py:89-90,132)."""
import struct
import zlib

import numpy as np
import pytest

from ann3depth_amd import matv5

sio = pytest.importorskip('scipy.io')


@pytest.mark.parametrize('compress', [False, True])
def test_numeric_arrays_match_scipy(tmp_path, compress):
    rng = np.random.default_rng(5)
    want = {
        'Position3DGrid': rng.uniform(0.5, 81.0, (55, 305, 4)),                   # make3d1's variable, its real shape
        'depthMap': rng.uniform(1.0, 80.0, (55, 305)).astype(np.float32),          # make3d2's
        'small_ints': np.arange(12, dtype=np.float64).reshape(3, 4),               # doubles a writer may store narrower
        'i16': rng.integers(-3000, 3000, (2, 3, 5)).astype(np.int16),
        'u8': rng.integers(0, 256, (7, 1)).astype(np.uint8),
        'i64': np.array([[2 ** 40, -2 ** 41]], dtype=np.int64),
        'flag': np.array([[True, False, True]]),
        'z': (rng.standard_normal((2, 2)) + 1j * rng.standard_normal((2, 2))),
        'scalar': np.array([[3.25]]),
        'empty': np.zeros((0, 3)),
        'name': 'Train400Depth',
    }
    p = str(tmp_path / 'x.mat')
    sio.savemat(p, want, do_compression=compress)
    got = matv5.loadmat(p)
    ref = sio.loadmat(p)
    assert got.pop('__skipped__') == []
    assert sorted(got) == sorted(want)
    for k in want:
        a, b = got[k], ref[k]
        assert a.shape == b.shape, k
        if k == 'name':
            assert list(a) == list(b) == ['Train400Depth']
            continue
        assert a.dtype == b.dtype, (k, a.dtype, b.dtype)
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(got['Position3DGrid'][..., 3], want['Position3DGrid'][..., 3])


def test_cells_and_structs_are_listed_not_returned(tmp_path):
    p = str(tmp_path / 'x.mat')
    sio.savemat(p, {'a': np.eye(2), 's': {'field': np.ones(3)}, 'c': np.array([np.ones(2), 'text'], dtype=object)})
    got = matv5.loadmat(p)
    assert sorted(got['__skipped__']) == ['c', 's']
    np.testing.assert_array_equal(got['a'], np.eye(2))


def _v5_file(order, elements):
    head = b'MATLAB 5.0 MAT-file, written by the test'.ljust(116) + b'\0' * 8
    head += struct.pack(order + 'H', 0x0100) + (b'IM' if order == '<' else b'MI')
    return head + b''.join(elements)


def _el(order, kind, data):
    return struct.pack(order + 'II', kind, len(data)) + data + b'\0' * (-len(data) % 8)


def _matrix(order, name, cls, dims, kind, values):
    flags = _el(order, 6, struct.pack(order + 'II', cls, 0))
    dim = _el(order, 5, struct.pack(order + f'{len(dims)}i', *dims))
    nm = _el(order, 1, name.encode())
    pr = _el(order, kind, np.asarray(values).astype(np.dtype(order + matv5.MI_TYPES[kind])).tobytes())
    return _el(order, matv5.MI_MATRIX, flags + dim + nm + pr)


@pytest.mark.parametrize('order', ['<', '>'])
def test_both_byte_orders_narrow_storage_and_small_elements(tmp_path, order):
    """A double array whose values MATLAB stored as uint8 (integer-valued data), in either byte order, plus a scalar whose
    name and data travel as small elements (data inside the 8-byte tag)."""
    vals = np.arange(6)
    m = _matrix(order, 'depthMap', 6, (2, 3), 2, vals)
    # small elements: name 'ab' (2 bytes of miINT8) and one int16 value, class int16
    flags = _el(order, 6, struct.pack(order + 'II', 10, 0))
    dim = _el(order, 5, struct.pack(order + '2i', 1, 1))
    small = lambda kind, data: (struct.pack(order + 'HH', *((kind, len(data)) if order == '<' else (len(data), kind))) +
                                data.ljust(4, b'\0'))
    s = _el(order, matv5.MI_MATRIX, flags + dim + small(1, b'ab') + small(3, struct.pack(order + 'h', -77)))
    p = tmp_path / 'x.mat'
    p.write_bytes(_v5_file(order, [m, s]))
    got = matv5.loadmat(str(p))
    assert got['depthMap'].dtype == np.float64
    np.testing.assert_array_equal(got['depthMap'], vals.reshape((2, 3), order='F').astype(np.float64))
    assert got['ab'].dtype == np.int16 and got['ab'].tolist() == [[-77]]
    if order == '<':                                    # scipy reads the hand-built file the same way
        ref = sio.loadmat(str(p))
        np.testing.assert_array_equal(ref['depthMap'], got['depthMap'])
        assert ref['ab'].tolist() == [[-77]]


def test_rejects_what_is_not_a_level5_file(tmp_path):
    p = tmp_path / 'x.mat'
    p.write_bytes(b'\x89HDF\r\n\x1a\n' + b'\0' * 200)
    with pytest.raises(matv5.MatReadError, match='HDF5'):
        matv5.loadmat(str(p))
    p.write_bytes(b'x' * 200)
    with pytest.raises(matv5.MatReadError, match='level 5'):
        matv5.loadmat(str(p))
    good = _v5_file('<', [_matrix('<', 'a', 6, (1, 2), 9, [1.0, 2.0])])
    p.write_bytes(good[:-4])
    with pytest.raises(matv5.MatReadError, match='past the end'):
        matv5.loadmat(str(p))
    bad = _v5_file('<', [struct.pack('<II', matv5.MI_COMPRESSED, 5) + b'notzl']) + b'\0' * 8
    p.write_bytes(bad)
    with pytest.raises(matv5.MatReadError, match='compressed'):
        matv5.loadmat(str(p))
    assert zlib  # (imported for readers of this file: the compressed case above is a broken zlib stream)
