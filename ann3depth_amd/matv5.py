"""MATLAB level-5 MAT-file reader — what `scipy.io.loadmat` does for the reference's Make3D processors
(tools/data_preprocessor.py:89-90 `['Position3DGrid']`, :132 `['depthMap']`): the numeric arrays of a .mat file written
by MATLAB 5 … 7.2 (`-v6` / `-v7`; a `-v7.3` file is HDF5, see hdf5.py).

Format ("MAT-File Format", MathWorks): a 128-byte header (116 bytes of text, subsystem offset, version 0x0100 and the
two characters 'MI' written in the writer's byte order), then data elements.  An element is a tag — 32-bit type and
32-bit byte count, or, when the upper half of the first word is non-zero, a "small" element whose (at most four) data
bytes share the tag's 8 bytes — followed by the data, padded to 8 bytes.  miCOMPRESSED (15) wraps one zlib stream that
inflates to one element; miMATRIX (14) holds sub-elements: array flags (class, complex / global / logical bits),
dimensions, name, the real part, and the imaginary part if complex.  The numeric part may be stored in a narrower type
than the array's class (MATLAB shrinks integer-valued doubles); values come back in the class's type, column-major
storage turned into a numpy array of shape `dimensions`.

Implemented: numeric, logical and char arrays, real or complex, compressed or not, either byte order.  Cell, struct,
object and sparse arrays are listed in `skipped` and not returned (the Make3D files hold none that the processors read).
Checked against files written by scipy.io.savemat (tests/test_matv5.py)."""
import struct
import zlib

import numpy as np

MI_TYPES = {1: 'i1', 2: 'u1', 3: 'i2', 4: 'u2', 5: 'i4', 6: 'u4', 7: 'f4', 9: 'f8', 12: 'i8', 13: 'u8', 16: 'u1', 17: 'u2',
            18: 'u4'}
MI_MATRIX, MI_COMPRESSED = 14, 15
MX_CLASSES = {4: 'u2', 6: 'f8', 7: 'f4', 8: 'i1', 9: 'u1', 10: 'i2', 11: 'u2', 12: 'i4', 13: 'u4', 14: 'i8', 15: 'u8'}
MX_CHAR = 4
MX_UNSUPPORTED = {1: 'cell', 2: 'struct', 3: 'object', 5: 'sparse', 16: 'function', 17: 'opaque'}
FLAG_COMPLEX = 0x0800          # (0x0200 marks a logical array: it comes back as its uint8 class, as scipy.io.loadmat returns it)


class MatReadError(ValueError):
    pass


def _element(buf, pos, order):
    """-> (type, data bytes, position of the next element)."""
    if pos + 8 > len(buf):
        raise MatReadError('truncated element tag')
    word, = struct.unpack_from(order + 'I', buf, pos)
    if word >> 16:                                   # small element: count in the upper half, data in the tag
        kind, count = word & 0xFFFF, word >> 16
        if count > 4:
            raise MatReadError(f'small element of {count} bytes')
        return kind, buf[pos + 4:pos + 4 + count], pos + 8
    kind, count = struct.unpack_from(order + 'II', buf, pos)
    end = pos + 8 + count
    if end > len(buf):
        raise MatReadError('element runs past the end of the file')
    nxt = end if kind == MI_COMPRESSED else pos + 8 + (count + 7) // 8 * 8        # compressed elements are not padded
    return kind, buf[pos + 8:end], nxt


def _numbers(kind, data, order):
    if kind not in MI_TYPES:
        raise MatReadError(f'numeric data of type {kind}')
    dt = np.dtype(order + MI_TYPES[kind])
    if len(data) % dt.itemsize:
        raise MatReadError('numeric data is not a whole number of items')
    return np.frombuffer(data, dt)


def _matrix(data, order):
    """miMATRIX payload -> (name, array) or (name, None) for an unsupported class."""
    if not data:                                     # an empty matrix element (unset struct field)
        return '', np.zeros((0, 0))
    kind, flags, pos = _element(data, 0, order)
    if kind != 6 or len(flags) < 8:
        raise MatReadError('array flags missing')
    word, = struct.unpack_from(order + 'I', flags, 0)
    cls, bits = word & 0xFF, word & 0xFF00
    kind, dims, pos = _element(data, pos, order)
    dims = tuple(int(d) for d in _numbers(kind, dims, order))
    kind, name, pos = _element(data, pos, order)
    name = bytes(name).decode('latin-1')
    if cls in MX_UNSUPPORTED:
        return name, None
    if cls not in MX_CLASSES:
        raise MatReadError(f'array class {cls}')
    kind, real, pos = _element(data, pos, order)
    if cls == MX_CHAR and kind == 16:                # UTF-8 text
        text = bytes(real).decode('utf-8')
        arr = np.array([ord(c) for c in text], dtype='u2')
    else:
        arr = _numbers(kind, real, order)
    if bits & FLAG_COMPLEX:
        kind, imag, pos = _element(data, pos, order)
        arr = arr.astype('c16' if cls == 6 else 'c8') + 1j * _numbers(kind, imag, order)
    else:
        arr = arr.astype(np.dtype(MX_CLASSES[cls]))      # native byte order, the class's own type
    n = int(np.prod(dims)) if dims else 0
    if arr.size != n:
        raise MatReadError(f'{name}: {arr.size} values for dimensions {dims}')
    arr = arr.reshape(dims, order='F')
    if cls == MX_CHAR:                               # rows of characters -> strings, as loadmat(chars_as_strings=True)
        rows = [''.join(map(chr, row)) for row in arr.reshape(dims[0], -1)] if arr.ndim >= 2 else []
        arr = np.array(rows)
    return name, arr


def loadmat(path):
    """-> {variable name: numpy array} of the file's numeric / logical / char variables; key '__skipped__' lists the
    variables of classes this reader does not return (cell, struct, object, sparse)."""
    with open(path, 'rb') as f:
        buf = f.read()
    if len(buf) < 128:
        raise MatReadError('shorter than a MAT-file header')
    if buf[:4] == b'\x89HDF' or b'MATLAB 7.3' in buf[:32]:
        raise MatReadError('a -v7.3 MAT-file is HDF5: read it with ann3depth_amd.hdf5')
    endian = bytes(buf[126:128])
    if endian == b'IM':
        order = '<'
    elif endian == b'MI':
        order = '>'
    else:
        raise MatReadError('no MAT-file level 5 header')
    version, = struct.unpack_from(order + 'H', buf, 124)
    if version != 0x0100:
        raise MatReadError(f'MAT-file version {version:#06x}')
    out, skipped = {}, []
    view = memoryview(buf)
    pos = 128
    while pos + 8 <= len(buf):
        kind, data, pos = _element(view, pos, order)
        if kind == MI_COMPRESSED:
            try:
                inner = zlib.decompress(bytes(data))
            except zlib.error as e:
                raise MatReadError(f'compressed element: {e}') from e
            kind, data, _ = _element(memoryview(inner), 0, order)
        if kind != MI_MATRIX:
            continue
        name, arr = _matrix(data, order)
        if arr is None:
            skipped.append(name)
        else:
            out[name] = arr
    out['__skipped__'] = skipped
    return out
