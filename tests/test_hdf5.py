"""ann3depth_amd/hdf5.py against a MATLAB-v7.3-shaped file assembled here byte by byte from the HDF5 File Format
Specification (no HDF5 library is installed; the reader shares no code with this writer), and the NYU preprocessor
(tools/data_preprocessor.py; reference: tools/data_preprocessor.py:167-210) run on it end to end."""
import os
import struct
import sys
import zlib

import numpy as np
import pytest

from ann3depth_amd import hdf5, imresize, png

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UNDEF = 0xffffffffffffffff


class Image:
    """The HDF5 part of the file (addresses relative to the superblock): pieces are appended 8-aligned."""

    def __init__(self):
        self.b = bytearray(96)                    # room for the version-0 superblock

    def add(self, data):
        while len(self.b) % 8:
            self.b.append(0)
        at = len(self.b)
        self.b += data
        return at


def msg(mtype, body, flags=0):
    body = body + b'\0' * (-len(body) % 8)
    return struct.pack('<HHB3x', mtype, len(body), flags) + body


def object_header(img, messages, split=None):
    """Version-1 object header; with `split` the messages from that index on live in a continuation block."""
    if split is None:
        body = b''.join(messages)
        return img.add(struct.pack('<BxHII4x', 1, len(messages), 1, len(body)) + body)
    tail = b''.join(messages[split:])
    tail_at = img.add(tail)
    head = b''.join(messages[:split]) + msg(0x0010, struct.pack('<QQ', tail_at, len(tail)))
    return img.add(struct.pack('<BxHII4x', 1, len(messages) + 1, 1, len(head)) + head)


def dataspace(shape):
    return msg(0x0001, struct.pack('<BBB5x', 1, len(shape), 0) + b''.join(struct.pack('<Q', d) for d in shape))


def datatype(dt):
    dt = np.dtype(dt)
    if dt.kind in 'iu':
        bits = 8 if dt.kind == 'i' else 0
        return msg(0x0003, struct.pack('<BBBBI', 0x10 | 0, bits, 0, 0, dt.itemsize) + struct.pack('<HH', 0, 8 * dt.itemsize))
    assert dt == np.float32
    return msg(0x0003, struct.pack('<BBBBI', 0x10 | 1, 0x20, 0x1f, 0, 4) + struct.pack('<HHBBBBI', 0, 32, 23, 8, 0, 23, 127))


REFTYPE = msg(0x0003, struct.pack('<BBBBI', 0x10 | 7, 0, 0, 0, 8))


def contiguous(img, arr, dtype_msg=None):
    arr = np.ascontiguousarray(arr)
    at = img.add(arr.tobytes())
    layout = msg(0x0008, struct.pack('<BBQQ', 3, 1, at, arr.nbytes))
    return object_header(img, [dataspace(arr.shape), dtype_msg or datatype(arr.dtype), layout])


def chunked(img, arr, cdims, filters, split=None, levels=1):
    """Chunked dataset, full chunks (edge chunks padded), filter pipeline `filters` (ids), version-1 chunk B-tree of one
    or two levels."""
    arr = np.ascontiguousarray(arr)
    rank = arr.ndim
    keys = []
    grid = [range(0, s, c) for s, c in zip(arr.shape, cdims)]
    for idx in np.ndindex(*[len(g) for g in grid]):
        offs = tuple(g[i] for g, i in zip(grid, idx))
        chunk = np.zeros(cdims, arr.dtype)
        sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, arr.shape))
        chunk[tuple(slice(0, s.stop - s.start) for s in sl)] = arr[sl]
        raw = chunk.tobytes()
        for fid in filters:
            if fid == 2:
                raw = np.frombuffer(raw, np.uint8).reshape(-1, arr.dtype.itemsize).T.tobytes()
            elif fid == 1:
                raw = zlib.compress(raw, 3)
            elif fid == 3:
                raw = raw + b'\xde\xad\xbe\xef'           # the checksum is not verified by the reader
        keys.append((offs, img.add(raw), len(raw)))

    def node(level, entries, last_offs):
        body = b'TREE' + struct.pack('<BBHQQ', 1, level, len(entries), UNDEF, UNDEF)
        for offs, child, size in entries:
            body += struct.pack('<II', size, 0) + b''.join(struct.pack('<Q', o) for o in offs) + struct.pack('<Q', 0)
            body += struct.pack('<Q', child)
        body += struct.pack('<II', 0, 0) + b''.join(struct.pack('<Q', o) for o in last_offs) + struct.pack('<Q', 0)
        return img.add(body)
    end = tuple(arr.shape)
    if levels == 1:
        root = node(0, keys, end)
    else:
        half = len(keys) // 2
        left, right = node(0, keys[:half], keys[half][0]), node(0, keys[half:], end)
        root = node(1, [(keys[0][0], left, 0), (keys[half][0], right, 0)], end)
    layout = msg(0x0008, struct.pack('<BBBQ', 3, 2, rank + 1, root) +
                 b''.join(struct.pack('<I', c) for c in cdims) + struct.pack('<I', arr.dtype.itemsize))
    names = {1: b'deflate\0', 2: b'shuffle\0', 3: b'fletcher32\0\0\0\0\0\0'}
    pipeline = struct.pack('<BB6x', 1, len(filters))
    for fid in filters:
        cd = [3] if fid == 1 else ([arr.dtype.itemsize] if fid == 2 else [])
        pipeline += struct.pack('<HHHH', fid, len(names[fid]), 1, len(cd)) + names[fid]
        pipeline += b''.join(struct.pack('<I', v) for v in cd) + (b'\0' * 4 if len(cd) % 2 else b'')
    msgs = [dataspace(arr.shape), datatype(arr.dtype), msg(0x000b, pipeline), layout]
    return object_header(img, msgs, split)


def group(img, links):
    """Old-style group: local heap with the names, one symbol-table node, a one-entry B-tree, symbol-table message."""
    names = sorted(links)
    heap_data = bytearray(b'\0' * 8)
    offsets = {}
    for n in names:
        offsets[n] = len(heap_data)
        heap_data += n.encode() + b'\0'
        heap_data += b'\0' * (-len(heap_data) % 8)
    data_at = img.add(bytes(heap_data))
    heap_at = img.add(b'HEAP' + struct.pack('<B3xQQQ', 0, len(heap_data), UNDEF, data_at))
    snod = b'SNOD' + struct.pack('<BxH', 1, len(names))
    for n in names:
        snod += struct.pack('<QQII16x', offsets[n], links[n], 0, 0)
    snod_at = img.add(snod)
    tree = b'TREE' + struct.pack('<BBHQQ', 0, 0, 1, UNDEF, UNDEF) + struct.pack('<QQQ', 0, snod_at, offsets[names[-1]])
    tree_at = img.add(tree)
    return object_header(img, [msg(0x0011, struct.pack('<QQ', tree_at, heap_at))])


def build_file(path, depths, images, filenames):
    img = Image()
    refs = {}
    for i, name in enumerate(filenames):
        chars = np.array([ord(ch) for ch in name], '<u2').reshape(-1, 1)
        refs['%c' % (ord('a') + i)] = contiguous(img, chars)
    ref_values = np.array([[refs[k] for k in sorted(refs)]], '<u8')
    links = {
        '#refs#': group(img, refs),
        'depths': chunked(img, depths, (2, 4, depths.shape[2]), [2, 1], split=2, levels=2),     # shuffle + deflate
        'images': chunked(img, images, (1, 3, 8, 4), [1, 3]),                                   # deflate + fletcher32
        'rawRgbFilenames': contiguous(img, ref_values, REFTYPE),
    }
    root = group(img, links)
    sb = hdf5.SIGNATURE + struct.pack('<BBBBBBBBHHI', 0, 0, 0, 0, 0, 8, 8, 0, 4, 16, 0)
    sb += struct.pack('<QQQQ', 0, UNDEF, len(img.b), UNDEF)           # base (as MATLAB writes it: 0), free space, EOF, driver
    sb += struct.pack('<QQII16x', 0, root, 0, 0)                        # root group symbol-table entry
    img.b[:len(sb)] = sb
    header = b'MATLAB 7.3 MAT-file, Platform: GLNXA64, Created on: test HDF5 schema 1.00 .'
    with open(path, 'wb') as f:
        f.write(header + b' ' * (512 - len(header)))
        f.write(bytes(img.b))


@pytest.fixture()
def nyu_like(tmp_path):
    rng = np.random.default_rng(5)
    n = 7
    depths = (rng.random((n, 8, 6)) * 9 + 0.7).astype('<f4')          # HDF5 order: MATLAB's [H=6? ...] reversed
    images = rng.integers(0, 256, (n, 3, 8, 6)).astype(np.uint8)
    names = ['kitchen_0004/r-%d.ppm' % (1000 + i) for i in range(n)]
    path = tmp_path / 'nyu_depth_v2_labeled.mat'
    build_file(str(path), depths, images, names)
    return str(path), depths, images, names


def test_reads_groups_chunked_filtered_datasets_and_references(nyu_like):
    path, depths, images, names = nyu_like
    with hdf5.File(path) as mat:
        assert sorted(mat.keys()) == ['#refs#', 'depths', 'images', 'rawRgbFilenames']
        d, im = mat['depths'], mat['images']
        assert d.shape == depths.shape and d.dtype == np.float32 and d.filters == [2, 1]
        assert im.shape == images.shape and im.dtype == np.uint8
        np.testing.assert_array_equal(d[:], depths)                    # two-level chunk tree, edge chunks, shuffle + deflate
        for i in range(len(depths)):
            np.testing.assert_array_equal(d[i], depths[i])
            np.testing.assert_array_equal(im[i], images[i])            # chunks narrower than the last axis
        np.testing.assert_array_equal(np.stack(list(im)), images)
        refs = mat['rawRgbFilenames'][0]
        assert len(refs) == len(names) and all(isinstance(r, hdf5.Reference) for r in refs)
        got = [''.join(map(chr, mat[r][:].T[0])) for r in refs]
        assert got == names
        assert ''.join(map(chr, mat['#refs#']['b'][:].T[0])) == names[1]
        with pytest.raises(NotImplementedError):
            d[1:3]


def test_rejects_what_it_does_not_implement(tmp_path):
    p = tmp_path / 'x.h5'
    p.write_bytes(hdf5.SIGNATURE + bytes([2]) + b'\0' * 100)
    with pytest.raises(NotImplementedError, match='superblock version 2'):
        hdf5.File(str(p))
    p.write_bytes(b'not hdf5' * 100)
    with pytest.raises(ValueError, match='no HDF5 superblock'):
        hdf5.File(str(p))


def test_preprocessor_writes_the_reference_layout(nyu_like, tmp_path):
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import data_preprocessor as pre
    path, depths, images, names = nyu_like
    data = tmp_path / 'data'
    os.makedirs(data / 'nyu' / 'unpacked')
    os.rename(path, data / 'nyu' / 'unpacked' / 'nyu_depth_v2_labeled.mat')
    env = {'DATA_DIR': str(data), 'WIDTH': '8', 'HEIGHT': '6', 'DHEIGHT': '3', 'LIMIT': '6'}
    logs = []
    assert pre.main(['nyu'], env, logs.append) == 0
    assert pre.settings(env)['d_width'] == 3 * 8 // 6                    # DWIDTH defaults to DHEIGHT * WIDTH // HEIGHT
    train, test = sorted(os.listdir(data / 'nyu' / 'train')), sorted(os.listdir(data / 'nyu' / 'test'))
    stem = lambda i: names[i].replace('/', '_').replace('.', '_')[:-4]
    assert test == sorted(f'{stem(i)}-{k}.png' for i in (0, 5) for k in ('depth', 'image'))         # c % 5 == 0 -> test
    assert train == sorted(f'{stem(i)}-{k}.png' for i in (1, 2, 3, 4) for k in ('depth', 'image'))   # LIMIT=6: sample 6 not written
    # sample 1: the image is the stored (3, 8, 6) array with channels last, unchanged in size, turned clockwise
    got = png.imread(str(data / 'nyu' / 'train' / f'{stem(1)}-image.png'))
    want = np.rot90(np.moveaxis(images[1], 0, 2), k=-1)
    assert got.shape == (6, 8, 3)
    np.testing.assert_array_equal(got, want)
    # its depth: min-max to 8 bits, (8, 6) -> (4, 3) with the antialiased triangle filter, turned clockwise
    got = png.imread(str(data / 'nyu' / 'train' / f'{stem(1)}-depth.png'))
    want = np.rot90(imresize.imresize(depths[1], (4, 3)), k=-1)
    assert got.shape == (3, 4) and got.dtype == np.uint8
    np.testing.assert_array_equal(got, want)
    # a second run refuses to overwrite, FORCE empties the directories first
    logs.clear()
    pre.main(['nyu'], env, logs.append)
    assert any('Directory is not empty' in str(l) for l in logs)
    pre.main(['nyu'], dict(env, FORCE='1', START='4'), logs.append)
    assert sorted(os.listdir(data / 'nyu' / 'train')) == sorted(f'{stem(4)}-{k}.png' for k in ('depth', 'image'))


def test_bytescale_and_pil_bilinear_known_answers():
    # bytescale (scipy/misc/pilutil.py): (x - min) * 255 / (max - min), + 0.5, truncated; uint8 passes through
    np.testing.assert_array_equal(imresize.bytescale(np.array([[1.0, 2.0], [3.0, 5.0]], np.float32)), [[0, 64], [128, 255]])
    u8 = np.array([[3, 250]], np.uint8)
    assert imresize.bytescale(u8) is u8
    np.testing.assert_array_equal(imresize.bytescale(np.full((2, 2), 7.0)), np.zeros((2, 2)))      # flat image: scale by 1
    row = np.array([[0, 255]], np.uint8)
    # Pillow's ImagingResample, triangle filter, 8-bit fixed point (hand-computed in the module docstring's terms):
    # upscale 2 -> 4: centres .25 .75 1.25 1.75, weights (1), (.75 .25), (.25 .75), (1)
    np.testing.assert_array_equal(imresize.imresize(row, (1, 4)), [[0, 64, 191, 255]])
    # downscale 2 -> 1: the filter is widened by the scale, both pixels weigh 1/2: (0 + 255)/2 + 1/2 -> 128
    np.testing.assert_array_equal(imresize.imresize(row, (1, 1)), [[128]])
    # same size: untouched; 2-D float input goes through bytescale first
    np.testing.assert_array_equal(imresize.imresize(row, (1, 2)), row)
    np.testing.assert_array_equal(imresize.imresize(np.array([[0.5, 1.5]]), (1, 2)), [[0, 255]])
    # channel axis: the first axis of length 3 (MATLAB's images arrive as (3, W, H))
    chw = np.arange(3 * 2 * 4, dtype=np.uint8).reshape(3, 2, 4)
    np.testing.assert_array_equal(imresize.imresize(chw, (2, 4)), np.moveaxis(chw, 0, 2))
    # a 4 -> 2 shrink: support 2, centres 1 and 3: weights (.75 .75 .25)/1.75 on pixels 0..2 and (.25 .75 .75)/1.75 on 1..3
    v = np.array([[10, 20, 30, 40]], np.uint8)
    w = np.array([0.75, 0.75, 0.25]) / 1.75
    k = (0.5 + w * (1 << 22)).astype(np.int64)
    first = ((1 << 21) + int((k * np.array([10, 20, 30])).sum())) >> 22
    last = ((1 << 21) + int((k[::-1] * np.array([20, 30, 40])).sum())) >> 22
    np.testing.assert_array_equal(imresize.imresize(v, (1, 2)), [[first, last]])
    assert (first, last) == (17, 33)
