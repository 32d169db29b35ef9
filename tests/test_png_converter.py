"""PNG codec and the TF-free `make convert` (tools/data_tf_converter.py of the reference)."""
import os
import struct
import zlib

import numpy as np
import pytest

from ann3depth_amd import png, tfrecord


def _png_with_filters(img, filters):
    """Encode with a chosen filter type per row (the encoder in png.py only writes filter 0)."""
    h, w, c = img.shape
    stride = w * c
    flat = img.reshape(h, stride).astype(np.int32)
    rows = []
    prev = np.zeros(stride, np.int32)
    for y in range(h):
        f = filters[y % len(filters)]
        cur = flat[y]
        left = np.concatenate([np.zeros(c, np.int32), cur[:-c]])
        upleft = np.concatenate([np.zeros(c, np.int32), prev[:-c]])
        if f == 0:
            pred = 0
        elif f == 1:
            pred = left
        elif f == 2:
            pred = prev
        elif f == 3:
            pred = (left + prev) >> 1
        else:
            p = left + prev - upleft
            pa, pb, pc = abs(p - left), abs(p - prev), abs(p - upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
        rows.append(bytes([f]) + ((cur - pred) & 255).astype(np.uint8).tobytes())
        prev = cur

    def chunk(kind, data):
        body = kind + data
        return struct.pack('>I', len(data)) + body + struct.pack('>I', zlib.crc32(body) & 0xFFFFFFFF)
    ct = {1: 0, 2: 4, 3: 2, 4: 6}[c]
    raw = zlib.compress(b''.join(rows))
    return (png.SIGNATURE + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, ct, 0, 0, 0)) +
            chunk(b'IDAT', raw[:len(raw) // 2]) + chunk(b'IDAT', raw[len(raw) // 2:]) + chunk(b'IEND', b''))


@pytest.mark.parametrize('c', [1, 2, 3, 4])
def test_decode_every_filter_type(c):
    rng = np.random.default_rng(c)
    img = rng.integers(0, 256, (11, 7, c)).astype(np.uint8)
    out = png.decode_png(_png_with_filters(img, [0, 1, 2, 3, 4]))
    np.testing.assert_array_equal(out, img[..., 0] if c == 1 else img)


def test_round_trip_and_pil_agreement(tmp_path):
    rng = np.random.default_rng(0)
    rgb = rng.integers(0, 256, (48, 64, 3)).astype(np.uint8)
    grey = rng.integers(0, 256, (6, 8)).astype(np.uint8)
    np.testing.assert_array_equal(png.decode_png(png.encode_png(rgb)), rgb)
    np.testing.assert_array_equal(png.decode_png(png.encode_png(grey)), grey)
    Image = pytest.importorskip('PIL.Image')
    p = str(tmp_path / 'x.png')
    Image.fromarray(rgb).save(p, optimize=True)                    # PIL picks adaptive filters
    np.testing.assert_array_equal(png.imread(p), rgb)
    np.testing.assert_array_equal(png.decode_png(open(p, 'rb').read()), rgb)          # own decoder on PIL's filters
    Image.fromarray(grey).convert('P').save(p)                     # palette image -> RGB, like smisc.imread
    np.testing.assert_array_equal(png.imread(p), np.asarray(Image.open(p).convert('RGB')))
    np.testing.assert_array_equal(png.decode_png(open(p, 'rb').read()), np.asarray(Image.open(p).convert('RGB')))
    png.imsave(p, rgb)
    np.testing.assert_array_equal(np.asarray(Image.open(p)), rgb)
    bad = bytearray(png.encode_png(rgb))
    bad[40] ^= 1
    with pytest.raises(ValueError, match='CRC'):
        png.decode_png(bytes(bad))


def test_convert_matches_the_reference_record_layout(tmp_path, monkeypatch):
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        'data_tf_converter', os.path.join(os.path.dirname(__file__), '..', 'tools', 'data_tf_converter.py'))
    conv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(conv)
    rng = np.random.default_rng(5)
    pairs = {}
    for split, names in (('train', ['b_0002', 'a_0001', 'c_0003']), ('test', ['t_0001'])):
        d = tmp_path / 'nyu' / split
        d.mkdir(parents=True)
        for n in names:
            img = rng.integers(0, 256, (48, 64, 3)).astype(np.uint8)
            dep = rng.integers(0, 256, (6, 8)).astype(np.uint8)
            png.imsave(str(d / f'{n}-image.png'), img)
            png.imsave(str(d / f'{n}-depth.png'), dep)
            pairs[(split, n)] = (img, dep)
    monkeypatch.setenv('DATA_DIR', str(tmp_path))
    assert conv.main([]) == 1
    assert conv.main(['nyu', '--del_raw']) == 0
    assert os.listdir(tmp_path / 'nyu' / 'train') == [] and os.listdir(tmp_path / 'nyu' / 'test') == []
    for split, names in (('train', ['a_0001', 'b_0002', 'c_0003']), ('test', ['t_0001'])):
        rf = tfrecord.RecordFile(str(tmp_path / 'nyu' / f'{split}.tfrecords'))
        frames = list(rf.frames())
        assert len(frames) == len(names)
        for (off, length), n in zip(frames, names):
            image, depth = rf.parse(off, length)
            img, dep = pairs[(split, n)]
            # RecordFile.parse applies the loader's +0.5 (src/data.py:84-85): stored value is png/255 - 0.5
            np.testing.assert_array_equal(image, (img.astype(np.float32) / np.float32(255) - np.float32(.5)) + np.float32(.5))
            np.testing.assert_array_equal(depth[..., 0], (dep.astype(np.float32) / np.float32(255) - np.float32(.5)) + np.float32(.5))
            assert image.shape == (48, 64, 3) and depth.shape == (6, 8, 1)
        rf.close()
