"""PNG without PIL / scipy.misc: what the reference's preprocessor writes (`smisc.imsave`, tools/data_preprocessor.py:
100-101) and its converter reads (`smisc.imread`, tools/data_tf_converter.py:36-37) — 8-bit grey / RGB (also grey+alpha,
RGBA and palette images), non-interlaced."""
import struct
import zlib

import numpy as np

SIGNATURE = b'\x89PNG\r\n\x1a\n'


def encode_png(img):
    """img uint8 [H,W] / [H,W,1] (grey) or [H,W,3] (RGB) -> PNG bytes (filter 0, zlib)."""
    img = np.ascontiguousarray(img, np.uint8)
    if img.ndim == 2:
        img = img[..., None]
    h, w, c = img.shape
    color_type = {1: 0, 3: 2}[c]

    def chunk(kind, data):
        body = kind + data
        return struct.pack('>I', len(data)) + body + struct.pack('>I', zlib.crc32(body) & 0xFFFFFFFF)
    raw = b''.join(b'\x00' + img[y].tobytes() for y in range(h))
    return (SIGNATURE + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, color_type, 0, 0, 0)) +
            chunk(b'IDAT', zlib.compress(raw, 6)) + chunk(b'IEND', b''))


def decode_png(data):
    """PNG bytes -> uint8 array, [H,W] for grey, [H,W,C] otherwise (palette images come back as RGB), like
    scipy.misc.imread.  Chunk CRCs are verified."""
    if data[:8] != SIGNATURE:
        raise ValueError('not a PNG file')
    pos, idat, ihdr, palette = 8, [], None, None
    while pos < len(data):
        n, kind = struct.unpack_from('>I4s', data, pos)
        body = data[pos + 8:pos + 8 + n]
        if zlib.crc32(kind + body) & 0xFFFFFFFF != struct.unpack_from('>I', data, pos + 8 + n)[0]:
            raise ValueError(f'PNG: bad CRC in {kind.decode("latin1")} chunk')
        pos += 12 + n
        if kind == b'IHDR':
            ihdr = struct.unpack('>IIBBBBB', body)
        elif kind == b'PLTE':
            palette = np.frombuffer(body, np.uint8).reshape(-1, 3)
        elif kind == b'IDAT':
            idat.append(body)
        elif kind == b'IEND':
            break
    if ihdr is None:
        raise ValueError('PNG: no IHDR chunk')
    w, h, depth, color_type, _, _, interlace = ihdr
    if depth != 8 or interlace != 0 or color_type not in (0, 2, 3, 4, 6):
        raise ValueError(f'PNG: unsupported format (bit depth {depth}, colour type {color_type}, interlace {interlace})')
    c = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color_type]
    stride = w * c
    raw = np.frombuffer(zlib.decompress(b''.join(idat)), np.uint8)
    if raw.size != h * (stride + 1):
        raise ValueError('PNG: image data has the wrong size')
    raw = raw.reshape(h, stride + 1)
    out = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    for y in range(h):
        f, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        if f == 0:
            cur = line
        elif f == 2:                                              # Up
            cur = (line + prev) & 255
        elif f == 1:                                              # Sub: running sum per channel, modulo 256
            cur = (np.cumsum(line.reshape(-1, c), axis=0) & 255).reshape(-1)
        elif f in (3, 4):                                         # Average / Paeth: serial along the row
            cur = np.zeros(stride, np.int32)
            lv, pv = line.tolist(), prev.tolist()
            cv = [0] * stride
            for i in range(stride):
                a = cv[i - c] if i >= c else 0
                b = pv[i]
                if f == 3:
                    pred = (a + b) >> 1
                else:
                    cc = pv[i - c] if i >= c else 0
                    p = a + b - cc
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - cc)
                    pred = a if pa <= pb and pa <= pc else (b if pb <= pc else cc)
                cv[i] = (lv[i] + pred) & 255
            cur = np.array(cv, np.int32)
        else:
            raise ValueError(f'PNG: unknown filter type {f}')
        out[y] = cur
        prev = cur
    if color_type == 3:
        if palette is None:
            raise ValueError('PNG: palette image without PLTE chunk')
        return palette[out]
    return out if c == 1 else out.reshape(h, w, c)


def imread(path):
    """Image file -> uint8 array.  Pillow does it when it is installed (C speed; PNG is lossless, so the pixels are the
    same); decode_png is the dependency-free fallback (Sub / Average / Paeth rows cost ~1 s per 480x640 RGB image)."""
    try:
        from PIL import Image
    except ImportError:
        with open(path, 'rb') as f:
            return decode_png(f.read())
    with Image.open(path) as im:
        if im.mode == 'P':
            im = im.convert('RGB')
        elif im.mode not in ('L', 'LA', 'RGB', 'RGBA'):
            raise ValueError(f'{path}: unsupported image mode {im.mode}')
        return np.asarray(im).copy()


def imsave(path, img):
    with open(path, 'wb') as f:
        f.write(encode_png(img))
