"""Pure-Python restatement of the TFRecord / tf.train.Example wire layout ann3depth reads and writes.

TEST INFRASTRUCTURE ONLY — see ``oracle/tf13_ops.py``.  PARITY UNPINNED (no sample shard ships with the
reference); the CRC is pinned by the RFC 3720 CRC32C check values, the protobuf encoding is cross-checked in
tests against ``google.protobuf`` with a dynamically built ``Example`` descriptor.

Writer side: ``/root/reference/tools/data_tf_converter.py:27-53`` (8 features, image/depth stored as
row-major HWC little-endian float32 of ``png/255 - 0.5``).  Reader side: ``/root/reference/src/data.py:70-86``
(``decode_raw`` float32, reshape, ``+ 0.5``).  Container framing is TF's record writer:
``u64 length | u32 masked_crc32c(length) | payload | u32 masked_crc32c(payload)``.
Slow byte-at-a-time loops: small cases only.
"""
import struct

import numpy as np

_POLY = 0x82F63B78
_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ (_POLY if _c & 1 else 0)
    _TABLE.append(_c)


def crc32c(data, crc=0):
    crc ^= 0xFFFFFFFF
    for b in bytes(data):
        crc = _TABLE[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def masked_crc(data):
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def frame(payload):
    head = struct.pack('<Q', len(payload))
    return head + struct.pack('<I', masked_crc(head)) + payload + struct.pack('<I', masked_crc(payload))


def unframe(buf):
    """Yield payloads from a bytes object holding whole records; raises ValueError on CRC mismatch."""
    pos = 0
    while pos < len(buf):
        head = buf[pos:pos + 8]
        (n,) = struct.unpack('<Q', head)
        (hc,) = struct.unpack('<I', buf[pos + 8:pos + 12])
        if hc != masked_crc(head):
            raise ValueError('corrupt record length')
        payload = buf[pos + 12:pos + 12 + n]
        (pc,) = struct.unpack('<I', buf[pos + 12 + n:pos + 16 + n])
        if pc != masked_crc(payload):
            raise ValueError('corrupt record payload')
        yield payload
        pos += 16 + n


def _varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(field, payload):
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _feature_int64(v):
    # Feature{ int64_list = 3 : Int64List{ value = 1 (packed) } }
    return _ld(3, _ld(1, _varint(v)))


def _feature_bytes(b):
    # Feature{ bytes_list = 1 : BytesList{ value = 1 } }
    return _ld(1, _ld(1, b))


def encode_example(image, depth):
    """image [H,W,3] / depth [H,W(,1)] float32 already in stored form (png/255 - 0.5)."""
    image = np.ascontiguousarray(image, dtype='<f4')
    depth = np.ascontiguousarray(depth, dtype='<f4')
    if depth.ndim < 3:
        depth = depth[..., None]
    feats = {
        'image_height': _feature_int64(image.shape[0]),
        'image_width': _feature_int64(image.shape[1]),
        'image_channels': _feature_int64(image.shape[2]),
        'depth_height': _feature_int64(depth.shape[0]),
        'depth_width': _feature_int64(depth.shape[1]),
        'depth_channels': _feature_int64(depth.shape[2]),
        'image': _feature_bytes(image.tobytes()),
        'depth': _feature_bytes(depth.tobytes()),
    }
    entries = b''
    for k in sorted(feats):            # protobuf map order is unspecified; sorted = deterministic
        entries += _ld(1, _ld(1, k.encode()) + _ld(2, feats[k]))
    return _ld(1, entries)             # Example{ features = 1 : Features{ feature = 1 : map } }


def _read_varint(buf, pos):
    v = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        v |= (b & 0x7F) << shift
        if not b & 0x80:
            return v, pos
        shift += 7


def _fields(buf):
    pos = 0
    while pos < len(buf):
        key, pos = _read_varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _read_varint(buf, pos)
        elif wt == 2:
            n, pos = _read_varint(buf, pos)
            v = buf[pos:pos + n]
            pos += n
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        else:
            raise ValueError('wire type %d' % wt)
        yield field, wt, v


def decode_example(payload):
    """-> dict name -> int | bytes for the Int64List / BytesList features (first value only)."""
    out = {}
    for f, _, features in _fields(payload):
        if f != 1:
            continue
        for f2, _, entry in _fields(features):
            if f2 != 1:
                continue
            key, feat = None, None
            for f3, _, v in _fields(entry):
                if f3 == 1:
                    key = bytes(v).decode()
                elif f3 == 2:
                    feat = v
            for kind, _, lst in _fields(feat):
                for f5, wt, v in _fields(lst):
                    if f5 != 1:
                        continue
                    if kind == 3:                        # Int64List: packed or unpacked varints
                        val = _read_varint(v, 0)[0] if wt == 2 else v
                        out[key] = val - (1 << 64) if val >> 63 else val
                    elif kind == 1:
                        out[key] = bytes(v)
                    break
    return out


def convert_img_depth(payload):
    """data._convert_img_depth (src/data.py:70-86) with the reference's hard-coded 480x640 replaced by the
    record's own size fields.  Returns (image [H,W,3], depth [H,W,1]) float32, '+ 0.5'."""
    ex = decode_example(payload)
    img = np.frombuffer(ex['image'], '<f4').reshape(ex['image_height'], ex['image_width'], ex['image_channels'])
    dep = np.frombuffer(ex['depth'], '<f4').reshape(ex['depth_height'], ex['depth_width'], ex['depth_channels'])
    return img + np.float32(0.5), dep + np.float32(0.5)
