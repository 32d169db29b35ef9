"""TF-free TFRecord reader/writer for the 8-feature float32 layout ann3depth uses
(writer side: tools/data_tf_converter.py:27-53; reader side: src/data.py:62-86).

Framing, CRC32C and the protobuf wire format are handled by liba3d.so (csrc/tfrecord.cc); this module only maps
files and moves numpy buffers.
"""
import ctypes
import mmap
import os

import numpy as np

from . import _lib
from ._lib import ExampleView, check


class TFRecordWriter:
    """Context manager mirroring tf.python_io.TFRecordWriter for (image, depth) pairs."""

    def __init__(self, path):
        self.path = path
        self.f = None
        self.buf = None

    def __enter__(self):
        os.makedirs(os.path.dirname(os.path.abspath(self.path)), exist_ok=True)
        self.f = open(self.path, 'wb')
        return self

    def __exit__(self, *exc):
        self.f.close()

    def write_example(self, image, depth):
        """image [H,W,C], depth [H,W] or [H,W,1] in STORED form: float32 `png/255 - 0.5`
        (tools/data_tf_converter.py:36-40)."""
        lib = _lib.load()
        image = np.ascontiguousarray(image, dtype='<f4')
        depth = np.ascontiguousarray(depth, dtype='<f4')
        if depth.ndim < 3:
            depth = depth[..., None]
        need = 16 + image.nbytes + depth.nbytes + 512
        if self.buf is None or len(self.buf) < need:
            self.buf = ctypes.create_string_buffer(need)
        n = lib.a3d_example_write(image.ctypes.data, *image.shape, depth.ctypes.data, *depth.shape, self.buf, need)
        if n < 0 or n > need:
            check(int(min(n, -1)), 'a3d_example_write')
        self.f.write(self.buf.raw[:n] if n < len(self.buf) else self.buf.raw)


class RecordFile:
    """A memory-mapped .tfrecords file; iterating yields (offset, length) of each payload after CRC checks."""

    def __init__(self, path, verify_crc=True):
        self.path = path
        self.verify = int(verify_crc)
        self.f = open(path, 'rb')
        self.size = os.fstat(self.f.fileno()).st_size
        self.mm = mmap.mmap(self.f.fileno(), 0, access=mmap.ACCESS_READ) if self.size else None
        self.base = (ctypes.addressof(ctypes.c_char.from_buffer_copy(b'\0')) if self.mm is None else
                     np.frombuffer(self.mm, np.uint8).ctypes.data)

    def close(self):
        if self.mm is not None:
            self.mm.close()
        self.f.close()

    def __iter__(self):
        lib = _lib.load()
        pos = 0
        off, ln, used = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
        while pos < self.size:
            check(lib.a3d_tfrecord_next(self.base + pos, self.size - pos, self.verify, ctypes.byref(off),
                                        ctypes.byref(ln), ctypes.byref(used)), f'a3d_tfrecord_next({self.path}@{pos})')
            yield pos + off.value, ln.value
            pos += used.value

    def frames(self):
        """Like iteration, but only the 12-byte length header of each frame is checked (cheap, serial); the payload
        CRC is left to verify_payload(), which reader threads run in parallel."""
        lib = _lib.load()
        pos = 0
        off, ln, used = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
        while pos < self.size:
            check(lib.a3d_tfrecord_next(self.base + pos, self.size - pos, 0, ctypes.byref(off), ctypes.byref(ln),
                                        ctypes.byref(used)), f'a3d_tfrecord_next({self.path}@{pos})')
            yield pos + off.value, ln.value
            pos += used.value

    def verify_payload(self, offset, length):
        lib = _lib.load()
        want = int.from_bytes(self.mm[offset + length:offset + length + 4], 'little')
        if lib.a3d_masked_crc32c(self.base + offset, length) != want:
            raise _lib.A3dError(f'{self.path}: corrupt payload at offset {offset}')

    def header(self, offset, length):
        """Parse the Example at (offset, length): -> (ExampleView, image shape, depth shape)."""
        lib = _lib.load()
        ev = ExampleView()
        check(lib.a3d_example_parse(self.base + offset, length, ctypes.byref(ev)), 'a3d_example_parse')
        ishape = (ev.image_height, ev.image_width, ev.image_channels)
        dshape = (ev.depth_height, ev.depth_width, ev.depth_channels)
        if ev.image_bytes != 4 * int(np.prod(ishape)) or ev.depth_bytes != 4 * int(np.prod(dshape)):
            raise _lib.A3dError(f'{self.path}: feature byte counts do not match the size features '
                                f'({ev.image_bytes} vs {ishape}, {ev.depth_bytes} vs {dshape})')
        return ev, ishape, dshape

    def parse_into(self, offset, length, image, depth):
        """data._convert_img_depth (src/data.py:70-86) straight into caller-owned float32 arrays (e.g. slots of a
        pinned staging pool): decode_raw + reshape + '+ 0.5'."""
        lib = _lib.load()
        ev, ishape, dshape = self.header(offset, length)
        if tuple(image.shape) != ishape or tuple(depth.shape) != dshape:
            raise ValueError(f'{self.path}: record is {ishape}/{dshape}, destination is '
                             f'{tuple(image.shape)}/{tuple(depth.shape)}: records of different sizes cannot be batched')
        check(lib.a3d_decode_raw_plus_half(ev.image, ev.image_bytes, image.ctypes.data), 'a3d_decode_raw_plus_half')
        check(lib.a3d_decode_raw_plus_half(ev.depth, ev.depth_bytes, depth.ctypes.data), 'a3d_decode_raw_plus_half')

    def decode_into(self, offset, length, image, depth, verify_crc=True):
        """Reader fast path: CRC check + parse + decode of the record whose payload is at (offset, length), in one pass
        over its bytes (a3d_record_decode).  image / depth are caller-owned float32 arrays of the record's shapes."""
        ev = ExampleView()
        check(_lib.load().a3d_record_decode(self.base + offset - 12, length + 16, int(verify_crc), image.ctypes.data,
                                            image.size, depth.ctypes.data, depth.size, ctypes.byref(ev)),
              f'a3d_record_decode({self.path}@{offset})')
        ishape = (ev.image_height, ev.image_width, ev.image_channels)
        dshape = (ev.depth_height, ev.depth_width, ev.depth_channels)
        if tuple(image.shape) != ishape or tuple(depth.shape) != dshape:
            raise ValueError(f'{self.path}: record is {ishape}/{dshape}, destination is '
                             f'{tuple(image.shape)}/{tuple(depth.shape)}: records of different sizes cannot be batched')

    def decode_into_u8(self, offset, length, image_u8, image_f32, depth_u8, depth_f32, verify_crc=True):
        """decode_into for converter-written records (a3d_record_decode_u8): a feature whose floats all are
        png_u8 / 255 - 0.5 lands in its uint8 array as the pixel values, any other in its float32 array (+ 0.5).
        Returns (image_is_u8, depth_is_u8)."""
        ev = ExampleView()
        kinds = ctypes.c_int(0)
        check(_lib.load().a3d_record_decode_u8(self.base + offset - 12, length + 16, int(verify_crc), image_u8.ctypes.data,
                                               image_f32.ctypes.data, image_f32.size, depth_u8.ctypes.data,
                                               depth_f32.ctypes.data, depth_f32.size, ctypes.byref(ev), ctypes.byref(kinds)),
              f'a3d_record_decode_u8({self.path}@{offset})')
        ishape = (ev.image_height, ev.image_width, ev.image_channels)
        dshape = (ev.depth_height, ev.depth_width, ev.depth_channels)
        if tuple(image_f32.shape) != ishape or tuple(depth_f32.shape) != dshape:
            raise ValueError(f'{self.path}: record is {ishape}/{dshape}, destination is '
                             f'{tuple(image_f32.shape)}/{tuple(depth_f32.shape)}: records of different sizes cannot be batched')
        return bool(kinds.value & 1), bool(kinds.value & 2)

    def parse(self, offset, length):
        """-> (image [H,W,C], depth [H,W,C']) float32 with '+ 0.5'.  Sizes come from the record's own size features
        (the reference hard-codes 480x640, src/data.py:84-85)."""
        _, ishape, dshape = self.header(offset, length)
        image = np.empty(ishape, np.float32)
        depth = np.empty(dshape, np.float32)
        self.parse_into(offset, length, image, depth)
        return image, depth


def crc32c(data):
    data = bytes(data)
    return _lib.load().a3d_crc32c(data, len(data))


def masked_crc32c(data):
    data = bytes(data)
    return _lib.load().a3d_masked_crc32c(data, len(data))
