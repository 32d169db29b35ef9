"""TensorBoard event files without TensorFlow: the scalar and image summaries the reference writes through
MonitoredTrainingSession(save_summaries_steps=sumfreq) — loss scalars (src/tfhelper.py:137-157), `Phase`
(src/models.py:361-365), `global_step/sec` (src/ann3depth.py:125) and the Input / Coarse / Fine / Target images
(src/models.py:292-296).  `make tb` of the reference (tensorboard --logdir=<ckptdir>) reads them unchanged.

An event file is a TFRecord stream (same framing / masked CRC32C as the dataset shards, computed by liba3d.so) of
`Event` protobufs, hand-encoded here: Event{1: wall_time f64, 2: step i64, 3: file_version | 5: Summary},
Summary{1: Value{1: tag, 2: simple_value f32 | 4: Image{1: h, 2: w, 3: colorspace, 4: png bytes}}}.
"""
import os
import socket
import struct
import time

import numpy as np

from . import tfrecord
from .png import encode_png  # noqa: F401  (re-exported; tests and the driver use summary.encode_png)


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


def _frame(payload):
    head = struct.pack('<Q', len(payload))
    return (head + struct.pack('<I', tfrecord.masked_crc32c(head)) + payload +
            struct.pack('<I', tfrecord.masked_crc32c(payload)))


def to_uint8(image):
    """tf.summary.image's float rule: all values >= 0 -> scale the max to 255; otherwise shift 0 to 127 and scale so
    the smallest value is 0 or the largest 255."""
    image = np.asarray(image, np.float32)
    lo, hi = float(image.min()), float(image.max())
    if lo >= 0:
        scale = 255.0 / hi if hi > 0 else 1.0
        out = image * scale
    else:
        scale = min(127.0 / -lo if lo < 0 else np.inf, 128.0 / hi if hi > 0 else np.inf)
        out = image * scale + 127.0
    return np.clip(out, 0, 255).astype(np.uint8)


class EventFileWriter:
    def __init__(self, logdir):
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, f'events.out.tfevents.{int(time.time())}.{socket.gethostname()}')
        self.f = open(self.path, 'ab')
        self._event(0, _ld(3, b'brain.Event:2'))

    def _event(self, step, what):
        ev = b'\x09' + struct.pack('<d', time.time()) + b'\x10' + _varint(step) + what
        self.f.write(_frame(ev))

    def add_scalars(self, step, scalars):
        values = b''.join(_ld(1, _ld(1, tag.encode()) + b'\x15' + struct.pack('<f', float(v)))
                          for tag, v in scalars.items())
        self._event(step, _ld(5, values))

    def add_images(self, step, tag, batch, max_outputs=3):
        """batch [N,H,W,C] float (C = 1 or 3): tags `<tag>/image/<i>` (`<tag>/image` when max_outputs == 1) like
        tf.summary.image."""
        values = b''
        for i in range(min(max_outputs, len(batch))):
            u8 = to_uint8(batch[i])
            h, w, c = u8.shape
            image = (b'\x08' + _varint(h) + b'\x10' + _varint(w) + b'\x18' + _varint(c) + _ld(4, encode_png(u8)))
            name = f'{tag}/image' if max_outputs == 1 else f'{tag}/image/{i}'
            values += _ld(1, _ld(1, name.encode()) + _ld(4, image))
        self._event(step, _ld(5, values))

    def flush(self):
        self.f.flush()

    def close(self):
        self.f.close()
