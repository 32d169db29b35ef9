"""Dataset plugin — the surface of the reference's ``src/data.py``: the ``Pipeline`` tuple, a registry keyed by
dataset name with a default, and ``inputs(datadir, dataset, batch_size, train_or_test, epochs)`` returning the
(inputs, targets) pair the driver hands to the model plugin (src/ann3depth.py:144-145).

The reference builds TF queue ops: ``string_input_producer`` (one file, no file shuffle, src/data.py:35-38) ->
``TFRecordReader`` (:18,62-67) -> ``_convert_img_depth`` (:70-86) -> ``shuffle_batch(capacity=20B,
min_after_dequeue=5B, num_threads=2)`` (:51-55).  Here the same stages run on host threads over a memory-mapped
file, with CRC / protobuf / decode in liba3d.so (the GIL is released inside those calls) and batches assembled in
pinned memory for asynchronous H2D copies.  With world_size > 1 each rank reads the records whose index is
congruent to its rank.
"""
import collections
import os
import threading

import numpy as np

from . import tfrecord

Pipeline = collections.namedtuple('Pipeline', ('files', 'labels', 'reader', 'convert'))


class OutOfRangeError(Exception):
    """All epochs consumed (tf.errors.OutOfRangeError: MonitoredSession turns it into should_stop())."""


def _files_tfrecords(base_dir, train_or_test='train'):
    return [os.path.join(base_dir, f'{train_or_test}.tfrecords')]          # src/data.py:58-59


def _read(files, epochs, rank=0, world=1):
    """TFRecordReader over string_input_producer(files, num_epochs=epochs, shuffle=False): yields
    (RecordFile, offset, length); epochs=None cycles forever."""
    epoch = 0
    while epochs is None or epoch < epochs:
        n = 0
        for path in files:
            rf = tfrecord.RecordFile(path)
            for i, (off, ln) in enumerate(rf):
                if i % world == rank:
                    n += 1
                    yield rf, off, ln
        if n == 0:
            raise OutOfRangeError(f'no records in {files}')
        epoch += 1


def _convert_img_depth(rf, off, ln):
    return rf.parse(off, ln)                                                # src/data.py:70-86


def _get_pipeline(dataset):
    default = Pipeline(files=_files_tfrecords, labels=None, reader=_read, convert=_convert_img_depth)
    pipelines = {'make3d1': default, 'make3d2': default, 'nyu': default}   # src/data.py:15-25
    return pipelines.get(dataset, default)


class ShuffleBatch:
    """tf.train.shuffle_batch: a bounded random-shuffle queue fed by `num_threads` producer threads.
    next_batch() blocks until more than min_after_dequeue elements would remain (or the producers finished), then
    removes batch_size uniformly chosen elements."""

    def __init__(self, records, convert, batch_size, capacity, min_after_dequeue, num_threads=2, seed=None):
        self.records, self.convert = records, convert
        self.B, self.capacity, self.min_after = batch_size, capacity, min_after_dequeue
        self.rng = np.random.default_rng(seed)
        self.queue = []
        self.cv = threading.Condition()
        self.src_lock = threading.Lock()
        self.live = num_threads
        self.error = None
        self.closed = False
        self.threads = [threading.Thread(target=self._produce, daemon=True) for _ in range(num_threads)]
        self.started = False

    def _produce(self):
        try:
            while True:
                with self.src_lock:                 # the reader op is shared: each record goes to one thread
                    try:
                        rec = next(self.records)
                    except StopIteration:
                        break
                item = self.convert(*rec)           # CRC-checked parse + decode in C, outside the GIL
                with self.cv:
                    while len(self.queue) >= self.capacity and not self.closed:
                        self.cv.wait()
                    if self.closed:
                        return
                    self.queue.append(item)
                    self.cv.notify_all()
        except BaseException as e:                  # surfaced by next_batch(): never swallow a corrupt record
            with self.cv:
                self.error = e
        finally:
            with self.cv:
                self.live -= 1
                self.cv.notify_all()

    def start(self):
        if not self.started:
            self.started = True
            for t in self.threads:
                t.start()

    def close(self):
        with self.cv:
            self.closed = True
            self.cv.notify_all()

    def next_batch(self, out_images=None, out_depths=None):
        self.start()
        with self.cv:
            while True:
                if self.error is not None:
                    raise self.error
                if len(self.queue) >= self.min_after + self.B:
                    break
                if self.live == 0:
                    if len(self.queue) >= self.B:
                        break
                    raise OutOfRangeError('input queue is closed and has insufficient elements')
                self.cv.wait()
            picks = []
            for _ in range(self.B):
                i = int(self.rng.integers(len(self.queue)))
                self.queue[i], self.queue[-1] = self.queue[-1], self.queue[i]
                picks.append(self.queue.pop())
            self.cv.notify_all()
        ishape, dshape = picks[0][0].shape, picks[0][1].shape
        if out_images is None:
            out_images = np.empty((self.B,) + ishape, np.float32)
            out_depths = np.empty((self.B,) + dshape, np.float32)
        for b, (img, dep) in enumerate(picks):
            if img.shape != ishape or dep.shape != dshape:
                raise ValueError(f'records of different sizes cannot be batched: {img.shape} vs {ishape}')
            out_images[b] = img
            out_depths[b] = dep
        return out_images, out_depths


class BatchHandle:
    """What `inputs()` returns in place of a TF tensor: one output (0 = inputs, 1 = targets) of a ShuffleBatch."""

    def __init__(self, pipeline, index):
        self.pipeline, self.index = pipeline, index


def inputs(datadir, dataset, batch_size=32, train_or_test='train', epochs=None, rank=0, world=1, seed=None):
    """src/data.py:28-55.  Returns (inputs, targets) handles; `inputs.pipeline.next_batch()` dequeues
    ([B,H,W,3], [B,H',W',1]) float32 arrays."""
    epochs = epochs if train_or_test == 'train' else 1
    pipeline = _get_pipeline(dataset)
    base_dir = os.path.join(datadir, dataset)
    files = pipeline.files(base_dir, train_or_test)
    for f in files:
        if not os.path.exists(f):
            raise FileNotFoundError(f)
    records = pipeline.reader(files, epochs, rank, world)
    sb = ShuffleBatch(records, pipeline.convert, batch_size, capacity=20 * batch_size,
                      min_after_dequeue=5 * batch_size, num_threads=2, seed=seed)
    return BatchHandle(sb, 0), BatchHandle(sb, 1)
