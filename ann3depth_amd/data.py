"""Dataset plugin — the surface of the reference's ``src/data.py``: the ``Pipeline`` tuple, a registry keyed by
dataset name with a default, and ``inputs(datadir, dataset, batch_size, train_or_test, epochs)`` returning the
(inputs, targets) pair the driver hands to the model plugin (src/ann3depth.py:144-145).

The reference builds TF queue ops: ``string_input_producer`` (one file, no file shuffle, src/data.py:35-38) ->
``TFRecordReader`` (:18,62-67) -> ``_convert_img_depth`` (:70-86) -> ``shuffle_batch(capacity=20B,
min_after_dequeue=5B, num_threads=2)`` (:51-55).  Here the same stages run on host threads over a memory-mapped
file: CRC check, protobuf parse and ``decode_raw + 0.5`` happen in liba3d.so (GIL released) and write straight
into a slot of a staging pool; the shuffle queue holds slot numbers, so a record is copied exactly once on the host
(into pinned memory when a GPU consumer asks for it) and then DMA'd to HBM.  With world_size > 1 each rank reads the
records whose index is congruent to its rank.

Records written by the converter hold png_u8 / 255 - 0.5 (tools/data_tf_converter.py:36-37): a feature whose floats all
have that form — checked bit for bit while the record is decoded — is staged as the uint8 pixel values instead (a quarter
of the bytes through pinned memory and the host link), and the consumer rebuilds exactly the float32 the loader's `+ 0.5`
(src/data.py:84-85) would have produced: `expand_u8` on the host, the resize kernel's table on the device.
"""
import collections
import ctypes
import os
import threading

import numpy as np

from . import _lib, tfrecord

Pipeline = collections.namedtuple('Pipeline', ('files', 'labels', 'reader', 'convert'))


class OutOfRangeError(Exception):
    """All epochs consumed (tf.errors.OutOfRangeError: MonitoredSession turns it into should_stop())."""


def _files_tfrecords(base_dir, train_or_test='train'):
    return [os.path.join(base_dir, f'{train_or_test}.tfrecords')]          # src/data.py:58-59


def _read(files, epochs, rank=0, world=1):
    """TFRecordReader over string_input_producer(files, num_epochs=epochs, shuffle=False): yields
    (RecordFile, offset, length); epochs=None cycles forever."""
    epoch = 0
    handles = {}
    while epochs is None or epoch < epochs:
        n = 0
        for path in files:
            rf = handles.get(path)
            if rf is None:
                rf = handles[path] = tfrecord.RecordFile(path)
            for i, (off, ln) in enumerate(rf.frames()):     # payload CRCs are checked by the reader threads
                if i % world == rank:
                    n += 1
                    yield rf, off, ln
        if n == 0:
            raise OutOfRangeError(f'no records in {files}')
        epoch += 1


def _convert_img_depth(rf, off, ln, image, depth):
    """src/data.py:70-86, decoding into the destination arrays; the record's payload CRC is checked in the same pass."""
    rf.decode_into(off, ln, image, depth)


def _get_pipeline(dataset):
    default = Pipeline(files=_files_tfrecords, labels=None, reader=_read, convert=_convert_img_depth)
    pipelines = {'make3d1': default, 'make3d2': default, 'nyu': default}   # src/data.py:15-25
    return pipelines.get(dataset, default)


def usable_cpus():
    """CPUs this process may actually use: the smallest of the host's count, the affinity mask and the cgroup's CPU quota
    (a GPU box of this pipeline shows 256 hardware threads and a quota of 16)."""
    n = os.cpu_count() or 2
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def default_reader_threads():
    """The reference uses num_threads=2 (src/data.py:55), enough for a 2017 GPU; an MI355X eats ~11.4 k records/s.  Half of
    the CPUs the process may use, at most eight: a reader that wakes needs the interpreter lock, which the thread that launches
    the training step takes ~50 times per step, so MORE readers than the decode rate needs make the loop slower (round 6 on
    a 16-CPU quota: 6 / 8 readers 0.97 of the resident step rate, 12 readers 0.89, 16 readers 0.91; round 5, with every
    reader woken at every dequeue and release, 12 readers 0.85 — profiles/r06_input_pipeline.json).  Each reader takes
    RECORDS_PER_CALL records per call into liba3d.so (decode outside the interpreter lock)."""
    env = os.environ.get('A3D_READER_THREADS')
    return int(env) if env else max(2, min(8, usable_cpus() // 2))


def expand_u8(k):
    """uint8 pixel values -> the float32 the reference's loader hands to the model: fl(fl(fl(k / 255) - 0.5) + 0.5), in
    numpy's float32 operations (converter: `.astype(np.float32) / 255. - .5`; loader: `+ 0.5`)."""
    lut = (np.arange(256, dtype=np.float32) / np.float32(255.) - np.float32(.5)) + np.float32(.5)
    return lut[np.asarray(k)]


class ShuffleBatch:
    """tf.train.shuffle_batch: a bounded random-shuffle queue fed by `num_threads` producer threads.
    dequeue() blocks until more than min_after_dequeue elements would remain (or the producers finished), then
    removes batch_size uniformly chosen elements.  Elements live in a slot pool (`images[slot]`, `depths[slot]`);
    the consumer gives slots back with release()."""

    def __init__(self, records, convert, batch_size, capacity, min_after_dequeue, num_threads=2, seed=None,
                 in_flight_batches=3):
        self.records, self.convert = records, convert
        self.B, self.capacity, self.min_after = batch_size, capacity, min_after_dequeue
        self.nslots = capacity + in_flight_batches * batch_size
        self.rng = np.random.default_rng(seed)
        self.queue = []                       # filled slot numbers
        self.free = None                      # free slot numbers (after allocate())
        self.images = self.depths = None
        self.images_u8 = self.depths_u8 = None    # uint8 twins of the pool (allocate(..., alloc_u8)): converter-written records
        self.kind = None                          # per slot: bit 0 image is in images_u8, bit 1 depth is in depths_u8
        # one lock, two wait sets: producers sleep on `cv` (room in the queue / a free slot), the consumer on `ready` (enough
        # elements).  A thread that wakes needs the interpreter lock, and so does the thread that launches the training step
        # ~50 times per step: nobody is woken who cannot make progress (notify_all on one condition woke every reader at
        # every dequeue and release: 0.85 of the resident step rate with 12 readers, tools/bench_input.py)
        self.lock = threading.Lock()
        self.cv = threading.Condition(self.lock)
        self.ready = threading.Condition(self.lock)
        self.src_lock = threading.Lock()
        self.live = num_threads
        self.error = None
        self.closed = False
        self.threads = [threading.Thread(target=self._produce, daemon=True) for _ in range(num_threads)]
        self.started = False
        self.first = None

    # ---- staging pool ----
    def shapes(self):
        """(image shape, depth shape) of the dataset, from its first record."""
        if self.first is None:
            try:
                self.first = next(self.records)
            except StopIteration:
                raise OutOfRangeError('empty dataset')
        rf, off, ln = self.first
        rf.verify_payload(off, ln)
        _, ishape, dshape = rf.header(off, ln)
        return ishape, dshape

    def allocate(self, alloc=None, alloc_u8=None):
        """Create the slot pool.  alloc(shape) -> float32 ndarray; default plain numpy, a GPU consumer passes an
        allocator of pinned memory.  alloc_u8(shape) -> uint8 ndarray turns on the uint8 staging of converter-written
        records (module docstring); a slot then holds each feature in ONE of its two arrays, see `kind`."""
        if self.images is not None:
            return
        ishape, dshape = self.shapes()
        alloc = alloc or (lambda shape: np.empty(shape, np.float32))
        self.images = alloc((self.nslots,) + tuple(ishape))
        self.depths = alloc((self.nslots,) + tuple(dshape))
        if alloc_u8 is not None and self.convert is _convert_img_depth and os.environ.get('A3D_NO_U8_RECORDS', '0') != '1':
            self.images_u8 = alloc_u8((self.nslots,) + tuple(ishape))
            self.depths_u8 = alloc_u8((self.nslots,) + tuple(dshape))
        self.kind = np.zeros(self.nslots, np.uint8)
        self.free = collections.deque(range(self.nslots))

    def materialise(self, slot, which):
        """Make sure feature `which` (0 image, 1 depth) of a slot exists as float32 (a batch that mixes uint8-staged and
        float32 records goes to the device as float32); returns the float32 array of the slot."""
        f32, u8 = (self.images, self.images_u8) if which == 0 else (self.depths, self.depths_u8)
        if self.kind[slot] & (1 << which):
            f32[slot] = expand_u8(u8[slot])
            self.kind[slot] &= ~(1 << which) & 0xff
        return f32[slot]

    RECORDS_PER_CALL = 8      # a reader thread decodes this many records per call into liba3d.so (one lock round trip each)

    def _next_records(self, want):
        """Up to `want` records (at least one, or StopIteration): each record goes to exactly one thread."""
        with self.src_lock:                    # the reader op is shared
            recs = []
            if self.first is not None:
                recs.append(self.first)
                self.first = None
            while len(recs) < want:
                try:
                    recs.append(next(self.records))
                except StopIteration:
                    if not recs:
                        raise
                    break
            return recs

    def _decode(self, recs, slots):
        """Records -> slots, in one call into the library when the pipeline's converter is the default one."""
        if self.convert is not _convert_img_depth:
            for (rf, off, ln), slot in zip(recs, slots):
                self.convert(rf, off, ln, self.images[slot], self.depths[slot])
            return
        n = len(recs)
        frames = (ctypes.c_void_p * n)(*[rf.base + off - 12 for rf, off, _ in recs])
        lens = (ctypes.c_size_t * n)(*[ln + 16 for _, _, ln in recs])
        ids = (ctypes.c_int32 * n)(*slots)
        kinds = (ctypes.c_int32 * n)()
        dims = (ctypes.c_int64 * 6)(*(tuple(self.images.shape[1:]) + tuple(self.depths.shape[1:])))
        u8 = self.images_u8 is not None
        _lib.check(_lib.load().a3d_records_decode(frames, lens, n, 1, dims, self.images_u8.ctypes.data if u8 else None,
                                                  self.images.ctypes.data, self.depths_u8.ctypes.data if u8 else None,
                                                  self.depths.ctypes.data, ids, len(self.images), kinds),
                   f'a3d_records_decode({recs[0][0].path}@{recs[0][1]})')
        for slot, k in zip(slots, kinds):
            self.kind[slot] = k

    def _produce(self):
        try:
            while True:
                try:
                    recs = self._next_records(self.RECORDS_PER_CALL)
                except StopIteration:
                    break
                while recs:
                    with self.cv:
                        while (len(self.queue) >= self.capacity or not self.free) and not self.closed:
                            self.cv.wait()
                        if self.closed:
                            return
                        room = max(1, min(len(recs), len(self.free), self.capacity - len(self.queue)))
                        slots = [self.free.popleft() for _ in range(room)]
                    self._decode(recs[:room], slots)                         # C code, outside the GIL
                    recs = recs[room:]
                    with self.cv:
                        self.queue.extend(slots)
                        if len(self.queue) >= self.min_after + self.B:      # (the consumer sleeps below that)
                            self.ready.notify()
        except BaseException as e:                  # surfaced by dequeue(): never swallow a corrupt record
            with self.cv:
                self.error = e
                self.ready.notify_all()
        finally:
            with self.cv:
                self.live -= 1
                self.ready.notify_all()

    def start(self):
        if not self.started:
            self.allocate()
            self.started = True
            for t in self.threads:
                t.start()

    def _wake_producers(self, nslots):
        # (lock held) as many readers as the slots that became usable give work to, not all of them
        self.cv.notify(max(1, -(-nslots // self.RECORDS_PER_CALL)))

    def close(self):
        with self.cv:
            self.closed = True
            self.cv.notify_all()
            self.ready.notify_all()

    def dequeue(self):
        """-> list of batch_size slot numbers (their contents stay valid until release())."""
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
                self.ready.wait()
            # batch_size uniformly chosen distinct elements (RandomShuffleQueue.dequeue_many), drawn in one call: removing
            # the chosen positions from the back keeps the positions still to be removed valid
            chosen = self.rng.choice(len(self.queue), self.B, replace=False)
            picks = [self.queue[i] for i in chosen]
            for i in sorted(chosen.tolist(), reverse=True):
                self.queue[i] = self.queue[-1]
                self.queue.pop()
            self._wake_producers(min(self.B, len(self.free)))
        return picks

    def release(self, slots):
        with self.cv:
            self.free.extend(slots)
            self._wake_producers(len(slots))

    def next_batch(self, out_images=None, out_depths=None):
        """Dequeue into freshly stacked (or caller-provided) host arrays: ([B,H,W,3], [B,H',W',1]) float32."""
        slots = self.dequeue()
        if out_images is None:
            out_images = np.empty((self.B,) + self.images.shape[1:], np.float32)
            out_depths = np.empty((self.B,) + self.depths.shape[1:], np.float32)
        for b, s in enumerate(slots):
            out_images[b] = self.materialise(s, 0)
            out_depths[b] = self.materialise(s, 1)
        self.release(slots)
        return out_images, out_depths


class BatchHandle:
    """What `inputs()` returns in place of a TF tensor: one output (0 = inputs, 1 = targets) of a ShuffleBatch."""

    def __init__(self, pipeline, index):
        self.pipeline, self.index = pipeline, index


def inputs(datadir, dataset, batch_size=32, train_or_test='train', epochs=None, rank=0, world=1, seed=None,
           num_threads=None):
    """src/data.py:28-55.  Returns (inputs, targets) handles; `inputs.pipeline.next_batch()` dequeues
    ([B,H,W,3], [B,H',W',1]) float32 arrays, `.dequeue()` the slots of the staging pool."""
    epochs = epochs if train_or_test == 'train' else 1
    pipeline = _get_pipeline(dataset)
    base_dir = os.path.join(datadir, dataset)
    files = pipeline.files(base_dir, train_or_test)
    for f in files:
        if not os.path.exists(f):
            raise FileNotFoundError(f)
    records = pipeline.reader(files, epochs, rank, world)
    sb = ShuffleBatch(records, pipeline.convert, batch_size, capacity=20 * batch_size,
                      min_after_dequeue=5 * batch_size, num_threads=num_threads or default_reader_threads(),
                      seed=seed)
    return BatchHandle(sb, 0), BatchHandle(sb, 1)
