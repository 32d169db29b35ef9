"""TensorFlow V2 checkpoints ("tensor bundles") without TensorFlow — what tf.train.Saver inside the reference's
MonitoredTrainingSession (src/ann3depth.py:113-125) leaves in the checkpoint directory:

    checkpoint                               text: model_checkpoint_path: "model.ckpt-<step>"
    model.ckpt-<step>.index                  a LevelDB-format table: "" -> BundleHeaderProto, name -> BundleEntryProto
    model.ckpt-<step>.data-00000-of-00001    the tensors' bytes, back to back

so that a run started with the reference can be continued here and a run made here can be opened with TensorFlow
tooling.  Format restated from the published sources (LevelDB table_format.md; tensorflow/core/util/tensor_bundle,
tensorflow/core/protobuf/tensor_bundle.proto at v1.3):

  table  = data blocks, metaindex block (empty), index block, 48-byte footer
  block  = entries, restart offsets (u32 each), restart count (u32); then 1 type byte (0 = uncompressed) and the
           masked CRC32C of contents + type byte
  entry  = varint shared-key-bytes, varint unshared-key-bytes, varint value-bytes, key suffix, value
  footer = metaindex BlockHandle, index BlockHandle (varint64 offset, varint64 size), zero padding to 40 bytes,
           magic 0xdb4775248b80fb57 little-endian
  BundleHeaderProto{1: num_shards, 2: endianness (0 = little), 3: VersionDef{1: producer}}
  BundleEntryProto{1: dtype, 2: TensorShapeProto{2: Dim{1: size}}, 3: shard_id, 4: offset, 5: size,
                   6: fixed32 masked CRC32C of the tensor bytes, 7: slices (partitioned variables — not read here)}

NOT PINNED by a TensorFlow-written file (none ships with the reference and TensorFlow cannot run here); the tests check
the writer against the reader, both against the constants above, and the reader on tables with shared key prefixes,
several blocks and several restart points that the writer itself never produces.
"""
import os
import struct

import numpy as np

from . import _lib

MAGIC = 0xdb4775248b80fb57
DT_FLOAT, DT_INT32, DT_INT64 = 1, 3, 9               # tensorflow/core/framework/types.proto
_DTYPES = {DT_FLOAT: np.dtype('<f4'), DT_INT32: np.dtype('<i4'), DT_INT64: np.dtype('<i8')}
_DT_OF = {v: k for k, v in _DTYPES.items()}
RESTART_INTERVAL = 16                                 # LevelDB default for data blocks; index blocks use 1
BLOCK_SIZE = 262144                                   # tensorflow/core/lib/io/table_options.h


def _masked_crc(buf, extend=b''):
    """Masked CRC32C of a bytes-like / ndarray (+ `extend`), computed by liba3d.so."""
    lib = _lib.load()
    if isinstance(buf, np.ndarray):
        data = np.ascontiguousarray(buf).reshape(-1).view(np.uint8)
        if extend:
            data = np.concatenate([data, np.frombuffer(extend, np.uint8)])
        return lib.a3d_masked_crc32c(data.ctypes.data, data.size)
    data = bytes(buf) + extend
    return lib.a3d_masked_crc32c(data, len(data))


def _varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | 0x80 if v else b)
        if not v:
            return bytes(out)


def _read_varint(buf, pos):
    shift = val = 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7


# ---- protobuf (just these two messages) ----
def _field(num, wire, payload):
    return _varint((num << 3) | wire) + payload


def _ld(num, payload):
    return _field(num, 2, _varint(len(payload)) + payload)


def encode_header(num_shards=1):
    return _field(1, 0, _varint(num_shards)) + _ld(3, _field(1, 0, _varint(1)))      # endianness 0 is the default


def encode_entry(dtype, shape, offset, size, crc):
    dims = b''.join(_ld(2, _field(1, 0, _varint(int(d)))) for d in shape)
    out = _field(1, 0, _varint(dtype)) + _ld(2, dims)
    if offset:
        out += _field(4, 0, _varint(offset))
    out += _field(5, 0, _varint(size)) + _field(6, 5, struct.pack('<I', crc))
    return out


def _decode_fields(buf):
    pos, out = 0, []
    while pos < len(buf):
        key, pos = _read_varint(buf, pos)
        num, wire = key >> 3, key & 7
        if wire == 0:
            val, pos = _read_varint(buf, pos)
        elif wire == 2:
            n, pos = _read_varint(buf, pos)
            val = bytes(buf[pos:pos + n])
            pos += n
        elif wire == 5:
            val = struct.unpack_from('<I', buf, pos)[0]
            pos += 4
        elif wire == 1:
            val = struct.unpack_from('<Q', buf, pos)[0]
            pos += 8
        else:
            raise ValueError(f'tensor bundle: unsupported protobuf wire type {wire}')
        out.append((num, val))
    return out


def decode_entry(buf):
    e = {'dtype': 0, 'shape': (), 'shard_id': 0, 'offset': 0, 'size': 0, 'crc32c': 0, 'slices': 0}
    for num, val in _decode_fields(buf):
        if num == 1:
            e['dtype'] = val
        elif num == 2:
            dims = []
            for n2, v2 in _decode_fields(val):
                if n2 == 2:
                    size = 0
                    for n3, v3 in _decode_fields(v2):
                        if n3 == 1:
                            size = v3
                    dims.append(size)
                elif n2 == 3 and v2:
                    raise ValueError('tensor bundle: unknown-rank shape')
            e['shape'] = tuple(dims)
        elif num == 3:
            e['shard_id'] = val
        elif num == 4:
            e['offset'] = val
        elif num == 5:
            e['size'] = val
        elif num == 6:
            e['crc32c'] = val
        elif num == 7:
            e['slices'] += 1
    return e


# ---- LevelDB table ----
class _BlockBuilder:
    def __init__(self, restart_interval):
        self.interval = restart_interval
        self.buf = bytearray()
        self.restarts = [0]
        self.count = 0
        self.last = b''

    def add(self, key, value):
        shared = 0
        if self.count < self.interval:
            n = min(len(key), len(self.last))
            while shared < n and key[shared] == self.last[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.count = 0
        self.buf += _varint(shared) + _varint(len(key) - shared) + _varint(len(value)) + key[shared:] + value
        self.last = key
        self.count += 1

    def size(self):
        return len(self.buf) + 4 * len(self.restarts) + 4

    def empty(self):
        return not self.buf

    def finish(self):
        return bytes(self.buf) + b''.join(struct.pack('<I', r) for r in self.restarts) + \
            struct.pack('<I', len(self.restarts))


def _shortest_separator(a, b):
    """BytewiseComparator::FindShortestSeparator: a short key k with a <= k < b."""
    n = min(len(a), len(b))
    i = 0
    while i < n and a[i] == b[i]:
        i += 1
    if i < n and a[i] < 0xFF and a[i] + 1 < b[i]:
        return a[:i] + bytes([a[i] + 1])
    return a


def _short_successor(a):
    for i, c in enumerate(a):
        if c != 0xFF:
            return a[:i] + bytes([c + 1])
    return a


def write_table(f, items, block_size=BLOCK_SIZE, restart_interval=RESTART_INTERVAL):
    """items: (key bytes, value bytes) in strictly increasing key order."""
    offset = 0
    index = _BlockBuilder(1)
    pending = None                              # (last key of the finished block, handle)

    def emit(contents):
        nonlocal offset
        trailer = b'\x00'
        f.write(contents + trailer + struct.pack('<I', _masked_crc(contents, trailer)))
        handle = _varint(offset) + _varint(len(contents))
        offset += len(contents) + 5
        return handle

    block = _BlockBuilder(restart_interval)
    last_key = None
    for key, value in items:
        if last_key is not None and key <= last_key:
            raise ValueError('table keys must be strictly increasing')
        if pending is not None:
            index.add(_shortest_separator(pending[0], key), pending[1])
            pending = None
        block.add(key, value)
        last_key = key
        if block.size() >= block_size:
            pending = (last_key, emit(block.finish()))
            block = _BlockBuilder(restart_interval)
    if not block.empty():
        pending = (last_key, emit(block.finish()))
    if pending is not None:
        index.add(_short_successor(pending[0]), pending[1])
    meta_handle = emit(_BlockBuilder(restart_interval).finish())
    index_handle = emit(index.finish())
    footer = meta_handle + index_handle
    footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', MAGIC)
    f.write(footer)


def _read_block(buf, offset, size, verify=True):
    contents = buf[offset:offset + size]
    ctype = buf[offset + size]
    if verify:
        stored = struct.unpack_from('<I', buf, offset + size + 1)[0]
        if stored != _masked_crc(bytes(contents), bytes([ctype])):
            raise ValueError(f'tensor bundle index: block checksum mismatch at offset {offset}')
    if ctype != 0:
        raise ValueError(f'tensor bundle index: compressed block (type {ctype}) not supported')
    n_restarts = struct.unpack_from('<I', contents, len(contents) - 4)[0]
    end = len(contents) - 4 - 4 * n_restarts
    pos, key, out = 0, b'', []
    while pos < end:
        shared, pos = _read_varint(contents, pos)
        unshared, pos = _read_varint(contents, pos)
        vlen, pos = _read_varint(contents, pos)
        key = key[:shared] + bytes(contents[pos:pos + unshared])
        pos += unshared
        out.append((key, bytes(contents[pos:pos + vlen])))
        pos += vlen
    return out


def read_table(path, verify=True):
    """All (key, value) pairs of a LevelDB-format table file, in key order."""
    with open(path, 'rb') as f:
        buf = f.read()
    if len(buf) < 48 or struct.unpack_from('<Q', buf, len(buf) - 8)[0] != MAGIC:
        raise ValueError(f'{path}: not a table file (bad magic)')
    footer = buf[len(buf) - 48:]
    _, pos = _read_varint(footer, 0)
    _, pos = _read_varint(footer, pos)                                  # metaindex handle: nothing in it
    ioff, pos = _read_varint(footer, pos)
    isize, pos = _read_varint(footer, pos)
    out = []
    for _, handle in _read_block(buf, ioff, isize, verify):
        boff, p2 = _read_varint(handle, 0)
        bsize, _ = _read_varint(handle, p2)
        out += _read_block(buf, boff, bsize, verify)
    return out


# ---- bundles ----
def data_path(prefix, shard=0, num_shards=1):
    return f'{prefix}.data-{shard:05d}-of-{num_shards:05d}'


def write_bundle(prefix, tensors):
    """tensors: name -> ndarray (float32 / int32 / int64).  Writes <prefix>.index and <prefix>.data-00000-of-00001."""
    names = sorted(tensors, key=lambda n: n.encode())
    items = [(b'', encode_header(1))]
    offset = 0
    tmp_data, tmp_index = data_path(prefix) + '.tmp', prefix + '.index.tmp'
    with open(tmp_data, 'wb') as f:
        for name in names:
            a = np.asarray(tensors[name], order='C')
            if a.dtype not in _DT_OF:
                a = a.astype({'f': np.float32, 'i': np.int64, 'u': np.int64, 'b': np.int32}[a.dtype.kind])
            a = a.astype(a.dtype.newbyteorder('<'), copy=False)
            f.write(a.tobytes())
            items.append((name.encode(), encode_entry(_DT_OF[a.dtype], a.shape, offset, a.nbytes, _masked_crc(a))))
            offset += a.nbytes
    with open(tmp_index, 'wb') as f:
        write_table(f, items)
    os.replace(tmp_data, data_path(prefix))
    os.replace(tmp_index, prefix + '.index')


def list_bundle(prefix, verify=True):
    """name -> entry dict (dtype, shape, shard_id, offset, size, crc32c, slices) and the header's shard count."""
    entries, num_shards = {}, 1
    for key, value in read_table(prefix + '.index', verify):
        if key == b'':
            for num, val in _decode_fields(value):
                if num == 1:
                    num_shards = val
                elif num == 2 and val != 0:
                    raise ValueError('tensor bundle: big-endian bundles are not supported')
        else:
            entries[key.decode('utf-8', 'surrogateescape')] = decode_entry(value)
    return entries, num_shards


def read_bundle(prefix, names=None, verify=True):
    """name -> ndarray for every (or the named) whole tensor of the bundle.  Partitioned variables (entries that
    carry slices, written under a variable partitioner such as the reference's dcnf, src/models.py:85-86) raise."""
    entries, num_shards = list_bundle(prefix, verify)
    out = {}
    maps = {}
    for name, e in entries.items():
        if names is not None and name not in names:
            continue
        if e['slices']:
            raise ValueError(f'tensor bundle: {name} is a partitioned variable (slices are not supported)')
        if e['dtype'] not in _DTYPES:
            if names is None:
                continue                                      # strings etc.: not something this build stores
            raise ValueError(f'tensor bundle: {name} has unsupported dtype {e["dtype"]}')
        shard = e['shard_id']
        if shard not in maps:
            maps[shard] = np.memmap(data_path(prefix, shard, num_shards), dtype=np.uint8, mode='r')
        raw = maps[shard][e['offset']:e['offset'] + e['size']]
        dt = _DTYPES[e['dtype']]
        if e['size'] != int(np.prod(e['shape'], dtype=np.int64)) * dt.itemsize:
            raise ValueError(f'tensor bundle: {name}: size {e["size"]} does not match shape {e["shape"]}')
        a = np.array(raw).view(dt).reshape(e['shape'])
        if verify and _masked_crc(a) != e['crc32c']:
            raise ValueError(f'tensor bundle: {name}: data checksum mismatch')
        out[name] = a
    return out


def is_bundle(prefix):
    return os.path.exists(prefix + '.index')
