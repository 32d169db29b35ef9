"""A minimal HDF5 reader — enough for MATLAB v7.3 files such as nyu_depth_v2_labeled.mat, which the reference opens with
h5py (tools/data_preprocessor.py:186-190: `mat['depths']`, `mat['images']`, `mat['rawRgbFilenames'][0]`, `mat[ref][:]`).

Restated from the HDF5 File Format Specification, version 2.0 (the format, not the library): superblock versions 0 and 1,
version-1 object headers with continuation blocks, old-style groups (symbol-table message -> version-1 B-tree of group
nodes + local heap), dataspace / datatype / data-layout (version 3: contiguous and chunked) / filter-pipeline messages,
version-1 chunk B-trees, the deflate / shuffle / fletcher32 filters, fixed-point, floating-point and object-reference
datatypes, little-endian.  That is what MATLAB's `save -v7.3` (HDF5 1.8, "earliest" format) writes.  Everything else
raises NotImplementedError with the name of the missing piece.

Unpinned: there is no HDF5 library in this image and no HDF5 file in the reference, so the reader is tested against a
file assembled byte by byte from the same specification (tests/test_hdf5.py), not against h5py's output.
"""
import mmap
import struct
import zlib

import numpy as np

SIGNATURE = b'\x89HDF\r\n\x1a\n'
UNDEF = 0xffffffffffffffff


class Reference(int):
    """An object reference: the file address of an object header (dereference with File[ref])."""


class File:
    def __init__(self, path):
        self._f = open(path, 'rb')
        self.buf = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ)
        start = 0
        while True:                                         # the superblock sits at 0, 512, 1024, 2048, ...
            if start + 8 > len(self.buf):
                raise ValueError(f'{path}: no HDF5 superblock')
            if self.buf[start:start + 8] == SIGNATURE:
                break
            start = 512 if start == 0 else start * 2
        b = self.buf
        version = b[start + 8]
        if version not in (0, 1):
            raise NotImplementedError(f'HDF5 superblock version {version}')
        self.so, self.sl = b[start + 13], b[start + 14]      # size of offsets / of lengths
        if self.so != 8 or self.sl != 8:
            raise NotImplementedError(f'HDF5 offsets of {self.so} bytes / lengths of {self.sl} bytes')
        pos = start + 24 + (4 if version == 1 else 0)
        # every address in the file is relative to the base address, which the library takes to be the superblock's own
        # address whatever the field says (a MATLAB file: 512 bytes of userblock in front)
        self.base = start
        root_entry = pos + 32                                # after base, free-space, end-of-file and driver addresses
        self.root = Group(self, self._u64(root_entry + 8))

    # ---- primitives
    def _u16(self, p):
        return struct.unpack_from('<H', self.buf, p)[0]

    def _u32(self, p):
        return struct.unpack_from('<I', self.buf, p)[0]

    def _u64(self, p):
        return struct.unpack_from('<Q', self.buf, p)[0]

    def at(self, addr):
        return addr + self.base

    # ---- h5py-like access
    def __getitem__(self, key):
        if isinstance(key, Reference):
            return open_object(self, int(key))
        node = self.root
        for part in [k for k in key.split('/') if k]:
            node = node[part]
        return node

    def keys(self):
        return self.root.keys()

    def close(self):
        self.buf.close()
        self._f.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def header_messages(f, addr):
    """(type, flags, payload offset, payload size) of every message of the version-1 object header at `addr`."""
    p = f.at(addr)
    if f.buf[p:p + 4] == b'OHDR':
        raise NotImplementedError('HDF5 version-2 object header')
    if f.buf[p] != 1:
        raise ValueError(f'object header version {f.buf[p]} at {addr}')
    total = f._u16(p + 2)
    blocks = [(p + 16, f._u32(p + 8))]                       # first block: right after the 12-byte prefix, 8-aligned
    out = []
    while blocks and len(out) < total:
        q, size = blocks.pop(0)
        end = q + size
        while q + 8 <= end and len(out) < total:
            mtype, msize, flags = f._u16(q), f._u16(q + 2), f.buf[q + 4]
            body = q + 8
            if mtype == 0x0010:                              # continuation: offset, length
                blocks.append((f.at(f._u64(body)), f._u64(body + 8)))
            out.append((mtype, flags, body, msize))
            q = body + msize
    return out


def open_object(f, addr):
    kinds = {m[0] for m in header_messages(f, addr)}
    if 0x0011 in kinds:
        return Group(f, addr)
    if 0x0008 in kinds:
        return Dataset(f, addr)
    raise NotImplementedError(f'object at {addr} is neither an old-style group nor a dataset')


class Group:
    def __init__(self, f, addr):
        self.f, self.addr = f, addr
        self._links = None

    def _load(self):
        if self._links is not None:
            return
        f = self.f
        self._links = {}
        for mtype, _, body, _ in header_messages(f, self.addr):
            if mtype == 0x0011:                              # symbol table: B-tree address, local heap address
                heap = f.at(f._u64(body + 8))
                if f.buf[heap:heap + 4] != b'HEAP':
                    raise ValueError('local heap signature')
                data = f.at(f._u64(heap + 24))
                self._walk(f.at(f._u64(body)), data)
                return
        raise NotImplementedError('group without a symbol-table message (new-style links)')

    def _walk(self, node, heap_data):
        f = self.f
        if f.buf[node:node + 4] == b'SNOD':
            n = f._u16(node + 6)
            for i in range(n):
                e = node + 8 + 40 * i
                name_at = heap_data + f._u64(e)
                end = f.buf.find(b'\0', name_at)
                self._links[f.buf[name_at:end].decode()] = f._u64(e + 8)
            return
        if f.buf[node:node + 4] != b'TREE' or f.buf[node + 4] != 0:
            raise ValueError('group B-tree node signature')
        used = f._u16(node + 6)
        p = node + 24 + 8                                    # past siblings and key 0; children alternate with keys
        for i in range(used):
            self._walk(f.at(f._u64(p + 16 * i)), heap_data)

    def keys(self):
        self._load()
        return list(self._links)

    def __contains__(self, name):
        self._load()
        return name in self._links

    def __getitem__(self, name):
        self._load()
        return open_object(self.f, self._links[name])


def _datatype(f, p):
    cls, version = f.buf[p] & 15, f.buf[p] >> 4
    bits0 = f.buf[p + 1]
    size = f._u32(p + 4)
    if version not in (1, 2, 3):
        raise NotImplementedError(f'datatype message version {version}')
    if cls in (0, 1) and bits0 & 1:
        raise NotImplementedError('big-endian datatype')
    if cls == 0:
        return np.dtype(('<i' if bits0 & 8 else '<u') + str(size))
    if cls == 1:
        if size not in (2, 4, 8):
            raise NotImplementedError(f'{size}-byte floating point')
        return np.dtype('<f' + str(size))
    if cls == 7:
        if bits0 & 15:
            raise NotImplementedError('dataset region references')
        return np.dtype('<u8'), True
    raise NotImplementedError(f'datatype class {cls}')


class Dataset:
    def __init__(self, f, addr):
        self.f, self.addr = f, addr
        self.is_reference = False
        self.filters = []
        self.layout = None
        for mtype, flags, body, size in header_messages(f, addr):
            if flags & 2 and mtype in (1, 3, 0x0b):
                raise NotImplementedError('shared header message')
            if mtype == 0x0001:
                version, rank, dflags = f.buf[body], f.buf[body + 1], f.buf[body + 2]
                if version not in (1, 2):
                    raise NotImplementedError(f'dataspace message version {version}')
                dims = body + (8 if version == 1 else 4)
                self.shape = tuple(f._u64(dims + 8 * i) for i in range(rank))
            elif mtype == 0x0003:
                dt = _datatype(f, body)
                if isinstance(dt, tuple):
                    self.dtype, self.is_reference = dt
                else:
                    self.dtype = dt
            elif mtype == 0x000b:
                self.filters = self._filters(body)
            elif mtype == 0x0008:
                version, cls = f.buf[body], f.buf[body + 1]
                if version != 3:
                    raise NotImplementedError(f'data layout message version {version}')
                if cls == 1:
                    self.layout = ('contiguous', f._u64(body + 2), f._u64(body + 10))
                elif cls == 2:
                    nd = f.buf[body + 2]
                    btree = f._u64(body + 3)
                    cdims = tuple(f._u32(body + 11 + 4 * i) for i in range(nd))
                    self.layout = ('chunked', btree, cdims[:-1])          # the last "dimension" is the element size
                elif cls == 0:
                    n = f._u16(body + 2)
                    self.layout = ('compact', body + 4, n)
                else:
                    raise NotImplementedError(f'data layout class {cls}')
        if self.layout is None or not hasattr(self, 'shape') or not hasattr(self, 'dtype'):
            raise ValueError(f'object at {addr} lacks a dataspace, datatype or layout message')
        self._chunks = None

    def _filters(self, body):
        f = self.f
        version, n = f.buf[body], f.buf[body + 1]
        if version not in (1, 2):
            raise NotImplementedError(f'filter pipeline message version {version}')
        p = body + (8 if version == 1 else 2)
        out = []
        for _ in range(n):
            fid = f._u16(p)
            if version == 1 or fid >= 256:
                name_len = f._u16(p + 2)
                p += 2
            else:
                name_len = 0
            ncd = f._u16(p + 4)
            p += 6
            p += (name_len + 7) // 8 * 8 if version == 1 else name_len
            p += 4 * ncd
            if version == 1 and ncd % 2:
                p += 4
            if fid not in (1, 2, 3):
                raise NotImplementedError(f'HDF5 filter {fid}')
            out.append(fid)
        return out

    # ---- chunk index
    def _walk_chunks(self, node, rank):
        f = self.f
        if f.buf[node:node + 4] != b'TREE' or f.buf[node + 4] != 1:
            raise ValueError('chunk B-tree node signature')
        level, used = f.buf[node + 5], f._u16(node + 6)
        key = 8 + 8 * (rank + 1)
        p = node + 24
        for i in range(used):
            k = p + i * (key + 8)
            child = f.at(f._u64(k + key))
            if level:
                self._walk_chunks(child, rank)
            else:
                offs = tuple(f._u64(k + 8 + 8 * d) for d in range(rank))
                self._chunks[offs] = (child, f._u32(k), f._u32(k + 4))

    def _chunk(self, offs, cdims):
        """The chunk that starts at element offsets `offs`, decoded; zeros where the file holds none."""
        if self._chunks is None:
            self._chunks = {}
            if self.layout[1] != UNDEF:
                self._walk_chunks(self.f.at(self.layout[1]), len(self.shape))
        hit = self._chunks.get(offs)
        if hit is None:
            return np.zeros(cdims, self.dtype)
        addr, nbytes, mask = hit
        raw = bytes(self.f.buf[addr:addr + nbytes])
        for i in reversed(range(len(self.filters))):         # the pipeline is undone last filter first
            if mask >> i & 1:
                continue
            fid = self.filters[i]
            if fid == 1:
                raw = zlib.decompress(raw)
            elif fid == 3:
                raw = raw[:-4]
            elif fid == 2:
                es = self.dtype.itemsize
                raw = np.frombuffer(raw, np.uint8).reshape(es, -1).T.tobytes()
        return np.frombuffer(raw, self.dtype, count=int(np.prod(cdims))).reshape(cdims)

    # ---- reads
    def _slab(self, first, count):
        """Elements [first, first + count) of the slowest axis (all of the others)."""
        kind = self.layout[0]
        row = int(np.prod(self.shape[1:], dtype=np.int64))
        if kind in ('contiguous', 'compact'):
            if kind == 'contiguous' and self.layout[1] == UNDEF:
                return np.zeros((count,) + self.shape[1:], self.dtype)
            start = (self.f.at(self.layout[1]) if kind == 'contiguous' else self.layout[1]) + first * row * self.dtype.itemsize
            a = np.frombuffer(self.f.buf, self.dtype, count=count * row, offset=start)
            return a.reshape((count,) + self.shape[1:]).copy()
        cdims = self.layout[2]
        out = np.empty((count,) + self.shape[1:], self.dtype)
        grid = [range(0, s, c) for s, c in zip(self.shape[1:], cdims[1:])]
        for c0 in range(first // cdims[0] * cdims[0], first + count, cdims[0]):
            lo, hi = max(c0, first), min(c0 + cdims[0], first + count, self.shape[0])
            for offs in np.ndindex(*[len(g) for g in grid]):
                o = tuple(g[i] for g, i in zip(grid, offs))
                chunk = self._chunk((c0,) + o, cdims)
                dst = tuple(slice(a, min(a + c, s)) for a, c, s in zip(o, cdims[1:], self.shape[1:]))
                src = tuple(slice(0, d.stop - d.start) for d in dst)
                out[(slice(lo - first, hi - first),) + dst] = chunk[(slice(lo - c0, hi - c0),) + src]
        return out

    def _wrap(self, a):
        if self.is_reference:
            flat = np.empty(a.size, object)
            for i, v in enumerate(a.reshape(-1)):
                flat[i] = Reference(int(v))
            return flat.reshape(a.shape)
        return a

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, key):
        if not self.shape:
            raise NotImplementedError('scalar datasets')
        if isinstance(key, (int, np.integer)):
            if key < 0:
                key += self.shape[0]
            return self._wrap(self._slab(int(key), 1)[0])
        if key == slice(None) or key is Ellipsis:
            return self._wrap(self._slab(0, self.shape[0]))
        raise NotImplementedError('only dataset[i] and dataset[:] are supported')

    def __iter__(self):
        for i in range(self.shape[0]):
            yield self[i]
