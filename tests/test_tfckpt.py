"""TensorFlow V2 checkpoint bundles without TensorFlow (ann3depth_amd/tfckpt.py): table format, bundle round trip,
corruption detection.  No TensorFlow-written file exists to pin these against (see the module docstring)."""
import io
import os
import struct

import numpy as np
import pytest

from ann3depth_amd import tfckpt


def test_table_layout_constants(tmp_path):
    buf = io.BytesIO()
    tfckpt.write_table(buf, [(b'', b'header'), (b'a/kernel', b'1'), (b'a/kernel/Adam', b'22')])
    raw = buf.getvalue()
    assert struct.unpack('<Q', raw[-8:])[0] == 0xdb4775248b80fb57                   # LevelDB table magic
    assert len(raw[-48:]) == 48
    # first data block: first entry has nothing shared, the third shares 'a/kernel' with the second
    assert raw[:3] == bytes([0, 0, 6]) and raw[3:9] == b'header'
    assert raw[9:12] == bytes([0, 8, 1]) and raw[12:20] == b'a/kernel' and raw[20:21] == b'1'
    assert raw[21:24] == bytes([8, 5, 2]) and raw[24:29] == b'/Adam'
    # block trailer: type 0 + masked crc32c(contents + type)
    n_block = 31 + 4 + 4
    assert raw[n_block] == 0
    assert struct.unpack_from('<I', raw, n_block + 1)[0] == tfckpt._masked_crc(raw[:n_block], b'\x00')
    p = tmp_path / 't.index'
    p.write_bytes(raw)
    assert tfckpt.read_table(str(p)) == [(b'', b'header'), (b'a/kernel', b'1'), (b'a/kernel/Adam', b'22')]


@pytest.mark.parametrize('block_size,restart', [(64, 2), (200, 16), (1 << 18, 16), (1, 1)])
def test_reader_handles_blocks_restarts_and_prefixes(tmp_path, block_size, restart):
    rng = np.random.default_rng(0)
    keys = sorted({('scope%d/layer_%d/%s' % (i % 3, i % 7, s)).encode() for i in range(60)
                   for s in ('kernel', 'bias', 'kernel/Adam', 'kernel/Adam_1')})
    items = [(b'', b'h')] + [(k, rng.bytes(int(rng.integers(0, 40)))) for k in keys]
    p = tmp_path / 't.index'
    with open(p, 'wb') as f:
        tfckpt.write_table(f, items, block_size=block_size, restart_interval=restart)
    assert tfckpt.read_table(str(p)) == items


def test_separators():
    assert tfckpt._shortest_separator(b'abc1', b'abd') == b'abc1'       # 'c'+1 == 'd': no shorter key in between
    assert tfckpt._shortest_separator(b'abc', b'abz') == b'abd'
    assert tfckpt._shortest_separator(b'abc', b'abcd') == b'abc'
    assert tfckpt._short_successor(b'abc') == b'b' and tfckpt._short_successor(b'\xff\xffa') == b'\xff\xffb'


def test_entry_protobuf_against_google_protobuf():
    """BundleEntryProto built from a descriptor declared here must parse what encode_entry writes, and vice versa."""
    pb = pytest.importorskip('google.protobuf')
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto(name='tb.proto', package='tb', syntax='proto3')
    dim = fd.message_type.add(name='Dim')
    dim.field.add(name='size', number=1, type=3, label=1)
    shp = fd.message_type.add(name='Shape')
    shp.field.add(name='dim', number=2, type=11, label=3, type_name='.tb.Dim')
    ent = fd.message_type.add(name='Entry')
    for name, num, typ in (('dtype', 1, 5), ('shard_id', 3, 5), ('offset', 4, 3), ('size', 5, 3), ('crc32c', 6, 7)):
        ent.field.add(name=name, number=num, type=typ, label=1)
    ent.field.add(name='shape', number=2, type=11, label=1, type_name='.tb.Shape')
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    Entry = message_factory.GetMessageClass(pool.FindMessageTypeByName('tb.Entry'))
    m = Entry()
    m.ParseFromString(tfckpt.encode_entry(1, (11, 11, 3, 96), 4096, 139392, 0xdeadbeef))
    assert (m.dtype, m.offset, m.size, m.crc32c) == (1, 4096, 139392, 0xdeadbeef)
    assert [d.size for d in m.shape.dim] == [11, 11, 3, 96]
    m2 = Entry(dtype=9, size=8, crc32c=7)
    m2.shape.SetInParent()
    e = tfckpt.decode_entry(m2.SerializeToString())
    assert (e['dtype'], e['shape'], e['offset'], e['size'], e['crc32c']) == (9, (), 0, 8, 7)


def test_bundle_round_trip_and_corruption(tmp_path):
    rng = np.random.default_rng(1)
    tensors = {
        'coarse/conv/conv2d_0/kernel': rng.standard_normal((11, 11, 3, 96)).astype(np.float32),
        'coarse/conv/conv2d_0/kernel/CoarseConv': rng.standard_normal((11, 11, 3, 96)).astype(np.float32),
        'coarse/conv/conv2d_0/bias': np.zeros(96, np.float32),
        'coarse/dense/dense_1/kernel': rng.standard_normal((64, 4070)).astype(np.float32),
        'CoarseConv/beta1_power': np.float32(0.81),
        'global_step': np.int64(62500),
    }
    prefix = str(tmp_path / 'model.ckpt-62500')
    tfckpt.write_bundle(prefix, tensors)
    assert sorted(os.listdir(tmp_path)) == ['model.ckpt-62500.data-00000-of-00001', 'model.ckpt-62500.index']
    assert tfckpt.is_bundle(prefix)
    entries, shards = tfckpt.list_bundle(prefix)
    assert shards == 1 and entries['global_step']['dtype'] == tfckpt.DT_INT64 and entries['global_step']['shape'] == ()
    assert entries['coarse/conv/conv2d_0/kernel']['shape'] == (11, 11, 3, 96)
    # tensors lie in the data file in key order, back to back
    order = sorted(tensors, key=lambda n: n.encode())
    off = 0
    for n in order:
        assert entries[n]['offset'] == off
        off += entries[n]['size']
    assert off == os.path.getsize(tfckpt.data_path(prefix))
    back = tfckpt.read_bundle(prefix)
    assert set(back) == set(tensors)
    for n, a in tensors.items():
        np.testing.assert_array_equal(back[n], a)
        assert back[n].dtype == np.asarray(a).dtype
    only = tfckpt.read_bundle(prefix, names={'global_step'})
    assert list(only) == ['global_step'] and int(only['global_step']) == 62500
    # flip one data byte: the tensor's checksum catches it
    with open(tfckpt.data_path(prefix), 'r+b') as f:
        f.seek(entries['coarse/dense/dense_1/kernel']['offset'] + 5)
        b = f.read(1)
        f.seek(-1, 1)
        f.write(bytes([b[0] ^ 1]))
    with pytest.raises(ValueError, match='data checksum'):
        tfckpt.read_bundle(prefix)
    # flip one index byte: the block checksum catches it
    raw = bytearray(open(prefix + '.index', 'rb').read())
    raw[10] ^= 1
    open(prefix + '.index', 'wb').write(bytes(raw))
    with pytest.raises(ValueError, match='block checksum'):
        tfckpt.list_bundle(prefix)
    with pytest.raises(ValueError, match='bad magic'):
        open(prefix + '.index', 'wb').write(bytes(raw[:-1]))
        tfckpt.list_bundle(prefix)


def test_partitioned_variables_are_refused(tmp_path):
    prefix = str(tmp_path / 'm')
    entry = tfckpt.encode_entry(1, (4,), 0, 16, 0) + tfckpt._ld(7, b'')          # one TensorSliceProto
    open(tfckpt.data_path(prefix), 'wb').write(b'\0' * 16)
    with open(prefix + '.index', 'wb') as f:
        tfckpt.write_table(f, [(b'', tfckpt.encode_header()), (b'unary/w', entry)])
    with pytest.raises(ValueError, match='partitioned'):
        tfckpt.read_bundle(prefix)
