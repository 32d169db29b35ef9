"""Host side of the dataset plugin (CPU): CRC32C, TFRecord framing and the Example wire format in liba3d.so against
published check values, the pure-Python oracle, and google.protobuf with a dynamically built Example descriptor."""
import os

import numpy as np
import pytest

from ann3depth_amd import _lib, data, tfrecord
from oracle import tfrecord as OT


def test_crc32c_known_answers():
    # RFC 3720 B.4 check values (little-endian CRC32C, Castagnoli polynomial)
    vectors = [(b'', 0x00000000), (b'123456789', 0xE3069283), (bytes(32), 0x8A9136AA), (b'\xff' * 32, 0x62A8AB43),
               (bytes(range(32)), 0x46DD794E), (bytes(range(31, -1, -1)), 0x113FDB5C)]
    for buf, want in vectors:
        assert tfrecord.crc32c(buf) == want
        assert OT.crc32c(buf) == want
    rng = np.random.default_rng(0)
    # (from 3 x 8 KiB on the hardware path runs three interleaved chains joined by the CRC's zero-shift operator)
    for n in [1, 7, 8, 9, 63, 64, 65, 1000, 4097, 8192, 24575, 24576, 24577, 24583, 49157, 100003]:
        b = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert tfrecord.crc32c(b) == OT.crc32c(b)
        assert tfrecord.crc32c(b[1:]) == OT.crc32c(b[1:])        # unaligned start
        assert tfrecord.masked_crc32c(b) == OT.masked_crc(b)


def _example_message_class():
    """tf.train.Example built from scratch with descriptor_pb2 (tensorflow/core/example/{example,feature}.proto)."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto(name='a3d_example.proto', package='tensorflow', syntax='proto3')
    T = descriptor_pb2.FieldDescriptorProto

    def msg(name):
        m = fd.message_type.add()
        m.name = name
        return m
    m = msg('BytesList'); m.field.add(name='value', number=1, type=T.TYPE_BYTES, label=T.LABEL_REPEATED)
    m = msg('FloatList'); m.field.add(name='value', number=1, type=T.TYPE_FLOAT, label=T.LABEL_REPEATED)
    m = msg('Int64List'); m.field.add(name='value', number=1, type=T.TYPE_INT64, label=T.LABEL_REPEATED)
    m = msg('Feature')
    m.oneof_decl.add(name='kind')
    for i, (n, t) in enumerate([('bytes_list', 'BytesList'), ('float_list', 'FloatList'), ('int64_list', 'Int64List')]):
        m.field.add(name=n, number=i + 1, type=T.TYPE_MESSAGE, type_name='.tensorflow.' + t, label=T.LABEL_OPTIONAL,
                    oneof_index=0)
    m = msg('Features')
    e = m.nested_type.add(name='FeatureEntry')
    e.options.map_entry = True
    e.field.add(name='key', number=1, type=T.TYPE_STRING, label=T.LABEL_OPTIONAL)
    e.field.add(name='value', number=2, type=T.TYPE_MESSAGE, type_name='.tensorflow.Feature', label=T.LABEL_OPTIONAL)
    m.field.add(name='feature', number=1, type=T.TYPE_MESSAGE, type_name='.tensorflow.Features.FeatureEntry',
                label=T.LABEL_REPEATED)
    m = msg('Example')
    m.field.add(name='features', number=1, type=T.TYPE_MESSAGE, type_name='.tensorflow.Features', label=T.LABEL_OPTIONAL)
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName('tensorflow.Example'))


def _stored(rng, h, w, c):
    return (rng.integers(0, 256, (h, w, c)).astype(np.float32) / np.float32(255.) - np.float32(.5))


def test_writer_matches_oracle_and_protobuf(tmp_path):
    rng = np.random.default_rng(1)
    img, dep = _stored(rng, 6, 8, 3), _stored(rng, 3, 4, 1)
    path = tmp_path / 'x.tfrecords'
    with tfrecord.TFRecordWriter(str(path)) as w:
        w.write_example(img, dep)
        w.write_example(img[::-1].copy(), dep[..., 0])          # 2-D depth gets the trailing axis (converter :39-40)
    raw = path.read_bytes()
    assert raw[:len(OT.frame(OT.encode_example(img, dep)))] == OT.frame(OT.encode_example(img, dep))   # byte-identical
    payloads = list(OT.unframe(raw))
    assert len(payloads) == 2
    Example = _example_message_class()
    ex = Example.FromString(payloads[0])
    f = ex.features.feature
    assert f['image_height'].int64_list.value == [6] and f['image_width'].int64_list.value == [8]
    assert f['image_channels'].int64_list.value == [3] and f['depth_channels'].int64_list.value == [1]
    assert f['depth_height'].int64_list.value == [3] and f['depth_width'].int64_list.value == [4]
    assert f['image'].bytes_list.value[0] == img.tobytes() and f['depth'].bytes_list.value[0] == dep.tobytes()


def test_reader_parses_protobuf_serialised_examples(tmp_path):
    """Records produced by the real protobuf library (any map order, unpacked or packed int64) parse identically."""
    rng = np.random.default_rng(2)
    img, dep = _stored(rng, 5, 7, 3), _stored(rng, 5, 7, 1)
    Example = _example_message_class()
    ex = Example()
    for k, v in [('depth_width', 7), ('image_height', 5), ('image_width', 7), ('image_channels', 3),
                 ('depth_height', 5), ('depth_channels', 1)]:
        ex.features.feature[k].int64_list.value.append(v)
    ex.features.feature['image'].bytes_list.value.append(img.tobytes())
    ex.features.feature['depth'].bytes_list.value.append(dep.tobytes())
    path = tmp_path / 'p.tfrecords'
    path.write_bytes(OT.frame(ex.SerializeToString()) * 3)
    rf = tfrecord.RecordFile(str(path))
    recs = list(rf)
    assert len(recs) == 3
    for off, ln in recs:
        i2, d2 = rf.parse(off, ln)
        np.testing.assert_array_equal(i2, img + np.float32(.5))            # src/data.py:84-85
        np.testing.assert_array_equal(d2, dep + np.float32(.5))
        oi, od = OT.convert_img_depth(rf.mm[off:off + ln])
        np.testing.assert_array_equal(i2, oi)
        np.testing.assert_array_equal(d2, od)
    rf.close()


def test_corruption_is_detected(tmp_path):
    rng = np.random.default_rng(3)
    good = OT.frame(OT.encode_example(_stored(rng, 4, 4, 3), _stored(rng, 4, 4, 1)))
    for pos in (3, 9, 40, len(good) - 2):                      # length, length-crc, payload, payload-crc
        bad = bytearray(good)
        bad[pos] ^= 0x10
        p = tmp_path / f'bad{pos}.tfrecords'
        p.write_bytes(bytes(bad))
        with pytest.raises(_lib.A3dError):
            list(tfrecord.RecordFile(str(p)))
    p = tmp_path / 'trunc.tfrecords'
    p.write_bytes(good[:-5])
    with pytest.raises(_lib.A3dError):
        list(tfrecord.RecordFile(str(p)))
    p = tmp_path / 'missing.tfrecords'                          # a valid frame whose Example lacks features
    p.write_bytes(OT.frame(b''))
    rf = tfrecord.RecordFile(str(p))
    (off, ln), = list(rf)
    with pytest.raises(_lib.A3dError, match='missing'):
        rf.parse(off, ln)
    p = tmp_path / 'empty.tfrecords'
    p.write_bytes(b'')
    assert list(tfrecord.RecordFile(str(p))) == []


def _write_dataset(root, name, n, h, w, dh, dw, seed=0):
    rng = np.random.default_rng(seed)
    d = os.path.join(root, name)
    os.makedirs(d, exist_ok=True)
    samples = []
    with tfrecord.TFRecordWriter(os.path.join(d, 'train.tfrecords')) as wr:
        for i in range(n):
            img, dep = _stored(rng, h, w, 3), _stored(rng, dh, dw, 1)
            img[0, 0, 0] = np.float32(i) / 255 - np.float32(.5)        # tag
            wr.write_example(img, dep)
            samples.append((img, dep))
    return samples


def test_inputs_pipeline_semantics(tmp_path):
    """data.inputs (src/data.py:28-55): file location, default pipeline for unknown names, shuffle_batch with
    capacity 20B / min_after_dequeue 5B, every record seen once per epoch, '+0.5' applied."""
    B, n = 4, 64
    samples = _write_dataset(str(tmp_path), 'nyu', n, 6, 8, 3, 4)
    inp, tgt = data.inputs(str(tmp_path), 'nyu', B, epochs=1, seed=7, num_threads=2)
    assert inp.pipeline is tgt.pipeline and (inp.index, tgt.index) == (0, 1)
    sb = inp.pipeline
    assert (sb.capacity, sb.min_after, len(sb.threads)) == (20 * B, 5 * B, 2)       # src/data.py:51-55
    assert sb.shapes() == ((6, 8, 3), (3, 4, 1))
    seen = []
    while True:
        try:
            imgs, deps = sb.next_batch()
        except data.OutOfRangeError:
            break
        assert imgs.shape == (B, 6, 8, 3) and deps.shape == (B, 3, 4, 1) and imgs.dtype == np.float32
        for b in range(B):
            tag = int(round(float(imgs[b, 0, 0, 0]) * 255))
            np.testing.assert_array_equal(imgs[b], samples[tag][0] + np.float32(.5))
            np.testing.assert_array_equal(deps[b], samples[tag][1] + np.float32(.5))
            seen.append(tag)
    assert sorted(seen) == list(range(n))                 # one epoch: each record exactly once
    assert seen != sorted(seen)                           # ... in shuffled order
    # unknown dataset names fall back to the default pipeline; missing files fail loudly
    assert data._get_pipeline('whatever') == data._get_pipeline('nyu')
    with pytest.raises(FileNotFoundError):
        data.inputs(str(tmp_path), 'make3d1', B)
    # test split forces epochs = 1 (src/data.py:30)
    os.rename(tmp_path / 'nyu' / 'train.tfrecords', tmp_path / 'nyu' / 'test.tfrecords')
    inp, _ = data.inputs(str(tmp_path), 'nyu', B, 'test', epochs=None)
    k = 0
    with pytest.raises(data.OutOfRangeError):
        while True:
            inp.pipeline.next_batch()
            k += 1
    assert k == n // B


def test_inputs_sharding_and_endless_epochs(tmp_path):
    B, n = 2, 12
    _write_dataset(str(tmp_path), 'nyu', n, 4, 4, 2, 2)
    tags = []
    for rank in range(2):
        inp, _ = data.inputs(str(tmp_path), 'nyu', B, epochs=1, rank=rank, world=2, seed=rank)
        got = []
        try:
            while True:
                imgs, _ = inp.pipeline.next_batch()
                got += [int(round(float(v) * 255)) for v in imgs[:, 0, 0, 0]]
        except data.OutOfRangeError:
            pass
        tags.append(got)
    assert sorted(tags[0]) == [0, 2, 4, 6, 8, 10] and sorted(tags[1]) == [1, 3, 5, 7, 9, 11]
    inp, _ = data.inputs(str(tmp_path), 'nyu', B, epochs=None, seed=1)       # epochs=None cycles forever
    for _ in range(3 * n // B):
        inp.pipeline.next_batch()
    inp.pipeline.close()


def test_pipeline_surfaces_corrupt_and_mismatched_records(tmp_path):
    """The reader threads verify each payload CRC inside the fused decode; a flipped bit or a record of another size
    must stop the pipeline with an error, never be batched."""
    B = 2
    _write_dataset(str(tmp_path), 'nyu', 16, 4, 4, 2, 2)
    path = tmp_path / 'nyu' / 'train.tfrecords'
    raw = bytearray(path.read_bytes())
    raw[len(raw) // 2] ^= 0x01                                   # somewhere inside a payload
    path.write_bytes(bytes(raw))
    inp, _ = data.inputs(str(tmp_path), 'nyu', B, epochs=1, seed=0, num_threads=2)
    with pytest.raises(_lib.A3dError, match='corrupt'):
        while True:
            inp.pipeline.next_batch()
    # records of two different sizes in one shard
    d2 = tmp_path / 'mixed'
    rng = np.random.default_rng(0)
    with tfrecord.TFRecordWriter(str(d2 / 'nyu' / 'train.tfrecords')) as w:
        for i in range(12):
            hw = (4, 4) if i < 11 else (4, 6)
            w.write_example(_stored(rng, hw[0], hw[1], 3), _stored(rng, 2, 2, 1))
    inp, _ = data.inputs(str(d2), 'nyu', B, epochs=1, seed=0, num_threads=1)
    with pytest.raises((_lib.A3dError, ValueError)):
        while True:
            inp.pipeline.next_batch()


def test_record_decode_u8_is_exact_or_declines(tmp_path):
    """a3d_record_decode_u8 (data.py: converter-written records travel as uint8 pixel values): a feature is delivered as
    uint8 only when EVERY float is bit for bit fl(fl(k/255) - 0.5); expand_u8 then gives exactly what decode_into's
    `+ 0.5` gives.  One float off by an ulp, a NaN, a value outside [-.5, .5]: the feature comes as float32 instead."""
    rng = np.random.default_rng(42)
    k_img = rng.integers(0, 256, (37, 41, 3), dtype=np.uint8)
    k_img.reshape(-1)[:256] = np.arange(256, dtype=np.uint8)                        # every pixel value once
    k_dep = rng.integers(0, 256, (9, 11, 1), dtype=np.uint8)
    conv = lambda k: k.astype(np.float32) / np.float32(255.) - np.float32(.5)         # tools/data_tf_converter.py:36-37
    good_img, good_dep = conv(k_img), conv(k_dep)
    off = good_img.copy()
    off[5, 6, 2] = np.nextafter(off[5, 6, 2], np.float32(-1))
    nan = good_dep.copy()
    nan[1, 1, 0] = np.nan
    big = good_dep.copy()
    big[8, 10, 0] = np.float32(0.75)
    cases = [(good_img, good_dep, True, True), (off, good_dep, False, True), (good_img, nan, True, False),
             (good_img, big, True, False), (off, (rng.random((9, 11, 1)) - .5).astype(np.float32), False, False)]
    path = str(tmp_path / 'u8.tfrecords')
    with tfrecord.TFRecordWriter(path) as w:
        for img, dep, _, _ in cases:
            w.write_example(img, dep)
    rf = tfrecord.RecordFile(path)
    frames = list(rf.frames())
    for (o, ln), (img, dep, img_u8, dep_u8) in zip(frames, cases):
        ref_i, ref_d = np.empty(img.shape, np.float32), np.empty(dep.shape, np.float32)
        rf.decode_into(o, ln, ref_i, ref_d)
        iu, iff = np.full(img.shape, 77, np.uint8), np.full(img.shape, -9, np.float32)
        du, dff = np.full(dep.shape, 77, np.uint8), np.full(dep.shape, -9, np.float32)
        assert rf.decode_into_u8(o, ln, iu, iff, du, dff) == (img_u8, dep_u8)
        got_i = data.expand_u8(iu) if img_u8 else iff
        got_d = data.expand_u8(du) if dep_u8 else dff
        np.testing.assert_array_equal(got_i.view(np.uint32), ref_i.view(np.uint32))   # bits, so that a NaN compares too
        np.testing.assert_array_equal(got_d.view(np.uint32), ref_d.view(np.uint32))
        if img_u8:
            np.testing.assert_array_equal(iu, k_img)
            assert (iff == -9).all()                              # the float32 destination of a uint8 feature is untouched
    # a flipped payload bit is still caught on this path
    raw = bytearray(open(path, 'rb').read())
    raw[frames[0][0] + 40] ^= 0x10
    bad = str(tmp_path / 'bad.tfrecords')
    open(bad, 'wb').write(bytes(raw))
    rfb = tfrecord.RecordFile(bad)
    o, ln = frames[0]
    with pytest.raises(_lib.A3dError, match='corrupt'):
        rfb.decode_into_u8(o, ln, np.empty(good_img.shape, np.uint8), np.empty(good_img.shape, np.float32),
                           np.empty(good_dep.shape, np.uint8), np.empty(good_dep.shape, np.float32))


def test_shuffle_batch_with_uint8_staging_hands_out_the_same_batches(tmp_path):
    """The staging pool with uint8 twins (what a GPU consumer allocates) against the plain float32 pool: every record comes
    out with the same float32 values, bit for bit — on a shard that mixes converter-written and arbitrary records."""
    rng = np.random.default_rng(3)
    os.makedirs(tmp_path / 'nyu')
    with tfrecord.TFRecordWriter(str(tmp_path / 'nyu' / 'train.tfrecords')) as w:
        for i in range(40):
            img = rng.integers(0, 256, (6, 8, 3)).astype(np.float32) / np.float32(255) - np.float32(.5)
            dep = rng.integers(0, 256, (3, 4, 1)).astype(np.float32) / np.float32(255) - np.float32(.5)
            if i % 5 == 0:
                img = (rng.random((6, 8, 3)) - .5).astype(np.float32)
            dep[0, 0, 0] = np.float32(i) / np.float32(255) - np.float32(.5)             # tag
            w.write_example(img, dep)
    batches = []
    for u8 in (False, True):
        inp, _ = data.inputs(str(tmp_path), 'nyu', 4, epochs=1, seed=11, num_threads=1)
        sb = inp.pipeline
        sb.allocate(None, (lambda shape: np.empty(shape, np.uint8)) if u8 else None)
        assert (sb.images_u8 is not None) == u8
        got = []
        try:
            while True:
                got.append(tuple(a.copy() for a in sb.next_batch()))
        except data.OutOfRangeError:
            pass
        batches.append(got)
    assert len(batches[0]) == len(batches[1]) == 10
    # (which records meet in a batch depends on how far the reader thread had got when the batch was drawn: compare by tag)
    by_tag = []
    for got in batches:
        recs = {}
        for imgs, deps in got:
            for b in range(4):
                recs[int(np.rint(deps[b, 0, 0, 0] * 255))] = (imgs[b], deps[b])
        assert sorted(recs) == list(range(40))
        by_tag.append(recs)
    for t in range(40):
        np.testing.assert_array_equal(by_tag[0][t][0], by_tag[1][t][0])
        np.testing.assert_array_equal(by_tag[0][t][1], by_tag[1][t][1])
