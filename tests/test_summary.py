"""TensorBoard event files (CPU): framing, Event / Summary encoding against google.protobuf with descriptors built
here from tensorflow/core/util/event.proto + framework/summary.proto, PNG payloads against Pillow."""
import io

import numpy as np

from ann3depth_amd import summary
from oracle import tfrecord as OT


def _classes():
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    T = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto(name='a3d_event.proto', package='tensorflow', syntax='proto3')
    s = fd.message_type.add(name='Summary')
    img = s.nested_type.add(name='Image')
    for i, (n, t) in enumerate([('height', T.TYPE_INT32), ('width', T.TYPE_INT32), ('colorspace', T.TYPE_INT32),
                                ('encoded_image_string', T.TYPE_BYTES)]):
        img.field.add(name=n, number=i + 1, type=t, label=T.LABEL_OPTIONAL)
    v = s.nested_type.add(name='Value')
    v.field.add(name='tag', number=1, type=T.TYPE_STRING, label=T.LABEL_OPTIONAL)
    v.field.add(name='simple_value', number=2, type=T.TYPE_FLOAT, label=T.LABEL_OPTIONAL)
    v.field.add(name='image', number=4, type=T.TYPE_MESSAGE, type_name='.tensorflow.Summary.Image', label=T.LABEL_OPTIONAL)
    s.field.add(name='value', number=1, type=T.TYPE_MESSAGE, type_name='.tensorflow.Summary.Value', label=T.LABEL_REPEATED)
    e = fd.message_type.add(name='Event')
    e.field.add(name='wall_time', number=1, type=T.TYPE_DOUBLE, label=T.LABEL_OPTIONAL)
    e.field.add(name='step', number=2, type=T.TYPE_INT64, label=T.LABEL_OPTIONAL)
    e.field.add(name='file_version', number=3, type=T.TYPE_STRING, label=T.LABEL_OPTIONAL)
    e.field.add(name='summary', number=5, type=T.TYPE_MESSAGE, type_name='.tensorflow.Summary', label=T.LABEL_OPTIONAL)
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName('tensorflow.Event'))


def test_event_file_roundtrip(tmp_path):
    from PIL import Image
    w = summary.EventFileWriter(str(tmp_path))
    w.add_scalars(100, {'loss/coarse_loss': 23122.5, 'optimizers/Phase': 1, 'global_step/sec': 240.25})
    rng = np.random.default_rng(0)
    pos = rng.random((4, 5, 7, 3)).astype(np.float32)
    neg = rng.standard_normal((2, 6, 4, 1)).astype(np.float32)
    w.add_images(100, 'summaries/Input', pos)
    w.add_images(200, 'summaries/Coarse', neg)
    w.close()
    Event = _classes()
    events = [Event.FromString(p) for p in OT.unframe(open(w.path, 'rb').read())]       # CRCs verified by the oracle
    assert events[0].file_version == 'brain.Event:2' and events[0].wall_time > 1e9
    sc = {v.tag: v.simple_value for v in events[1].summary.value}
    assert events[1].step == 100 and sc['optimizers/Phase'] == 1 and abs(sc['loss/coarse_loss'] - 23122.5) < 1e-2
    imgs = events[2].summary.value
    assert [v.tag for v in imgs] == ['summaries/Input/image/0', 'summaries/Input/image/1', 'summaries/Input/image/2']
    for i, v in enumerate(imgs):
        assert (v.image.height, v.image.width, v.image.colorspace) == (5, 7, 3)
        got = np.asarray(Image.open(io.BytesIO(v.image.encoded_image_string)))
        np.testing.assert_array_equal(got, summary.to_uint8(pos[i]))
        assert got.max() == 255                                     # positive images: max scaled to 255
    g = events[3].summary.value[0]
    assert events[3].step == 200 and g.tag == 'summaries/Coarse/image/0' and g.image.colorspace == 1
    got = np.asarray(Image.open(io.BytesIO(g.image.encoded_image_string)))
    np.testing.assert_array_equal(got, summary.to_uint8(neg[0])[..., 0])
    zero = summary.to_uint8(np.array([[[-1.0], [0.0], [2.0]]], np.float32))   # signed images: 0 -> 127
    assert zero[0, 1, 0] == 127 and zero.min() >= 0 and zero.max() <= 255
