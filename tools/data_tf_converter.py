"""PNG pairs -> TFRecord shards without TensorFlow: the replacement of the reference's tools/data_tf_converter.py
(`make convert`).  Same interface: DATA_DIR in the environment, dataset names as arguments, `--del_raw` removes the
PNGs that were converted.

    <DATA_DIR>/<dataset>/{train,test}/<name>-image.png + <name>-depth.png  ->  <DATA_DIR>/<dataset>/{train,test}.tfrecords

Each record is the 8-feature Example of tools/data_tf_converter.py:41-51 with `png / 255 - 0.5` float32 payloads
(:36-37); pairs are taken in sorted order (the reference uses glob order, which is arbitrary).
"""
import glob
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ann3depth_amd import png, tfrecord  # noqa: E402


def convert(datadir, dataset, remove=False, log=print):
    counts = {}
    for split in ('test', 'train'):
        directory = os.path.join(datadir, dataset, split)
        depth_paths = sorted(glob.glob(os.path.join(directory, '*-depth.png')))
        with tfrecord.TFRecordWriter(os.path.join(datadir, dataset, split + '.tfrecords')) as writer:
            for depth_path in depth_paths:
                image_path = depth_path[:-9] + 'image.png'
                depth = png.imread(depth_path).astype(np.float32) / np.float32(255.) - np.float32(.5)
                img = png.imread(image_path).astype(np.float32) / np.float32(255.) - np.float32(.5)
                if img.ndim < 3:
                    raise ValueError(f'{image_path}: the record layout needs an image with a channel axis')
                if depth.ndim < 3:
                    depth = depth[..., None]
                writer.write_example(img, depth)
                if remove:
                    os.remove(depth_path)
                    os.remove(image_path)
        counts[split] = len(depth_paths)
        log(f'{dataset}/{split}.tfrecords: {len(depth_paths)} records')
    return counts


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    datasets = [d for d in argv if d != '--del_raw']
    if not datasets:
        print('Provide a dataset ( data_tf_converter.py DATASET [--del_raw] )')
        return 1
    for dataset in datasets:
        convert(os.environ['DATA_DIR'], dataset, remove='--del_raw' in argv)
    return 0


if __name__ == '__main__':
    sys.exit(main())
