"""tools/data_preprocessor.py — the Make3D and MNIST processors (reference: tools/data_preprocessor.py:65-164), on tiny
data sets the test writes with independent libraries: JPEG files by Pillow, level-5 .mat files by scipy.io.savemat."""
import os
import sys

import numpy as np
import pytest

from ann3depth_amd import imresize, png

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import data_preprocessor as pre  # noqa: E402

Image = pytest.importorskip('PIL.Image')
sio = pytest.importorskip('scipy.io')

ENV = {'WIDTH': '16', 'HEIGHT': '12', 'DHEIGHT': '5'}


def jpeg(path, rng, shape):
    arr = rng.integers(0, 256, shape).astype(np.uint8)
    Image.fromarray(arr).save(path, quality=92)
    with Image.open(path) as im:                       # what any JPEG decoder returns for this file
        return np.array(im)


def test_make3d1_layout_names_and_values(tmp_path):
    rng = np.random.default_rng(1)
    data = tmp_path / 'data'
    un = data / 'make3d1' / 'unpacked'
    for d in ('Train400Img', 'Train400Depth', 'Test134', 'Test134Depth'):
        os.makedirs(un / d)
    ids = ['10.21op2-p-015t000', '10.21op2-p-046t000', '10.21op3-p-139t000']
    imgs, grids = {}, {}
    for k, i in enumerate(ids):
        split_img, split_depth = ('Train400Img', 'Train400Depth') if k < 2 else ('Test134', 'Test134Depth')
        imgs[i] = jpeg(str(un / split_img / f'img-{i}.jpg'), rng, (40, 30, 3))
        grids[i] = rng.uniform(0.9, 81.0, (11, 9, 4))
        sio.savemat(str(un / split_depth / f'depth_sph_corr-{i}.mat'), {'Position3DGrid': grids[i]}, do_compression=k % 2 == 0)
    (un / 'Train400Img' / 'Thumbs.db').write_bytes(b'x')                  # the reference's filter drops these
    (un / 'Train400Depth' / 'readme.txt').write_bytes(b'x')
    env = dict(ENV, DATA_DIR=str(data))
    logs = []
    assert pre.main(['make3d1'], env, logs.append) == 0
    # named after the depth file: between its first '-' and its first '.'  (tools/data_preprocessor.py:86)
    stem = lambda i: f'depth_sph_corr-{i}.mat'[len('depth_sph_corr-'):].split('.')[0]
    assert stem(ids[0]) == '10'                                           # ... which is all the reference keeps of these ids
    train = sorted(os.listdir(data / 'make3d1' / 'train'))
    # both training samples get the stem '10': the second overwrites the first, as in the reference
    assert train == ['10-depth.png', '10-image.png']
    assert sorted(os.listdir(data / 'make3d1' / 'test')) == ['10-depth.png', '10-image.png']
    got = png.imread(str(data / 'make3d1' / 'train' / '10-image.png'))
    want = imresize.imresize(imgs[ids[1]], (16, 12))                      # size = (WIDTH, HEIGHT) read as (rows, cols), as there
    assert got.shape == (16, 12, 3)
    np.testing.assert_array_equal(got, want)
    got = png.imread(str(data / 'make3d1' / 'test' / '10-depth.png'))
    cfg = pre.settings(env)
    want = imresize.imresize(grids[ids[2]][..., 3], (cfg['d_width'], cfg['d_height']))
    assert got.shape == (5 * 16 // 12, 5) and got.dtype == np.uint8
    np.testing.assert_array_equal(got, want)


def test_make3d2_turns_the_image_and_skips_bad_samples(tmp_path):
    rng = np.random.default_rng(2)
    data = tmp_path / 'data'
    un = data / 'make3d2' / 'unpacked'
    for d in ('Dataset3_Images', 'Dataset3_Depths', 'Dataset2_Images', 'Dataset2_Depths'):
        os.makedirs(un / d)
    imgs, maps = {}, {}
    for split, names in (('Dataset3', ['a01', 'a02', 'a03']), ('Dataset2', ['b01'])):
        for n in names:
            imgs[n] = jpeg(str(un / f'{split}_Images' / f'img-{n}.jpg'), rng, (24, 36, 3))
            maps[n] = rng.uniform(1.0, 80.0, (13, 17)).astype(np.float32)
            sio.savemat(str(un / f'{split}_Depths' / f'depth-{n}.mat'), {'depthMap': maps[n]})
    # a depth file that is not a MAT file: ValueError -> reported and skipped (tools/data_preprocessor.py:134-136)
    (un / 'Dataset3_Depths' / 'depth-a02.mat').write_bytes(b'garbage' * 40)
    env = dict(ENV, DATA_DIR=str(data))
    logs = []
    assert pre.main(['make3d2'], env, logs.append) == 0
    assert sorted(os.listdir(data / 'make3d2' / 'train')) == ['a01-depth.png', 'a01-image.png', 'a03-depth.png', 'a03-image.png']
    assert sorted(os.listdir(data / 'make3d2' / 'test')) == ['b01-depth.png', 'b01-image.png']
    assert any('Skipping sample 1, depth-a02.mat and img-a02.jpg' in str(l) for l in logs)
    got = png.imread(str(data / 'make3d2' / 'train' / 'a03-image.png'))
    np.testing.assert_array_equal(got, imresize.imresize(np.rot90(imgs['a03'], k=-1), (16, 12)))
    got = png.imread(str(data / 'make3d2' / 'test' / 'b01-depth.png'))
    cfg = pre.settings(env)
    np.testing.assert_array_equal(got, imresize.imresize(maps['b01'], (cfg['d_width'], cfg['d_height'])))
    # START / LIMIT slice the listings (depths[START:LIMIT]); a second run needs FORCE
    logs.clear()
    pre.main(['make3d2'], env, logs.append)
    assert any('Directory is not empty' in str(l) for l in logs)
    pre.main(['make3d2'], dict(env, FORCE='1', START='2', LIMIT='3'), logs.append)
    assert sorted(os.listdir(data / 'make3d2' / 'train')) == ['a03-depth.png', 'a03-image.png']
    assert os.listdir(data / 'make3d2' / 'test') == []


def test_mnist_files_are_moved(tmp_path):
    data = tmp_path / 'data'
    un = data / 'mnist' / 'unpacked'
    os.makedirs(un)
    for fn in ('train-images-idx3-ubyte', 'train-labels-idx1-ubyte', 't10k-images-idx3-ubyte', 't10k-labels-idx1-ubyte', 'README'):
        (un / fn).write_bytes(fn.encode())
    logs = []
    assert pre.main(['mnist'], {'DATA_DIR': str(data)}, logs.append) == 0
    assert sorted(os.listdir(data / 'mnist' / 'train')) == ['train-images-idx3-ubyte', 'train-labels-idx1-ubyte']
    assert sorted(os.listdir(data / 'mnist' / 'test')) == ['t10k-images-idx3-ubyte', 't10k-labels-idx1-ubyte']
    assert os.listdir(un) == ['README'] and 'Skipping README' in logs
    assert (data / 'mnist' / 'test' / 't10k-labels-idx1-ubyte').read_bytes() == b't10k-labels-idx1-ubyte'
