"""Pins the restated Pillow resampler (ann3depth_amd/imresize.py) and the PNG writer against Pillow itself — the library
scipy.misc.imresize / imsave call underneath in the reference's preprocessor (tools/data_preprocessor.py:195-208)."""
import numpy as np
import pytest

from ann3depth_amd import imresize, png

Image = pytest.importorskip('PIL.Image')
BILINEAR = getattr(Image, 'Resampling', Image).BILINEAR

SHAPES = [  # (rows, cols) -> (rows, cols)
    ((480, 640), (55, 73)),          # the preprocessor's default depth size (Makefile:50-51)
    ((480, 640), (228, 304)), ((48, 64), (480, 640)), ((37, 53), (37, 11)), ((9, 7), (31, 7)), ((100, 5), (7, 6)),
    ((64, 48), (64, 48)), ((2, 2), (5, 9)), ((1, 17), (1, 4)),
]


@pytest.mark.parametrize('src,dst', SHAPES)
@pytest.mark.parametrize('channels', [None, 3])
def test_imresize_equals_pillow_bilinear(src, dst, channels):
    rng = np.random.default_rng(hash((src, dst, channels)) & 0xFFFF)
    shape = src if channels is None else src + (channels,)
    u8 = rng.integers(0, 256, shape).astype(np.uint8)
    want = np.asarray(Image.fromarray(u8).resize((dst[1], dst[0]), BILINEAR))
    np.testing.assert_array_equal(imresize.imresize(u8, dst), want)


def test_imresize_of_float_depth_goes_through_bytescale_like_toimage():
    """scipy.misc.imresize(float array): toimage() scales to the image's own min..max first (bytescale), THEN Pillow
    resamples the 8-bit image — what strips the metric scale from the depth maps (tools/data_preprocessor.py:199-202)."""
    rng = np.random.default_rng(3)
    depth = rng.uniform(0.7, 9.9, (480, 640)).astype(np.float32)
    u8 = imresize.bytescale(depth)
    assert u8.min() == 0 and u8.max() == 255
    want = np.asarray(Image.fromarray(u8).resize((73, 55), BILINEAR))
    np.testing.assert_array_equal(imresize.imresize(depth, (55, 73)), want)


def test_extreme_values_saturate_like_pillow():
    u8 = np.zeros((16, 16), np.uint8)
    u8[::2] = 255
    for dst in ((5, 5), (40, 40), (16, 3)):
        want = np.asarray(Image.fromarray(u8).resize((dst[1], dst[0]), BILINEAR))
        np.testing.assert_array_equal(imresize.imresize(u8, dst), want)


@pytest.mark.parametrize('shape', [(55, 73), (48, 64, 3)])
def test_imsave_is_read_back_by_pillow(tmp_path, shape):
    rng = np.random.default_rng(len(shape))
    u8 = rng.integers(0, 256, shape).astype(np.uint8)
    p = str(tmp_path / 'x.png')
    imresize.imsave(p, u8)
    with Image.open(p) as im:
        assert im.mode == ('L' if len(shape) == 2 else 'RGB')
        np.testing.assert_array_equal(np.asarray(im), u8)
    np.testing.assert_array_equal(png.imread(p), u8)
