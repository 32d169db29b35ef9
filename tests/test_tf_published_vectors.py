"""The oracle against the known-answer vectors of TensorFlow's own unit tests (tests/golden/tf13_published_vectors.py)."""
import os
import sys

import numpy as np
import pytest

from oracle import tf13_ops as T

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
import tf13_published_vectors as V  # noqa: E402


def seq(shape):
    return np.arange(1, int(np.prod(shape)) + 1, dtype=np.float32).reshape(shape)


@pytest.mark.parametrize('case', V.CONV2D_FWD, ids=lambda c: c[0])
def test_conv2d_forward(case):
    _, xs, ws, stride, pad, want = case
    y = T.conv2d_fwd(seq(xs), seq(ws), np.zeros(ws[-1], np.float32), stride, pad, False)
    np.testing.assert_array_equal(y.ravel(), np.array(want, np.float32))


@pytest.mark.parametrize('case', V.CONV2D_BACKPROP_INPUT, ids=lambda c: c[0])
def test_conv2d_backprop_input(case):
    _, xs, ws, os_, stride, pad, want = case
    dx = T.conv2d_bwd_data(seq(os_), seq(ws), xs, stride, pad)
    np.testing.assert_array_equal(dx.ravel(), np.array(want, np.float32))


@pytest.mark.parametrize('case', V.CONV2D_BACKPROP_FILTER, ids=lambda c: c[0])
def test_conv2d_backprop_filter(case):
    _, xs, ws, os_, stride, pad, want = case
    dw, db = T.conv2d_bwd_filter(seq(xs), seq(os_), ws, stride, pad)
    np.testing.assert_array_equal(dw.ravel(), np.array(want, np.float32))
    np.testing.assert_array_equal(db, seq(os_).sum(axis=(0, 1, 2)))                  # BiasAddGrad


def test_maxpool_valid():
    xs, want = V.MAXPOOL_VALID
    np.testing.assert_array_equal(T.maxpool2x2_fwd(seq(xs)).ravel(), np.array(want, np.float32))


@pytest.mark.parametrize('case', V.RESIZE_BILINEAR, ids=lambda c: c[0])
def test_resize_bilinear(case):
    _, xs, data, h, w, want = case
    y = T.resize_bilinear_tf1(np.array(data, np.float32).reshape(xs), h, w)
    np.testing.assert_array_equal(y.ravel(), np.array(want, np.float32))


@pytest.mark.parametrize('case', V.EXTRACT_PATCHES_2X2, ids=lambda c: c[0])
def test_extract_image_patches(case):
    pad, want = case
    x = np.array([1, 2, 3, 4], np.float32).reshape(1, 2, 2, 1)
    want = np.array(want, np.float32)
    np.testing.assert_array_equal(T.extract_patches(x, 2, 1, pad).reshape(want.shape), want)
