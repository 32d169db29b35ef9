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


def adam_update_numpy(param, g_t, t, m, v, alpha, beta1, beta2, epsilon):
    """adam_test.py's reference recurrence, in float64."""
    alpha_t = alpha * np.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
    m_t = beta1 * m + (1 - beta1) * g_t
    v_t = beta2 * v + (1 - beta2) * g_t * g_t
    return param - alpha_t * m_t / (np.sqrt(v_t) + epsilon), m_t, v_t


def test_adam_basic():
    """AdamOptimizerTest.testBasic: the oracle's ApplyAdam (fp32, TF's operation order) follows the test's float64
    recurrence to the test's own float32 tolerance, beta powers included."""
    c = V.ADAM_TEST_BASIC
    opt = T.AdamTF1(c['lr'], c['beta1'], c['beta2'], c['epsilon'])
    var = {'v0': np.array(c['var0'], np.float32), 'v1': np.array(c['var1'], np.float32)}
    g = {'v0': np.array(c['grads0'], np.float32), 'v1': np.array(c['grads1'], np.float32)}
    ref = {k: (np.array(c['var' + k[1]], np.float64), 0.0, 0.0) for k in var}
    for t in range(1, c['steps'] + 1):
        assert abs(float(opt.beta1_power) - c['beta1'] ** t) < 1e-6 and abs(float(opt.beta2_power) - c['beta2'] ** t) < 1e-6
        opt.apply(var, g)
        for k in var:
            ref[k] = adam_update_numpy(ref[k][0], np.array(c['grads' + k[1]], np.float64), t, ref[k][1], ref[k][2],
                                       c['lr'], c['beta1'], c['beta2'], c['epsilon'])
            np.testing.assert_allclose(var[k], ref[k][0], rtol=1e-6, atol=1e-6)


def test_maxpool_grad_ties_go_to_the_first_maximum():
    c = V.MAXPOOL_GRAD_DIRECT1
    x = np.array(c['input_data'], np.float32).reshape(c['input_sizes'])
    dy = np.array(c['output_backprop'], np.float32).reshape(1, 3, 3, 1)
    np.testing.assert_array_equal(T.maxpool_grad(x, dy, c['window'], c['stride']).ravel(),
                                  np.array(c['expected_input_backprop'], np.float32))
    # the 2x2 / stride-2 routine the path uses is that rule: equal to the general one on inputs full of ties
    rng = np.random.default_rng(3)
    xt = rng.integers(0, 3, (2, 6, 8, 5)).astype(np.float32)
    dyt = rng.standard_normal((2, 3, 4, 5)).astype(np.float32)
    np.testing.assert_array_equal(T.maxpool2x2_bwd(xt, dyt), T.maxpool_grad(xt, dyt, 2, 2))
    ones = np.ones((1, 4, 4, 1), np.float32)
    got = T.maxpool2x2_bwd(ones, np.array([[[[5.]], [[6.]]], [[[7.]], [[8.]]]], np.float32).reshape(1, 2, 2, 1))
    np.testing.assert_array_equal(got.reshape(4, 4), [[5, 0, 6, 0], [0, 0, 0, 0], [7, 0, 8, 0], [0, 0, 0, 0]])


def test_histogram_fixed_width():
    c = V.HISTOGRAM_FIXED_WIDTH
    np.testing.assert_array_equal(T.histogram_fixed_width(np.array(c['new_values'], np.float32), c['value_range'], c['nbins']),
                                  c['expected'])


@pytest.mark.parametrize('keep_prob', V.DROPOUT_KEEP_PROBS)
def test_dropout_values(keep_prob):
    x = np.ones((40, 30), np.float32)
    keep = np.random.default_rng(0).random(x.shape) < keep_prob
    y = T.dropout_fwd(x, keep, rate=1.0 - keep_prob)
    vals = np.unique(y)
    assert len(vals) == 2 and vals[0] == 0
    np.testing.assert_allclose(vals[1], 1 / keep_prob, rtol=1e-6)


def test_gradient_descent_basic():
    """GradientDescentOptimizerTest.testBasic against the update rule the DCNF oracle and sgd_kernel apply
    (var -= lr * g in the variable's dtype); the HIP kernel runs the same vector in tests/test_gpu_dcnf.py."""
    c = V.SGD_TEST_BASIC
    for k in ('0', '1'):
        var = np.array(c['var' + k], np.float32)
        var -= np.float32(c['learning_rate']) * np.array(c['grads' + k], np.float32)
        np.testing.assert_allclose(var, c['expected' + k], rtol=1e-6)


def test_scatter_nd_update_documented_example_and_the_crf_matrix():
    """The documented rank-1 example through numpy's index assignment (what oracle.dcnf.crf_matrix uses for get_A's two
    scatter_nd_update calls), then the matrix itself: R symmetric with r on the pair positions, A = I + D - R."""
    from oracle import dcnf as OD
    c = V.SCATTER_ND_UPDATE_DOC
    ref = np.array(c['ref'])
    ref[np.array(c['indices'])[:, 0]] = c['updates']
    np.testing.assert_array_equal(ref, c['expected'])
    left, right = OD.pair_indices()
    r = np.arange(1, len(left) + 1, dtype=np.float64)
    A = OD.crf_matrix(r)
    R = np.zeros_like(A)
    for (i, j), v in zip(zip(left, right), r):          # scatter_nd_update: one element per index pair, later wins
        R[i, j] = v
    for (i, j), v in zip(zip(right, left), r):
        R[i, j] = v
    np.testing.assert_array_equal(A, np.eye(A.shape[0]) + np.diag(R.sum(axis=1)) - R)
    np.testing.assert_array_equal(R, R.T)
