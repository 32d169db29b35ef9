"""CPU restatement of the TensorFlow-1.3 ops that ann3depth's training path instantiates.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it, and only as the
checker.  The product path (``ann3depth_amd``) never imports this package.

PARITY UNPINNED.  The reference (shoeffner/ann3depth) ships no tests, golden vectors or fixtures for
this path, and its arithmetic lives in the un-vendored dependency ``tensorflow==1.3.0``
(``/root/reference/requirements-cpu.txt:1``), which cannot be installed or imported here.  Every
function below restates the published TF-1.3 kernel semantics for one call site in
``/root/reference/src/models.py`` / ``src/data.py`` and is cross-checked in ``tests/`` against an
independent implementation (torch-CPU float64 with explicit padding), hand-computed known answers and the
known-answer vectors of TensorFlow 1.3's own unit tests for conv2d / its two gradients / max-pool / bilinear
resize (``tests/golden/tf13_published_vectors.py``), but not against output of TensorFlow run here.

Layouts are TensorFlow's: activations NHWC, conv filters HWIO, dense kernels [in, out].
All functions are pure numpy and work in the dtype of their inputs (float32 to mirror the reference,
float64 to act as a high-precision truth for tolerance tests).
"""
import math

import numpy as np


# --------------------------------------------------------------------------------------------
# padding arithmetic  (TF-1.3 common_shape_fns / GetWindowedOutputSize)
# --------------------------------------------------------------------------------------------
def conv_out_size(in_size, k, stride, padding):
    """Output extent and (before, after) padding of one spatial axis.

    VALID: out = (in - k)//s + 1, no padding.  SAME: out = ceil(in/s),
    pad = max((out-1)*s + k - in, 0), before = pad//2, after = pad - before.
    Used by tf.layers.conv2d (src/models.py:64-72,211-223,241-251), max_pooling2d
    (:65,68,73,213,216,243) and extract_image_patches (:53-57).
    """
    padding = padding.upper()
    if padding == 'VALID':
        return (in_size - k) // stride + 1, 0, 0
    if padding == 'SAME':
        out = -(-in_size // stride)
        pad = max((out - 1) * stride + k - in_size, 0)
        return out, pad // 2, pad - pad // 2
    raise ValueError(padding)


def _pad_hw(x, pt, pb, pl, pr):
    if pt == pb == pl == pr == 0:
        return x
    return np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))


def _windows(xp, R, S, stride, Ho, Wo):
    """Strided [B,Ho,Wo,R,S,C] view of a padded NHWC tensor."""
    B, Hp, Wp, C = xp.shape
    sb, sh, sw, sc = xp.strides
    return np.lib.stride_tricks.as_strided(
        xp, (B, Ho, Wo, R, S, C), (sb, sh * stride, sw * stride, sh, sw, sc), writeable=False)


# --------------------------------------------------------------------------------------------
# conv2d  (Conv2D + BiasAdd + Relu; Conv2DBackpropInput / Conv2DBackpropFilter / BiasAddGrad / ReluGrad)
# --------------------------------------------------------------------------------------------
def conv2d_fwd(x, w, b=None, stride=1, padding='VALID', relu=False):
    """tf.layers.conv2d(x, Cout, k, (s,s), padding, activation) — src/models.py:211-223,241-251,64-72.

    x [B,H,W,Cin], w [R,S,Cin,Cout] (HWIO), b [Cout].  Returns y [B,Ho,Wo,Cout].
    """
    B, H, W, C = x.shape
    R, S, C2, K = w.shape
    assert C == C2
    Ho, pt, pb = conv_out_size(H, R, stride, padding)
    Wo, pl, pr = conv_out_size(W, S, stride, padding)
    xp = _pad_hw(x, pt, pb, pl, pr)
    cols = _windows(xp, R, S, stride, Ho, Wo).reshape(B * Ho * Wo, R * S * C)
    y = cols @ w.reshape(R * S * C, K)
    if b is not None:
        y = y + b
    if relu:
        y = np.maximum(y, 0)
    return y.reshape(B, Ho, Wo, K).astype(x.dtype, copy=False)


def conv2d_bwd_filter(x, dz, wshape, stride=1, padding='VALID'):
    """Conv2DBackpropFilter + BiasAddGrad.  dz is the gradient wrt the pre-activation output.

    Returns (dw [R,S,Cin,Cout], db [Cout]).
    """
    B, H, W, C = x.shape
    R, S, _, K = wshape
    Ho, pt, pb = conv_out_size(H, R, stride, padding)
    Wo, pl, pr = conv_out_size(W, S, stride, padding)
    xp = _pad_hw(x, pt, pb, pl, pr)
    cols = _windows(xp, R, S, stride, Ho, Wo).reshape(B * Ho * Wo, R * S * C)
    dz2 = dz.reshape(B * Ho * Wo, K)
    dw = (cols.T @ dz2).reshape(R, S, C, K)
    db = dz2.sum(axis=0)
    return dw.astype(x.dtype, copy=False), db.astype(x.dtype, copy=False)


def conv2d_bwd_data(dz, w, xshape, stride=1, padding='VALID'):
    """Conv2DBackpropInput.  Returns dx [B,H,W,Cin]."""
    B, H, W, C = xshape
    R, S, _, K = w.shape
    Ho, pt, pb = conv_out_size(H, R, stride, padding)
    Wo, pl, pr = conv_out_size(W, S, stride, padding)
    dxp = np.zeros((B, H + pt + pb, W + pl + pr, C), dtype=dz.dtype)
    dz2 = dz.reshape(B * Ho * Wo, K)
    for r in range(R):
        for s in range(S):
            contrib = (dz2 @ w[r, s].T).reshape(B, Ho, Wo, C)
            dxp[:, r:r + (Ho - 1) * stride + 1:stride, s:s + (Wo - 1) * stride + 1:stride, :] += contrib
    return dxp[:, pt:pt + H, pl:pl + W, :]


def relu_grad(dy, y):
    """ReluGrad: dy * (y > 0)."""
    return dy * (y > 0)


# --------------------------------------------------------------------------------------------
# max_pooling2d(x, 2, 2)  VALID  (MaxPool / MaxPoolGrad) — src/models.py:65,68,73,213,216,243
# --------------------------------------------------------------------------------------------
def maxpool2x2_fwd(x):
    B, H, W, C = x.shape
    Ho, Wo = H // 2, W // 2
    v = x[:, :Ho * 2, :Wo * 2, :].reshape(B, Ho, 2, Wo, 2, C)
    return v.max(axis=(2, 4))


def maxpool2x2_bwd(x, dy):
    """Gradient goes to the first maximum of each window in (row, col) scan order (strict '>' update
    in TF's CPU SpatialMaxPoolWithArgMaxHelper); rows/cols cut off by VALID flooring get 0."""
    B, H, W, C = x.shape
    Ho, Wo = H // 2, W // 2
    v = x[:, :Ho * 2, :Wo * 2, :].reshape(B, Ho, 2, Wo, 2, C).transpose(0, 1, 3, 5, 2, 4)
    v = v.reshape(B, Ho, Wo, C, 4)
    arg = v.argmax(axis=-1)                       # numpy argmax = first maximum
    onehot = (arg[..., None] == np.arange(4)).astype(dy.dtype)
    g = onehot * dy[..., None]                    # [B,Ho,Wo,C,4]
    g = g.reshape(B, Ho, Wo, C, 2, 2).transpose(0, 1, 4, 2, 5, 3).reshape(B, Ho * 2, Wo * 2, C)
    dx = np.zeros_like(x, dtype=dy.dtype)
    dx[:, :Ho * 2, :Wo * 2, :] = g
    return dx


def maxpool_grad(x, dy, k, stride):
    """MaxPoolGrad for any square window / stride, VALID: each window's gradient goes to its FIRST maximum in (row, col)
    scan order.  Slow reference (loops); maxpool2x2_bwd above is the vectorised 2x2 / stride-2 case the path uses, and
    tests pin the two to each other and this one to TensorFlow's _testMaxPoolGradDirect1."""
    B, H, W, C = x.shape
    Ho, Wo = (H - k) // stride + 1, (W - k) // stride + 1
    dx = np.zeros_like(x, dtype=dy.dtype)
    for b in range(B):
        for i in range(Ho):
            for j in range(Wo):
                for c in range(C):
                    win = x[b, i * stride:i * stride + k, j * stride:j * stride + k, c]
                    r, q = np.unravel_index(int(np.argmax(win)), win.shape)      # numpy argmax = first maximum
                    dx[b, i * stride + r, j * stride + q, c] += dy[b, i, j, c]
    return dx


def histogram_fixed_width(values, value_range, nbins):
    """tf.histogram_fixed_width (TF 1.3 histogram_ops.py): scaled = (v - lo) / (hi - lo), index = floor(nbins * scaled)
    clipped to [0, nbins - 1], counted — all in the values' dtype.  Used by the reference's color_histogram
    (src/models.py:95-100; oracle/dcnf.py restates that call with nbins = 256 over [0, 2^24))."""
    v = np.asarray(values)
    dt = v.dtype.type if v.dtype.kind == 'f' else np.float32
    v = v.astype(dt)
    lo, hi = dt(value_range[0]), dt(value_range[1])
    scaled = (v - lo) / (hi - lo)
    idx = np.clip(np.floor(dt(nbins) * scaled).astype(np.int64), 0, nbins - 1)
    return np.bincount(idx.ravel(), minlength=nbins).astype(np.int32)


# --------------------------------------------------------------------------------------------
# dense / dropout — src/models.py:80-82,228-231
# --------------------------------------------------------------------------------------------
def dense_fwd(x, w, b=None, act=None):
    """tf.layers.dense: x[B,in] @ w[in,out] + b, optional 'relu' / 'sigmoid'."""
    y = x @ w
    if b is not None:
        y = y + b
    if act == 'relu':
        y = np.maximum(y, 0)
    elif act == 'sigmoid':
        y = 1 / (1 + np.exp(-y))
    elif act is not None:
        raise ValueError(act)
    return y.astype(x.dtype, copy=False)


def dense_bwd(x, w, dz):
    """Returns (dx, dw, db) for z = x@w+b given dz (gradient wrt pre-activation)."""
    return dz @ w.T, x.T @ dz, dz.sum(axis=0)


def dropout_fwd(x, keep_mask, rate=0.5):
    """tf.layers.dropout(x, rate=.5, training=True) — src/models.py:230.

    TF-1.3 nn.dropout: binary = floor(keep_prob + U[0,1)); y = x / keep_prob * binary.
    TF's Philox stream is not reproducible, so the keep mask is an INPUT (bool / {0,1}).
    """
    keep = 1.0 - rate
    return (x / x.dtype.type(keep)) * keep_mask.astype(x.dtype)


def dropout_bwd(dy, keep_mask, rate=0.5):
    keep = 1.0 - rate
    return (dy / dy.dtype.type(keep)) * keep_mask.astype(dy.dtype)


# --------------------------------------------------------------------------------------------
# tf.image.resize_images(x, [h, w])  = ResizeBilinear(align_corners=False), legacy mapping
# src/models.py:180-181,282-283
# --------------------------------------------------------------------------------------------
def _interp_weights(out_size, in_size):
    """TF-1.3 resize_bilinear_op.cc compute_interpolation_weights: float32 arithmetic,
    scale = in/out, src = i*scale, lower = int(src), upper = min(lower+1, in-1), lerp = src-lower."""
    scale = np.float32(in_size) / np.float32(out_size)
    src = np.arange(out_size, dtype=np.float32) * scale
    lower = src.astype(np.int64)
    upper = np.minimum(lower + 1, in_size - 1)
    lerp = src - lower.astype(np.float32)
    return lower, upper, lerp.astype(np.float32)


def resize_bilinear_tf1(x, out_h, out_w):
    """Bilinear resize, align_corners=False, NO half-pixel offset, no antialias.

    out = top + (bottom - top) * y_lerp with top = tl + (tr - tl) * x_lerp (each op rounded to the
    working dtype, no fused multiply-add), as resize_bilinear_op.cc resize_image().
    """
    B, H, W, C = x.shape
    if (H, W) == (out_h, out_w):
        return x.copy()
    ylo, yhi, yl = _interp_weights(out_h, H)
    xlo, xhi, xl = _interp_weights(out_w, W)
    yl = yl.astype(x.dtype)[None, :, None, None]
    xl = xl.astype(x.dtype)[None, None, :, None]
    tl = x[:, ylo][:, :, xlo]
    tr = x[:, ylo][:, :, xhi]
    bl = x[:, yhi][:, :, xlo]
    br = x[:, yhi][:, :, xhi]
    top = tl + (tr - tl) * xl
    bot = bl + (br - bl) * xl
    return top + (bot - top) * yl


# --------------------------------------------------------------------------------------------
# tf.extract_image_patches(ksizes=100x100, strides=40x40, SAME) — src/models.py:53-59
# --------------------------------------------------------------------------------------------
def extract_patches(x, k, stride, padding='SAME'):
    """Returns [B, n_rows*n_cols, k, k, C] (the reshape of src/models.py:58-59); zero padded."""
    B, H, W, C = x.shape
    Ho, pt, pb = conv_out_size(H, k, stride, padding)
    Wo, pl, pr = conv_out_size(W, k, stride, padding)
    xp = _pad_hw(x, pt, pb, pl, pr)
    win = _windows(xp, k, k, stride, Ho, Wo)
    return np.ascontiguousarray(win).reshape(B, Ho * Wo, k, k, C)


# --------------------------------------------------------------------------------------------
# scale-invariant log loss — src/models.py:255-275
# --------------------------------------------------------------------------------------------
SILOG_EPS = 1e-8
SILOG_LAMBDA = 0.5
SILOG_N = 74 * 55


def _masked_log(v):
    with np.errstate(invalid='ignore', divide='ignore'):
        lv = np.log(v + v.dtype.type(SILOG_EPS))
    nan = np.isnan(lv)
    return np.where(nan, v.dtype.type(0), lv), nan


def silog_loss_fwd(outputs, targets):
    """loss = mean_b( sum_i d_i^2 - (0.5/4070) * (sum_i d_i)^2 ),  d = log(o+1e-8) - log(t+1e-8),
    NaN logs (argument < 0) replaced by 0; -inf (argument == 0) kept.  Not divided by n.
    The constant 0.5/(74*55) is a Python float folded to the tensor dtype (src/models.py:269)."""
    B = outputs.shape[0]
    o = outputs.reshape(B, -1)
    t = targets.reshape(B, -1)
    lo, _ = _masked_log(o)
    lt, _ = _masked_log(t)
    d = lo - lt
    c = o.dtype.type(SILOG_LAMBDA / SILOG_N)
    per = (d * d).sum(axis=1) - c * np.square(d.sum(axis=1))
    return per.mean(dtype=o.dtype)


def silog_loss_bwd(outputs, targets, dloss=1.0):
    """d loss / d outputs.  tf.where routes the gradient only to the selected branch, so elements
    whose log was NaN get 0; others get (2 d_i - 2*c*sum d) / (B * (o_i + eps))."""
    B = outputs.shape[0]
    o = outputs.reshape(B, -1)
    t = targets.reshape(B, -1)
    lo, nan_o = _masked_log(o)
    lt, _ = _masked_log(t)
    d = lo - lt
    c = o.dtype.type(SILOG_LAMBDA / SILOG_N)
    sd = d.sum(axis=1, keepdims=True)
    g = (2 * d - 2 * c * sd) * o.dtype.type(dloss / B)
    with np.errstate(invalid='ignore', divide='ignore'):
        g = g / (o + o.dtype.type(SILOG_EPS))
    g = np.where(nan_o, o.dtype.type(0), g)
    return g.reshape(outputs.shape).astype(outputs.dtype, copy=False)


# --------------------------------------------------------------------------------------------
# tf.train.AdamOptimizer(lr, beta1, beta2).apply_gradients  (ApplyAdam) — src/models.py:309
# --------------------------------------------------------------------------------------------
class AdamTF1:
    """One tf.train.AdamOptimizer instance: shared beta powers, per-variable m / v slots.

    ApplyAdam (training_ops.cc, TF 1.3), all in the variable dtype:
        alpha = lr * sqrt(1 - beta2_power) / (1 - beta1_power)
        m += (g - m) * (1 - beta1);  v += (g*g - v) * (1 - beta2)
        var -= (m * alpha) / (sqrt(v) + eps)
    beta powers start at beta1 / beta2 and are multiplied AFTER all variables were applied.
    With the reference's beta2 = 1 (src/models.py:309) alpha is exactly 0: weights never move.
    """

    def __init__(self, lr, beta1=0.9, beta2=0.999, eps=1e-8, dtype=np.float32):
        self.dtype = np.dtype(dtype).type
        self.lr, self.beta1, self.beta2, self.eps = (self.dtype(v) for v in (lr, beta1, beta2, eps))
        self.beta1_power = self.dtype(beta1)
        self.beta2_power = self.dtype(beta2)
        self.m, self.v = {}, {}

    def alpha(self):
        one = self.dtype(1)
        return self.lr * np.sqrt(one - self.beta2_power) / (one - self.beta1_power)

    def apply(self, variables, grads):
        """variables/grads: dict name -> ndarray; variables are updated in place."""
        one = self.dtype(1)
        alpha = self.alpha()
        for name, g in grads.items():
            var = variables[name]
            m = self.m.setdefault(name, np.zeros_like(var))
            v = self.v.setdefault(name, np.zeros_like(var))
            m += (g - m) * (one - self.beta1)
            v += (g * g - v) * (one - self.beta2)
            var -= (m * alpha) / (np.sqrt(v) + self.eps)
        self.beta1_power = self.beta1_power * self.beta1
        self.beta2_power = self.beta2_power * self.beta2


# --------------------------------------------------------------------------------------------
# glorot_uniform (tf.layers default kernel_initializer) — values are always test INPUTS, this only
# gives them the reference's scale.
# --------------------------------------------------------------------------------------------
def glorot_uniform(rng, shape, dtype=np.float32):
    if len(shape) == 2:
        fan_in, fan_out = shape
    else:
        rf = int(np.prod(shape[:-2]))
        fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    limit = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-limit, limit, size=shape).astype(dtype)
