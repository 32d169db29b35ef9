"""CPU restatement of the DCNF unary conv stack ("DCNF-lite", Liu et al. 2015) of ann3depth.

TEST INFRASTRUCTURE ONLY — see ``oracle/tf13_ops.py``.  PARITY UNPINNED.
Follows ``/root/reference/src/models.py:14-18,50-89,179-183``: resize to 240x320, 48 overlapping 100x100
patches per image (stride 40, SAME zero padding), one shared conv stack per patch (make_template,
src/models.py:61), one scalar z per patch.  The pairwise part and the CRF loss (src/models.py:91-177) are
not on the north-star path.
"""
import numpy as np

from . import tf13_ops as T

IMG_H, IMG_W = 240, 320      # src/models.py:180
PATCH = 100                  # src/models.py:15
SP = 40                      # src/models.py:16 (patch stride)

# TF default layer names inside the 'unary/unary_layers' template scope.
PREFIX = 'unary/unary_layers/'
CONVS = [  # (name, kernel shape); all VALID, stride 1, ReLU   src/models.py:64-72
    ('conv2d', (11, 11, 3, 64)),
    ('conv2d_1', (5, 5, 64, 256)),
    ('conv2d_2', (3, 3, 256, 256)),
    ('conv2d_3', (3, 3, 256, 256)),
    ('conv2d_4', (3, 3, 256, 256)),
]
POOL_AFTER = {'conv2d', 'conv2d_1', 'conv2d_4'}       # src/models.py:65,68,73
DENSES = [('dense', (12544, 128), 'relu'), ('dense_1', (128, 16), 'sigmoid'), ('dense_2', (16, 1), None)]


def param_shapes():
    s = {}
    for n, ks in CONVS:
        s[PREFIX + n + '/kernel'] = ks
        s[PREFIX + n + '/bias'] = (ks[-1],)
    for n, (i, o), _ in DENSES:
        s[PREFIX + n + '/kernel'] = (i, o)
        s[PREFIX + n + '/bias'] = (o,)
    return s


def init_params(seed=3000, dtype=np.float32):
    rng = np.random.default_rng(seed)
    p = {}
    for name, shape in param_shapes().items():
        p[name] = T.glorot_uniform(rng, shape, dtype) if name.endswith('/kernel') else np.zeros(shape, dtype)
    return p


def patches(images):
    """dcnf.__call__ resize (src/models.py:180) + dcnf.patches (:50-59): [B,H,W,3] -> [B*48,100,100,3]."""
    x = T.resize_bilinear_tf1(images, IMG_H, IMG_W)
    pt = T.extract_patches(x, PATCH, SP, 'SAME')
    return pt.reshape(-1, PATCH, PATCH, x.shape[-1])


def unary_forward(p, patch_batch):
    """unary_part_patch over a batch of patches (src/models.py:61-83).  Returns dict of activations;
    'z' is [P,1]."""
    a = {'x': patch_batch}
    t = patch_batch
    for n, _ in CONVS:
        t = T.conv2d_fwd(t, p[PREFIX + n + '/kernel'], p[PREFIX + n + '/bias'], 1, 'VALID', True)
        a[n] = t
        if n in POOL_AFTER:
            t = T.maxpool2x2_fwd(t)
            a[n + '/pool'] = t
    t = t.reshape(t.shape[0], -1)
    a['flat'] = t
    for n, _, act in DENSES:
        t = T.dense_fwd(t, p[PREFIX + n + '/kernel'], p[PREFIX + n + '/bias'], act)
        a[n] = t
    a['z'] = t
    return a


def unary_backward(p, a, dz):
    """Gradients of sum(z * dz) wrt all unary variables (dz [P,1] is a synthetic upstream gradient)."""
    g = {}
    d = dz
    inputs = ['flat', 'dense', 'dense_1']
    for (n, _, act), xin in reversed(list(zip(DENSES, inputs))):
        y = a[n]
        if act == 'relu':
            d = T.relu_grad(d, y)
        elif act == 'sigmoid':
            d = d * y * (1 - y)
        d, g[PREFIX + n + '/kernel'], g[PREFIX + n + '/bias'] = T.dense_bwd(a[xin], p[PREFIX + n + '/kernel'], d)
    prev_out = {'conv2d': 'x', 'conv2d_1': 'conv2d/pool', 'conv2d_2': 'conv2d_1/pool',
                'conv2d_3': 'conv2d_2', 'conv2d_4': 'conv2d_3'}
    d = d.reshape(a['conv2d_4/pool'].shape)
    for n, ks in reversed(CONVS):
        if n in POOL_AFTER:
            d = T.maxpool2x2_bwd(a[n], d)
        dzc = T.relu_grad(d, a[n])
        x = a[prev_out[n]]
        g[PREFIX + n + '/kernel'], g[PREFIX + n + '/bias'] = T.conv2d_bwd_filter(x, dzc, ks, 1, 'VALID')
        if n != 'conv2d':
            d = T.conv2d_bwd_data(dzc, p[PREFIX + n + '/kernel'], x.shape, 1, 'VALID')
    return g


def forward(p, images):
    """dcnf unary z for a batch of images: [B,H,W,3] -> [B,48,1] (src/models.py:85-89,183)."""
    pb = patches(images)
    z = unary_forward(p, pb)['z']
    return z.reshape(images.shape[0], -1, 1)


# =====================================================================================================
# Pairwise part and CRF negative log-likelihood — src/models.py:20-48,91-177,185-200
# =====================================================================================================
# TF-1.3 semantics this restatement assumes (UNVERIFIED here, TensorFlow cannot run):
#   * tf.scatter_nd_update on the non-trainable Variable R (src/models.py:138-141) has no registered gradient
#     (state_grad.py lists ScatterNdUpdate as NotDifferentiable), so A = I + D - R is a CONSTANT for the optimizer:
#     only z (the unary stack) receives a gradient, the pairwise dense layer never trains;
#   * tf.histogram_fixed_width computes floor(nbins * (v - lo) / (hi - lo)) in float32 and clips to [0, nbins-1];
#   * matrix_determinant / matrix_inverse are LU-based (any correctly rounded-ish LU agrees to ~1e-6 here).
GAMMA = 1.0          # src/models.py:17
EPSILON = 1e-7       # src/models.py:18
N_ROWS, N_COLS = IMG_H // SP, IMG_W // SP      # 6 x 8 superpixels of 40x40 (src/models.py:32-35)
N_SP = N_ROWS * N_COLS
PAIR_PREFIX = 'pairwise/pairwise_layers/dense/'


def pair_indices():
    """src/models.py:20-30: interior checkerboard superpixels and their four neighbours: 48 (left, right) pairs."""
    left, right = [], []
    for row in range(1, N_ROWS - 1):
        for col in range(2 - (row & 1), N_COLS - 1, 2):
            pixel = row * N_COLS + col
            for addend in (-N_COLS, N_COLS, -1, 1):
                left.append(pixel)
                right.append(pixel + addend)
    return np.array(left), np.array(right)


def superpixels(x):
    """extract_image_patches 40x40 / stride 40 (src/models.py:37-48): [B,240,320,C] -> [B,48,1600,C]."""
    B, H, W, C = x.shape
    v = x.reshape(B, N_ROWS, SP, N_COLS, SP, C).transpose(0, 1, 3, 2, 4, 5)
    return v.reshape(B, N_SP, SP * SP, C)


def color_histogram(sp):
    """src/models.py:95-100 for all superpixels: [B,48,1600,3] -> [B,48,256] float counts."""
    dt = sp.dtype
    values = (sp * np.array([16777216., 65536., 256.], dt)).sum(axis=-1)
    B, P, _ = values.shape
    hist = np.zeros((B, P, 256), dt)
    for b in range(B):
        for p in range(P):
            hist[b, p] = T.histogram_fixed_width(values[b, p], (0.0, 16777216.0), 256)
    return hist


def similarity(features, pairs):
    """src/models.py:102-106: exp(-gamma * ||f[left] - f[right]||_2) per pair."""
    diff = features[:, pairs[0]] - features[:, pairs[1]]
    return np.exp(-features.dtype.type(GAMMA) * np.sqrt((diff * diff).sum(axis=2)))


def pairwise_init(seed=3001, dtype=np.float32):
    rng = np.random.default_rng(seed)
    return {PAIR_PREFIX + 'kernel': T.glorot_uniform(rng, (2, 1), dtype), PAIR_PREFIX + 'bias': np.zeros((1,), dtype)}


def pairwise_forward(p, images240):
    """pairwise_part (src/models.py:108-127): images already resized to 240x320 -> r [B,48,1] and the two
    similarity features [B,48,2]."""
    sp = superpixels(images240)
    pairs = pair_indices()
    cdiff = similarity(sp.mean(axis=-1), pairs)
    hdiff = similarity(color_histogram(sp), pairs)
    sims = np.stack([cdiff, hdiff], axis=-1)
    r = sims @ p[PAIR_PREFIX + 'kernel'] + p[PAIR_PREFIX + 'bias']
    return r, sims


def crf_matrix(r):
    """get_A (src/models.py:136-143) for one image: r [48] -> A = I + D - R [48,48]."""
    left, right = pair_indices()
    n = r.shape[0]
    R = np.zeros((n, n), r.dtype)
    R[left, right] = r
    R[right, left] = r
    return np.eye(n, dtype=r.dtype) + np.diag(R.sum(axis=1)) - R


def crf_loss(depths240, z, r):
    """loss_part (src/models.py:129-177).  Returns (mean loss, per-image losses, d mean_loss / d z [B,48,1])
    with A treated as a constant (see the assumptions above)."""
    dt = z.dtype
    eps = dt.type(EPSILON)
    y = superpixels(depths240).mean(axis=2)                       # [B,48,1]
    B, n = z.shape[0], z.shape[1]
    losses = np.zeros(B, dt)
    dz = np.zeros_like(z)
    fac0 = dt.type(np.pi ** (n / 2))
    for b in range(B):
        A = crf_matrix(r[b, :, 0])
        zb, yb = z[b, :, 0], y[b, :, 0]
        energy = yb @ A @ yb - 2 * (zb @ yb) + zb @ zb
        with np.errstate(invalid='ignore'):
            fac = fac0 / (np.sqrt(np.linalg.det(A).astype(dt)) + eps)
        invA = np.linalg.inv(A).astype(dt) + eps
        g = zb @ invA @ zb - zb @ zb
        ex = np.exp(g)
        Z = fac * ex + eps
        u = np.exp(-energy) / Z
        losses[b] = -np.log(u + eps)
        # d loss_b / d z, A constant
        dE = -2 * yb + 2 * zb
        dg = invA @ zb + invA.T @ zb - 2 * zb
        dZ = fac * ex * dg
        du = u * (-dE) - (u / Z) * dZ
        dz[b, :, 0] = (-du / (u + eps)) / dt.type(B)
    return losses.mean(dtype=dt), losses, dz
