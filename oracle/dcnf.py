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
