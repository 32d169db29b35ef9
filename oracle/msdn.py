"""CPU restatement of ann3depth's MSDN model function (Eigen et al. 2014 coarse+fine).

TEST INFRASTRUCTURE ONLY — see the header of ``oracle/tf13_ops.py``.  PARITY UNPINNED: follows
``/root/reference/src/models.py:203-367`` line by line with TF-1.3 op semantics, but the reference ships no
golden vectors and TensorFlow 1.3 cannot run here.

Variable names are the reference's TF variable names (what a TF checkpoint of the reference would hold).
"""
import numpy as np

from . import tf13_ops as T

NET_H, NET_W = 228, 304          # src/models.py:282
OUT_H, OUT_W = 55, 74            # src/models.py:283 (and 74*55 in the loss, :269)
SAMPLES_COARSE = 2000000         # src/models.py:302
SAMPLES_FINE = 1500000           # src/models.py:303

# name -> (kernel shape, stride, padding, relu)     src/models.py:211-223,241-251
CONVS = {
    'coarse/conv/conv2d_0': ((11, 11, 3, 96), 4, 'VALID', True),
    'coarse/conv/conv2d_1': ((5, 5, 96, 256), 1, 'SAME', True),
    'coarse/conv/conv2d_2': ((3, 3, 256, 384), 1, 'SAME', True),
    'coarse/conv/conv2d_3': ((3, 3, 384, 384), 1, 'SAME', True),
    'coarse/conv/conv2d_4': ((3, 3, 384, 256), 2, 'VALID', True),
    'fine/first/conv2d': ((9, 9, 3, 63), 2, 'VALID', True),
    'fine/second/conv2d': ((5, 5, 64, 64), 1, 'SAME', True),
    'fine/third': ((5, 5, 64, 1), 1, 'SAME', False),
}
DENSES = {
    'coarse/dense/dense_0': (12288, 4096),      # src/models.py:228
    'coarse/dense/dense_1': (4096, 55 * 74),    # src/models.py:231
}
COARSE_CONV_VARS = [n for n in CONVS if n.startswith('coarse/conv')]
COARSE_DENSE_VARS = list(DENSES)
FINE_A_VARS = ['fine/first/conv2d', 'fine/third']
FINE_B_VARS = ['fine/second/conv2d']


def param_shapes():
    shapes = {}
    for n, (ks, _, _, _) in CONVS.items():
        shapes[n + '/kernel'] = ks
        shapes[n + '/bias'] = (ks[-1],)
    for n, (i, o) in DENSES.items():
        shapes[n + '/kernel'] = (i, o)
        shapes[n + '/bias'] = (o,)
    return shapes


def init_params(seed=3000, dtype=np.float32):
    """glorot-uniform kernels, zero biases (tf.layers defaults); numpy PCG64 stream, fixed order."""
    rng = np.random.default_rng(seed)
    p = {}
    for name, shape in param_shapes().items():
        if name.endswith('/kernel'):
            p[name] = T.glorot_uniform(rng, shape, dtype)
        else:
            p[name] = np.zeros(shape, dtype)
    return p


def phase_of(global_step, batchsize):
    """src/models.py:301-305,348-365: 1 = coarse, 2 = fine, 3 = neither (only global_step += 1)."""
    steps_coarse = SAMPLES_COARSE // batchsize
    steps_fine = SAMPLES_FINE // batchsize
    if global_step < steps_coarse:
        return 1
    if global_step < steps_coarse + steps_fine:
        return 2
    return 3


def _conv(p, name, x):
    ks, stride, pad, relu = CONVS[name]
    return T.conv2d_fwd(x, p[name + '/kernel'], p[name + '/bias'], stride, pad, relu)


def forward(p, images, depths, keep_mask):
    """msdn.__call__ forward (src/models.py:277-290).  images [B,H,W,3], depths [B,H',W',1] as delivered
    by data.inputs; keep_mask [B,4096] is the dropout keep mask.  Returns a dict of every tensor the
    backward pass needs plus 'coarse', 'fine', 'loss_coarse', 'loss_fine'."""
    a = {}
    a['images'] = x = T.resize_bilinear_tf1(images, NET_H, NET_W)
    a['depths'] = t = T.resize_bilinear_tf1(depths, OUT_H, OUT_W)
    B = x.shape[0]
    # coarse (src/models.py:208-236)
    a['c0'] = _conv(p, 'coarse/conv/conv2d_0', x)                # [B,55,74,96]
    a['p0'] = T.maxpool2x2_fwd(a['c0'])                          # [B,27,37,96]
    a['c1'] = _conv(p, 'coarse/conv/conv2d_1', a['p0'])          # [B,27,37,256]
    a['p1'] = T.maxpool2x2_fwd(a['c1'])                          # [B,13,18,256]
    a['c2'] = _conv(p, 'coarse/conv/conv2d_2', a['p1'])          # [B,13,18,384]
    a['c3'] = _conv(p, 'coarse/conv/conv2d_3', a['c2'])          # [B,13,18,384]
    a['c4'] = _conv(p, 'coarse/conv/conv2d_4', a['c3'])          # [B,6,8,256]
    a['flat'] = a['c4'].reshape(B, -1)                           # NHWC flatten, 12288
    a['d0'] = T.dense_fwd(a['flat'], p['coarse/dense/dense_0/kernel'], p['coarse/dense/dense_0/bias'], 'relu')
    a['keep_mask'] = keep_mask
    # keep_mask None = the plugin called with train=False: tf.layers.dropout(training=False) is the identity (:230)
    a['drop'] = a['d0'] if keep_mask is None else T.dropout_fwd(a['d0'], keep_mask)
    a['d1'] = T.dense_fwd(a['drop'], p['coarse/dense/dense_1/kernel'], p['coarse/dense/dense_1/bias'])
    a['coarse'] = coarse = a['d1'].reshape(B, OUT_H, OUT_W, 1)
    # fine (src/models.py:238-253)
    a['f1'] = _conv(p, 'fine/first/conv2d', x)                   # [B,110,148,63]
    a['fp'] = T.maxpool2x2_fwd(a['f1'])                          # [B,55,74,63]
    a['cat'] = np.concatenate([a['fp'], coarse], axis=-1)        # [B,55,74,64]
    a['f2'] = _conv(p, 'fine/second/conv2d', a['cat'])           # [B,55,74,64]
    a['fine'] = _conv(p, 'fine/third', a['f2'])                  # [B,55,74,1]
    a['loss_coarse'] = T.silog_loss_fwd(coarse, t)
    a['loss_fine'] = T.silog_loss_fwd(a['fine'], t)
    return a


def _conv_bwd(p, name, x, y, dy, need_dx=True):
    """dy is the gradient wrt the layer OUTPUT (post-activation)."""
    ks, stride, pad, relu = CONVS[name]
    dz = T.relu_grad(dy, y) if relu else dy
    dw, db = T.conv2d_bwd_filter(x, dz, ks, stride, pad)
    dx = T.conv2d_bwd_data(dz, p[name + '/kernel'], x.shape, stride, pad) if need_dx else None
    return dx, dw, db


def backward_coarse(p, a):
    """Gradients of loss_coarse wrt coarse/* (src/models.py:318-324)."""
    g = {}
    B = a['coarse'].shape[0]
    dcoarse = T.silog_loss_bwd(a['coarse'], a['depths'])
    dz1 = dcoarse.reshape(B, -1)
    ddrop, g['coarse/dense/dense_1/kernel'], g['coarse/dense/dense_1/bias'] = \
        T.dense_bwd(a['drop'], p['coarse/dense/dense_1/kernel'], dz1)
    dd0 = ddrop if a['keep_mask'] is None else T.dropout_bwd(ddrop, a['keep_mask'])
    dz0 = T.relu_grad(dd0, a['d0'])
    dflat, g['coarse/dense/dense_0/kernel'], g['coarse/dense/dense_0/bias'] = \
        T.dense_bwd(a['flat'], p['coarse/dense/dense_0/kernel'], dz0)
    d = dflat.reshape(a['c4'].shape)
    # (layer, its input, its output, output was max-pooled before the next layer)
    chain = [('coarse/conv/conv2d_4', 'c3', 'c4', False), ('coarse/conv/conv2d_3', 'c2', 'c3', False),
             ('coarse/conv/conv2d_2', 'p1', 'c2', False), ('coarse/conv/conv2d_1', 'p0', 'c1', True),
             ('coarse/conv/conv2d_0', 'images', 'c0', True)]
    for name, xin, yout, pooled in chain:
        if pooled:
            d = T.maxpool2x2_bwd(a[yout], d)     # d arrives wrt the pooled tensor
        last = name == 'coarse/conv/conv2d_0'
        d, g[name + '/kernel'], g[name + '/bias'] = _conv_bwd(p, name, a[xin], a[yout], d, need_dx=not last)
    return g


def backward_fine(p, a):
    """Gradients of loss_fine wrt fine/* (src/models.py:333-338); nothing flows into coarse/*."""
    g = {}
    dfine = T.silog_loss_bwd(a['fine'], a['depths'])
    d, g['fine/third/kernel'], g['fine/third/bias'] = _conv_bwd(p, 'fine/third', a['f2'], a['fine'], dfine)
    d, g['fine/second/conv2d/kernel'], g['fine/second/conv2d/bias'] = \
        _conv_bwd(p, 'fine/second/conv2d', a['cat'], a['f2'], d)
    d = T.maxpool2x2_bwd(a['f1'], d[..., :63])
    _, g['fine/first/conv2d/kernel'], g['fine/first/conv2d/bias'] = \
        _conv_bwd(p, 'fine/first/conv2d', a['images'], a['f1'], d, need_dx=False)
    return g


class Trainer:
    """State of one replica: params, global_step, the four Adam optimizers of src/models.py:318-338."""

    def __init__(self, params, batchsize, global_step=0, beta2=1.0):
        dt = next(iter(params.values())).dtype
        self.p = params
        self.batchsize = batchsize
        self.global_step = global_step
        mk = lambda lr: T.AdamTF1(lr, 0.9, beta2, 1e-8, dt)   # AdamOptimizer(rate, momentum, 1)
        self.opt = {'CoarseConv': mk(0.001), 'CoarseDense': mk(0.1), 'FineA': mk(0.001), 'FineB': mk(0.01)}

    def step(self, images, depths, keep_mask):
        """One session.run(train_op): both forwards + both losses always run; gradients and Adam only in
        the active phase; global_step += 1 in every phase (src/models.py:329,343,356)."""
        a = forward(self.p, images, depths, keep_mask)
        phase = phase_of(self.global_step, self.batchsize)
        grads = {}
        if phase == 1:
            grads = backward_coarse(self.p, a)
            self._apply('CoarseConv', COARSE_CONV_VARS, grads)
            self._apply('CoarseDense', COARSE_DENSE_VARS, grads)
        elif phase == 2:
            grads = backward_fine(self.p, a)
            self._apply('FineA', FINE_A_VARS, grads)
            self._apply('FineB', FINE_B_VARS, grads)
        self.global_step += 1
        return a, grads, phase

    def _apply(self, opt, layers, grads):
        names = [l + s for l in layers for s in ('/kernel', '/bias')]
        self.opt[opt].apply(self.p, {n: grads[n] for n in names})
