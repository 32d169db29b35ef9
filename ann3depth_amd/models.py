"""Model plugins — the surface of the reference's ``src/models.py``: module attributes ``msdn`` and ``dcnf`` that
the driver looks up with ``getattr(models, args.model)`` (src/ann3depth.py:143) and calls as
``model(inputs, targets)`` (src/ann3depth.py:145, src/models.py:179,277) to obtain one train op, executed once per
step by the driver loop (src/ann3depth.py:126-127).

Underneath there is no graph and no autograd: a replica owns flat HBM buffers (weights / gradients / Adam slots per
optimizer group, activations) and ``step()`` enqueues the fixed sequence of HIP kernels of forward, backward and
optimizer on the current stream.  Variable names are the reference's TF variable names.
"""
import collections
import math

import numpy as np
import torch

from . import ops

NET_H, NET_W = 228, 304          # src/models.py:282
OUT_H, OUT_W = 55, 74            # src/models.py:283
SAMPLES_COARSE = 2000000         # src/models.py:302
SAMPLES_FINE = 1500000           # src/models.py:303

ConvSpec = collections.namedtuple('ConvSpec', 'name cin cout k stride padding relu')

MSDN_CONVS = [  # src/models.py:211-223,241-251
    ConvSpec('coarse/conv/conv2d_0', 3, 96, 11, 4, 'VALID', True),
    ConvSpec('coarse/conv/conv2d_1', 96, 256, 5, 1, 'SAME', True),
    ConvSpec('coarse/conv/conv2d_2', 256, 384, 3, 1, 'SAME', True),
    ConvSpec('coarse/conv/conv2d_3', 384, 384, 3, 1, 'SAME', True),
    ConvSpec('coarse/conv/conv2d_4', 384, 256, 3, 2, 'VALID', True),
    ConvSpec('fine/first/conv2d', 3, 63, 9, 2, 'VALID', True),
    ConvSpec('fine/second/conv2d', 64, 64, 5, 1, 'SAME', True),
    ConvSpec('fine/third', 64, 1, 5, 1, 'SAME', False),
]
MSDN_DENSES = [('coarse/dense/dense_0', 12288, 4096), ('coarse/dense/dense_1', 4096, OUT_H * OUT_W)]

# optimizer name -> (learning rate, variable scopes)   src/models.py:318-338
MSDN_OPTIMIZERS = collections.OrderedDict([
    ('CoarseConv', (0.001, ['coarse/conv'])),
    ('CoarseDense', (0.1, ['coarse/dense'])),
    ('FineA', (0.001, ['fine/first', 'fine/third'])),
    ('FineB', (0.01, ['fine/second'])),
])


def glorot_uniform(rng, shape):
    """tf.layers default kernel_initializer (glorot_uniform): limit = sqrt(6/(fan_in+fan_out)), fans include k*k."""
    if len(shape) == 2:
        fan_in, fan_out = shape
    else:
        rf = int(np.prod(shape[:-2]))
        fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    limit = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-limit, limit, size=shape).astype(np.float32)


def phase_of(global_step, batchsize):
    """src/models.py:301-305,348-365: 1 = coarse, 2 = fine, 3 = only global_step += 1."""
    steps_coarse = SAMPLES_COARSE // batchsize
    steps_fine = SAMPLES_FINE // batchsize
    if global_step < steps_coarse:
        return 1
    if global_step < steps_coarse + steps_fine:
        return 2
    return 3


class ParamGroup:
    """All variables of one tf.train.AdamOptimizer instance in ONE flat buffer (plus gradient and the m / v slots),
    so the optimizer is a single streaming kernel and the data-parallel all-reduce a single bucket."""
    ALIGN = 64   # elements; keeps every tensor 256-byte aligned for 16-byte vector loads

    def __init__(self, name, lr, shapes, device, beta1=0.9, beta2=1.0, eps=1e-8):
        self.name, self.lr, self.beta1, self.beta2, self.eps = name, lr, beta1, beta2, eps
        self.offsets = collections.OrderedDict()
        off = 0
        for n, shp in shapes.items():
            self.offsets[n] = (off, tuple(shp))
            off += -(-int(np.prod(shp)) // self.ALIGN) * self.ALIGN
        self.count = off
        self.var = torch.zeros(off, device=device)
        self.grad = torch.zeros(off, device=device)
        self.m = torch.zeros(off, device=device)
        self.v = torch.zeros(off, device=device)
        # beta powers are host scalars, updated after each apply like AdamOptimizer._finish
        self.beta1_power = np.float32(beta1)
        self.beta2_power = np.float32(beta2)

    def view(self, buf, name):
        off, shp = self.offsets[name]
        return buf[off:off + int(np.prod(shp))].view(shp)

    def apply(self, grad_scale=1.0):
        ops.adam_apply_tf1(self.var, self.m, self.v, self.grad, self.lr, self.beta1, self.beta2, self.eps,
                           float(self.beta1_power), float(self.beta2_power), grad_scale)
        self.beta1_power = self.beta1_power * np.float32(self.beta1)
        self.beta2_power = self.beta2_power * np.float32(self.beta2)


class MSDNReplica:
    """One data-parallel replica of the MSDN training graph (src/models.py:203-367) on one GPU."""

    def __init__(self, batchsize, device='cuda', params=None, seed=3000, global_step=0, beta2=1.0, reducer=None):
        self.B = B = batchsize
        self.device = torch.device(device)
        self.global_step = global_step
        self.reducer = reducer
        dev = self.device
        shapes = collections.OrderedDict()
        for c in MSDN_CONVS:
            shapes[c.name + '/kernel'] = (c.k, c.k, c.cin, c.cout)
            shapes[c.name + '/bias'] = (c.cout,)
        for n, i, o in MSDN_DENSES:
            shapes[n + '/kernel'] = (i, o)
            shapes[n + '/bias'] = (o,)
        self.shapes = shapes
        self.groups = collections.OrderedDict()
        self.group_of = {}
        for gname, (lr, scopes) in MSDN_OPTIMIZERS.items():
            gshapes = collections.OrderedDict((n, s) for n, s in shapes.items()
                                              if any(n.startswith(sc + '/') for sc in scopes))
            self.groups[gname] = ParamGroup(gname, lr, gshapes, dev, beta2=beta2)
            for n in gshapes:
                self.group_of[n] = gname
        if params is None:
            rng = np.random.default_rng(seed)
            params = {n: (glorot_uniform(rng, s) if n.endswith('/kernel') else np.zeros(s, np.float32))
                      for n, s in shapes.items()}
        self.load_params(params)

        def buf(*shape):
            return torch.empty(shape, device=dev)
        self.conv = {c.name: c for c in MSDN_CONVS}
        # activations
        self.x = buf(B, NET_H, NET_W, 3)
        self.t = buf(B, OUT_H, OUT_W, 1)
        self.c0 = buf(B, 55, 74, 96); self.p0 = buf(B, 27, 37, 96)
        self.c1 = buf(B, 27, 37, 256); self.p1 = buf(B, 13, 18, 256)
        self.c2 = buf(B, 13, 18, 384); self.c3 = buf(B, 13, 18, 384); self.c4 = buf(B, 6, 8, 256)
        self.drop = buf(B, 4096)
        self.coarse = buf(B, OUT_H, OUT_W, 1)
        self.f1 = buf(B, 110, 148, 63)
        self.cat = buf(B, OUT_H, OUT_W, 64)
        self.f2 = buf(B, OUT_H, OUT_W, 64)
        self.fine = buf(B, OUT_H, OUT_W, 1)
        self.loss_coarse = buf(1); self.loss_fine = buf(1)
        self.ws_c = buf(2 * B); self.ws_f = buf(2 * B)
        # gradients wrt pre-activations
        self.dz1 = buf(B, OUT_H * OUT_W); self.dz0 = buf(B, 4096)
        self.dc4 = buf(B, 6, 8, 256); self.dc3 = buf(B, 13, 18, 384); self.dc2 = buf(B, 13, 18, 384)
        self.dp1 = buf(B, 13, 18, 256); self.dc1 = buf(B, 27, 37, 256)
        self.dp0 = buf(B, 27, 37, 96); self.dc0 = buf(B, 55, 74, 96)
        self.dfine = buf(B, OUT_H, OUT_W, 1); self.df2 = buf(B, OUT_H, OUT_W, 64)
        self.dcat = buf(B, OUT_H, OUT_W, 64); self.df1 = buf(B, 110, 148, 63)
        # descriptors
        D = ops.conv_desc
        self.d = {
            'coarse/conv/conv2d_0': D(B, NET_H, NET_W, 3, 96, 11, 11, 4, 'VALID'),
            'coarse/conv/conv2d_1': D(B, 27, 37, 96, 256, 5, 5, 1, 'SAME'),
            'coarse/conv/conv2d_2': D(B, 13, 18, 256, 384, 3, 3, 1, 'SAME'),
            'coarse/conv/conv2d_3': D(B, 13, 18, 384, 384, 3, 3, 1, 'SAME'),
            'coarse/conv/conv2d_4': D(B, 13, 18, 384, 256, 3, 3, 2, 'VALID'),
            'fine/first/conv2d': D(B, NET_H, NET_W, 3, 63, 9, 9, 2, 'VALID'),
            'fine/second/conv2d': D(B, OUT_H, OUT_W, 64, 64, 5, 5, 1, 'SAME'),
            'fine/third': D(B, OUT_H, OUT_W, 64, 1, 5, 5, 1, 'SAME'),
        }

    # ---- variables ----
    def load_params(self, params):
        for n, shp in self.shapes.items():
            g = self.groups[self.group_of[n]]
            a = np.asarray(params[n], np.float32)
            assert a.shape == tuple(shp), (n, a.shape, shp)
            g.view(g.var, n).copy_(torch.from_numpy(np.ascontiguousarray(a)))

    def var(self, name):
        g = self.groups[self.group_of[name]]
        return g.view(g.var, name)

    def grad(self, name):
        g = self.groups[self.group_of[name]]
        return g.view(g.grad, name)

    def slot(self, name, which):
        g = self.groups[self.group_of[name]]
        return g.view(g.m if which == 'm' else g.v, name)

    def state_dict(self):
        """name -> tensor for every variable, Adam slot (TF slot naming '<var>/<Optimizer>[_1]') and global_step."""
        sd = collections.OrderedDict()
        for n in self.shapes:
            sd[n] = self.var(n)
            sd[n + '/' + self.group_of[n]] = self.slot(n, 'm')
            sd[n + '/' + self.group_of[n] + '_1'] = self.slot(n, 'v')
        for gname, g in self.groups.items():
            sd[gname + '/beta1_power'] = torch.tensor(float(g.beta1_power))
            sd[gname + '/beta2_power'] = torch.tensor(float(g.beta2_power))
        sd['global_step'] = torch.tensor(self.global_step, dtype=torch.int64)
        return sd

    def load_state_dict(self, sd):
        for n in self.shapes:
            self.var(n).copy_(sd[n])
            self.slot(n, 'm').copy_(sd[n + '/' + self.group_of[n]])
            self.slot(n, 'v').copy_(sd[n + '/' + self.group_of[n] + '_1'])
        for gname, g in self.groups.items():
            g.beta1_power = np.float32(sd[gname + '/beta1_power'].item())
            g.beta2_power = np.float32(sd[gname + '/beta2_power'].item())
        self.global_step = int(sd['global_step'].item())

    def _kb(self, name):
        return self.var(name + '/kernel'), self.var(name + '/bias')

    def _conv(self, name, x, y):
        w, b = self._kb(name)
        ops.conv2d_fwd(self.d[name], x, w, b, y, 'relu' if self.conv[name].relu else None)

    # ---- forward: src/models.py:277-290 ----
    def forward(self, images, depths, keep_mask):
        ops.resize_bilinear_tf1(images, self.x)
        ops.resize_bilinear_tf1(depths, self.t)
        B = self.B
        self._conv('coarse/conv/conv2d_0', self.x, self.c0)
        ops.maxpool2x2_fwd(self.c0, self.p0)
        self._conv('coarse/conv/conv2d_1', self.p0, self.c1)
        ops.maxpool2x2_fwd(self.c1, self.p1)
        self._conv('coarse/conv/conv2d_2', self.p1, self.c2)
        self._conv('coarse/conv/conv2d_3', self.c2, self.c3)
        self._conv('coarse/conv/conv2d_4', self.c3, self.c4)
        w, b = self._kb('coarse/dense/dense_0')
        ops.dense_fwd(self.c4.view(B, -1), w, b, self.drop, 'relu', drop_keep=keep_mask)     # relu + dropout fused
        w, b = self._kb('coarse/dense/dense_1')
        ops.dense_fwd(self.drop, w, b, self.coarse.view(B, -1))
        self._conv('fine/first/conv2d', self.x, self.f1)
        ops.maxpool2x2_fwd(self.f1, self.cat, extra=self.coarse)                              # pool + concat fused
        self._conv('fine/second/conv2d', self.cat, self.f2)
        self._conv('fine/third', self.f2, self.fine)
        ops.silog_loss_fwd(self.coarse, self.t, self.loss_coarse, self.ws_c)
        ops.silog_loss_fwd(self.fine, self.t, self.loss_fine, self.ws_f)

    # ---- backward of loss_coarse wrt coarse/* : src/models.py:318-324 ----
    def backward_coarse(self, after_dense=None):
        B = self.B
        G = self.grad
        ops.silog_loss_bwd(self.coarse, self.t, self.ws_c, self.dz1.view(B, OUT_H, OUT_W, 1))
        n = 'coarse/dense/dense_1'
        ops.dense_bwd_filter(self.drop, self.dz1, G(n + '/kernel'), G(n + '/bias'))
        # dropout-grad (x2 on kept units) and dense_0's ReluGrad in one mask: drop > 0  <=>  kept and relu active
        ops.dense_bwd_data(self.dz1, self.var(n + '/kernel'), self.dz0, mask=self.drop, scale=2.0)
        n = 'coarse/dense/dense_0'
        flat = self.c4.view(B, -1)
        ops.dense_bwd_filter(flat, self.dz0, G(n + '/kernel'), G(n + '/bias'))
        ops.dense_bwd_data(self.dz0, self.var(n + '/kernel'), self.dc4.view(B, -1), mask=flat, scale=1.0)
        if after_dense is not None:
            after_dense()          # dense gradients are complete: their all-reduce can overlap the conv backward
        n = 'coarse/conv/conv2d_4'
        ops.conv2d_bwd_filter(self.d[n], self.c3, self.dc4, G(n + '/kernel'), G(n + '/bias'))
        ops.conv2d_bwd_data(self.d[n], self.dc4, self.var(n + '/kernel'), self.dc3, relu_mask=self.c3)
        n = 'coarse/conv/conv2d_3'
        ops.conv2d_bwd_filter(self.d[n], self.c2, self.dc3, G(n + '/kernel'), G(n + '/bias'))
        ops.conv2d_bwd_data(self.d[n], self.dc3, self.var(n + '/kernel'), self.dc2, relu_mask=self.c2)
        n = 'coarse/conv/conv2d_2'
        ops.conv2d_bwd_filter(self.d[n], self.p1, self.dc2, G(n + '/kernel'), G(n + '/bias'))
        ops.conv2d_bwd_data(self.d[n], self.dc2, self.var(n + '/kernel'), self.dp1)
        ops.maxpool2x2_bwd(self.c1, self.dp1, self.dc1, relu_mask=True)
        n = 'coarse/conv/conv2d_1'
        ops.conv2d_bwd_filter(self.d[n], self.p0, self.dc1, G(n + '/kernel'), G(n + '/bias'))
        ops.conv2d_bwd_data(self.d[n], self.dc1, self.var(n + '/kernel'), self.dp0)
        ops.maxpool2x2_bwd(self.c0, self.dp0, self.dc0, relu_mask=True)
        n = 'coarse/conv/conv2d_0'
        ops.conv2d_bwd_filter(self.d[n], self.x, self.dc0, G(n + '/kernel'), G(n + '/bias'))

    # ---- backward of loss_fine wrt fine/* : src/models.py:333-338 ----
    def backward_fine(self):
        G = self.grad
        ops.silog_loss_bwd(self.fine, self.t, self.ws_f, self.dfine)
        n = 'fine/third'
        ops.conv2d_bwd_filter(self.d[n], self.f2, self.dfine, G(n + '/kernel'), G(n + '/bias'))
        ops.conv2d_bwd_data(self.d[n], self.dfine, self.var(n + '/kernel'), self.df2, relu_mask=self.f2)
        n = 'fine/second/conv2d'
        ops.conv2d_bwd_filter(self.d[n], self.cat, self.df2, G(n + '/kernel'), G(n + '/bias'))
        ops.conv2d_bwd_data(self.d[n], self.df2, self.var(n + '/kernel'), self.dcat)
        ops.maxpool2x2_bwd(self.f1, self.dcat, self.df1, relu_mask=True)      # reads channels 0..62 of dcat
        n = 'fine/first/conv2d'
        ops.conv2d_bwd_filter(self.d[n], self.x, self.df1, G(n + '/kernel'), G(n + '/bias'))

    # ---- one session.run(train_op) ----
    def step(self, images, depths, keep_mask):
        """images [B,H,W,3], depths [B,H',W',1] float32, keep_mask [B,4096] bool/uint8, all on this device.
        Both forwards and both losses run in every phase; gradients + Adam only for the active phase;
        global_step += 1 always (src/models.py:329,343,356)."""
        if keep_mask.dtype != torch.uint8:
            keep_mask = keep_mask.to(torch.uint8)
        self.forward(images, depths, keep_mask)
        phase = phase_of(self.global_step, self.B)
        red = self.reducer
        scale = 1.0 / red.world_size if red is not None else 1.0
        if phase == 1:
            gc, gd = self.groups['CoarseConv'], self.groups['CoarseDense']
            self.backward_coarse(after_dense=(lambda: red.start(gd.grad)) if red is not None else None)
            if red is not None:
                red.start(gc.grad)
                red.finish()
            gc.apply(scale)
            gd.apply(scale)
        elif phase == 2:
            ga, gb = self.groups['FineA'], self.groups['FineB']
            self.backward_fine()
            if red is not None:
                red.start(ga.grad)
                red.start(gb.grad)
                red.finish()
            ga.apply(scale)
            gb.apply(scale)
        self.global_step += 1
        return {'coarse_loss': self.loss_coarse, 'fine_loss': self.loss_fine, 'phase': phase}
