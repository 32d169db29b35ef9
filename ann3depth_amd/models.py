"""Model plugins — the surface of the reference's ``src/models.py``: module attributes ``msdn`` and ``dcnf`` that
the driver looks up with ``getattr(models, args.model)`` (src/ann3depth.py:143) and calls as
``model(inputs, targets)`` (src/ann3depth.py:145, src/models.py:179,277) to obtain one train op, executed once per
step by the driver loop (src/ann3depth.py:126-127).

Underneath there is no graph and no autograd: a replica owns flat HBM buffers (weights / gradients / Adam slots per
optimizer group, activations) and ``step()`` enqueues the fixed sequence of HIP kernels of forward, backward and
optimizer on the current stream.  Variable names are the reference's TF variable names.
"""
import collections
import contextlib
import ctypes
import math
import os

import numpy as np
import torch

from . import _lib, ops

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

    def __init__(self, name, lr, shapes, device, beta1=0.9, beta2=1.0, eps=1e-8, slots=True, multiple=1):
        """multiple: the flat length is padded (with zeros that stay zeros) to a multiple of multiple * ALIGN elements, so
        that `multiple` data-parallel ranks can cut any bucket of it into equal, 256-byte-aligned slices."""
        self.name, self.lr, self.beta1, self.beta2, self.eps = name, lr, beta1, beta2, eps
        self.offsets = collections.OrderedDict()
        off = 0
        for n, shp in shapes.items():
            self.offsets[n] = (off, tuple(shp))
            off += -(-int(np.prod(shp)) // self.ALIGN) * self.ALIGN
        off = -(-off // (multiple * self.ALIGN)) * (multiple * self.ALIGN)
        self.count = off
        self.var = torch.zeros(off, device=device)
        self.grad = torch.zeros(off, device=device)
        self.m = torch.zeros(off, device=device) if slots else None      # slots=False: plain gradient descent
        self.v = torch.zeros(off, device=device) if slots else None
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

    def apply_slice(self, a, b, grad_scale=1.0, poisoned=None):
        """ApplyAdam on elements [a, b) only (a data-parallel rank's slice of a reduce-scattered bucket); the caller
        advances the beta powers once per step with advance()."""
        ops.adam_apply_tf1(self.var[a:b], self.m[a:b], self.v[a:b], self.grad[a:b], self.lr, self.beta1, self.beta2,
                           self.eps, float(self.beta1_power), float(self.beta2_power), grad_scale, poisoned=poisoned)

    def frozen(self):
        """True for the optimizer the reference builds, AdamOptimizer(rate, 0.9, beta2 = 1): alpha = 0, only m moves."""
        return self.beta2 == 1.0 and float(self.beta2_power) == 1.0

    def advance(self):
        """The beta-power bookkeeping of one ApplyAdam whose tensor work was fused elsewhere (dense_bwd_filter_adam_tf1)."""
        self.beta1_power = self.beta1_power * np.float32(self.beta1)
        self.beta2_power = self.beta2_power * np.float32(self.beta2)

    def apply_sgd(self, grad_scale=1.0):
        """tf.train.GradientDescentOptimizer(lr): var -= lr * grad_scale * g."""
        ops.sgd_apply(self.var, self.grad, self.lr * grad_scale)


class MSDNReplica:
    """One data-parallel replica of the MSDN training graph (src/models.py:203-367) on one GPU."""

    def __init__(self, batchsize, device='cuda', params=None, seed=3000, global_step=0, beta2=1.0, reducer=None,
                 precision='fp32', keep_dense_grads=True):
        """precision: arithmetic of the conv contractions — 'fp32' (exact, the parity default), 'bf16x3' (split
        operands on the bf16 matrix cores, ~1e-5) or 'bf16' (BASELINE config 5).  Tensors stay float32 in HBM; the
        Cin = 3 layers, the dense layers and everything element-wise always compute in fp32."""
        # keep_dense_grads=False (the training driver and bench.py on one GPU): under the reference's frozen optimizer the
        # gradient of the two dense kernels (268 MB) goes straight from the matrix cores into ApplyAdam's m slot
        # (ops.dense_bwd_filter_adam_tf1) and is never written; grad('coarse/dense/...') is then stale.  A data-parallel
        # replica needs the gradient for its all-reduce and always keeps it.
        self.keep_dense_grads = keep_dense_grads or reducer is not None or batchsize > 64
        # 'bf16s' = BASELINE config 5 proper: bf16 arithmetic AND bf16 storage — the conv stacks' activations and their
        # gradients, and a bf16 copy of every kernel with at least 8 input channels, live in HBM as bf16; fp32 stay the
        # master weights, the Adam slots, all filter / bias gradients and split-K slabs, the resized input, the 3-channel
        # layers' filters, fine/second's output (fine/third is a single-output-channel stencil on fp32), and the
        # tensors of the dense layers' small side (x, y, dz, dx: a few MB; dense_0's 201 MB of weights are read as bf16).
        self.bf16s = precision == 'bf16s'
        self.fine_first_bf16 = self.conv0_image = self.fuse_casts = False
        self._c4_32_fresh = False
        if self.bf16s:
            precision = 'bf16'
        self.precision = precision
        self.B = B = batchsize
        self.device = torch.device(device)
        self.global_step = global_step
        self.reducer = reducer
        dev = self.device
        # Second HIP stream (A3D_OVERLAP=0 turns it off): the fine network's forward — MFMA-bound, and in the coarse phase
        # needed only for its loss — runs beside the stretch of the coarse chain that is HBM-bound (dense forward, loss,
        # dense backward + ApplyAdam, conv2d_4 backward) and is joined before the large conv backward GEMMs start.
        # Measured on one MI355X at B = 32 (round 2): coarse-phase step 3.42 -> 3.32 ms.  The two kinds of kernel share
        # CUs only as far as the register file allows (two 8-wave GEMM blocks fill a SIMD's 512 registers), so the gain
        # is a quarter of the stretch, not all of it.  The dominant bwd-filter GEMMs run after the join, alone.
        self.overlap = os.environ.get('A3D_OVERLAP', '1') == '1'
        # A3D_SHARE_CU: which fine-network GEMMs are launched with A3D_HINT_SHARE_CU (at most two wavefronts per SIMD, so
        # that the dense layers' weight-streaming kernels find room on every CU): 1 = both (default: 2.939 ms per step,
        # alternating runs), 3 = fine/second only (2.947), 2 = fine/first only (2.98), 0 = none (2.985).  (fine/first's
        # few-channel kernel keeps three chunks in flight per wave under the hint; with its stand-alone schedule it lost.)
        # Starting fine/first earlier (beside the coarse conv stack) or later (after the dense forward) both lose: an
        # MFMA-bound grid beside another MFMA-bound grid only takes CU slots from it.
        share = int(os.environ.get('A3D_SHARE_CU', '1')) if (self.overlap and dev.type == 'cuda') else 0
        self._share_names = {0: (), 1: ('fine/first/conv2d', 'fine/second/conv2d'), 2: ('fine/first/conv2d',),
                             3: ('fine/second/conv2d',)}[share]
        self._shared_desc = {}
        # the few-channel forwards' filters in the layout their kernels read, repacked when the weights change instead of on
        # every call (a3d_conv2d_fwd_prepare_filter; A3D_PREPARED_FILTERS=0: per call, as before round 5)
        self.prepare_filters = os.environ.get('A3D_PREPARED_FILTERS', '1') != '0' and self.device.type == 'cuda'
        self._prep = {}
        self._alone = ()          # fine-network GEMMs of the current forward that nothing runs beside (fine phase: fine/second)
        self._deferred = None     # (all-reduce handle, group, grad scale): CoarseDense bucket still in flight, see step()
        # data-parallel replicas under the reference's frozen optimizer: the dense bucket is reduce-scattered and each
        # rank keeps the Adam `m` slot of its own slices only (dp.py); gather_state() reassembles it
        self._m_sharded = False
        # A3D_DP_DENSE=allreduce forces the exchange without aliasing buffers (all-reduce + replicated ApplyAdam, what the
        # flagged beta2 < 1 mode always uses); the default, reduce-scatter, is taken only if the backend passes the
        # reducer's own check of the in-place collectives (dp.GradReducer.inplace_ok: RCCL has not run them on this pipeline)
        self.dense_exchange = None
        if reducer is not None:
            want = os.environ.get('A3D_DP_DENSE', 'scatter')
            if want not in ('scatter', 'allreduce'):
                raise ValueError(f'A3D_DP_DENSE={want!r}: scatter or allreduce')
            self.dense_exchange = 'reduce_scatter' if (want == 'scatter' and reducer.inplace_ok(dev)) else 'all_reduce'
            if want == 'scatter' and self.dense_exchange == 'all_reduce':
                print('MSDNReplica: in-place reduce-scatter failed its self-check on this backend; the dense bucket is '
                      'all-reduced instead', flush=True)
        self._dense_pieces = None
        self._poison = None       # device flag: a non-finite gradient reached one of this rank's slices
        self._poison_seen = []    # [(event, pinned host copy of the all-rank MAX of the flag)] of earlier steps
        # conv + ReLU + max pool in one kernel inside step(): the pre-pool activations c0, c1, f1 are never written
        # (the network being trained keeps one byte per pool window instead, see forward())
        self.fuse_pool = precision == 'fp32' and os.environ.get('A3D_NO_FUSED_POOL', '0') != '1'
        # bf16 storage: conv2d_1's pool in the LDS-DMA kernel's epilogue (argmax bytes + a bf16 MaxPoolGrad by index)
        self.pool1_fused = self.bf16s and os.environ.get('A3D_BF16S_POOL1', '1') != '0'
        self.pool0_fused = self.bf16s and os.environ.get('A3D_BF16S_POOL0', '1') != '0'
        self.side = None
        if self.overlap and dev.type == 'cuda':
            # below the main stream's queue priority: a CU slot that frees up goes to the HBM-bound kernel first
            handle = ctypes.c_void_p()
            with torch.cuda.device(dev):
                ops.check(_lib.load().a3d_stream_create(int(os.environ.get('A3D_SIDE_PRIORITY', '1')), ctypes.byref(handle)),
                          'a3d_stream_create')
            self.side = torch.cuda.ExternalStream(handle.value, device=dev)
            self._side_handle = handle.value       # ExternalStream does not own it: returned in __del__
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
            world = reducer.world_size if (reducer is not None and gname == 'CoarseDense') else 1
            self.groups[gname] = ParamGroup(gname, lr, gshapes, dev, beta2=beta2, multiple=world)
            for n in gshapes:
                self.group_of[n] = gname
        if params is None:
            rng = np.random.default_rng(seed)
            params = {n: (glorot_uniform(rng, s) if n.endswith('/kernel') else np.zeros(s, np.float32))
                      for n, s in shapes.items()}
        self.load_params(params)

        def buf(*shape):
            return torch.empty(shape, device=dev)

        def abuf(*shape):           # an activation (or activation gradient) of the conv stacks: bf16 under 'bf16s'
            return torch.empty(shape, device=dev, dtype=torch.bfloat16 if self.bf16s else torch.float32)
        self.conv = {c.name: c for c in MSDN_CONVS}
        # activations
        self.x = buf(B, NET_H, NET_W, 3)
        self.t = buf(B, OUT_H, OUT_W, 1)
        # c0 / c1 exist only where a forward writes them: never under 'bf16s' with its pools in the conv epilogues (83 MB at B = 64)
        self.c0 = None if self.pool0_fused else abuf(B, 55, 74, 96)
        self.c1 = None if self.pool1_fused else abuf(B, 27, 37, 256)
        self.p0 = abuf(B, 27, 37, 96); self.p1 = abuf(B, 13, 18, 256)
        self.c2 = abuf(B, 13, 18, 384); self.c3 = abuf(B, 13, 18, 384); self.c4 = abuf(B, 6, 8, 256)
        self.drop = buf(B, 4096)
        self.coarse = buf(B, OUT_H, OUT_W, 1)
        self.f1 = buf(B, 110, 148, 63) if not self.bf16s else None        # 'bf16s': never written (fused with its pool)
        self.cat = abuf(B, OUT_H, OUT_W, 64)
        self.f2 = buf(B, OUT_H, OUT_W, 64)
        self.fine = buf(B, OUT_H, OUT_W, 1)
        self.loss_coarse = buf(1); self.loss_fine = buf(1)
        # window positions of the pool maxima (all MaxPoolGrad needs besides the pooled values, see forward())
        self.a0 = torch.empty((B, 27, 37, 96), dtype=torch.uint8, device=dev)
        self.a1 = torch.empty((B, 13, 18, 256), dtype=torch.uint8, device=dev)
        self.af1 = torch.empty((B, OUT_H, OUT_W, 63), dtype=torch.uint8, device=dev)
        self.pooled_fwd = None            # which network ran conv + pool fused in the last forward
        self.ws_c = ops.silog_ws(B, dev); self.ws_f = ops.silog_ws(B, dev)   # per-sample sums + arrival ticket + partials
        # gradients wrt pre-activations
        self.dz1 = buf(B, OUT_H * OUT_W); self.dz0 = buf(B, 4096)
        self.dc4 = abuf(B, 6, 8, 256); self.dc3 = abuf(B, 13, 18, 384); self.dc2 = abuf(B, 13, 18, 384)
        self.dp1 = abuf(B, 13, 18, 256); self.dc1 = abuf(B, 27, 37, 256)
        self.dp0 = abuf(B, 27, 37, 96); self.dc0 = None            # dc0 (50 MB at B = 32): allocated only if the unfused path runs
        # fine/third's backward writes df2 in the type fine/second's backward reads it: bf16 under 'bf16s' (round 5)
        self.dfine = buf(B, OUT_H, OUT_W, 1); self.df2 = abuf(B, OUT_H, OUT_W, 64)
        self.dcat = abuf(B, OUT_H, OUT_W, 64); self.df1 = None      # df1 (131 MB at B = 32): allocated only if the unfused path runs
        if self.bf16s:
            self.c4_32 = buf(B, 6, 8, 256)                                          # the fp32 copy of c4 dense_0's filter gradient reads
            self.dz0_16 = torch.empty((B, 4096), device=dev, dtype=torch.bfloat16)  # dense_0's bwd-data takes dz0 as bf16
            # dense_1 ([4096, 4070]: rows of 8140 bytes, not whole 16-byte pieces) on the LDS-DMA kernel too: a bf16 copy of
            # its kernel with rows of 4072 elements (the two pad columns zero, the bias padded alike); x (= drop), dz and y cross
            # to that layout and back by a3d_cast_rows (0.5 - 1 MB each).  Its master weights, gradient and ApplyAdam stay fp32.
            # bf16 x / dz into the dense layers: the LDS-DMA kernel takes them as a weight stream of at most 64 rows (ring_plan,
            # igemm_host.hip); a larger batch keeps the dense layers' small side fp32 (B = 65..383 had no kernel)
            self.dense_bf16_x = B <= 64
            # round 5: the casts between the bf16 conv stack and the fp32 dense side leave through the reductions that produce
            # their sources (a3d_second_output) instead of five 5-us launches; A3D_BF16S_FUSE_CASTS=0: separate launches
            self.fuse_casts = os.environ.get('A3D_BF16S_FUSE_CASTS', '1') != '0'
            self.dense1_bf16 = os.environ.get('A3D_BF16S_DENSE1', '1') != '0' and self.dense_bf16_x
            if not self.dense_bf16_x:
                self.dc4_32 = buf(B, 6, 8, 256)
            NP = (OUT_H * OUT_W + 7) // 8 * 8
            self.w1pad = torch.zeros((4096, NP), device=dev, dtype=torch.bfloat16)
            self.b1pad = torch.zeros((1, NP), device=dev)
            self.drop16 = torch.empty((B, 4096), device=dev, dtype=torch.bfloat16)
            self.y1pad = torch.empty((B, NP), device=dev)
            self.dz1_16 = torch.zeros((B, NP), device=dev, dtype=torch.bfloat16)       # (columns 4070.. stay zero)
        # descriptors
        def D(*a):
            return ops.conv_desc(*a, precision=precision)
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
        # conv -> ReLU -> pool blocks on the 3-channel image: filter gradient straight from the POOLED map's gradient (MaxPoolGrad
        # by index + ReluGrad while the kernel stages its operand): dc0 / df1 (50 / 131 MB at B = 32) are never written or read.
        # fp32 arithmetic on the fp32 image in every precision mode.  A3D_FEWCH_POOLED=0: the separate launches.
        self.d_few = {}
        if os.environ.get('A3D_FEWCH_POOLED', '1') != '0' and dev.type == 'cuda':
            # 'bf16s' (config 5): the same launch in bf16 arithmetic (fewch16.hip: fp32 image, bf16 pooled tensors, both operands
            # transposed into LDS); A3D_BF16S_FEWCH16=0: fp32 arithmetic for fine/first, igemm_bf16 + MaxPoolGrad for conv2d_0
            few_prec = 'bf16' if (self.bf16s and os.environ.get('A3D_BF16S_FEWCH16', '1') != '0') else 'fp32'
            for n, dd in (('coarse/conv/conv2d_0', ops.conv_desc(B, NET_H, NET_W, 3, 96, 11, 11, 4, 'VALID', precision=few_prec)),
                          ('fine/first/conv2d', ops.conv_desc(B, NET_H, NET_W, 3, 63, 9, 9, 2, 'VALID', precision=few_prec))):
                if ops.conv2d_bwd_filter_pooled_supported(dd):
                    self.d_few[n] = dd
            if self.bf16s and os.environ.get('A3D_BF16S_FEWCH0', '1') == '0':
                # A3D_BF16S_FEWCH0=0: config 5's conv2d_0 on igemm_bf16 + MaxPoolGrad by index (78 + 5 + 18 us at B = 64; the fused
                # kernel: 68 + 11 since its staging lost its lane masks, 91 + 11 before)
                self.d_few.pop('coarse/conv/conv2d_0', None)
        # bf16 storage: per layer, which tensors of the forward / bwd-data / bwd-filter call are bf16 (ops.STORE_*), and
        # the bf16 copies of the kernels (refreshed whenever the fp32 masters change)
        self.store = {}
        self.wcopy = {}
        if self.bf16s:
            X, W, Y = ops.STORE_X, ops.STORE_W, ops.STORE_Y
            for n in ('coarse/conv/conv2d_1', 'coarse/conv/conv2d_2', 'coarse/conv/conv2d_3', 'coarse/conv/conv2d_4'):
                self.store[n] = {'fwd': X | W | Y, 'bwd_d': X | W | Y, 'bwd_f': X | Y}
            # conv2d_0 takes the fp32 image through the bf16 kernel (window runs of 36 floats, 16-byte loads) and writes c0
            # as bf16; its filter stays fp32 (padded per call).  At B = 64: forward 93 us + pool against 204 us for the fp32
            # conv + pool fusion, bwd-filter 108 us against 221 us.
            n = 'coarse/conv/conv2d_0'
            self.store[n] = {'fwd': Y, 'bwd_d': 0, 'bwd_f': Y}
            # fine/first keeps fp32 arithmetic (its window runs start 24 bytes apart: 8-byte loads only, which the bf16
            # kernel does not take): conv + ReLU + max pool in one kernel, of which only the pooled map (bf16) and an argmax
            # byte per window reach HBM — f1, the largest activation of the model, is never written; its gradient comes
            # back fp32
            for n in ('fine/first/conv2d',):
                self.d[n] = ops.with_storage(self.d[n], 0)
                self.d[n].precision = ops.PREC['fp32']
                self.store[n] = {'fwd': Y, 'bwd_d': 0, 'bwd_f': 0}
            # f2 stays fp32 (fine/third is a single-output-channel stencil on fp32); df2 comes back from that layer's fused
            # backward as bf16, so fine/second's bwd-data and bwd-filter read two bf16 operands (the LDS-DMA kernel)
            self.store['fine/second/conv2d'] = {'fwd': X | W, 'bwd_d': W | X | Y, 'bwd_f': X | Y}
            # ... except where fine/first has no backward (coarse phase, and once nothing trains any more): there its forward
            # runs on the bf16 pipe from a 4-channel bf16 copy of the image (8-byte pixels: window runs 16 bytes apart),
            # 0.41 -> 0.10 ms at B = 64, conv + ReLU + max pool in one launch (f1 is never written there either)
            # round 5: also where it trains (fine phase) — config 5's arithmetic is bf16 throughout, the fp32 few-channel kernel
            # took 0.38 ms of that phase's 1.73 at B = 64.  A3D_BF16S_FINE1=0: the fp32 conv + pool there, as before.
            self.fine_first_bf16 = os.environ.get('A3D_BF16S_FINE1', '1') != '0'
            self.x4 = torch.empty((B, NET_H, NET_W, 4), device=dev, dtype=torch.bfloat16)
            # round 5: conv2d_0's forward from the same 4-channel bf16 image (conv3.hip's bf16 form, straight from L2) instead
            # of the fp32 image through igemm_bf16's window runs.  A3D_BF16S_CONV0_IMAGE=0: as before.
            self.conv0_image = os.environ.get('A3D_BF16S_CONV0_IMAGE', '1') != '0' and self.pool0_fused
            self.w0_4 = torch.zeros((11, 11, 4, 96), device=dev)
            self.d0_4 = ops.with_storage(ops.conv_desc(B, NET_H, NET_W, 4, 96, 11, 11, 4, 'VALID', ldy=96, precision='bf16'),
                                         X | Y)
            self.w4 = torch.zeros((9, 9, 4, 63), device=dev)
            self.d4 = ops.with_storage(ops.conv_desc(B, NET_H, NET_W, 4, 63, 9, 9, 2, 'VALID', ldy=64, precision='bf16'),
                                       X | Y)
            for n in list(self.store) + ['coarse/dense/dense_0']:
                if n == 'coarse/dense/dense_0' or self.store[n]['fwd'] & W:
                    self.wcopy[n] = torch.empty(self.shapes[n + '/kernel'], device=dev, dtype=torch.bfloat16)
            self.refresh_weight_copies()

    def __del__(self):
        handle, self._side_handle = getattr(self, '_side_handle', None), None
        if handle:
            try:
                torch.cuda.synchronize(self.device)
                ops.drop_workspace(handle)
                _lib.load().a3d_stream_destroy(handle)
            except Exception:          # noqa: BLE001 - interpreter shutdown: the driver reclaims the stream
                pass

    def _desc(self, name, which):
        """The layer's conv descriptor for 'fwd' | 'bwd_d' | 'bwd_f', with that call's storage bits."""
        d = self.d[name]
        bits = self.store.get(name, {}).get(which, 0)
        if which == 'fwd' and name in self._share_names and name not in self._alone:
            # a fine-network GEMM on the side stream beside the dense layers' weight streams: leave those room on every CU
            key = (name, bits)
            if key not in self._shared_desc:
                e = ops.with_storage(d, bits)
                e.hints = ops.HINT_SHARE_CU
                self._shared_desc[key] = e
            return self._shared_desc[key]
        return ops.with_storage(d, bits) if bits else d

    def _w(self, name, which='fwd'):
        """The kernel a conv / dense call reads: the bf16 copy where this call takes one, else the fp32 master."""
        if name in self.wcopy and (name not in self.store or self.store[name][which] & ops.STORE_W):
            return self.wcopy[name]
        return self._v(name + '/kernel')

    def _prepared(self, key, desc, w):
        """(descriptor, filter) for a few-channel forward: the prepared copy and the descriptor that says so, where the launch has one."""
        if not self.prepare_filters:
            return desc, w
        pf = self._prep.get(key)
        if pf is None:
            pf = self._prep[key] = ops.PreparedFilter(desc, self.device)
            pf.source = w
            if pf.ok:
                pf.refresh(w)
        return (pf.desc_prepared, pf.buf) if pf.ok else (desc, w)

    def _refresh_prepared(self):
        for pf in self._prep.values():
            if pf.ok:
                pf.refresh(pf.source)

    def refresh_weight_copies(self):
        for n, c in self.wcopy.items():
            ops.cast_bf16(self._v(n + '/kernel'), c)
        if self.bf16s:
            self.w4[:, :, :3, :] = self._v('fine/first/conv2d/kernel')
            self.w0_4[:, :, :3, :] = self._v('coarse/conv/conv2d_0/kernel')
            n = 'coarse/dense/dense_1'
            ops.cast_rows(self._v(n + '/kernel'), self.w1pad)
            ops.cast_rows(self._v(n + '/bias').view(1, -1), self.b1pad)

    def _weights_moved(self, *groups):
        """After an ApplyAdam that can change `var` (any optimizer but the reference's frozen beta2 = 1 one): the bf16
        copies the next forward / bwd-data read must follow the fp32 masters."""
        if not all(g.frozen() for g in groups):
            if self.wcopy:
                self.refresh_weight_copies()
            self._refresh_prepared()

    # ---- variables ----
    def settle(self):
        """Finish what step() left in flight across the step boundary (the CoarseDense all-reduce and its ApplyAdam).
        Every accessor below and the next forward call it; call it yourself before touching `groups[...]` directly."""
        if self._deferred is not None:
            works, group, scale = self._deferred
            self._deferred = None
            if works and isinstance(works[0], tuple):          # reduce-scattered pieces: (handle, first, last) of my slice
                self._poll_poison()
                for work, a, b in works:
                    self.reducer.wait(work)
                    group.apply_slice(a, b, scale, poisoned=self._poison)
                group.advance()
                self._m_sharded = True
                self._watch_poison()
                return
            for work in works:
                self.reducer.wait(work)
            group.apply(scale)
            self._weights_moved(group)

    def _dense_buckets(self):
        """The CoarseDense flat buffer cut for reduce-scatter, in the order the backward completes it: (early, late) where
        `early` = [(c, count)] holds dense_1's gradient only (c = its first element rounded UP to a slice boundary) and goes
        out as soon as dense_1's bwd-filter is done, `late` = [0, c) in three pieces, complete after dense_0's."""
        if self._dense_pieces is None:
            gd = self.groups['CoarseDense']
            q = self.reducer.world_size * ParamGroup.ALIGN
            c = -(-gd.offsets['coarse/dense/dense_1/kernel'][0] // q) * q
            third = -(-(c // 3) // q) * q
            self._dense_pieces = ([(c, gd.count)], [(0, third), (third, 2 * third), (2 * third, c)])
        return self._dense_pieces

    def _my_slice(self, a, b):
        n = (b - a) // self.reducer.world_size
        return a + self.reducer.rank * n, a + (self.reducer.rank + 1) * n

    def _watch_poison(self):
        """The reference's optimizer changes a weight in exactly one case: a non-finite gradient turns it into NaN.  Under
        sharding only the slice's owner sees that; the MAX of the per-rank flags (a 4-byte all-reduce per step) tells
        everybody, read on the host two steps later (no stall), and _resync() then copies the owners' var / v slices."""
        red = self.reducer
        work = red.any(self._poison)
        red.wait(work)
        if not hasattr(self, '_poison_ring'):       # four pinned words, reused (at most two reads are ever outstanding)
            ring = torch.empty(4, dtype=torch.int32)
            self._poison_ring = (ring.pin_memory() if self.device.type == 'cuda' else ring, 0)
        ring, nxt = self._poison_ring
        self._poison_ring = (ring, (nxt + 1) % 4)
        host = ring[nxt:nxt + 1]
        host.copy_(self._poison, non_blocking=True)
        ev = None
        if self.device.type == 'cuda':
            ev = torch.cuda.Event()
            ev.record()
        self._poison_seen.append((ev, host))

    def _poll_poison(self, block=False):
        while self._poison_seen and (block or len(self._poison_seen) >= 2):
            ev, host = self._poison_seen.pop(0)
            if ev is not None:
                ev.synchronize()
            if int(host.item()):
                self._resync()

    def _resync(self):
        """Every rank takes the slice owners' var and v (and m) of the dense group; clears the poison flag."""
        gd = self.groups['CoarseDense']
        early, late = self._dense_buckets()
        for a, b in early + late:
            for buf in (gd.var, gd.v, gd.m):
                self.reducer.all_gather(buf[a:b])
        self._poison.zero_()
        self._poison_seen = []
        self._m_sharded = False

    def gather_state(self):
        """COLLECTIVE under a data-parallel reducer (every rank must call it): reassembles the sharded Adam `m` slot of the
        dense group on all ranks.  The driver calls it before the chief writes a checkpoint; a no-op otherwise."""
        self.settle()
        if self._m_sharded:
            self._poll_poison(block=True)
        if self._m_sharded:
            gd = self.groups['CoarseDense']
            early, late = self._dense_buckets()
            for a, b in early + late:
                self.reducer.all_gather(gd.m[a:b])
            self._m_sharded = False

    def load_params(self, params):
        self.settle()
        for n, shp in self.shapes.items():
            g = self.groups[self.group_of[n]]
            a = np.asarray(params[n], np.float32)
            assert a.shape == tuple(shp), (n, a.shape, shp)
            g.view(g.var, n).copy_(torch.from_numpy(np.ascontiguousarray(a)))
        if getattr(self, 'wcopy', None):
            self.refresh_weight_copies()
        if getattr(self, '_prep', None):
            self._refresh_prepared()

    def _v(self, name):
        g = self.groups[self.group_of[name]]
        return g.view(g.var, name)

    def _g(self, name):
        g = self.groups[self.group_of[name]]
        return g.view(g.grad, name)

    def var(self, name):
        self.settle()
        return self._v(name)

    def grad(self, name):
        self.settle()
        return self._g(name)

    def slot(self, name, which):
        self.gather_state()        # collective under a data-parallel reducer whose dense m slot is sharded
        g = self.groups[self.group_of[name]]
        return g.view(g.m if which == 'm' else g.v, name)

    def state_dict(self):
        """name -> tensor for every variable, Adam slot (TF slot naming '<var>/<Optimizer>[_1]') and global_step.
        Collective under a data-parallel reducer (gather_state)."""
        self.gather_state()
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
        self.gather_state()
        for n in self.shapes:
            self.var(n).copy_(sd[n])
            self.slot(n, 'm').copy_(sd[n + '/' + self.group_of[n]])
            self.slot(n, 'v').copy_(sd[n + '/' + self.group_of[n] + '_1'])
        for gname, g in self.groups.items():
            g.beta1_power = np.float32(sd[gname + '/beta1_power'].item())
            g.beta2_power = np.float32(sd[gname + '/beta2_power'].item())
        self.global_step = int(sd['global_step'].item())
        if self.wcopy:
            self.refresh_weight_copies()
        self._refresh_prepared()

    uses_dropout = True

    def load_tf_variables(self, tensors):
        """Restore from the tensors of a TensorFlow checkpoint (tfckpt.read_bundle): every model variable by its TF
        name; Adam slots `<var>/<Optimizer>` / `<var>/<Optimizer>_1` and `global_step` when present.
        Beta powers: `<Optimizer>/beta{1,2}_power` is THIS build's name for them (tf_variables below).  TensorFlow 1.3
        creates them as plain variables `beta1_power`, `beta1_power_1`, ... under whatever name scope apply_gradients
        ran in (for the reference: inside the tf.cond of src/models.py:340-345), so a reference checkpoint's powers are
        not looked up; they are rebuilt from global_step and the phase schedule instead — exactly, since the power is
        beta multiplied into itself once per applied step in float32 (and beta2 = 1 keeps its power at 1)."""
        missing = [n for n in self.shapes if n not in tensors]
        if missing:
            raise KeyError(f'TensorFlow checkpoint lacks {len(missing)} model variables, e.g. {missing[:3]}')
        self.load_params(tensors)
        for n in self.shapes:
            for which, suffix in (('m', ''), ('v', '_1')):
                key = n + '/' + self.group_of[n] + suffix
                if key in tensors:
                    self.slot(n, which).copy_(torch.from_numpy(np.ascontiguousarray(tensors[key], np.float32)))
        if 'global_step' in tensors:
            self.global_step = int(np.asarray(tensors['global_step']).reshape(-1)[0])
        steps_coarse, steps_fine = SAMPLES_COARSE // self.B, SAMPLES_FINE // self.B
        applied = {'CoarseConv': min(self.global_step, steps_coarse), 'CoarseDense': min(self.global_step, steps_coarse),
                   'FineA': min(max(self.global_step - steps_coarse, 0), steps_fine),
                   'FineB': min(max(self.global_step - steps_coarse, 0), steps_fine)}
        for gname, g in self.groups.items():
            for attr, beta in (('beta1_power', g.beta1), ('beta2_power', g.beta2)):
                key = f'{gname}/{attr}'
                if key in tensors:
                    setattr(g, attr, np.float32(np.asarray(tensors[key]).reshape(-1)[0]))
                else:
                    power = np.float32(beta)
                    for _ in range(min(applied[gname], 4096)):       # beta^(t+1); underflows to its limit long before
                        power = power * np.float32(beta)
                    setattr(g, attr, power)

    def tf_variables(self):
        """name -> ndarray in TensorFlow's naming, for tfckpt.write_bundle."""
        out = {}
        for k, v in self.state_dict().items():
            a = v.detach().cpu().numpy()
            out[k] = a.astype(np.int64) if k == 'global_step' else a.astype(np.float32)
        return out

    def summary_scalars(self, out):
        """Tags as the reference's name scopes 'loss' (src/models.py:288) and 'optimizers' (:347)."""
        return {'loss/coarse_loss': float(out['coarse_loss']), 'loss/fine_loss': float(out['fine_loss']),
                'optimizers/Phase': out['phase']}

    def summary_images(self):
        """src/models.py:292-296: (tag, batch, max_outputs)."""
        return [('summaries/Input', self.x, 3), ('summaries/Coarse', self.coarse, 3), ('summaries/Fine', self.fine, 3),
                ('summaries/Target', self.t, 3)]

    def broadcast_state(self, dist, src=0):
        """Non-chief replicas take the chief's variables, slots, beta powers and global_step."""
        self.gather_state()
        for g in self.groups.values():
            for buf in (g.var, g.m, g.v):
                dist.broadcast(buf, src)
        st = torch.tensor([self.global_step] + [float(x) for g in self.groups.values()
                                                for x in (g.beta1_power, g.beta2_power)],
                          dtype=torch.float64, device=self.device)
        dist.broadcast(st, src)
        self.global_step = int(st[0].item())
        for i, g in enumerate(self.groups.values()):
            g.beta1_power, g.beta2_power = np.float32(st[1 + 2 * i].item()), np.float32(st[2 + 2 * i].item())
        if self.wcopy:
            self.refresh_weight_copies()
        self._refresh_prepared()

    def _kb(self, name):
        return self._v(name + '/kernel'), self._v(name + '/bias')

    FEW_CHANNEL = ('coarse/conv/conv2d_0', 'fine/first/conv2d')

    def _fwd_operands(self, name):
        d, w = self._desc(name, 'fwd'), self._w(name)
        if name in self.FEW_CHANNEL:
            d, w = self._prepared((name, d.storage, d.hints, d.precision), d, w)
        return d, w

    def _conv(self, name, x, y):
        d, w = self._fwd_operands(name)
        ops.conv2d_fwd(d, x, w, self._v(name + '/bias'), y, 'relu' if self.conv[name].relu else None)

    def _pool(self, x, y, extra=None, c=None):
        if self.bf16s:
            ops.maxpool2x2_fwd_bf16(x, y, extra=extra, c=c)
        else:
            ops.maxpool2x2_fwd(x, y, extra=extra)

    def _pool_bwd(self, x, dy, dx, c=None):
        if self.bf16s:
            ops.maxpool2x2_bwd_bf16(x, dy, dx, relu_mask=True, c=c)
        else:
            ops.maxpool2x2_bwd(x, dy, dx, relu_mask=True)

    @contextlib.contextmanager
    def _beside(self):
        """Enqueue the body on the side stream, ordered after everything enqueued on the current stream so far."""
        if self.side is None:
            yield
            return
        self.side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.side):
            yield

    def _join(self):
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)

    # ---- forward: src/models.py:277-290 ----
    def prepool_equivalent(self, which):
        """'c0' | 'c1' | 'f1' as far as a backward pass can tell: the tensor itself when the last forward wrote it,
        otherwise zeros with each pool window's maximum at its recorded position (same MaxPoolGrad routing, same
        ReluGrad mask).  For tests and debugging."""
        src = {'c0': (self.c0, self.p0, self.a0, 1), 'c1': (self.c1, self.p1, self.a1, 1),
               'f1': (self.f1, self.cat, self.af1, 2)}[which]
        full, pooled, arg, phase = src
        if self.pooled_fwd != phase and not (self.bf16s and which == 'f1') and not (self.pool1_fused and which == 'c1') and not (self.pool0_fused and which == 'c0'):
            return full
        if full is None:          # 'bf16s': the tensor does not exist at all
            full = torch.empty((pooled.shape[0],) + {'c0': (55, 74, 96), 'c1': (27, 37, 256), 'f1': (110, 148, 63)}[which],
                               device=self.device)
        pooled = pooled.float()
        c = arg.shape[-1]
        out = torch.zeros_like(full)
        ph, pw = arg.shape[1], arg.shape[2]
        win = out[:, :2 * ph, :2 * pw, :].reshape(full.shape[0], ph, 2, pw, 2, c)
        a = arg.long()
        for pos in range(4):
            win[:, :, pos >> 1, :, pos & 1, :] = torch.where(a == pos, pooled[..., :c], torch.zeros_like(pooled[..., :c]))
        return out

    def _conv_pool(self, name, x, y_pooled, argmax=None):
        d, w = self._fwd_operands(name)
        ops.conv2d_pool_fwd(d, x, w, self._v(name + '/bias'), y_pooled, 'relu', argmax)

    def _fine_first_image_form(self, argmax=None):
        """'bf16s': fine/first on the bf16 pipe from the 4-channel bf16 image, conv + ReLU + pool in one launch."""
        d, w = self._prepared(('fine/first/image', 0, 0, 0), self.d4, self.w4)
        ops.conv2d_pool_fwd(d, self.x4, w, self._v('fine/first/conv2d/bias'), self.cat, 'relu', argmax)

    def forward(self, images, depths, keep_mask, join=True, phase=None):
        """join=False leaves the fine network's forward in flight on the side stream (step() joins later).
        phase (1 coarse / 2 fine / 3 none trained) runs every conv that feeds a max pool fused with it: the pre-pool
        activations c0, c1, f1 are then NOT written.  The network being trained also records the position of each
        maximum (a0, a1 / af1): MaxPoolGrad routes dy to that position and the fused ReluGrad only asks whether the
        maximum is positive, so the backward needs nothing else of c0 / c1 / f1.  phase None keeps every activation."""
        if images.shape[:3] == depths.shape[:3]:
            ops.resize_bilinear_tf1_pair(images, self.x, depths, self.t)
        else:
            ops.resize_bilinear_tf1(images, self.x)
            ops.resize_bilinear_tf1(depths, self.t)
        B = self.B
        self.dropout_on = keep_mask is not None            # None: the plugin was called with train=False
        # fine phase: the main queue has nothing left to run beside fine/second (the coarse backward does not exist there and
        # the fine backward waits for this forward), so the launch takes the whole CU instead of leaving room (A3D_HINT_SHARE_CU)
        self._alone = ('fine/second/conv2d',) if phase == 2 else ()
        if self.bf16s and (phase in (1, 3) or self.fine_first_bf16 or self.conv0_image) and not (self.fuse_pool and phase in (1, 2, 3)):
            # the fine network's 4-channel bf16 image, on the main stream: the side stream's chain (fine/first .. loss) is
            # the longer one at the join, the main stream idles there
            ops.pad_channels_bf16(self.x, self.x4)
        fused = self.fuse_pool and phase in (1, 2, 3)
        self.pooled_fwd = phase if fused else None
        lean_fine = fused
        if fused:
            train = phase == 1
            self._conv_pool('coarse/conv/conv2d_0', self.x, self.p0, self.a0 if train else None)
            self._conv_pool('coarse/conv/conv2d_1', self.p0, self.p1, self.a1 if train else None)
        elif self.bf16s:
            if self.pool0_fused and self.conv0_image:
                d, w = self._prepared(('conv2d_0/image', 0, 0, 0), self.d0_4, self.w0_4)
                ops.conv2d_pool_fwd(d, self.x4, w, self._v('coarse/conv/conv2d_0/bias'), self.p0, 'relu', self.a0)
            elif self.pool0_fused:
                self._conv_pool('coarse/conv/conv2d_0', self.x, self.p0, self.a0)
            else:
                self._conv('coarse/conv/conv2d_0', self.x, self.c0)
                self._pool(self.c0, self.p0)
            if self.pool1_fused:        # conv + ReLU + pool in the LDS-DMA kernel's epilogue: c1 (33 MB at B = 64) is never written
                self._conv_pool('coarse/conv/conv2d_1', self.p0, self.p1, self.a1)
            else:
                self._conv('coarse/conv/conv2d_1', self.p0, self.c1)
                self._pool(self.c1, self.p1)
        else:
            self._conv('coarse/conv/conv2d_0', self.x, self.c0)
            self._pool(self.c0, self.p0)
            self._conv('coarse/conv/conv2d_1', self.p0, self.c1)
            self._pool(self.c1, self.p1)
        def fine_first():
            with self._beside():    # beside the two weight-streaming dense layers
                if lean_fine:
                    self._conv_pool('fine/first/conv2d', self.x, self.cat, self.af1 if phase == 2 else None)   # cat[..., :63]
                elif self.bf16s and phase in (1, 3):        # conv + ReLU + pool in one launch: f1 is never written
                    self._fine_first_image_form()
                elif self.bf16s and self.fine_first_bf16:   # the same launch where the layer trains: it also records the window positions
                    self._fine_first_image_form(self.af1)
                elif self.bf16s:
                    self._conv_pool('fine/first/conv2d', self.x, self.cat, self.af1)
                else:
                    self._conv('fine/first/conv2d', self.x, self.f1)
        # where the side stream starts: ahead of conv2d_4's forward.  The fine forward is the longer chain of the stretch
        # (bf16 storage: fine/first runs 0.27 ms beside the dense layers; started after conv2d_4 it left the main queue
        # waiting 41 us at the join: 1.43 -> 1.39 ms; fp32: 26 us at the join, 2.93 -> 2.91 ms).  One layer earlier still is
        # slower again (1.41 / 2.94 ms), two layers 1.44: MFMA-bound grids beside each other only trade CU slots.
        self._conv('coarse/conv/conv2d_2', self.p1, self.c2)
        self._conv('coarse/conv/conv2d_3', self.c2, self.c3)
        fine_first()
        self._c4_32_fresh = False
        if self.bf16s and self.dense_bf16_x and self.fuse_casts and phase == 1:
            # c4 also as float32 (dense_0's filter gradient reads it in that type) from the same reduction
            n4 = 'coarse/conv/conv2d_4'
            ops.conv2d_fwd(self._desc(n4, 'fwd'), self.c3, self._w(n4), self._v(n4 + '/bias'), self.c4, 'relu',
                           out2=ops.second_output(self.c4_32.view(-1, 256)))
            self._c4_32_fresh = True
        else:
            self._conv('coarse/conv/conv2d_4', self.c3, self.c4)
        if not self._sharded_in_flight():
            self.settle()           # the previous step's dense-layer update is due now, not earlier
        w, b = self._kb('coarse/dense/dense_0')
        if self.bf16s and self.dense_bf16_x:   # dense_0 streams its 100 MB bf16 weight copy against c4 as it stands (bf16): LDS-DMA kernel
            # (its bf16 copy for dense_1 leaves through the same reduction: a3d_second_output, round 5)
            ops.dense_fwd_ex(self.c4.view(B, -1), self.wcopy['coarse/dense/dense_0'], b, self.drop, 'relu',
                             drop_keep=keep_mask, precision='bf16', storage=ops.STORE_W | ops.STORE_X,
                             out2=ops.second_output(self.drop16) if (self.dense1_bf16 and self.fuse_casts) else None)
        elif self.bf16s:            # more than 64 rows: the LDS-DMA kernel's weight-stream tiles do not apply; fp32 x, bf16 weight copy
            ops.cast_bf16(self.c4, self.c4_32)
            ops.dense_fwd_ex(self.c4_32.view(B, -1), self.wcopy['coarse/dense/dense_0'], b, self.drop, 'relu',
                             drop_keep=keep_mask, precision='bf16', storage=ops.STORE_W)
        else:
            ops.dense_fwd(self.c4.view(B, -1), w, b, self.drop, 'relu', drop_keep=keep_mask)     # relu + dropout fused
        w, b = self._kb('coarse/dense/dense_1')
        cat_done = False
        if self.bf16s and self.dense1_bf16 and self.fuse_casts:
            # the padded GEMM (4072 columns) writes the 4070-column depth map AND channel 63 of the concat buffer from its
            # reduction: no cast_rows, no copy_channel launch
            ops.dense_fwd_ex(self.drop16, self.w1pad, self.b1pad, self.coarse.view(B, -1), None, precision='bf16',
                             storage=ops.STORE_W | ops.STORE_X, n=self.w1pad.shape[1],
                             out2=ops.second_output(self.cat, cols=OUT_H * OUT_W, step=64, offset=63, ld=OUT_H * OUT_W))
            cat_done = True
        elif self.bf16s and self.dense1_bf16:
            ops.cast_bf16(self.drop, self.drop16)
            ops.dense_fwd_ex(self.drop16, self.w1pad, self.b1pad, self.y1pad, None, precision='bf16',
                             storage=ops.STORE_W | ops.STORE_X)
            ops.cast_rows(self.y1pad, self.coarse.view(B, -1))
        else:
            ops.dense_fwd(self.drop, w, b, self.coarse.view(B, -1))
        with self._beside():
            if cat_done:
                pass
            elif lean_fine or self.bf16s:
                ops.copy_channel(self.coarse, 0, self.cat, 63)                                # tf.concat([pool, coarse])
            else:
                self._pool(self.f1, self.cat, extra=self.coarse, c=63)                        # pool + concat fused
            self._conv('fine/second/conv2d', self.cat, self.f2)
            self._conv('fine/third', self.f2, self.fine)
            ops.silog_loss_fwd(self.fine, self.t, self.loss_fine, self.ws_f)
        ops.silog_loss_fwd(self.coarse, self.t, self.loss_coarse, self.ws_c)
        if join:
            self._join()

    def _sharded_in_flight(self):
        """The deferred dense bucket is a reduce-scatter feeding only the m slot: nothing in the forward needs it, it is due
        when the next backward is about to overwrite the gradient buffer it reads (backward_coarse settles first)."""
        d = self._deferred
        return d is not None and bool(d[0]) and isinstance(d[0][0], tuple)

    def _fused_dense_adam(self):
        return not self.keep_dense_grads and self.groups['CoarseDense'].frozen()

    def _bwd_data(self, name, dz, dx, relu_mask=None):
        ops.conv2d_bwd_data(self._desc(name, 'bwd_d'), dz, self._w(name, 'bwd_d'), dx, relu_mask=relu_mask)

    def _bwd_filter(self, name, x, dz):
        if name in self.d:
            ops.conv2d_bwd_filter(self._desc(name, 'bwd_f'), x, dz, self._g(name + '/kernel'), self._g(name + '/bias'))
        elif self._fused_dense_adam():
            g = self.groups[self.group_of[name + '/kernel']]
            kw = [g.view(buf, name + '/kernel') for buf in (g.var, g.m, g.v)]
            kb = [g.view(buf, name + '/bias') for buf in (g.var, g.m, g.v)]
            ops.dense_bwd_filter_adam_tf1(x, dz, *kw, *kb, g.lr, g.beta1, g.beta2, float(g.beta1_power),
                                          float(g.beta2_power), 1.0, precision='bf16' if self.bf16s else 'fp32')
        else:
            ops.dense_bwd_filter(x, dz, self._g(name + '/kernel'), self._g(name + '/bias'))

    # ---- backward of loss_coarse wrt coarse/* : src/models.py:318-324 ----
    def backward_coarse(self, after_dense=None, after_conv2=None, after_dense1=None):
        B = self.B
        fused16 = self.bf16s and self.dense1_bf16 and self.fuse_casts
        ops.silog_loss_bwd(self.coarse, self.t, self.ws_c, self.dz1.view(B, OUT_H, OUT_W, 1),
                           dout16=self.dz1_16 if fused16 else None)       # (dz1 a second time as bf16 rows of 4072: dense_1's bwd-data)
        self.settle()              # a reduce-scatter of the previous step may still be reading the dense gradient buffer
        n = 'coarse/dense/dense_1'
        self._bwd_filter(n, self.drop, self.dz1)
        if after_dense1 is not None:
            after_dense1()         # dense_1's gradient (67 MB) is complete: first piece of the dense bucket
        # dropout-grad (x2 on kept units) and dense_0's ReluGrad in one mask: drop > 0  <=>  kept and relu active
        # (train=False, src/models.py:230: tf.layers.dropout is the identity; only the ReluGrad remains)
        if self.bf16s and self.dense1_bf16:
            if not fused16:
                ops.cast_rows(self.dz1, self.dz1_16)
            ops.dense_bwd_data_ex(self.dz1_16, self.w1pad, self.dz0, mask=self.drop, scale=2.0 if self.dropout_on else 1.0,
                                  precision='bf16', storage=ops.STORE_W | ops.STORE_Y,
                                  out2=ops.second_output(self.dz0_16) if fused16 else None)     # dz0 as bf16 too: dense_0's bwd-data
        else:
            ops.dense_bwd_data(self.dz1, self._v(n + '/kernel'), self.dz0, mask=self.drop, scale=2.0 if self.dropout_on else 1.0)
        n = 'coarse/dense/dense_0'
        if self.bf16s and self.dense_bf16_x and not self._c4_32_fresh:   # the filter gradient takes c4 on the dense layers' fp32 side
            ops.cast_bf16(self.c4, self.c4_32)      # (B > 64: the forward already made this copy; fused casts: conv2d_4's reduction did)
        flat = (self.c4_32 if self.bf16s else self.c4).view(B, -1)
        self._bwd_filter(n, flat, self.dz0)
        if self.bf16s and self.dense_bf16_x:   # dz0 -> bf16 (1 MB); dc4 leaves as bf16, masked by the bf16 c4: no fp32 detour
            if not fused16:
                ops.cast_bf16(self.dz0, self.dz0_16)
            ops.dense_bwd_data_ex(self.dz0_16, self.wcopy[n], self.dc4.view(B, -1), mask=self.c4.view(B, -1), scale=1.0,
                                  precision='bf16', storage=ops.STORE_W | ops.STORE_X | ops.STORE_Y)
        elif self.bf16s:            # B > 64: fp32 dz0 and mask against the bf16 weight copy, dc4 through an fp32 buffer
            ops.dense_bwd_data_ex(self.dz0, self.wcopy[n], self.dc4_32.view(B, -1), mask=flat, scale=1.0, precision='bf16',
                                  storage=ops.STORE_W)
            ops.cast_bf16(self.dc4_32, self.dc4)
        else:
            ops.dense_bwd_data(self.dz0, self._v(n + '/kernel'), self.dc4.view(B, -1), mask=flat, scale=1.0)
        if after_dense is not None:
            after_dense()          # dense gradients are complete: their all-reduce can overlap the conv backward
        n = 'coarse/conv/conv2d_4'
        self._bwd_filter(n, self.c3, self.dc4)
        self._bwd_data(n, self.dc4, self.dc3, relu_mask=self.c3)
        self._join()               # the fine forward has had the dense / conv2d_4 stretch; the big GEMMs below run alone
        # (round 4: the join costs the main queue ~25 us — the side stream's loss kernel ends 14 us after conv2d_4's backward,
        # plus the cross-queue signal.  Joining later — after conv2d_3's filter gradient, after its bwd-data, at the end of the
        # backward — was measured at 2.97 ms against 2.92: a stream-K grid sized for all CU slots starts lopsided when the
        # side stream's blocks still hold some.  bf16 storage: 1.504 against 1.506, noise.)
        # (measured, round 3: the filter-gradient GEMMs on the side stream beside the bwd-data chain — to fill its tails and
        # launch gaps — make the step 2 % SLOWER, 3.20 vs 3.14 ms: two MFMA-bound grids only take CU slots from each other)
        dw = self._bwd_filter
        n = 'coarse/conv/conv2d_3'
        dw(n, self.c2, self.dc3)
        self._bwd_data(n, self.dc3, self.dc2, relu_mask=self.c2)
        n = 'coarse/conv/conv2d_2'
        dw(n, self.p1, self.dc2)
        if after_conv2 is not None:
            after_conv2()          # gradients of conv2d_2..4 (the tail of the CoarseConv buffer) are complete
        self._bwd_data(n, self.dc2, self.dp1)
        if self.pooled_fwd == 1 or self.pool1_fused:
            ops.maxpool2x2_bwd_idx(self.a1, self.p1, self.dp1, self.dc1, relu_mask=True)
        else:
            self._pool_bwd(self.c1, self.dp1, self.dc1)
        n = 'coarse/conv/conv2d_1'
        dw(n, self.p0, self.dc1)
        self._bwd_data(n, self.dc1, self.dp0)
        n = 'coarse/conv/conv2d_0'
        if (self.pooled_fwd == 1 or self.pool0_fused) and n in self.d_few:
            ops.conv2d_bwd_filter_pooled(self.d_few[n], self.x, self.dp0, self.p0, self.a0, self._g(n + '/kernel'),
                                         self._g(n + '/bias'))
            return
        if self.dc0 is None:
            self.dc0 = torch.empty((self.B, 55, 74, 96), device=self.device, dtype=self.dp0.dtype)
        if self.pooled_fwd == 1 or self.pool0_fused:
            ops.maxpool2x2_bwd_idx(self.a0, self.p0, self.dp0, self.dc0, relu_mask=True)
        else:
            self._pool_bwd(self.c0, self.dp0, self.dc0)
        dw(n, self.x, self.dc0)

    # ---- backward of loss_fine wrt fine/* : src/models.py:333-338 ----
    def backward_fine(self):
        self._join()                                                          # the fine forward ran on the side stream
        ops.silog_loss_bwd(self.fine, self.t, self.ws_f, self.dfine)
        n = 'fine/third'
        if ops.conv2d_bwd_both_supported(self.d[n]):
            # filter, bias and input gradient (+ fine/second's ReluGrad) in one pass over f2
            ops.conv2d_bwd_both(self.d[n], self.f2, self.dfine, self._v(n + '/kernel'), self._g(n + '/kernel'),
                                self._g(n + '/bias'), self.df2, relu_mask=True)
        else:
            assert not self.bf16s, 'bf16 storage: df2 is a bf16 tensor only the fused backward writes'
            self._bwd_filter(n, self.f2, self.dfine)
            self._bwd_data(n, self.dfine, self.df2, relu_mask=self.f2)
        n = 'fine/second/conv2d'
        self._bwd_filter(n, self.cat, self.df2)
        self._bwd_data(n, self.df2, self.dcat)
        n = 'fine/first/conv2d'
        if (self.pooled_fwd == 2 or self.bf16s) and n in self.d_few:          # reads channels 0..62 of dcat / cat
            ops.conv2d_bwd_filter_pooled(self.d_few[n], self.x, self.dcat, self.cat, self.af1, self._g(n + '/kernel'),
                                         self._g(n + '/bias'))
            return
        if self.df1 is None:
            self.df1 = torch.empty((self.B, 110, 148, 63), device=self.device)
        if self.pooled_fwd == 2 or self.bf16s:                                 # both read channels 0..62 of dcat
            ops.maxpool2x2_bwd_idx(self.af1, self.cat, self.dcat, self.df1, relu_mask=True)
        else:
            self._pool_bwd(self.f1, self.dcat, self.df1, c=63)
        self._bwd_filter(n, self.x, self.df1)

    # ---- one session.run(train_op) ----
    def step(self, images, depths, keep_mask):
        """images [B,H,W,3], depths [B,H',W',1] float32, keep_mask [B,4096] bool/uint8, all on this device.
        Both forwards and both losses run in every phase; gradients + Adam only for the active phase;
        global_step += 1 always (src/models.py:329,343,356)."""
        if keep_mask is not None and keep_mask.dtype != torch.uint8:
            keep_mask = keep_mask.to(torch.uint8)
        phase = phase_of(self.global_step, self.B)
        self.forward(images, depths, keep_mask, join=False, phase=phase)
        red = self.reducer
        scale = 1.0 / red.world_size if red is not None else 1.0
        if phase == 1:
            gc, gd = self.groups['CoarseConv'], self.groups['CoarseDense']
            if red is None:
                fused = self._fused_dense_adam()       # decided before the backward: it is what the dense layers ran
                self.backward_coarse()
                gc.apply(scale)
                if fused:
                    gd.advance()                       # ApplyAdam of coarse/dense/* already happened inside the backward
                else:
                    gd.apply(scale)
                self._weights_moved(gc, gd)
            else:
                # dense bucket (268 MB): reduced while the conv backward runs AND, being due only before the next
                # step's dense_0, while that step's conv forward runs (settle()); conv bucket (15 MB): waited for here
                # conv bucket in two pieces: conv2d_2..4 (12.4 of 15 MB, the tail of the flat buffer) goes out as
                # soon as conv2d_2's filter gradient exists and rides under the conv2d_1 / conv2d_0 backward; only
                # the 2.6 MB head is reduced on the critical path
                handle, tail = [], []
                cut = gc.offsets['coarse/conv/conv2d_2/kernel'][0]
                if gd.frozen() and self.dense_exchange == 'reduce_scatter':
                    # the reference's optimizer: reduce-scatter, each rank updates m on its own slices only (dp.py)
                    early, late = self._dense_buckets()
                    if self._poison is None:
                        self._poison = torch.zeros(1, dtype=torch.int32, device=self.device)

                    def scatter(pieces):
                        for a, b in pieces:
                            work, _ = red.reduce_scatter(gd.grad[a:b])
                            handle.append((work,) + self._my_slice(a, b))
                    after_dense1, after_dense = (lambda: scatter(early)), (lambda: scatter(late))
                else:
                    # dense bucket in backward production order: dense_1 (67 MB) leaves as soon as its filter
                    # gradient exists, dense_0 (201 MB) in three pieces after its own — four collectives of 50-67 MB
                    # pipeline over the xGMI links where one 268 MB ring would serialise behind its own reduce-scatter
                    d1 = gd.offsets['coarse/dense/dense_1/kernel'][0]
                    third = -(-(d1 // 3) // ParamGroup.ALIGN) * ParamGroup.ALIGN
                    pieces0 = [(0, third), (third, 2 * third), (2 * third, d1)]
                    after_dense1 = lambda: handle.append(red.start(gd.grad[d1:]))
                    after_dense = lambda: handle.extend(red.start(gd.grad[a:b]) for a, b in pieces0)
                self.backward_coarse(after_dense1=after_dense1, after_dense=after_dense,
                                     after_conv2=lambda: tail.append(red.start(gc.grad[cut:], urgent=True)))
                head = red.start(gc.grad[:cut], urgent=True)       # (urgent: not queued behind the dense reduce-scatter, dp.py)
                red.wait(tail[0])
                red.wait(head)
                gc.apply(scale)
                self._weights_moved(gc)
                self._deferred = (handle, gd, scale)
        elif phase == 2:
            ga, gb = self.groups['FineA'], self.groups['FineB']
            self.backward_fine()
            if red is not None:
                red.start(ga.grad)
                red.start(gb.grad)
                red.finish()
            ga.apply(scale)
            gb.apply(scale)
            self._weights_moved(ga, gb)
        self._join()
        self.global_step += 1
        return {'coarse_loss': self.loss_coarse, 'fine_loss': self.loss_fine, 'phase': phase}


# =====================================================================================================
# DCNF unary stack ("DCNF-lite", Liu et al. 2015) — src/models.py:14-18,50-89,179-183
# =====================================================================================================
DCNF_IMG_H, DCNF_IMG_W = 240, 320      # src/models.py:180
DCNF_PATCH, DCNF_SP = 100, 40          # src/models.py:15-16
DCNF_PREFIX = 'unary/unary_layers/'
DCNF_CONVS = [('conv2d', 3, 64, 11), ('conv2d_1', 64, 256, 5), ('conv2d_2', 256, 256, 3),
              ('conv2d_3', 256, 256, 3), ('conv2d_4', 256, 256, 3)]        # all VALID, stride 1, ReLU (:64-72)
DCNF_POOL_AFTER = ('conv2d', 'conv2d_1', 'conv2d_4')                       # :65,68,73
DCNF_DENSES = [('dense', 12544, 128, 'relu'), ('dense_1', 128, 16, 'sigmoid'), ('dense_2', 16, 1, None)]  # :80-82


class DCNFUnary:
    """The shared-weight unary conv stack over all patches of a batch at once (the reference maps it over the
    batch with tf.map_fn, src/models.py:89): images [B,H,W,3] -> z [B,48,1].  backward() takes a synthetic
    upstream gradient dz (DCNFReplica feeds it the CRF loss gradient)."""

    def __init__(self, batchsize, device='cuda', params=None, seed=3000, precision='fp32'):
        self.B = batchsize
        self.device = dev = torch.device(device)
        self.rows, _ = ops.same_pad(DCNF_IMG_H, DCNF_PATCH, DCNF_SP)
        self.cols, _ = ops.same_pad(DCNF_IMG_W, DCNF_PATCH, DCNF_SP)
        self.P = P = batchsize * self.rows * self.cols
        shapes = collections.OrderedDict()
        for n, ci, co, k in DCNF_CONVS:
            shapes[DCNF_PREFIX + n + '/kernel'] = (k, k, ci, co)
            shapes[DCNF_PREFIX + n + '/bias'] = (co,)
        for n, i, o, _ in DCNF_DENSES:
            shapes[DCNF_PREFIX + n + '/kernel'] = (i, o)
            shapes[DCNF_PREFIX + n + '/bias'] = (o,)
        self.shapes = shapes
        self.group = ParamGroup('unary', 0.1, shapes, dev, slots=False)    # GradientDescentOptimizer(0.1), src/models.py:198
        if params is None:
            rng = np.random.default_rng(seed)
            params = {n: (glorot_uniform(rng, s) if n.endswith('/kernel') else np.zeros(s, np.float32))
                      for n, s in shapes.items()}
        for n, s in shapes.items():
            self.group.view(self.group.var, n).copy_(torch.from_numpy(np.ascontiguousarray(params[n], np.float32)))

        def buf(*shape):
            return torch.empty(shape, device=dev)
        self.resized = buf(batchsize, DCNF_IMG_H, DCNF_IMG_W, 3)
        self.act = collections.OrderedDict()
        self.dact = {}
        self.desc = {}
        # conv + ReLU + max pool in one kernel (fp32): the pre-pool activations (1.6 + 1.3 GB at batch 16) are never
        # written, one argmax byte per pool window serves the backward (see MSDNReplica.forward)
        self.fuse_pool = precision == 'fp32' and os.environ.get('A3D_NO_FUSED_POOL', '0') != '1'
        self.argmax, self.conv_hw = {}, {}
        self.few_pooled = False
        self.act['x'] = buf(P, DCNF_PATCH, DCNF_PATCH, 3)
        h = DCNF_PATCH
        for n, ci, co, k in DCNF_CONVS:
            self.desc[n] = ops.conv_desc(P, h, h, ci, co, k, k, 1, 'VALID', precision=precision)
            h = h - k + 1
            self.conv_hw[n] = h
            if not (self.fuse_pool and n in DCNF_POOL_AFTER):
                self.act[n] = buf(P, h, h, co)
            if n in DCNF_POOL_AFTER:
                h //= 2
                self.act[n + '/pool'] = buf(P, h, h, co)
                if self.fuse_pool:
                    self.argmax[n] = torch.empty((P, h, h, co), dtype=torch.uint8, device=dev)
        for n, i, o, _ in DCNF_DENSES:
            self.act[n] = buf(P, o)
        self.z = self.act['dense_2']
        self.few_pooled = (self.fuse_pool and dev.type == 'cuda' and os.environ.get('A3D_FEWCH_POOLED', '1') != '0'
                           and ops.conv2d_bwd_filter_pooled_supported(self.desc['conv2d']))

    def var(self, name):
        return self.group.view(self.group.var, DCNF_PREFIX + name)

    def grad(self, name):
        return self.group.view(self.group.grad, DCNF_PREFIX + name)

    def forward(self, images):
        ops.resize_bilinear_tf1(images, self.resized)                           # src/models.py:180
        ops.extract_patches(self.resized, DCNF_PATCH, DCNF_SP, self.act['x'])   # :50-59
        t = self.act['x']
        for n, _, _, _ in DCNF_CONVS:
            if self.fuse_pool and n in DCNF_POOL_AFTER:
                t = ops.conv2d_pool_fwd(self.desc[n], t, self.var(n + '/kernel'), self.var(n + '/bias'),
                                        self.act[n + '/pool'], 'relu', self.argmax[n])
                continue
            ops.conv2d_fwd(self.desc[n], t, self.var(n + '/kernel'), self.var(n + '/bias'), self.act[n], 'relu')
            t = self.act[n]
            if n in DCNF_POOL_AFTER:
                ops.maxpool2x2_fwd(t, self.act[n + '/pool'])
                t = self.act[n + '/pool']
        t = t.view(self.P, -1)
        for n, _, _, act in DCNF_DENSES:
            ops.dense_fwd(t, self.var(n + '/kernel'), self.var(n + '/bias'), self.act[n], act)
            t = self.act[n]
        return self.z.view(self.B, self.rows * self.cols, 1)

    def activations(self):
        """name -> numpy array of every activation; a pre-pool activation that the fused conv + pool never wrote comes
        back as zeros with each window's maximum at its recorded position (all a backward pass can see of it)."""
        out = {k: v.cpu().numpy() for k, v in self.act.items()}
        for n, arg in self.argmax.items():
            pooled = self.act[n + '/pool']
            hw, c = self.conv_hw[n], pooled.shape[-1]
            full = torch.zeros((self.P, hw, hw, c), device=self.device)
            ph = pooled.shape[1]
            win = full[:, :2 * ph, :2 * ph, :].reshape(self.P, ph, 2, ph, 2, c)
            a = arg.long()
            for pos in range(4):
                win[:, :, pos >> 1, :, pos & 1, :] = torch.where(a == pos, pooled, torch.zeros_like(pooled))
            out[n] = full.cpu().numpy()
        return out

    def _dbuf(self, key, like):
        if key not in self.dact:
            self.dact[key] = torch.empty_like(like)
        return self.dact[key]

    def backward(self, dz):
        """Gradients of sum(z * dz) wrt every unary variable, into the flat gradient buffer."""
        P = self.P
        a = self.act
        flat = a['conv2d_4/pool'].view(P, -1)
        d2 = dz.reshape(P, 1).contiguous()
        # dense_2 (linear)
        ops.dense_bwd_filter(a['dense_1'], d2, self.grad('dense_2/kernel'), self.grad('dense_2/bias'))
        d1 = self._dbuf('dense_1', a['dense_1'])
        ops.dense_bwd_data(d2, self.var('dense_2/kernel'), d1, mask=a['dense_1'], mask_act='sigmoid')  # y(1-y) fused
        ops.dense_bwd_filter(a['dense'], d1, self.grad('dense_1/kernel'), self.grad('dense_1/bias'))
        d0 = self._dbuf('dense', a['dense'])
        ops.dense_bwd_data(d1, self.var('dense_1/kernel'), d0, mask=a['dense'], scale=1.0)     # ReluGrad fused
        ops.dense_bwd_filter(flat, d0, self.grad('dense/kernel'), self.grad('dense/bias'))
        dflat = self._dbuf('flat', a['conv2d_4/pool'])
        ops.dense_bwd_data(d0, self.var('dense/kernel'), dflat.view(P, -1))
        d = dflat
        inputs = {'conv2d': 'x', 'conv2d_1': 'conv2d/pool', 'conv2d_2': 'conv2d_1/pool', 'conv2d_3': 'conv2d_2',
                  'conv2d_4': 'conv2d_3'}
        for n, _, _, _ in reversed(DCNF_CONVS):
            if n == 'conv2d' and self.fuse_pool and self.few_pooled:
                # the first conv's filter gradient straight from the pooled map's gradient: its 768 x 90 x 90 x 64 pre-pool
                # gradient (1.6 GB at batch 16) is never written
                pool = a[n + '/pool']
                ops.conv2d_bwd_filter_pooled(self.desc[n], a['x'], d.view(pool.shape), pool, self.argmax[n],
                                             self.grad(n + '/kernel'), self.grad(n + '/bias'))
                continue
            if n in DCNF_POOL_AFTER and self.fuse_pool:
                hw, co = self.conv_hw[n], a[n + '/pool'].shape[-1]
                if n not in self.dact:
                    self.dact[n] = torch.empty((P, hw, hw, co), device=self.device)
                dzc = self.dact[n]
                ops.maxpool2x2_bwd_idx(self.argmax[n], a[n + '/pool'], d.view(a[n + '/pool'].shape), dzc, relu_mask=True)
            elif n in DCNF_POOL_AFTER:
                dzc = self._dbuf(n, a[n])
                ops.maxpool2x2_bwd(a[n], d, dzc, relu_mask=True)
            else:
                dzc = d      # the producer (bwd-data of the next conv) already applied this layer's ReluGrad
            x = a[inputs[n]]
            ops.conv2d_bwd_filter(self.desc[n], x, dzc, self.grad(n + '/kernel'), self.grad(n + '/bias'))
            if n != 'conv2d':
                dx = self._dbuf('in:' + n, x)
                prev_is_plain_relu = inputs[n] in ('conv2d_2', 'conv2d_3')     # no pool between: fuse its ReluGrad
                ops.conv2d_bwd_data(self.desc[n], dzc, self.var(n + '/kernel'), dx,
                                    relu_mask=x if prev_is_plain_relu else None)
                d = dx


DCNF_GAMMA, DCNF_EPSILON = 1.0, 1e-7    # src/models.py:17-18
DCNF_PAIR_PREFIX = 'pairwise/pairwise_layers/dense/'


def dcnf_pair_indices(rows, cols):
    """src/models.py:20-30: every interior superpixel on one colour of the checkerboard, paired with its four
    neighbours (up, down, left, right): two int lists (left, right)."""
    left, right = [], []
    for row in range(1, rows - 1):
        for col in range(2 - (row & 1), cols - 1, 2):
            pixel = row * cols + col
            for addend in (-cols, cols, -1, 1):
                left.append(pixel)
                right.append(pixel + addend)
    return left, right


class DCNFReplica:
    """The whole DCNF train step (src/models.py:179-200): resize to 240x320, unary z over 48 patches, pairwise r over 48
    superpixel pairs, CRF negative log-likelihood, gradient descent (0.1) on what receives a gradient.

    TF-1.3 semantics assumed (the oracle states the same, oracle/dcnf.py): scatter_nd_update has no gradient, so the CRF
    matrix A is a constant for the optimizer — the pairwise dense layer never moves and only the unary stack trains."""
    uses_dropout = False

    def __init__(self, batchsize, device='cuda', params=None, seed=3000, global_step=0, reducer=None,
                 precision='fp32'):
        self.B = batchsize
        self.device = dev = torch.device(device)
        self.reducer = reducer
        self.global_step = global_step
        self.unary = DCNFUnary(batchsize, dev, params=params, seed=seed, precision=precision)
        self.rows, self.cols = DCNF_IMG_H // DCNF_SP, DCNF_IMG_W // DCNF_SP
        assert (self.rows, self.cols) == (self.unary.rows, self.unary.cols)
        self.nsp = self.rows * self.cols
        left, right = dcnf_pair_indices(self.rows, self.cols)
        self.left = torch.tensor(left, dtype=torch.int32, device=dev)
        self.right = torch.tensor(right, dtype=torch.int32, device=dev)
        pshapes = collections.OrderedDict([(DCNF_PAIR_PREFIX + 'kernel', (2, 1)), (DCNF_PAIR_PREFIX + 'bias', (1,))])
        self.pair_group = ParamGroup('pairwise', 0.1, pshapes, dev, slots=False)
        if params is None or DCNF_PAIR_PREFIX + 'kernel' not in params:
            rng = np.random.default_rng(seed + 1)
            pw = {DCNF_PAIR_PREFIX + 'kernel': glorot_uniform(rng, (2, 1)),
                  DCNF_PAIR_PREFIX + 'bias': np.zeros((1,), np.float32)}
        else:
            pw = params
        for n in pshapes:
            self.pair_group.view(self.pair_group.var, n).copy_(
                torch.from_numpy(np.ascontiguousarray(pw[n], np.float32)))
        self.groups = collections.OrderedDict([('unary', self.unary.group), ('pairwise', self.pair_group)])
        self.depths240 = torch.empty((batchsize, DCNF_IMG_H, DCNF_IMG_W, 1), device=dev)
        self.hist = torch.empty((batchsize, self.nsp, 256), device=dev)
        self.y = torch.empty((batchsize, self.nsp, 1), device=dev)
        self.output = torch.empty((batchsize, DCNF_IMG_H, DCNF_IMG_W, 1), device=dev)
        self.sims = self.r = self.loss = self.loss_per_image = self.dz = None

    def pair_var(self, name):
        return self.pair_group.view(self.pair_group.var, DCNF_PAIR_PREFIX + name)

    def forward(self, images, depths):
        """z, r and the loss; leaves d loss / d z in self.dz."""
        self.unary.forward(images)                                                        # src/models.py:180,183
        return self.forward_crf(depths)

    def forward_crf(self, depths):
        """Everything after the unary stack: target superpixels, pairwise r, CRF loss and d loss / d z."""
        u = self.unary
        ops.resize_bilinear_tf1(depths, self.depths240)                                   # src/models.py:181
        ops.superpixel_hist(u.resized, DCNF_SP, self.hist)                                # :112-113
        self.sims, self.r = ops.pair_similarity(u.resized, DCNF_SP, self.hist, self.left, self.right,
                                                self.pair_var('kernel'), self.pair_var('bias'), DCNF_GAMMA)  # :115-127
        ops.superpixel_mean(self.depths240, DCNF_SP, self.y)                              # :131-132
        self.loss, self.loss_per_image, self.dz = ops.crf_loss(u.z.view(self.B, self.nsp), self.y.view(self.B, self.nsp),
                                                               self.r, self.left, self.right, DCNF_EPSILON)  # :129-177
        return self.loss

    def step(self, images, depths, keep_mask=None):
        self.forward(images, depths)
        self.unary.backward(self.dz)
        red = self.reducer
        scale = 1.0
        if red is not None:
            red.start(self.unary.group.grad)
            red.finish()
            scale = 1.0 / red.world_size
        self.unary.group.apply_sgd(scale)                                                 # :198-200
        self.global_step += 1
        return {'mean_loss': self.loss}

    def summary_scalars(self, out):
        return {'loss/mean_loss': float(out['mean_loss'])}                                # src/models.py:174

    def summary_images(self):
        """src/models.py:187-196; max_outputs=1."""
        ops.resize_bilinear_tf1(self.unary.z.view(self.B, self.rows, self.cols, 1), self.output)
        return [('summaries/Output', self.output, 1), ('summaries/Input', self.unary.resized, 1),
                ('summaries/Target', self.depths240, 1)]

    def broadcast_state(self, dist, src=0):
        for g in self.groups.values():
            dist.broadcast(g.var, src)
        st = torch.tensor([self.global_step], dtype=torch.float64, device=self.device)
        dist.broadcast(st, src)
        self.global_step = int(st[0].item())

    def gather_state(self):
        """Nothing is sharded here (the driver calls this on every replica before a checkpoint)."""

    def state_dict(self):
        sd = collections.OrderedDict()
        for g in self.groups.values():
            for n in g.offsets:
                sd[n] = g.view(g.var, n)
        sd['global_step'] = torch.tensor(self.global_step, dtype=torch.int64)
        return sd

    def load_state_dict(self, sd):
        for g in self.groups.values():
            for n in g.offsets:
                g.view(g.var, n).copy_(sd[n])
        self.global_step = int(sd['global_step'].item())

    def load_tf_variables(self, tensors):
        """Restore from a TensorFlow checkpoint's whole tensors (the reference's dcnf partitions its variables, which
        tfckpt.read_bundle refuses; a bundle written by this build loads)."""
        for g in self.groups.values():
            for n in g.offsets:
                g.view(g.var, n).copy_(torch.from_numpy(np.ascontiguousarray(tensors[n], np.float32)))
        if 'global_step' in tensors:
            self.global_step = int(np.asarray(tensors['global_step']).reshape(-1)[0])

    def tf_variables(self):
        out = {}
        for k, v in self.state_dict().items():
            a = v.detach().cpu().numpy()
            out[k] = a.astype(np.int64) if k == 'global_step' else a.astype(np.float32)
        return out


# =====================================================================================================
# plugin surface: models.msdn / models.dcnf  (src/models.py:370-371)
# =====================================================================================================
class TrainOp:
    """What `model(inputs, targets)` returns: run() is one `session.run(train_op)`.

    Input side: the shuffle queue's staging pool is pinned memory; a dequeued batch is B slot numbers, DMA'd to HBM
    on a side stream into one of two device batch buffers.  Batch k+1 is in flight while step k computes, so host
    decode, PCIe transfer and the training step overlap."""

    def __init__(self, replica, pipeline, seed=0):
        self.replica, self.pipeline, self.seed = replica, pipeline, seed
        dev = replica.device
        self.keep = torch.empty((replica.B, 4096), dtype=torch.uint8, device=dev) if replica.uses_dropout else None
        pipeline.allocate(lambda shape: torch.empty(shape, dtype=torch.float32).pin_memory().numpy(),
                          lambda shape: torch.empty(shape, dtype=torch.uint8).pin_memory().numpy())
        self.pool = (torch.from_numpy(pipeline.images), torch.from_numpy(pipeline.depths))      # pinned views
        # converter-written records are staged as uint8 pixel values (data.py): their own pinned pool and device buffers;
        # the resize kernel rebuilds the loader's float32 from them bit for bit
        self.pool_u8 = None
        if pipeline.images_u8 is not None:
            self.pool_u8 = (torch.from_numpy(pipeline.images_u8), torch.from_numpy(pipeline.depths_u8))
        self.dev = [tuple(torch.empty((replica.B,) + tuple(p.shape[1:]), device=dev) for p in self.pool)
                    for _ in range(2)]
        self.dev_u8 = [tuple(torch.empty((replica.B,) + tuple(p.shape[1:]), device=dev, dtype=torch.uint8) for p in self.pool)
                       for _ in range(2)] if self.pool_u8 is not None else None
        self.cur = [None, None]             # the (images, depths) tensors batch i was copied into: uint8 or float32 each
        self.copy_stream = torch.cuda.Stream(device=dev)
        self.copied = [None, None]          # event: the DMA into buffer i finished
        self.consumed = [None, None]        # event: the step that read buffer i finished
        self.held = [[], []]                # pool slots buffer i was filled from
        self.k = 0                          # batches consumed
        self.end = None                     # (batch index, exception) once the pipeline ran dry
        self.last = None
        self._prefetch(0)
        self._prefetch(1)

    def _prefetch(self, j):
        """Dequeue batch j and start its DMA into device buffer j & 1 on the copy stream."""
        if self.end is not None:
            return
        i = j & 1
        try:
            slots = self.pipeline.dequeue()
        except BaseException as e:          # raised to the caller when batch j would have been consumed
            self.end = (j, e)
            return
        if self.consumed[i] is not None:
            self.copy_stream.wait_event(self.consumed[i])
        pl = self.pipeline
        cur = []
        kinds = int(np.bitwise_and.reduce(pl.kind[slots])) if self.pool_u8 is not None else 0
        for which in (0, 1):                # images, depths: uint8 when EVERY record of the batch staged that feature so
            as_u8 = bool(kinds & (1 << which))
            if not as_u8 and self.pool_u8 is not None:
                for s in slots:
                    pl.materialise(s, which)                    # (a mixed batch: the uint8 records are expanded on the host)
            cur.append((self.dev_u8 if as_u8 else self.dev)[i][which])
        ids = (ctypes.c_int32 * len(slots))(*slots)
        with torch.cuda.stream(self.copy_stream):
            for which in (0, 1):
                src = (self.pool_u8 if cur[which].dtype == torch.uint8 else self.pool)[which]
                ops.check(_lib.load().a3d_h2d_gather(cur[which].data_ptr(), src.data_ptr(), ids, len(slots), len(src),
                                                     src[0].numel() * src.element_size(), self.copy_stream.cuda_stream),
                          'a3d_h2d_gather')                     # B copies from ONE call (no per-record host-language work)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        self.cur[i] = tuple(cur)
        self.copied[i] = ev
        self.held[i] = slots

    def run(self):
        r = self.replica
        i = self.k & 1
        if self.end is not None and self.end[0] == self.k:
            raise self.end[1]
        cur = torch.cuda.current_stream()
        cur.wait_event(self.copied[i])
        if self.keep is not None:
            ops.dropout_keep_mask(self.keep, self.seed, r.global_step)
        self.last = r.step(self.cur[i][0], self.cur[i][1], self.keep)
        ev = torch.cuda.Event()
        ev.record(cur)
        self.consumed[i] = ev
        self.copied[i].synchronize()        # issued a whole step ago: the staging slots can go back to the readers
        self.pipeline.release(self.held[i])
        self.held[i] = []
        self.k += 1
        self._prefetch(self.k + 1)          # into the buffer this step just read; the DMA waits for `consumed`
        return self.last

    @property
    def global_step(self):
        return self.replica.global_step


class _MultiScaleDeepNetwork:
    """Eigen et al. (2014) coarse+fine network — plugin wrapper around MSDNReplica (src/models.py:203-367)."""
    beta2 = 1.0          # the reference's AdamOptimizer(rate, momentum, 1): alpha == 0, weights never move
    reducer = None       # set by the driver when world_size > 1
    seed = 3000
    precision = 'fp32'   # --precision: 'fp32' | 'bf16x3' | 'bf16' (see MSDNReplica)

    def __call__(self, images, depths, train=True):
        assert images.pipeline is depths.pipeline, 'inputs and targets must come from the same data.inputs() call'
        self.train = train
        replica = MSDNReplica(images.pipeline.B, device=torch.device('cuda', torch.cuda.current_device()),
                              seed=self.seed, beta2=self.beta2, reducer=self.reducer, precision=self.precision,
                              keep_dense_grads=False)        # one GPU + the reference's optimizer: dW feeds ApplyAdam directly
        replica.uses_dropout = bool(train)                   # train=False: tf.layers.dropout(training=False), src/models.py:230
        if self.reducer is not None:                         # replicas start from rank 0's weights
            for g in replica.groups.values():
                self.reducer.broadcast(g.var)
        return TrainOp(replica, images.pipeline, seed=self.seed + 1000 * (self.reducer.rank if self.reducer else 0))


class _DistributedConvolutionalNeuralFields:
    """Liu et al. (2015) — plugin wrapper around DCNFReplica (src/models.py:13-200).  The reference pins its layers
    to /job:worker/task:{1,2,3} (src/models.py:63-79); here every replica holds the whole model and replicas are
    data-parallel, like msdn."""
    reducer = None
    seed = 3000
    precision = 'fp32'
    beta2 = None         # accepted for symmetry with msdn; gradient descent has no beta

    def __call__(self, images, depths, train=True):
        assert images.pipeline is depths.pipeline, 'inputs and targets must come from the same data.inputs() call'
        self.train = train
        replica = DCNFReplica(images.pipeline.B, device=torch.device('cuda', torch.cuda.current_device()),
                              seed=self.seed, reducer=self.reducer, precision=self.precision)
        if self.reducer is not None:
            for g in replica.groups.values():
                self.reducer.broadcast(g.var)
        return TrainOp(replica, images.pipeline, seed=self.seed)


dcnf = _DistributedConvolutionalNeuralFields()
msdn = _MultiScaleDeepNetwork()
