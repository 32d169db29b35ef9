"""Operator layer: the TF ops ann3depth's model functions instantiate, as calls into liba3d.so on torch tensors.

PyTorch only owns memory and streams here.  Every function enqueues HIP kernels on the current stream through the
C ABI (include/a3d.h) and returns without synchronising.  Layouts are TensorFlow's (NHWC / HWIO / [in,out]).
"""
import ctypes

import torch

from . import _lib
from ._lib import ConvDesc, check

ACT = {None: 0, 'relu': 1, 'sigmoid': 2}
PREC = {'fp32': 0, 'f32': 0, 'bf16x3': 1, 'bf16': 2}


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class Workspace:
    """Scratch for split-K slabs and partial reductions.  One per stream; grows on demand (never inside a graph
    capture: call reserve() with the largest need first)."""

    def __init__(self):
        self.buf = None

    def reserve(self, nbytes, device):
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError(f'workspace of {nbytes} bytes must be reserved before graph capture')
            self.buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        return self.buf

    def get(self, nbytes, device):
        buf = self.reserve(nbytes, device)
        return ctypes.c_void_p(buf.data_ptr()), buf.numel()


_WS = {}


def _ws():
    """The workspace of the current stream (kernels of different streams may run concurrently)."""
    key = torch.cuda.current_stream().cuda_stream
    w = _WS.get(key)
    if w is None:
        w = _WS[key] = Workspace()
    return w


def drop_workspace(stream_handle):
    """Forget the workspace of a stream that is being destroyed (its split-K scratch goes back to the allocator, and a
    later stream that happens to get the same handle value starts with a workspace of its own)."""
    _WS.pop(stream_handle, None)


def same_pad(in_size, k, stride):
    out = -(-in_size // stride)
    pad = max((out - 1) * stride + k - in_size, 0)
    return out, pad // 2


STORE_X, STORE_W, STORE_Y = 1, 2, 4      # a3d_conv_desc.storage bits: which tensors are bf16 in memory


HINT_SHARE_CU = 1                    # a3d_conv_desc.hints
HINT_W_PREPARED = 2


class PreparedFilter:
    """The filter of a few-channel forward in the layout its kernel reads (a3d_conv2d_fwd_prepare_filter), kept beside the
    stored filter so that the repack runs when the weights change instead of on every call.  desc: the descriptor the forward
    uses WITHOUT the hint; .desc_prepared is the same descriptor with A3D_HINT_W_PREPARED, .buf the prepared copy."""

    def __init__(self, desc, device):
        lib = _lib.load()
        nbytes = lib.a3d_conv2d_fwd_prepared_filter_bytes(ctypes.byref(desc))
        self.ok = nbytes > 0
        self.desc = desc
        if self.ok:
            self.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
            self.desc_prepared = ConvDesc.from_buffer_copy(desc)
            self.desc_prepared.hints = desc.hints | HINT_W_PREPARED

    def refresh(self, w):
        check(_lib.load().a3d_conv2d_fwd_prepare_filter(ctypes.byref(self.desc), _ptr(w), _ptr(self.buf), self.buf.numel(), _stream()),
              'a3d_conv2d_fwd_prepare_filter')


def conv_desc(n, h, w, c, k, r, s, stride, padding, ldx=None, ldy=None, precision='fp32', storage=0, hints=0):
    """Descriptor of tf.layers.conv2d(x[n,h,w,c], k, (r,s), (stride,stride), padding).  precision selects the
    arithmetic of the contraction: 'fp32' (exact, default), 'bf16x3' (split operands) or 'bf16'."""
    padding = padding.upper()
    if padding == 'SAME':
        ho, pt = same_pad(h, r, stride)
        wo, pl = same_pad(w, s, stride)
    elif padding == 'VALID':
        ho, pt = (h - r) // stride + 1, 0
        wo, pl = (w - s) // stride + 1, 0
    else:
        raise ValueError(padding)
    return ConvDesc(n=n, h=h, w=w, c=c, k=k, r=r, s=s, stride=stride, pad_t=pt, pad_l=pl, ho=ho, wo=wo,
                    ldx=ldx or c, ldy=ldy or k, precision=PREC[precision], storage=storage, hints=hints)


def second_output(t, cols=None, step=1, offset=0, ld=None):
    """a3d_second_output for tensor t: element (row, col < cols) of the launch's output also goes to
    t.flat[(row * ld + col) * step + offset], in t's type (float32 or bfloat16).  Default: t as a [rows, ld] matrix."""
    if t is None:
        return None
    ld = t.shape[-1] if ld is None else ld
    o = _lib.SecondOutput(t.data_ptr(), ld, step, offset, int(t.dtype == torch.bfloat16), ld if cols is None else cols)
    o._keep = t
    return o


def _o2(out2):
    return None if out2 is None else ctypes.byref(out2)


def conv2d_fwd(d, x, w, bias, y, act=None, out2=None):
    lib = _lib.load()
    ws, n = _ws().get(lib.a3d_conv2d_fwd_ws_bytes(ctypes.byref(d)), x.device)
    if out2 is not None:
        check(lib.a3d_conv2d_fwd_ex2(ctypes.byref(d), _ptr(x), _ptr(w), _ptr(bias), _ptr(y), ACT[act], _o2(out2), ws, n, _stream()),
              'a3d_conv2d_fwd_ex2')
        return y
    check(lib.a3d_conv2d_fwd(ctypes.byref(d), _ptr(x), _ptr(w), _ptr(bias), _ptr(y), ACT[act], ws, n, _stream()),
          'a3d_conv2d_fwd')
    return y


def conv2d_pool_fwd(d, x, w, bias, y_pooled, act='relu', argmax=None):
    """maxpool2x2(act(conv2d(x) + bias)) in one kernel; y_pooled [n, ho//2, wo//2, >= k] (last dim = pixel stride).
    argmax: optional uint8 [n, ho//2, wo//2, k], receives the window position of each maximum (for maxpool2x2_bwd_idx)."""
    lib = _lib.load()
    ws, n = _ws().get(lib.a3d_conv2d_fwd_ws_bytes(ctypes.byref(d)), x.device)
    check(lib.a3d_conv2d_pool_fwd(ctypes.byref(d), _ptr(x), _ptr(w), _ptr(bias), _ptr(y_pooled), y_pooled.shape[-1],
                                  _ptr(argmax), ACT[act], ws, n, _stream()), 'a3d_conv2d_pool_fwd')
    return y_pooled


def maxpool2x2_bwd_idx(argmax, y_pooled, dy, dx, relu_mask=True):
    """MaxPoolGrad (+ ReluGrad) from the argmax positions and pooled values of conv2d_pool_fwd; dx [n,h,w,c] dense,
    y_pooled / dy: last dim = pixel stride (>= c)."""
    n, h, w, c = dx.shape
    if y_pooled.dtype == torch.bfloat16 and dx.dtype == torch.bfloat16:      # bf16 storage throughout (conv2d_1 of config 5)
        assert dy.dtype == torch.bfloat16
        check(_lib.load().a3d_maxpool2x2_bwd_idx_bf16s(n, h, w, c, _ptr(argmax), _ptr(y_pooled), y_pooled.shape[-1], _ptr(dy),
                                                       dy.shape[-1], _ptr(dx), int(relu_mask), _stream()),
              'a3d_maxpool2x2_bwd_idx_bf16s')
        return dx
    if y_pooled.dtype == torch.bfloat16:         # bf16 storage: pooled values and dy bf16, dx float32
        assert dy.dtype == torch.bfloat16 and dx.dtype == torch.float32
        check(_lib.load().a3d_maxpool2x2_bwd_idx_bf16(n, h, w, c, _ptr(argmax), _ptr(y_pooled), y_pooled.shape[-1], _ptr(dy),
                                                      dy.shape[-1], _ptr(dx), int(relu_mask), _stream()),
              'a3d_maxpool2x2_bwd_idx_bf16')
        return dx
    check(_lib.load().a3d_maxpool2x2_bwd_idx(n, h, w, c, _ptr(argmax), _ptr(y_pooled), y_pooled.shape[-1], _ptr(dy),
                                             dy.shape[-1], _ptr(dx), int(relu_mask), _stream()),
          'a3d_maxpool2x2_bwd_idx')
    return dx


def copy_channel(src, c_src, dst, c_dst):
    """dst[..., c_dst] = src[..., c_src] (same pixel count; last dims are the pixel strides)."""
    npix = src.numel() // src.shape[-1]
    assert dst.numel() // dst.shape[-1] == npix
    fn = _lib.load().a3d_copy_channel_bf16 if dst.dtype == torch.bfloat16 else _lib.load().a3d_copy_channel
    check(fn(npix, _ptr(src), src.shape[-1], c_src, _ptr(dst), dst.shape[-1], c_dst, _stream()), 'a3d_copy_channel')
    return dst


def conv2d_bwd_data(d, dz, w, dx, relu_mask=None):
    lib = _lib.load()
    ws, n = _ws().get(lib.a3d_conv2d_bwd_data_ws_bytes(ctypes.byref(d)), dz.device)
    check(lib.a3d_conv2d_bwd_data(ctypes.byref(d), _ptr(dz), _ptr(w), _ptr(dx), _ptr(relu_mask), ws, n, _stream()),
          'a3d_conv2d_bwd_data')
    return dx


def conv2d_bwd_filter(d, x, dz, dw, db=None):
    lib = _lib.load()
    ws, n = _ws().get(lib.a3d_conv2d_bwd_filter_ws_bytes(ctypes.byref(d)), x.device)
    check(lib.a3d_conv2d_bwd_filter(ctypes.byref(d), _ptr(x), _ptr(dz), _ptr(dw), _ptr(db), ws, n, _stream()),
          'a3d_conv2d_bwd_filter')
    return dw, db


def conv2d_bwd_filter_pooled_supported(d):
    return _lib.load().a3d_conv2d_bwd_filter_pooled_ws_bytes(ctypes.byref(d)) > 0


def conv2d_bwd_filter_pooled(d, x, dpool, pooled, argmax, dw, db=None):
    """Filter and bias gradient of a conv -> ReLU -> 2x2 max pool block from the gradient of the POOLED map: MaxPoolGrad (by the
    recorded window positions) and ReluGrad (pooled > 0; pooled=None: none) happen while the kernel stages its operand."""
    lib = _lib.load()
    ws, n = _ws().get(lib.a3d_conv2d_bwd_filter_pooled_ws_bytes(ctypes.byref(d)), x.device)
    assert pooled is None or (pooled.dtype == dpool.dtype and pooled.shape[-1] == dpool.shape[-1])
    check(lib.a3d_conv2d_bwd_filter_pooled(ctypes.byref(d), _ptr(x), _ptr(dpool), dpool.shape[-1], _ptr(pooled), _ptr(argmax),
                                           argmax.shape[-1], int(dpool.dtype == torch.bfloat16), _ptr(dw), _ptr(db), ws, n,
                                           _stream()), 'a3d_conv2d_bwd_filter_pooled')
    return dw, db


_BOTH_STATE = {}


def conv2d_bwd_both_supported(d):
    return bool(_lib.load().a3d_conv2d_bwd_both_supported(ctypes.byref(d)))


def conv2d_bwd_both(d, x, dz, w, dw, db, dx, relu_mask=True):
    """Filter gradient, bias gradient and input gradient (times x > 0 if relu_mask) of a single-output-channel conv in one
    pass over x (a3d_conv2d_bwd_both); dx may be a bfloat16 tensor.  The launch's arrival counters live in a small zeroed
    buffer per stream (every call leaves it zero)."""
    lib = _lib.load()
    key = (torch.cuda.current_stream().cuda_stream, x.device)
    state = _BOTH_STATE.get(key)
    if state is None:
        state = _BOTH_STATE[key] = torch.zeros(64, dtype=torch.int32, device=x.device)
    ws, n = _ws().get(lib.a3d_conv2d_bwd_both_ws_bytes(ctypes.byref(d)), x.device)
    check(lib.a3d_conv2d_bwd_both(ctypes.byref(d), _ptr(x), _ptr(dz), _ptr(w), _ptr(dw), _ptr(db), _ptr(dx), dx.shape[-1],
                                  int(dx.dtype == torch.bfloat16), int(bool(relu_mask)), _ptr(state), ws, n, _stream()),
          'a3d_conv2d_bwd_both')
    return dw, db, dx


def dense_fwd(x, w, bias, y, act=None, drop_keep=None):
    m, k = x.shape
    n = w.shape[1]
    lib = _lib.load()
    ws, nb = _ws().get(lib.a3d_dense_fwd_ws_bytes(m, k, n), x.device)
    check(lib.a3d_dense_fwd(m, k, n, _ptr(x), _ptr(w), _ptr(bias), _ptr(y), ACT[act], _ptr(drop_keep), ws, nb,
                            _stream()), 'a3d_dense_fwd')
    return y


def dense_bwd_data(dz, w, dx, mask=None, scale=1.0, mask_act='relu'):
    """dx = dz @ w^T, optionally times the activation gradient of the layer below given its output `mask`."""
    m, n = dz.shape
    k = w.shape[0]
    lib = _lib.load()
    ws, nb = _ws().get(lib.a3d_dense_bwd_data_ws_bytes(m, k, n), dz.device)
    check(lib.a3d_dense_bwd_data(m, k, n, _ptr(dz), _ptr(w), _ptr(dx), _ptr(mask), ACT[mask_act], scale, ws, nb,
                                 _stream()),
          'a3d_dense_bwd_data')
    return dx


def dense_bwd_filter(x, dz, dw, db=None):
    m, k = x.shape
    n = dz.shape[1]
    lib = _lib.load()
    ws, nb = _ws().get(lib.a3d_dense_bwd_filter_ws_bytes(m, k, n), x.device)
    check(lib.a3d_dense_bwd_filter(m, k, n, _ptr(x), _ptr(dz), _ptr(dw), _ptr(db), ws, nb, _stream()),
          'a3d_dense_bwd_filter')
    return dw, db


def maxpool2x2_fwd(x, y, extra=None):
    """y[..., :c] = max_pool(x); if extra is given, y[..., c] = extra (fused concat).  y's last dim is its pixel
    stride."""
    n, h, w, c = x.shape
    check(_lib.load().a3d_maxpool2x2_fwd(n, h, w, c, _ptr(x), _ptr(y), y.shape[-1], _ptr(extra), _stream()),
          'a3d_maxpool2x2_fwd')
    return y


def maxpool2x2_bwd(x, dy, dx, relu_mask=True):
    """dy's last dim is its pixel stride (>= c): only its first c channels are read."""
    n, h, w, c = x.shape
    check(_lib.load().a3d_maxpool2x2_bwd(n, h, w, c, _ptr(x), _ptr(dy), dy.shape[-1], _ptr(dx), int(relu_mask),
                                         _stream()), 'a3d_maxpool2x2_bwd')
    return dx


def resize_bilinear_tf1(x, y):
    """x float32, or uint8 pixel values k of a converter-written record (the kernel reads fl(fl(fl(k/255) - .5) + .5))."""
    n, h, w, c = x.shape
    if x.dtype == torch.uint8:
        check(_lib.load().a3d_resize_bilinear_tf1_ex(n, h, w, c, _ptr(x), 1, y.shape[1], y.shape[2], _ptr(y), 0, None, 0, 0, 0,
                                                     None, _stream()), 'a3d_resize_bilinear_tf1_ex')
        return y
    check(_lib.load().a3d_resize_bilinear_tf1(n, h, w, c, _ptr(x), y.shape[1], y.shape[2], _ptr(y), _stream()),
          'a3d_resize_bilinear_tf1')
    return y


def resize_bilinear_tf1_pair(x0, y0, x1, y1):
    """Both resizes of a step in one launch: x0 -> y0 and x1 -> y1, stored tensors of the same batch and size; each
    float32 or uint8 (see resize_bilinear_tf1)."""
    n, h, w, c0 = x0.shape
    assert x1.shape[:3] == (n, h, w)
    if x0.dtype == torch.uint8 or x1.dtype == torch.uint8:
        check(_lib.load().a3d_resize_bilinear_tf1_ex(n, h, w, c0, _ptr(x0), int(x0.dtype == torch.uint8), y0.shape[1],
                                                     y0.shape[2], _ptr(y0), x1.shape[3], _ptr(x1),
                                                     int(x1.dtype == torch.uint8), y1.shape[1], y1.shape[2], _ptr(y1),
                                                     _stream()), 'a3d_resize_bilinear_tf1_ex')
        return
    check(_lib.load().a3d_resize_bilinear_tf1_pair(n, h, w, c0, _ptr(x0), y0.shape[1], y0.shape[2], _ptr(y0), x1.shape[3],
                                                   _ptr(x1), y1.shape[1], y1.shape[2], _ptr(y1), _stream()),
          'a3d_resize_bilinear_tf1_pair')


def extract_patches(x, k, stride, y):
    n, h, w, c = x.shape
    check(_lib.load().a3d_extract_patches(n, h, w, c, _ptr(x), k, stride, _ptr(y), _stream()), 'a3d_extract_patches')
    return y


SILOG_PARTS = 8                      # A3D_SILOG_PARTS (include/a3d.h)


def silog_ws(b, device):
    """Workspace of silog_loss_fwd / _bwd for a batch of b: A3D_SILOG_WS_FLOATS(b) zeros (the ticket must start at zero)."""
    return torch.zeros(2 * b + 1 + 2 * b * SILOG_PARTS, device=device)


def silog_loss_fwd(out, tgt, loss, ws):
    b = out.shape[0]
    assert ws.numel() >= 2 * b + 1 + 2 * b * SILOG_PARTS, 'silog workspace too small: allocate it with ops.silog_ws(b, device)'
    npix = out.numel() // b
    check(_lib.load().a3d_silog_loss_fwd(b, npix, _ptr(out), _ptr(tgt), _ptr(loss), _ptr(ws), _stream()),
          'a3d_silog_loss_fwd')
    return loss


def silog_loss_bwd(out, tgt, ws, dout, dout16=None):
    """dout16: optional bfloat16 [b, ld >= npix], receives the same gradient (columns npix.. stay as they are: keep them zero)."""
    b = out.shape[0]
    npix = out.numel() // b
    check(_lib.load().a3d_silog_loss_bwd_ex(b, npix, _ptr(out), _ptr(tgt), _ptr(ws), _ptr(dout), _ptr(dout16),
                                            0 if dout16 is None else dout16.shape[-1], _stream()), 'a3d_silog_loss_bwd_ex')
    return dout


def adam_apply_tf1(var, m, v, g, lr, beta1, beta2, eps, beta1_power, beta2_power, grad_scale=1.0, poisoned=None):
    """poisoned: optional int32[1] device tensor; bit 0 is set when the update left a non-finite weight behind."""
    if poisoned is not None:
        check(_lib.load().a3d_adam_apply_tf1_flag(var.numel(), _ptr(var), _ptr(m), _ptr(v), _ptr(g), lr, beta1, beta2, eps,
                                                  beta1_power, beta2_power, grad_scale, _ptr(poisoned), _stream()),
              'a3d_adam_apply_tf1_flag')
        return
    check(_lib.load().a3d_adam_apply_tf1(var.numel(), _ptr(var), _ptr(m), _ptr(v), _ptr(g), lr, beta1, beta2, eps,
                                         beta1_power, beta2_power, grad_scale, _stream()), 'a3d_adam_apply_tf1')


def dense_bwd_filter_adam_tf1(x, dz, var_w, m_w, v_w, var_b, m_b, v_b, lr, beta1, beta2, beta1_power, beta2_power,
                              grad_scale=1.0, precision='fp32'):
    """dense_bwd_filter + ApplyAdam(beta2 = 1) of one dense layer in one pass; the gradient is not materialised.
    precision 'bf16': x and dz rounded to bf16 for the contraction (batches above 32 rows; config 5)."""
    m, k = x.shape
    n = dz.shape[1]
    check(_lib.load().a3d_dense_bwd_filter_adam_tf1_ex(m, k, n, _ptr(x), _ptr(dz), _ptr(var_w), _ptr(m_w), _ptr(v_w),
                                                       _ptr(var_b), _ptr(m_b), _ptr(v_b), lr, beta1, beta2, beta1_power,
                                                       beta2_power, grad_scale, PREC[precision], _stream()),
          'a3d_dense_bwd_filter_adam_tf1_ex')


def with_storage(d, storage):
    """Copy of a conv descriptor with other storage bits (forward / bwd-data / bwd-filter mark different tensors)."""
    e = ConvDesc()
    ctypes.pointer(e)[0] = d
    e.storage = storage
    return e


def _dense_ws(lib, m, k, n, precision, storage, device):
    d = conv_desc(m, 1, 1, k, n, 1, 1, 1, 'VALID', precision=precision, storage=storage)
    need = max(lib.a3d_conv2d_fwd_ws_bytes(ctypes.byref(d)), lib.a3d_conv2d_bwd_data_ws_bytes(ctypes.byref(d)))
    return _ws().get(need, device)


def dense_fwd_ex(x, w, bias, y, act=None, drop_keep=None, precision='fp32', storage=0, out2=None, n=None):
    """dense_fwd with the arithmetic / weight storage of BASELINE config 5 (w may be the layer's bf16 copy).  n: the GEMM's
    column count where it is wider than y (w padded to whole 16-byte pieces): y then receives its own y.shape[1] columns at its
    own pitch; out2: a3d_second_output (second_output())."""
    m, k = x.shape
    ncols = y.shape[1]
    n = ncols if n is None else n
    lib = _lib.load()
    ws, nb = _dense_ws(lib, m, k, n, precision, storage, x.device)
    check(lib.a3d_dense_fwd_ex2(m, k, n, _ptr(x), _ptr(w), _ptr(bias), _ptr(y), ncols, ncols, ACT[act], _ptr(drop_keep),
                                PREC[precision], storage, _o2(out2), ws, nb, _stream()), 'a3d_dense_fwd_ex2')
    return y


def dense_bwd_data_ex(dz, w, dx, mask=None, mask_act='relu', scale=1.0, precision='fp32', storage=0, out2=None):
    m, n = dz.shape
    k = dx.shape[1]
    lib = _lib.load()
    ws, nb = _dense_ws(lib, m, k, n, precision, storage, dz.device)
    check(lib.a3d_dense_bwd_data_ex2(m, k, n, _ptr(dz), _ptr(w), _ptr(dx), _ptr(mask), ACT[mask_act], scale, PREC[precision],
                                     storage, _o2(out2), ws, nb, _stream()), 'a3d_dense_bwd_data_ex2')
    return dx


def pad_channels_bf16(src, dst):
    """float32 [..., c] -> bfloat16 [..., 4], missing channels zero."""
    assert src.dtype == torch.float32 and dst.dtype == torch.bfloat16 and dst.shape[-1] == 4 and src.shape[:-1] == dst.shape[:-1]
    check(_lib.load().a3d_pad_channels_bf16(src.numel() // src.shape[-1], src.shape[-1], _ptr(src), 4, _ptr(dst), _stream()),
          'a3d_pad_channels_bf16')
    return dst


def cast_rows(src, dst, cols=None):
    """dst[r, :cols] = src[r, :cols] (float32 / bfloat16 on either side), dst[r, cols:] = 0: 2-D tensors whose last
    dimensions are their row pitches."""
    rows = src.shape[0]
    cols = cols if cols is not None else min(src.shape[1], dst.shape[1])
    assert dst.shape[0] == rows and src.dim() == dst.dim() == 2 and src.is_contiguous() and dst.is_contiguous()
    check(_lib.load().a3d_cast_rows(rows, cols, _ptr(src), src.shape[1], int(src.dtype == torch.bfloat16), _ptr(dst),
                                    dst.shape[1], int(dst.dtype == torch.bfloat16), _stream()), 'a3d_cast_rows')
    return dst


def cast_bf16(src, dst):
    """float32 -> bfloat16 or back, by dst's dtype (round to nearest even)."""
    assert src.numel() == dst.numel() and {src.dtype, dst.dtype} == {torch.float32, torch.bfloat16}
    check(_lib.load().a3d_cast_bf16(src.numel(), _ptr(src), _ptr(dst), int(dst.dtype == torch.bfloat16), _stream()),
          'a3d_cast_bf16')
    return dst


def maxpool2x2_fwd_bf16(x, y, extra=None, c=None):
    """bf16 tensors; x's and y's last dims are their pixel strides, c (default: x's) the channels pooled."""
    n, h, w, ldx = x.shape
    check(_lib.load().a3d_maxpool2x2_fwd_bf16(n, h, w, c or ldx, _ptr(x), ldx, _ptr(y), y.shape[-1], _ptr(extra), _stream()),
          'a3d_maxpool2x2_fwd_bf16')
    return y


def maxpool2x2_bwd_bf16(x, dy, dx, relu_mask=True, c=None):
    n, h, w, ldx = x.shape
    assert dx.shape == x.shape
    check(_lib.load().a3d_maxpool2x2_bwd_bf16(n, h, w, c or ldx, _ptr(x), ldx, _ptr(dy), dy.shape[-1], _ptr(dx),
                                              int(relu_mask), _stream()), 'a3d_maxpool2x2_bwd_bf16')
    return dx


def sgd_apply(var, g, lr):
    """tf.train.GradientDescentOptimizer (src/models.py:198)."""
    check(_lib.load().a3d_sgd_apply(var.numel(), _ptr(var), _ptr(g), lr, _stream()), 'a3d_sgd_apply')


def superpixel_mean(x, sp, out=None):
    """[n,h,w,c] -> [n,(h/sp)*(w/sp),c] block means (src/models.py:110,132)."""
    n, h, w, c = x.shape
    out = out if out is not None else torch.empty((n, (h // sp) * (w // sp), c), dtype=torch.float32, device=x.device)
    check(_lib.load().a3d_superpixel_mean(n, h, w, c, _ptr(x), sp, _ptr(out), _stream()), 'a3d_superpixel_mean')
    return out


def superpixel_hist(x, sp, out=None):
    """color_histogram of every superpixel (src/models.py:95-100): [n,h,w,3] -> [n,P,256]."""
    n, h, w, c = x.shape
    assert c == 3
    out = out if out is not None else torch.empty((n, (h // sp) * (w // sp), 256), dtype=torch.float32, device=x.device)
    check(_lib.load().a3d_superpixel_hist(n, h, w, _ptr(x), sp, _ptr(out), _stream()), 'a3d_superpixel_hist')
    return out


def pair_similarity(x, sp, hist, left, right, dense_w, dense_b, gamma=1.0):
    """pairwise_part (src/models.py:108-127): returns (sims [n,Q,2], r [n,Q])."""
    n, h, w, _ = x.shape
    q = left.numel()
    sims = torch.empty((n, q, 2), dtype=torch.float32, device=x.device)
    r = torch.empty((n, q), dtype=torch.float32, device=x.device)
    check(_lib.load().a3d_pair_similarity(n, h, w, _ptr(x), sp, _ptr(hist), _ptr(left), _ptr(right), q, _ptr(dense_w),
                                          _ptr(dense_b), gamma, _ptr(sims), _ptr(r), _stream()), 'a3d_pair_similarity')
    return sims, r


def crf_loss(z, y, r, left, right, eps=1e-7):
    """loss_part (src/models.py:129-177): returns (mean loss [1], per-image loss [n], d mean / d z [n,P])."""
    n, nsp = z.shape[0], z.shape[1]
    per = torch.empty((n,), dtype=torch.float32, device=z.device)
    mean = torch.empty((1,), dtype=torch.float32, device=z.device)
    dz = torch.empty((n, nsp), dtype=torch.float32, device=z.device)
    check(_lib.load().a3d_crf_loss(n, nsp, _ptr(z), _ptr(y), _ptr(r), _ptr(left), _ptr(right), left.numel(), eps,
                                   _ptr(per), _ptr(mean), _ptr(dz), _stream()), 'a3d_crf_loss')
    return mean, per, dz


def dropout_keep_mask(keep, seed, step, rate=0.5):
    """Fill the uint8 tensor `keep` with the Bernoulli(1-rate) keep mask of training step `step`."""
    check(_lib.load().a3d_dropout_keep_mask(keep.numel(), seed, step, rate, _ptr(keep), _stream()),
          'a3d_dropout_keep_mask')
    return keep
