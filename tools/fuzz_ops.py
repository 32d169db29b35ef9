"""Randomised cross-check of the conv / dense entry points against torch float64 on the GPU box (not part of the test
suite; the shapes that ever failed live in tests/test_gpu_ops.py).     python tools/fuzz_ops.py [seconds] [seed]
Shapes are drawn to hit the planner's corners: odd sizes, K tails, split-K factors that are not powers of two, strides
1/2/4, SAME and VALID, channel counts that are / are not multiples of 4, and problems big enough for every tile class."""
import os
os.environ.setdefault('A3D_TUNING', '1')   # the library reads its A3D_FORCE_* switches per launch only then
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ann3depth_amd import ops  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
R5 = float(os.environ.get('FUZZ_R5', '5'))       # weight of round 5's kernel classes (x 5 %): FUZZ_R5=19 fuzzes almost only those
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
TOL = 3e-6
TOL_POOL = 2e-5
TOL_BF16_OUT, TOL_BF16_DW = 4e-3, 1e-4      # bf16 storage: an output rounded to bf16 (2^-9 per element); fp32 accumulation


def rel(a, b, scale=None):
    """||a - b|| relative to ||scale||, where `scale` is the same contraction over absolute values (no cancellation):
    the yardstick fp32 accumulation error is proportional to.  Defaults to ||b||."""
    return float((a.double() - b).norm() / max(float((b if scale is None else scale).norm()), 1e-30))


def conv_case():
    st = int(rng.choice([1, 1, 1, 2, 2, 4]))
    ks = int(rng.choice([1, 2, 3, 3, 5, 7, 9, 11]))
    pad = str(rng.choice(['SAME', 'VALID']))
    c = int(rng.choice([1, 3, 3, 4, 5, 8, 16, 24, 63, 64, 96, 130]))
    k = int(rng.choice([1, 2, 4, 7, 16, 48, 63, 64, 96, 128, 200]))
    n = int(rng.integers(1, 40))
    h = int(rng.integers(ks, 70))
    w = int(rng.integers(ks, 70))
    if n * h * w * max(c, k) > 3e7:
        n = max(1, int(3e7 / (h * w * max(c, k))))
    return n, h, w, c, k, ks, st, pad


def run_conv(n, h, w, c, k, ks, st, pad):
    d = ops.conv_desc(n, h, w, c, k, ks, ks, st, pad)
    if d.ho <= 0 or d.wo <= 0:
        return 0.0
    g = torch.Generator(device='cuda').manual_seed(int(rng.integers(1 << 30)))
    x = torch.randn((n, h, w, c), device='cuda', generator=g)
    wt = torch.randn((ks, ks, c, k), device='cuda', generator=g) / np.sqrt(ks * ks * c)
    b = torch.randn((k,), device='cuda', generator=g)
    y = torch.full((n, d.ho, d.wo, k), float('nan'), device='cuda')
    ops.conv2d_fwd(d, x, wt, b, y, None)
    # torch float64 reference with TF padding
    out_h, out_w = d.ho, d.wo
    ph = max((out_h - 1) * st + ks - h, 0) if pad == 'SAME' else 0
    pw = max((out_w - 1) * st + ks - w, 0) if pad == 'SAME' else 0
    xd = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    wd = wt.double().permute(3, 2, 0, 1).contiguous().requires_grad_(True)
    xp = F.pad(xd, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
    ref = F.conv2d(xp, wd, b.double(), stride=st)
    assert ref.shape[2:] == (out_h, out_w), (ref.shape, out_h, out_w)
    xa = x.double().abs().permute(0, 3, 1, 2).requires_grad_(True)
    wa = wt.double().abs().permute(3, 2, 0, 1).contiguous().requires_grad_(True)
    ref_abs = F.conv2d(F.pad(xa, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2)), wa, b.double().abs(), stride=st)
    err = rel(y.permute(0, 3, 1, 2), ref.detach(), ref_abs.detach())
    dz = torch.randn((n, d.ho, d.wo, k), device='cuda', generator=g)
    gx, gw = torch.autograd.grad(ref, [xd, wd], dz.double().permute(0, 3, 1, 2))
    gxa, gwa = torch.autograd.grad(ref_abs, [xa, wa], dz.double().abs().permute(0, 3, 1, 2))
    dw = torch.full_like(wt, float('nan'))
    db = torch.full_like(b, float('nan'))
    ops.conv2d_bwd_filter(d, x, dz, dw, db)
    err = max(err, rel(dw, gw.permute(2, 3, 1, 0), gwa.permute(2, 3, 1, 0)),
              rel(db, dz.double().sum((0, 1, 2)), dz.double().abs().sum((0, 1, 2))))
    dx = torch.full_like(x, float('nan'))
    ops.conv2d_bwd_data(d, dz, wt, dx)
    err = max(err, rel(dx, gx.permute(0, 2, 3, 1), gxa.permute(0, 2, 3, 1)) if float(gxa.norm()) > 0
              else float(dx.abs().max()))
    return err


def run_conv_bf16s(n, h, w, c, k, ks, st, pad):
    """bf16 storage (BASELINE config 5): x, the filter copy, y, dz and dx bf16, the filter gradient fp32 — against torch
    float64 on the bf16-ROUNDED operands.  -> (error of the bf16 outputs, error of the fp32 filter gradient)"""
    bf = torch.bfloat16
    d = ops.conv_desc(n, h, w, c, k, ks, ks, st, pad, precision='bf16')
    if d.ho <= 0 or d.wo <= 0:
        return 0.0, 0.0
    g = torch.Generator(device='cuda').manual_seed(int(rng.integers(1 << 30)))
    x = torch.randn((n, h, w, c), device='cuda', generator=g).to(bf)
    wt = (torch.randn((ks, ks, c, k), device='cuda', generator=g) / np.sqrt(ks * ks * c)).to(bf)
    b = torch.randn((k,), device='cuda', generator=g)
    X, W, Y = ops.STORE_X, ops.STORE_W, ops.STORE_Y
    y = torch.full((n, d.ho, d.wo, k), float('nan'), device='cuda', dtype=bf)
    ops.conv2d_fwd(ops.with_storage(d, X | W | Y), x, wt, b, y, None)
    ph = max((d.ho - 1) * st + ks - h, 0) if pad == 'SAME' else 0
    pw = max((d.wo - 1) * st + ks - w, 0) if pad == 'SAME' else 0
    xd = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    wd = wt.double().permute(3, 2, 0, 1).contiguous().requires_grad_(True)
    ref = F.conv2d(F.pad(xd, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2)), wd, b.double(), stride=st)
    e16 = rel(y.permute(0, 3, 1, 2), ref.detach())
    dz = torch.randn((n, d.ho, d.wo, k), device='cuda', generator=g).to(bf)
    gx, gw = torch.autograd.grad(ref, [xd, wd], dz.double().permute(0, 3, 1, 2))
    dw = torch.full(wt.shape, float('nan'), device='cuda')
    db = torch.full_like(b, float('nan'))
    ops.conv2d_bwd_filter(ops.with_storage(d, X | Y), x, dz, dw, db)
    e32 = max(rel(dw, gw.permute(2, 3, 1, 0)) if float(gw.norm()) > 0 else float(dw.abs().max()),
              rel(db, dz.double().sum((0, 1, 2))))
    dx = torch.full(x.shape, float('nan'), device='cuda', dtype=bf)
    ops.conv2d_bwd_data(ops.with_storage(d, X | W | Y), dz, wt, dx)
    e16 = max(e16, rel(dx, gx.permute(0, 2, 3, 1)) if float(gx.norm()) > 0 else float(dx.float().abs().max()))
    return e16, e32


def conv_case_bf16s():
    st = int(rng.choice([1, 1, 2, 2]))
    ks = int(rng.choice([1, 2, 3, 3, 5]))
    pad = str(rng.choice(['SAME', 'VALID']))
    c = int(rng.choice([8, 16, 24, 64, 96, 128, 256]))
    k = int(rng.choice([8, 16, 48, 64, 96, 128, 200, 256]))
    n = int(rng.integers(1, 48))
    h = int(rng.integers(ks, 40))
    w = int(rng.integers(ks, 40))
    if n * h * w * max(c, k) > 2e7:
        n = max(1, int(2e7 / (h * w * max(c, k))))
    return n, h, w, c, k, ks, st, pad


def run_conv_pool(n, h, w, c, k, ks, st, pad):
    """conv + ReLU + 2x2 max pool in one launch (the headline step's conv2d_0 / conv2d_1 / fine/first) against the two
    launches (whose conv may add its K range in another order: split-K, stream-K): the pooled values to fp32 accumulation
    error, and the recorded position must hold the window's maximum (to the same error; ties may resolve differently when
    the two sums differ in their last bits)."""
    d = ops.conv_desc(n, h, w, c, k, ks, ks, st, pad)
    if d.ho < 2 or d.wo < 2 or k == 1:
        return 0.0
    g = torch.Generator(device='cuda').manual_seed(int(rng.integers(1 << 30)))
    x = torch.randn((n, h, w, c), device='cuda', generator=g)
    wt = torch.randn((ks, ks, c, k), device='cuda', generator=g) / np.sqrt(ks * ks * c)
    b = torch.randn((k,), device='cuda', generator=g)
    y = torch.empty((n, d.ho, d.wo, k), device='cuda')
    ops.conv2d_fwd(d, x, wt, b, y, 'relu')
    ph, pw = d.ho // 2, d.wo // 2
    pooled = torch.full((n, ph, pw, k), float('nan'), device='cuda')
    arg = torch.full((n, ph, pw, k), 9, device='cuda', dtype=torch.uint8)
    ops.conv2d_pool_fwd(d, x, wt, b, pooled, 'relu', arg)
    win = y[:, :2 * ph, :2 * pw].reshape(n, ph, 2, pw, 2, k).permute(0, 1, 3, 5, 2, 4).reshape(n, ph, pw, k, 4)
    want = win.max(-1).values.double()
    if int(arg.max()) > 3 or not bool(torch.isfinite(pooled).all()):
        return 1.0
    at_arg = win.gather(-1, arg.long().unsqueeze(-1)).squeeze(-1).double()
    scale = max(float(y.double().norm()) / np.sqrt(y.numel()), 1e-30) * np.sqrt(want.numel())     # typical magnitude
    return max(float((pooled.double() - want).norm()), float((at_arg - want).norm())) / scale


def run_conv_pool_bf16s(n, h, w, c, k, ks, st, pad):
    """bf16 storage with the pool fused (config 5's conv2d_1: x, w, pooled y bf16, the LDS-DMA kernel): the pooled map against
    torch float64 on the rounded operands to bf16 rounding; the recorded position must hold a value within that rounding of
    the window's maximum; and the by-index MaxPoolGrad to a bf16 gradient must put exactly dy (or 0 under the ReLU mask) at
    that position and zeros elsewhere.  stride 1 only, k a multiple of 16 (whole 16-byte pieces of argmax bytes)."""
    bf = torch.bfloat16
    k = max(16, k // 16 * 16)
    d = ops.conv_desc(n, h, w, c, k, ks, ks, 1, pad, precision='bf16')
    if d.ho < 2 or d.wo < 2:
        return 0.0
    g = torch.Generator(device='cuda').manual_seed(int(rng.integers(1 << 30)))
    x = torch.randn((n, h, w, c), device='cuda', generator=g).to(bf)
    wt = (torch.randn((ks, ks, c, k), device='cuda', generator=g) / np.sqrt(ks * ks * c)).to(bf)
    b = torch.randn((k,), device='cuda', generator=g) * 0.1
    X, W, Y = ops.STORE_X, ops.STORE_W, ops.STORE_Y
    ph, pw = d.ho // 2, d.wo // 2
    pooled = torch.full((n, ph, pw, k), float('nan'), device='cuda', dtype=bf)
    arg = torch.full((n, ph, pw, k), 9, device='cuda', dtype=torch.uint8)
    ops.conv2d_pool_fwd(ops.with_storage(d, X | W | Y), x, wt, b, pooled, 'relu', arg)
    if int(arg.max()) > 3 or not bool(torch.isfinite(pooled.float()).all()):
        return 1.0
    pt = max((d.ho - 1) + ks - h, 0) if pad == 'SAME' else 0
    pl = max((d.wo - 1) + ks - w, 0) if pad == 'SAME' else 0
    ref = F.conv2d(F.pad(x.double().permute(0, 3, 1, 2), (pl // 2, pl - pl // 2, pt // 2, pt - pt // 2)),
                   wt.double().permute(3, 2, 0, 1).contiguous(), b.double(), stride=1).clamp_min(0).permute(0, 2, 3, 1)
    win = ref[:, :2 * ph, :2 * pw].reshape(n, ph, 2, pw, 2, k).permute(0, 1, 3, 5, 2, 4).reshape(n, ph, pw, k, 4)
    want = win.max(-1).values
    at_arg = win.gather(-1, arg.long().unsqueeze(-1)).squeeze(-1)
    scale = max(float(want.norm()), 1e-30)
    err = max(float((pooled.double() - want).norm()), float((at_arg - want).norm())) / scale
    dy = torch.randn((n, ph, pw, k), device='cuda', generator=g).to(bf)
    dx = torch.full((n, d.ho, d.wo, k), float('nan'), device='cuda', dtype=bf)
    ops.maxpool2x2_bwd_idx(arg, pooled, dy, dx, relu_mask=True)
    routed = torch.where(pooled.float() > 0, dy.float(), torch.zeros_like(dy.float()))
    exp = torch.zeros((n, ph, pw, k, 4), device='cuda')
    exp.scatter_(-1, arg.long().unsqueeze(-1), routed.unsqueeze(-1))
    full = torch.zeros((n, d.ho, d.wo, k), device='cuda')
    full[:, :2 * ph, :2 * pw] = exp.reshape(n, ph, pw, k, 2, 2).permute(0, 1, 4, 2, 5, 3).reshape(n, 2 * ph, 2 * pw, k)
    if not torch.equal(dx.float(), full):
        return 1.0
    return err


def run_dense_adam():
    """a3d_dense_bwd_filter_adam_tf1 (the gradient never written; the reference's frozen optimizer) against
    a3d_dense_bwd_filter + a3d_adam_apply_tf1 over two steps: m, v, var bit for bit (same MFMA sum order, the same
    separate fp32 operations).  Batches of at most 64 rows, k * n >= 64 k (below that the two-pass path is a GEMM with
    another sum order).  -> 0.0 or 1.0"""
    m = int(rng.integers(1, 65))
    k = int(rng.choice([64, 96, 130, 257, 512, 1028, 4488]))
    n = int(rng.choice([64, 66, 257, 520, 772, 1030, 1538, 4070]))
    while k * n < 65536:
        k *= 2
    g = torch.Generator(device='cuda').manual_seed(int(rng.integers(1 << 30)))
    x = torch.randn((m, k), device='cuda', generator=g)
    fused = [torch.randn((k, n), device='cuda', generator=g), torch.zeros((k, n), device='cuda'),
             torch.rand((k, n), device='cuda', generator=g) * 0.01, torch.randn((n,), device='cuda', generator=g),
             torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')]
    plain = [t.clone() for t in fused]
    b1p, ok = np.float32(0.9), True
    for step in range(2):
        dz = torch.randn((m, n), device='cuda', generator=g)
        if step == 1:
            dz[0, 3 % n] = float('inf')                       # the poison path (v, var take a NaN)
        ops.dense_bwd_filter_adam_tf1(x, dz, *fused, 0.1, 0.9, 1.0, float(b1p), 1.0, 0.5)
        dw, db = torch.empty((k, n), device='cuda'), torch.empty(n, device='cuda')
        ops.dense_bwd_filter(x, dz, dw, db)
        ops.adam_apply_tf1(plain[0], plain[1], plain[2], dw, 0.1, 0.9, 1.0, 1e-8, float(b1p), 1.0, 0.5)
        ops.adam_apply_tf1(plain[3], plain[4], plain[5], db, 0.1, 0.9, 1.0, 1e-8, float(b1p), 1.0, 0.5)
        b1p = b1p * np.float32(0.9)
        for a, b in zip(fused, plain):
            ok &= bool(torch.equal(torch.nan_to_num(a, nan=12345.0), torch.nan_to_num(b, nan=12345.0)))
    return (m, k, n), 0.0 if ok else 1.0


def run_dense():
    m = int(rng.choice([1, 2, 5, 16, 32, 33, 64, 200]))
    k = int(rng.choice([1, 3, 16, 100, 128, 1000, 4096, 12288]))
    n = int(rng.choice([1, 2, 16, 63, 128, 1000, 4070, 4096]))
    print('dense case', m, k, n, flush=True) if os.environ.get('FUZZ_VERBOSE') else None
    g = torch.Generator(device='cuda').manual_seed(int(rng.integers(1 << 30)))
    x = torch.randn((m, k), device='cuda', generator=g)
    w = torch.randn((k, n), device='cuda', generator=g) / np.sqrt(k)
    b = torch.randn((n,), device='cuda', generator=g)
    y = torch.full((m, n), float('nan'), device='cuda')
    ops.dense_fwd(x, w, b, y, 'relu')
    xa, wa, za = x.double().abs(), w.double().abs(), None
    err = rel(y, torch.relu(x.double() @ w.double() + b.double()), xa @ wa + b.double().abs())
    dz = torch.randn((m, n), device='cuda', generator=g)
    za = dz.double().abs()
    dx = torch.full_like(x, float('nan'))
    ops.dense_bwd_data(dz, w, dx)
    err = max(err, rel(dx, dz.double() @ w.double().t(), za @ wa.t()))
    dw = torch.full_like(w, float('nan'))
    db = torch.full_like(b, float('nan'))
    ops.dense_bwd_filter(x, dz, dw, db)
    err = max(err, rel(dw, x.double().t() @ dz.double(), xa.t() @ za), rel(db, dz.double().sum(0), za.sum(0)))
    return (m, k, n), err


RAN = {}


def ran(what):
    RAN[what] = RAN.get(what, 0) + 1


def fewch_case():
    """Few-channel VALID convolutions the LDS-staged filter-gradient kernels take (fewch.hip / fewch16.hip): 1..4 channels,
    S*C >= 27, 33..96 filters, input rows of whole 16-byte pieces, rows short enough for the staging registers."""
    c = int(rng.choice([3, 3, 3, 4, 2]))
    ks = int(rng.choice([k for k in (7, 9, 9, 11, 11, 13, 14) if k * c >= 27]))
    st = int(rng.choice([1, 2, 2, 4, 4]))
    k = int(rng.choice([33, 40, 48, 63, 64, 80, 96]))
    while True:
        w = int(rng.integers(ks + 2 * st, 320))
        if (w * c) % 4 == 0:
            break
    h = int(rng.integers(ks + 2 * st, 120))
    n = int(rng.integers(1, 9))
    return n, h, w, c, k, ks, st


def run_fewch(n, h, w, c, k, ks, st):
    """a3d_conv2d_bwd_filter (plain) and a3d_conv2d_bwd_filter_pooled (MaxPoolGrad + ReluGrad fused; float32 and bf16 pooled
    tensors; fp32 and bf16 arithmetic) against torch float64 on the gradient the separate path would materialise.
    -> error relative to TOL (bf16 arithmetic: against the oracle on the bf16-rounded image, same yardstick)"""
    bf = torch.bfloat16
    d = ops.conv_desc(n, h, w, c, k, ks, ks, st, 'VALID')
    if d.ho < 2 or d.wo < 2:
        return 0.0
    g = torch.Generator(device='cuda').manual_seed(int(rng.integers(1 << 30)))
    x = torch.randn((n, h, w, c), device='cuda', generator=g)
    ph, pw = d.ho // 2, d.wo // 2
    ld = int(rng.choice([k, (k + 7) // 8 * 8, (k + 3) // 4 * 4 + 4]))
    ld = (ld + 3) // 4 * 4
    lda = int(rng.choice([k, ld]))
    pooled = torch.randn((n, ph, pw, ld), device='cuda', generator=g)
    dpool = torch.randn((n, ph, pw, ld), device='cuda', generator=g)
    arg = torch.randint(0, 4, (n, ph, pw, lda), device='cuda', generator=g, dtype=torch.uint8)

    def reference(xs, dp, pl):
        gsel = torch.where(pl[..., :k] > 0, dp[..., :k], torch.zeros_like(dp[..., :k])).double()
        exp = torch.zeros((n, ph, pw, k, 4), device='cuda', dtype=torch.float64)
        exp.scatter_(-1, arg[..., :k].long().unsqueeze(-1), gsel.unsqueeze(-1))
        dz = torch.zeros((n, d.ho, d.wo, k), device='cuda', dtype=torch.float64)
        dz[:, :2 * ph, :2 * pw] = exp.reshape(n, ph, pw, k, 2, 2).permute(0, 1, 4, 2, 5, 3).reshape(n, 2 * ph, 2 * pw, k)
        xd = xs.double().permute(0, 3, 1, 2)
        wd = torch.zeros((k, c, ks, ks), device='cuda', dtype=torch.float64, requires_grad=True)
        ref = F.conv2d(xd, wd, stride=st)
        gw, = torch.autograd.grad(ref, [wd], dz.permute(0, 3, 1, 2))
        wa = torch.zeros_like(wd).requires_grad_(True)
        gwa, = torch.autograd.grad(F.conv2d(xd.abs(), wa, stride=st), [wa], dz.abs().permute(0, 3, 1, 2))
        return dz, gw.permute(2, 3, 1, 0), gwa.permute(2, 3, 1, 0)

    err = 0.0
    dw = torch.full((ks, ks, c, k), float('nan'), device='cuda')
    db = torch.full((k,), float('nan'), device='cuda')
    if ops.conv2d_bwd_filter_pooled_supported(d):
        ran('fewch fp32 arithmetic (pooled f32 + bf16 sources, plain)')
        for dt in (torch.float32, bf):
            dp, pl = dpool.to(dt), pooled.to(dt)
            dz, gw, gwa = reference(x, dp.float(), pl.float())
            ops.conv2d_bwd_filter_pooled(d, x, dp, pl, arg, dw.fill_(float('nan')), db.fill_(float('nan')))
            err = max(err, rel(dw, gw, gwa), rel(db, dz.sum((0, 1, 2)), dz.abs().sum((0, 1, 2))))
        # the plain form on the materialised gradient
        dzf = dz.float()
        dz64, gw, gwa = reference(x, dpool.to(bf).float(), pooled.to(bf).float())
        ops.conv2d_bwd_filter(d, x, dzf, dw.fill_(float('nan')), db.fill_(float('nan')))
        err = max(err, rel(dw, gw, gwa))
    d16 = ops.conv_desc(n, h, w, c, k, ks, ks, st, 'VALID', precision='bf16')
    if ops.conv2d_bwd_filter_pooled_supported(d16):
        ran('fewch16 bf16 arithmetic')
        dp, pl = dpool.to(bf), pooled.to(bf)
        dz, gw, gwa = reference(x.to(bf), dp.float(), pl.float())
        ops.conv2d_bwd_filter_pooled(d16, x, dp, pl, arg, dw.fill_(float('nan')), db.fill_(float('nan')))
        err = max(err, rel(dw, gw, gwa), rel(db, dz.sum((0, 1, 2)), dz.abs().sum((0, 1, 2))))
    return err


def run_bwd_both():
    """a3d_conv2d_bwd_both (one output channel, 5x5: filter, bias and input gradient + ReluGrad in one pass) against torch
    float64; float32 and bf16 dx; twice on one state buffer, bit-identical."""
    bf = torch.bfloat16
    c = int(rng.choice([2, 8, 24, 40, 64, 64, 64]))
    ldx = c + int(rng.choice([0, 0, 2, 8]))
    n, h, w = int(rng.integers(1, 12)), int(rng.integers(5, 60)), int(rng.integers(5, 78))
    pad = str(rng.choice(['SAME', 'VALID']))
    d = ops.conv_desc(n, h, w, c, 1, 5, 5, 1, pad, ldx=ldx)
    if d.ho < 1 or d.wo < 1 or not ops.conv2d_bwd_both_supported(d):
        return (n, h, w, c, pad), 0.0
    g = torch.Generator(device='cuda').manual_seed(int(rng.integers(1 << 30)))
    xb = torch.randn((n, h, w, ldx), device='cuda', generator=g)
    wt = torch.randn((5, 5, c, 1), device='cuda', generator=g) / np.sqrt(25 * c)
    dz = torch.randn((n, d.ho, d.wo, 1), device='cuda', generator=g)
    pt = 2 if pad == 'SAME' else 0
    xd = xb[..., :c].double().permute(0, 3, 1, 2).requires_grad_(True)
    wd = wt.double().permute(3, 2, 0, 1).contiguous().requires_grad_(True)
    ref = F.conv2d(F.pad(xd, (pt, pt, pt, pt)), wd)
    gx, gw = torch.autograd.grad(ref, [xd, wd], dz.double().permute(0, 3, 1, 2))
    xa = xb[..., :c].double().abs().permute(0, 3, 1, 2).requires_grad_(True)
    wa = wt.double().abs().permute(3, 2, 0, 1).contiguous().requires_grad_(True)
    gxa, gwa = torch.autograd.grad(F.conv2d(F.pad(xa, (pt, pt, pt, pt)), wa), [xa, wa], dz.double().abs().permute(0, 3, 1, 2))
    mask = (xb[..., :c] > 0).double()
    err = 0.0
    ran('bwd_both')
    for dt, tol_scale in ((torch.float32, 1.0), (bf, TOL / TOL_BF16_OUT)):
        outs = []
        for _ in range(2):
            dw = torch.full_like(wt, float('nan'))
            db = torch.full((1,), float('nan'), device='cuda')
            dx = torch.full((n, h, w, c), float('nan'), device='cuda', dtype=dt)
            ops.conv2d_bwd_both(d, xb, dz, wt, dw, db, dx, relu_mask=True)
            outs.append((dw, db, dx))
        if not all(torch.equal(a, b) for a, b in zip(outs[0], outs[1])):
            return (n, h, w, c, pad), 1.0
        dw, db, dx = outs[0]
        err = max(err, rel(dw, gw.permute(2, 3, 1, 0), gwa.permute(2, 3, 1, 0)), rel(db, dz.double().sum().reshape(1), dz.double().abs().sum().reshape(1)),
                  rel(dx.float(), gx.permute(0, 2, 3, 1) * mask, gxa.permute(0, 2, 3, 1)) * tol_scale)
    return (n, h, w, c, pad), err


def run_image_form(n, h, w, c, k, ks, st):
    """The bf16 image form of the few-channel forward (conv3b_fwd_kernel for >= 33 filters): 4-channel bf16 image, with and
    without the fused pool and a prepared filter, against torch float64 on the rounded operands (outputs bf16: 4e-3)."""
    bf = torch.bfloat16
    if st % 2 or c > 3:
        return 0.0
    ldy = (k + 7) // 8 * 8
    d = ops.with_storage(ops.conv_desc(n, h, w, 4, k, ks, ks, st, 'VALID', ldy=ldy, precision='bf16'), ops.STORE_X | ops.STORE_Y)
    if d.ho < 2 or d.wo < 2:
        return 0.0
    g = torch.Generator(device='cuda').manual_seed(int(rng.integers(1 << 30)))
    x3 = torch.rand((n, h, w, 3), device='cuda', generator=g)
    x4 = torch.empty((n, h, w, 4), device='cuda', dtype=bf)
    ops.pad_channels_bf16(x3, x4)
    w4 = torch.zeros((ks, ks, 4, k), device='cuda')
    w4[:, :, :3] = torch.randn((ks, ks, 3, k), device='cuda', generator=g) / np.sqrt(ks * ks * 3)
    w4[:, :, 3] = 5.0                                   # the pad channel of the filter: anything (its pixels are zero)
    b = torch.randn((k,), device='cuda', generator=g) * 0.1
    ran('bf16 image form, >= 33 filters (conv3b)' if k >= 33 else 'bf16 image form (igemm_bf16)')
    ref = F.conv2d(x4[..., :3].double().permute(0, 3, 1, 2), w4[:, :, :3].to(bf).double().permute(3, 2, 0, 1).contiguous(), b.double(),
                   stride=st).clamp_min(0).permute(0, 2, 3, 1)
    y = torch.full((n, d.ho, d.wo, ldy), -3.0, device='cuda', dtype=bf)
    pf = ops.PreparedFilter(d, x4.device)
    if pf.ok and rng.random() < 0.5:
        pf.refresh(w4)
        ops.conv2d_fwd(pf.desc_prepared, x4, pf.buf, b, y, 'relu')
    else:
        ops.conv2d_fwd(d, x4, w4, b, y, 'relu')
    err = rel(y[..., :k].float(), ref)
    ph, pw = d.ho // 2, d.wo // 2
    pooled = torch.full((n, ph, pw, ldy), -3.0, device='cuda', dtype=bf)
    arg = torch.full((n, ph, pw, k), 9, device='cuda', dtype=torch.uint8)
    ops.conv2d_pool_fwd(d, x4, w4, b, pooled, 'relu', arg)
    win = y[:, :2 * ph, :2 * pw, :k].float().reshape(n, ph, 2, pw, 2, k).permute(0, 1, 3, 5, 2, 4).reshape(n, ph, pw, k, 4)
    want, want_arg = win.max(-1).values, (win == win.max(-1, keepdim=True).values).float().argmax(-1)
    if not (torch.equal(pooled[..., :k].float(), want) and torch.equal(arg.long(), want_arg)):
        return 1.0
    return err * TOL / TOL_BF16_OUT


def force_plan():
    """Half of the cases pin a tile config and either a split-K factor or a stream-K grid (the planner's own picks cover
    only a few of the combinations the kernels support)."""
    for v in ('A3D_FORCE_CFG', 'A3D_FORCE_SPLITK', 'A3D_FORCE_STREAMK', 'A3D_FORCE_SK_SLICED'):
        os.environ.pop(v, None)
    u = rng.random()
    if u < 0.5:
        return 'auto'
    os.environ['A3D_FORCE_CFG'] = str(int(rng.integers(0, 12)))      # 11: the second-generation kernel (its twin where it does not apply)
    if u < 0.75:
        os.environ['A3D_FORCE_SPLITK'] = str(int(rng.choice([1, 2, 3, 5, 8, 13])))
        return 'cfg%s sk%s' % (os.environ['A3D_FORCE_CFG'], os.environ['A3D_FORCE_SPLITK'])
    os.environ['A3D_FORCE_STREAMK'] = str(int(rng.choice([1, 2, 3, 7, 8, 32, 64, 100, 256, 512, 700])))
    if rng.random() < 0.5:        # bwd-filter shares cut from per-XCD K slices (where the shape allows it)
        os.environ['A3D_FORCE_SK_SLICED'] = '1'
    return 'cfg%s streamk%s%s' % (os.environ['A3D_FORCE_CFG'], os.environ['A3D_FORCE_STREAMK'],
                                  ' sliced' if 'A3D_FORCE_SK_SLICED' in os.environ else '')


t_end = time.time() + budget
count, worst = 0, (0.0, None)
while time.time() < t_end:
    forced = force_plan()
    u = rng.random()
    if u < 0.05 * R5:
        # round 5's kernels: their own plans, no forced tiles
        for v in ('A3D_FORCE_CFG', 'A3D_FORCE_SPLITK', 'A3D_FORCE_STREAMK', 'A3D_FORCE_SK_SLICED'):
            os.environ.pop(v, None)
        forced = 'auto'
        kind = int(rng.integers(0, 3))
        if kind == 0:
            case = fewch_case()
            err = run_fewch(*case)
            case = ('fewch / fewch16 filter gradient',) + case
        elif kind == 1:
            case, err = run_bwd_both()
            case = ('bwd_both',) + case
        else:
            case = fewch_case()
            err = run_image_form(*case)
            case = ('bf16 image form',) + case
    elif u < 0.05 * R5 + 0.05:
        for v in ('A3D_FORCE_CFG', 'A3D_FORCE_SPLITK', 'A3D_FORCE_STREAMK', 'A3D_FORCE_SK_SLICED'):
            os.environ.pop(v, None)                       # both paths on the weight-streaming kernels
        forced = 'auto'
        case, err = run_dense_adam()
        case = ('dense dW+Adam',) + case
    elif u < 0.25:
        try:
            case, err = run_dense()
        except Exception:
            print('EXCEPTION in a dense case under', forced, {k: v for k, v in os.environ.items() if k.startswith('A3D_')}, flush=True)
            raise
        case = ('dense',) + case
    elif u < 0.45:
        # bf16 storage on the bf16 kernels (a pinned tile configuration does not apply to them; split-K factors do)
        os.environ.pop('A3D_FORCE_STREAMK', None)
        case = conv_case_bf16s()
        e16, e32 = run_conv_bf16s(*case)
        err = max(e16 * TOL / TOL_BF16_OUT, e32 * TOL / TOL_BF16_DW)      # each judged against its own tolerance
        case = ('conv bf16s',) + case
    elif u < 0.50:
        for v in ('A3D_FORCE_CFG', 'A3D_FORCE_SPLITK', 'A3D_FORCE_STREAMK', 'A3D_FORCE_SK_SLICED'):
            os.environ.pop(v, None)
        forced = 'auto'
        case = conv_case_bf16s()
        err = run_conv_pool_bf16s(*case) * TOL / TOL_BF16_OUT
        case = ('conv+pool bf16s',) + case
    elif u < 0.58:
        os.environ.pop('A3D_FORCE_STREAMK', None)          # the fused pool takes whole K ranges only
        os.environ.pop('A3D_FORCE_SPLITK', None)
        case = conv_case()
        err = run_conv_pool(*case) * TOL / TOL_POOL          # two fp32 sums of up to 16 k products against each other
        case = ('conv+pool',) + case
    else:
        case = conv_case()
        err = run_conv(*case)
        case = ('conv',) + case
    count += 1
    case = case + (forced,)
    if not (err <= TOL):
        print('FAIL', case, err, flush=True)
    if err > worst[0]:
        worst = (err, case)
print(f'{count} cases, worst rel-L2 {worst[0]:.2e} at {worst[1]}')
print('round-5 kernels exercised:', RAN)
sys.exit(0 if worst[0] <= TOL else 1)
