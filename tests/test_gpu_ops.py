"""Parity of every HIP op (through the C ABI) against the numpy oracle on the same seeded inputs.
Bit-exact where the kernel mirrors the oracle's fp32 operation order (pool, resize, patches, Adam); rel-L2 <= 1e-5
against the float64 oracle for fp32-accumulating contractions (conv / dense / loss)."""
import numpy as np
import pytest
import torch

from oracle import tf13_ops as T

pytestmark = pytest.mark.gpu

RTOL_F32 = 1e-5     # fp32 MFMA accumulation vs float64 truth, K up to a few thousand


def dev(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).cuda()


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


@pytest.fixture(scope='module')
def ops():
    from ann3depth_amd import ops
    return ops


CONV_CASES = [
    # n, h, w, c, k, ksize, stride, padding            (edge cases of SURVEY 8c golden plan + every MSDN/DCNF layer kind)
    (2, 35, 47, 3, 96, 11, 4, 'VALID'),      # conv2d_0 kind: Cin=3 scalar gather (odd row length), N=96 tile
    (2, 35, 48, 3, 96, 11, 4, 'VALID'),      # conv2d_0 kind in window-run form: 8-byte runs of 33 -> 34 floats
    (2, 20, 32, 3, 16, 4, 4, 'VALID'),       # window-run form with 16-byte runs (12 floats, no padding needed)
    (2, 21, 32, 3, 24, 5, 4, 'VALID'),       # 16-byte runs of 15 -> 16 floats (8-byte runs would not save a tile)
    (1, 228, 304, 3, 96, 11, 4, 'VALID'),    # conv2d_0 itself (one image): last run ends exactly at the row end
    (2, 27, 37, 96, 256, 5, 1, 'SAME'),      # conv2d_1 at full spatial size
    (3, 13, 18, 256, 384, 3, 1, 'SAME'),     # conv2d_2
    (2, 13, 18, 384, 256, 3, 2, 'VALID'),    # conv2d_4: stride-2 VALID 13x18 -> 6x8
    (2, 40, 52, 3, 63, 9, 2, 'VALID'),       # fine/first: Cout=63
    (2, 21, 30, 64, 64, 5, 1, 'SAME'),       # fine/second
    (2, 21, 30, 64, 1, 5, 1, 'SAME'),        # fine/third: Cout=1 (direct stencil kernels)
    (3, 55, 74, 64, 1, 5, 1, 'SAME'),        # fine/third at full spatial size (74 = 9 strips of 8 + 2)
    (2, 9, 13, 40, 1, 5, 1, 'VALID'),        # stencil path with Cin < 64 and VALID padding
    (2, 12, 17, 64, 1, 5, 1, 'VALID'),       # 64 channels, VALID: the forward's taps-as-columns product on the matrix cores, a band of 8 + 4 rows cut short
    (1, 19, 70, 64, 1, 5, 1, 'SAME'),        # ... three bands (8 + 8 + 3 rows), rows of three 32-pixel pieces with 26 idle lanes
    (1, 10, 11, 8, 12, 4, 2, 'SAME'),        # asymmetric SAME padding, stride 2
    (1, 9, 9, 5, 7, 3, 1, 'SAME'),           # Cin, Cout not multiples of 4
    (5, 24, 24, 64, 256, 5, 1, 'VALID'),     # dcnf conv2d_1 kind
    (1, 1, 1, 16, 8, 1, 1, 'VALID'),         # degenerate 1x1
    (2, 13, 14, 4, 8, 3, 4, 'VALID'),        # stride 4 > kernel 3: some input pixels receive no gradient at all
    (2, 12, 16, 8, 8, 2, 2, 'SAME'),         # stride 2, kernel 2
    (32, 13, 18, 384, 256, 3, 2, 'VALID'),   # conv2d_4 at batch 32: the four bwd-data parity classes as ONE launch
    (20, 31, 33, 64, 48, 3, 2, 'SAME'),      # the same single-launch path with odd sizes, SAME padding, unequal classes
    (24, 26, 30, 5, 7, 5, 2, 'SAME'),        # ... and with scalar operands (Cin, Cout not multiples of 4), 5x5 taps
    # few-channel forward straight from L2 (conv3.hip): run starts 12 / 4 / 32 bytes apart, K tails, Cout off the tile width
    (2, 30, 35, 3, 64, 11, 1, 'VALID'),      # DCNF's first conv kind (stride 1: runs start 12 bytes apart)
    (2, 30, 36, 3, 64, 11, 1, 'VALID'),      # ... with rows of whole 16-byte pieces: the filter gradient from LDS-staged rows (fewch.hip)
    (3, 23, 29, 1, 40, 5, 1, 'VALID'),       # one channel: runs start 4 bytes apart, run of 5 -> 8 floats
    (2, 20, 27, 4, 70, 3, 2, 'VALID'),       # four channels, 70 filters (three column tiles, 26 of 96 columns idle)
    (65, 15, 15, 2, 33, 7, 4, 'VALID'),      # 65 images of 3 x 3 outputs: row tiles crossing images, M tail
    (2, 12, 15, 1, 40, 3, 1, 'VALID'),       # a run of ONE 4-group (3 taps -> 4): the chunk's two halves lie in different filter rows
    (3, 9, 14, 2, 36, 2, 1, 'VALID'),        # ... and a run that fills its 4-group exactly
    # bwd-filter window runs at any 4-byte address (stride 1 x 3 channels) must not be taken where SAME pads on the right /
    # below only (even kernel: pad_t = pad_l = 0): the last column's run would read the next row instead of zeros
    (3, 12, 27, 3, 16, 2, 1, 'SAME'),
    (4, 25, 38, 1, 2, 2, 1, 'SAME'),
]


@pytest.mark.parametrize('n,h,w,c,k,ks,st,pad', CONV_CASES)
def test_conv2d_fwd_bwd(ops, n, h, w, c, k, ks, st, pad):
    rng = np.random.default_rng(100 + h * w + c + k)
    x = rng.standard_normal((n, h, w, c)).astype(np.float32)
    wt = (rng.standard_normal((ks, ks, c, k)) / np.sqrt(ks * ks * c)).astype(np.float32)
    b = rng.standard_normal(k).astype(np.float32)
    d = ops.conv_desc(n, h, w, c, k, ks, ks, st, pad)
    x64, w64, b64 = x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64)
    y_ref = T.conv2d_fwd(x64, w64, b64, st, pad, relu=False)
    assert (d.ho, d.wo) == y_ref.shape[1:3]
    xd, wd, bd = dev(x), dev(wt), dev(b)
    y = torch.empty((n, d.ho, d.wo, k), device='cuda')
    ops.conv2d_fwd(d, xd, wd, bd, y, None)
    assert rel_l2(y.cpu().numpy(), y_ref) < RTOL_F32
    ops.conv2d_fwd(d, xd, wd, bd, y, 'relu')
    yr = y.cpu().numpy()
    assert rel_l2(yr, np.maximum(y_ref, 0)) < RTOL_F32
    ops.conv2d_fwd(d, xd, wd, None, y, None)
    assert rel_l2(y.cpu().numpy(), y_ref - b64) < RTOL_F32

    dz = rng.standard_normal(y_ref.shape).astype(np.float32)
    dw_ref, db_ref = T.conv2d_bwd_filter(x64, dz.astype(np.float64), wt.shape, st, pad)
    dx_ref = T.conv2d_bwd_data(dz.astype(np.float64), w64, x.shape, st, pad)
    dzd = dev(dz)
    dw = torch.empty_like(wd)
    db = torch.empty_like(bd)
    ops.conv2d_bwd_filter(d, xd, dzd, dw, db)
    assert rel_l2(dw.cpu().numpy(), dw_ref) < RTOL_F32
    assert rel_l2(db.cpu().numpy(), db_ref) < RTOL_F32
    dx = torch.full_like(xd, float('nan'))
    ops.conv2d_bwd_data(d, dzd, wd, dx)
    assert rel_l2(dx.cpu().numpy(), dx_ref) < RTOL_F32
    # fused ReluGrad of the previous layer: mask = this layer's input
    ops.conv2d_bwd_data(d, dzd, wd, dx, relu_mask=xd)
    assert rel_l2(dx.cpu().numpy(), dx_ref * (x > 0)) < RTOL_F32


@pytest.mark.parametrize('form', ['conv3', 'window runs (bf16 arithmetic)', 'bf16 image'])
def test_prepared_filter_gives_the_same_bits_as_the_per_call_repack(ops, form):
    """A3D_HINT_W_PREPARED (a3d_conv2d_fwd_prepare_filter): the few-channel forwards repack their filter per call; hoisted into a
    caller-owned copy the launch must give the same bits, with and without the fused pool; a forward that reads its filter as
    stored has no prepared form and refuses the hint."""
    import ctypes
    from ann3depth_amd import _lib
    rng = np.random.default_rng(len(form))
    n, h, w, k, ks, st = 3, 60, 80, 63, 9, 2
    wt = dev((rng.standard_normal((ks, ks, 3, k)) / np.sqrt(ks * ks * 3)).astype(np.float32))
    b = dev(rng.standard_normal(k).astype(np.float32) * 0.1)
    if form == 'conv3':
        d = ops.conv_desc(n, h, w, 3, k, ks, ks, st, 'VALID')
        x, filt, ydt = dev(rng.random((n, h, w, 3)).astype(np.float32)), wt, torch.float32
    elif form == 'bf16 image':
        d = ops.with_storage(ops.conv_desc(n, h, w, 4, k, ks, ks, st, 'VALID', ldy=64, precision='bf16'), ops.STORE_X | ops.STORE_Y)
        x = torch.empty((n, h, w, 4), device='cuda', dtype=torch.bfloat16)
        ops.pad_channels_bf16(dev(rng.random((n, h, w, 3)).astype(np.float32)), x)
        filt = torch.zeros((ks, ks, 4, k), device='cuda')
        filt[:, :, :3, :] = wt
        ydt = torch.bfloat16
    else:
        n, h, w, k, ks, st = 2, 100, 132, 96, 11, 4
        wt = dev((rng.standard_normal((ks, ks, 3, k)) / np.sqrt(ks * ks * 3)).astype(np.float32))
        b = dev(rng.standard_normal(k).astype(np.float32) * 0.1)
        d = ops.with_storage(ops.conv_desc(n, h, w, 3, k, ks, ks, st, 'VALID', precision='bf16'), ops.STORE_Y)
        x, filt, ydt = dev(rng.random((n, h, w, 3)).astype(np.float32)), wt, torch.bfloat16
    pf = ops.PreparedFilter(d, x.device)
    assert pf.ok
    pf.refresh(filt)
    ld = d.ldy
    y0 = torch.full((n, d.ho, d.wo, ld), -1.0, device='cuda', dtype=ydt)
    y1 = torch.full_like(y0, -2.0)
    ops.conv2d_fwd(d, x, filt, b, y0, 'relu')
    ops.conv2d_fwd(pf.desc_prepared, x, pf.buf, b, y1, 'relu')
    assert torch.equal(y0[..., :k], y1[..., :k])
    p0 = torch.full((n, d.ho // 2, d.wo // 2, ld), -1.0, device='cuda', dtype=ydt)
    p1 = torch.full_like(p0, -2.0)
    a0 = torch.full((n, d.ho // 2, d.wo // 2, k), 9, device='cuda', dtype=torch.uint8)
    a1 = torch.full_like(a0, 8)
    ops.conv2d_pool_fwd(d, x, filt, b, p0, 'relu', a0)
    ops.conv2d_pool_fwd(pf.desc_prepared, x, pf.buf, b, p1, 'relu', a1)
    assert torch.equal(p0[..., :k], p1[..., :k]) and torch.equal(a0, a1)
    # a layer that reads its filter as stored: nothing to prepare, and the hint is refused instead of misread
    dg = ops.conv_desc(2, 13, 18, 256, 384, 3, 3, 1, 'SAME')
    assert _lib.load().a3d_conv2d_fwd_prepared_filter_bytes(ctypes.byref(dg)) == 0 and not ops.PreparedFilter(dg, x.device).ok
    dg.hints = ops.HINT_W_PREPARED
    with pytest.raises(_lib.A3dError):
        ops.conv2d_fwd(dg, torch.zeros((2, 13, 18, 256), device='cuda'), torch.zeros((3, 3, 256, 384), device='cuda'), None,
                       torch.zeros((2, 13, 18, 384), device='cuda'), None)


def test_second_outputs_leave_through_the_reduction(ops):
    """a3d_second_output (a3d_dense_fwd_ex2 / a3d_dense_bwd_data_ex2 / a3d_conv2d_fwd_ex2): a second copy of a launch's finished
    output in another type / pitch / place equals the cast or copy launch it replaces bit for bit — config 5's five tensors that
    cross between the bf16 conv stack and the fp32 dense side (dense_0's dropout output as bf16, dense_1's padded GEMM into the
    4070-column depth map and channel 63 of the concat buffer, dz0 as bf16, conv2d_4's output as float32)."""
    rng = np.random.default_rng(77)
    bf = torch.bfloat16
    B, K, N, NP = 64, 4096, 4070, 4072
    x16 = torch.from_numpy(rng.standard_normal((B, K)).astype(np.float32)).cuda().to(bf)
    wpad = torch.zeros((K, NP), device='cuda', dtype=bf)
    wpad[:, :N] = torch.from_numpy((rng.standard_normal((K, N)) / 64).astype(np.float32)).cuda().to(bf)
    bpad = torch.zeros((1, NP), device='cuda')
    bpad[0, :N] = dev(rng.standard_normal(N).astype(np.float32))
    st = ops.STORE_W | ops.STORE_X
    # dense_1's form: padded GEMM -> y of 4070 columns + channel 63 of a [B, 55, 74, 64] bf16 buffer
    ypad = torch.empty((B, NP), device='cuda')
    ops.dense_fwd_ex(x16, wpad, bpad, ypad, None, precision='bf16', storage=st)
    y = torch.full((B, N), float('nan'), device='cuda')
    cat = torch.full((B, 55, 74, 64), 5.0, device='cuda', dtype=bf)
    ops.dense_fwd_ex(x16, wpad, bpad, y, None, precision='bf16', storage=st, n=NP,
                     out2=ops.second_output(cat, cols=N, step=64, offset=63, ld=N))
    assert torch.equal(y, ypad[:, :N])
    assert torch.equal(cat[..., 63].reshape(B, N), ypad[:, :N].to(bf)) and bool((cat[..., :63] == 5.0).all())
    # dense_0's form: relu + dropout, second output = the same values as bf16
    keep = torch.from_numpy((rng.random((B, NP)) >= 0.5).astype(np.uint8)).cuda()
    y1 = torch.empty((B, NP), device='cuda')
    y2 = torch.empty((B, NP), device='cuda')
    y16 = torch.empty((B, NP), device='cuda', dtype=bf)
    ops.dense_fwd_ex(x16, wpad, bpad, y1, 'relu', drop_keep=keep, precision='bf16', storage=st)
    ops.dense_fwd_ex(x16, wpad, bpad, y2, 'relu', drop_keep=keep, precision='bf16', storage=st, out2=ops.second_output(y16))
    assert torch.equal(y1, y2) and torch.equal(y16, y1.to(bf))
    # bwd-data: dx float32 (masked, scaled) + its bf16 copy
    dz16 = torch.from_numpy(rng.standard_normal((B, NP)).astype(np.float32)).cuda().to(bf)
    mask = dev(rng.standard_normal((B, K)).astype(np.float32))
    dx1 = torch.empty((B, K), device='cuda')
    dx2 = torch.empty((B, K), device='cuda')
    dx16 = torch.empty((B, K), device='cuda', dtype=bf)
    ops.dense_bwd_data_ex(dz16, wpad, dx1, mask=mask, scale=2.0, precision='bf16', storage=ops.STORE_W | ops.STORE_Y)
    ops.dense_bwd_data_ex(dz16, wpad, dx2, mask=mask, scale=2.0, precision='bf16', storage=ops.STORE_W | ops.STORE_Y,
                          out2=ops.second_output(dx16))
    assert torch.equal(dx1, dx2) and torch.equal(dx16, dx1.to(bf))
    # a conv forward to a bf16 tensor with a float32 second copy: with a reduction stage (conv2d_4 at batch 64) and without one
    for n, h, w, c, k, ks, stv in ((64, 13, 18, 384, 256, 3, 2), (2, 27, 37, 96, 256, 5, 1)):
        X, W, Y = ops.STORE_X, ops.STORE_W, ops.STORE_Y
        d = ops.with_storage(ops.conv_desc(n, h, w, c, k, ks, ks, stv, 'VALID', precision='bf16'), X | W | Y)
        xc = torch.from_numpy(rng.standard_normal((n, h, w, c)).astype(np.float32)).cuda().to(bf)
        wc = torch.from_numpy((rng.standard_normal((ks, ks, c, k)) / np.sqrt(ks * ks * c)).astype(np.float32)).cuda().to(bf)
        bc = dev(rng.standard_normal(k).astype(np.float32) * 0.1)
        ya = torch.empty((n, d.ho, d.wo, k), device='cuda', dtype=bf)
        yb = torch.empty_like(ya)
        y32 = torch.full((n, d.ho, d.wo, k), float('nan'), device='cuda')
        ops.conv2d_fwd(d, xc, wc, bc, ya, 'relu')
        ops.conv2d_fwd(d, xc, wc, bc, yb, 'relu', out2=ops.second_output(y32.view(-1, k)))
        assert torch.equal(ya, yb) and torch.equal(y32, ya.float())


POOLED_BWDF_CASES = [
    # n, h, w, c, k, ksize, stride, ld (pixel stride of the pooled tensors), argmax stride
    (2, 35, 48, 3, 96, 11, 4, 96, 96),       # conv2d_0 kind: 7 x 10 conv outputs, the odd last row has no pool window
    (2, 40, 52, 3, 63, 9, 2, 64, 63),        # fine/first kind: 63 filters inside the 64-channel concat buffer
    (3, 30, 36, 3, 64, 11, 1, 64, 64),       # DCNF's first conv kind (stride 1), 20 x 26 outputs
    (1, 228, 304, 3, 96, 11, 4, 96, 96),     # conv2d_0 itself, one image
]


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('n,h,w,c,k,ks,st,ld,lda', POOLED_BWDF_CASES)
def test_conv2d_bwd_filter_with_the_pool_gradient_fused(ops, n, h, w, c, k, ks, st, ld, lda, dtype):
    """a3d_conv2d_bwd_filter_pooled (MaxPoolGrad by index + ReluGrad + Conv2DBackpropFilter + BiasAddGrad in one launch) against
    the float64 oracle on the gradient the separate path materialises: dz[window position argmax] = dpool * (pooled > 0)."""
    rng = np.random.default_rng(900 + h * w + k)
    x = rng.standard_normal((n, h, w, c)).astype(np.float32)
    d = ops.conv_desc(n, h, w, c, k, ks, ks, st, 'VALID')
    assert ops.conv2d_bwd_filter_pooled_supported(d)
    ph, pw = d.ho // 2, d.wo // 2
    tdt = torch.bfloat16 if dtype == 'bf16' else torch.float32
    pooled = rng.standard_normal((n, ph, pw, ld)).astype(np.float32)          # about half of the maxima <= 0: ReluGrad is live
    dpool = rng.standard_normal((n, ph, pw, ld)).astype(np.float32)
    arg = rng.integers(0, 4, (n, ph, pw, lda)).astype(np.uint8)
    pooled_d, dpool_d = dev(pooled).to(tdt), dev(dpool).to(tdt)
    pooled_r, dpool_r = pooled_d.float().cpu().numpy(), dpool_d.float().cpu().numpy()
    dz = np.zeros((n, d.ho, d.wo, k), np.float64)
    g = np.where(pooled_r[..., :k] > 0, dpool_r[..., :k], 0.0)
    for pos in range(4):
        dz[:, pos >> 1:2 * ph:2, pos & 1:2 * pw:2, :] = np.where(arg[..., :k] == pos, g, 0.0)
    dw_ref, db_ref = T.conv2d_bwd_filter(x.astype(np.float64), dz, (ks, ks, c, k), st, 'VALID')
    xd = dev(x)
    dw = torch.full((ks, ks, c, k), float('nan'), device='cuda')
    db = torch.full((k,), float('nan'), device='cuda')
    ops.conv2d_bwd_filter_pooled(d, xd, dpool_d, pooled_d, torch.from_numpy(arg).cuda(), dw, db)
    assert rel_l2(dw.cpu().numpy(), dw_ref) < RTOL_F32
    assert rel_l2(db.cpu().numpy(), db_ref) < RTOL_F32
    # the same launch twice gives the same bits (fixed summation order)
    dw2 = torch.empty_like(dw)
    ops.conv2d_bwd_filter_pooled(d, xd, dpool_d, pooled_d, torch.from_numpy(arg).cuda(), dw2, None)
    np.testing.assert_array_equal(dw.cpu().numpy(), dw2.cpu().numpy())
    # and it equals the two launches it replaces to rounding (float32 sources only: the separate kernel takes no bf16 here)
    if dtype == 'f32' and ld == k and lda == k:
        dzd = torch.empty((n, d.ho, d.wo, k), device='cuda')
        ops.maxpool2x2_bwd_idx(torch.from_numpy(arg).cuda(), pooled_d, dpool_d, dzd, relu_mask=True)
        dw3 = torch.empty_like(dw)
        ops.conv2d_bwd_filter(d, xd, dzd, dw3, None)
        assert rel_l2(dw3.cpu().numpy(), dw.cpu().numpy()) < 1e-5


@pytest.mark.parametrize('n,h,w,c,k,ks,st,ld,lda', POOLED_BWDF_CASES[:2] + [(3, 228, 304, 3, 63, 9, 2, 64, 63)])
def test_conv2d_bwd_filter_pooled_on_the_bf16_matrix_cores(ops, n, h, w, c, k, ks, st, ld, lda):
    """Config 5's form of a3d_conv2d_bwd_filter_pooled (fewch16.hip): bf16 arithmetic on the float32 image and bf16 pooled
    tensors, both operands transposed into LDS.  Against the float64 oracle ON THE ROUNDED OPERANDS (x rounded to bf16, the
    gradient as the bf16 tensor holds it) the only error left is the fp32 accumulation: 1e-5; against the unrounded image it
    is the mode's bf16 rounding.  fine/first at full width runs two segments per output row."""
    rng = np.random.default_rng(1900 + h * w + k)
    x = rng.standard_normal((n, h, w, c)).astype(np.float32)
    d = ops.conv_desc(n, h, w, c, k, ks, ks, st, 'VALID', precision='bf16')
    assert ops.conv2d_bwd_filter_pooled_supported(d)
    ph, pw = d.ho // 2, d.wo // 2
    bf = torch.bfloat16
    pooled_d = dev(rng.standard_normal((n, ph, pw, ld)).astype(np.float32)).to(bf)
    dpool_d = dev(rng.standard_normal((n, ph, pw, ld)).astype(np.float32)).to(bf)
    arg = rng.integers(0, 4, (n, ph, pw, lda)).astype(np.uint8)
    pooled_r, dpool_r = pooled_d.float().cpu().numpy(), dpool_d.float().cpu().numpy()
    dz = np.zeros((n, d.ho, d.wo, k), np.float64)
    g = np.where(pooled_r[..., :k] > 0, dpool_r[..., :k], 0.0)
    for pos in range(4):
        dz[:, pos >> 1:2 * ph:2, pos & 1:2 * pw:2, :] = np.where(arg[..., :k] == pos, g, 0.0)
    xr = dev(x).to(bf).double().cpu().numpy()
    dw_ref, db_ref = T.conv2d_bwd_filter(xr, dz, (ks, ks, c, k), st, 'VALID')
    dw = torch.full((ks, ks, c, k), float('nan'), device='cuda')
    db = torch.full((k,), float('nan'), device='cuda')
    ops.conv2d_bwd_filter_pooled(d, dev(x), dpool_d, pooled_d, torch.from_numpy(arg).cuda(), dw, db)
    assert rel_l2(dw.cpu().numpy(), dw_ref) < RTOL_F32
    assert rel_l2(db.cpu().numpy(), db_ref) < RTOL_F32
    dw_exact, _ = T.conv2d_bwd_filter(x.astype(np.float64), dz, (ks, ks, c, k), st, 'VALID')
    assert rel_l2(dw.cpu().numpy(), dw_exact) < 4e-3
    dw2 = torch.empty_like(dw)
    ops.conv2d_bwd_filter_pooled(d, dev(x), dpool_d, pooled_d, torch.from_numpy(arg).cuda(), dw2, None)
    np.testing.assert_array_equal(dw.cpu().numpy(), dw2.cpu().numpy())


BOTH_CASES = [
    # n, h, w, c, padding, ldx, lddx          (single-output-channel 5x5 convs: fine/third, src/models.py:250-251)
    (2, 21, 30, 64, 'SAME', 64, 64),
    (3, 55, 74, 64, 'SAME', 64, 64),         # fine/third at full spatial size: 74 = 18 runs of 4 + 2, 55 = 13 bands of 4 + 3
    (2, 9, 13, 40, 'VALID', 40, 40),         # fewer channels than lanes, no padding: output smaller than the input
    (5, 7, 6, 64, 'SAME', 64, 72),           # fewer rows than two bands' reach; dx with a wider pixel stride
    (70, 5, 9, 24, 'SAME', 32, 24),          # more than 63 groups' worth of blocks at the default group size; strided x
    (2, 12, 17, 64, 'VALID', 64, 64),        # 64 channels, no padding: the matrix-core form (round 6) with an output smaller than the input
    (1, 19, 70, 64, 'SAME', 64, 64),         # ... three bands of 7 / 7 / 5 rows, pieces of 32 pixels that straddle rows
    (300, 6, 5, 64, 'SAME', 64, 64),         # ... 300 one-band blocks: ten groups, a last piece of 30 pixels
]


@pytest.mark.parametrize('dx16', [False, True])
@pytest.mark.parametrize('n,h,w,c,pad,ldx,lddx', BOTH_CASES)
def test_conv2d_bwd_both_single_output_channel(ops, n, h, w, c, pad, ldx, lddx, dx16):
    """a3d_conv2d_bwd_both: Conv2DBackpropFilter + BiasAddGrad + Conv2DBackpropInput + ReluGrad of a Cout = 1 conv in one pass
    over x, against the float64 oracle; dx as float32 or bf16 (config 5 hands it to fine/second's backward in that form);
    run twice on one state buffer (the arrival counters must come back to zero) and bit-identical both times."""
    rng = np.random.default_rng(700 + n * h * w + c)
    xbuf = rng.standard_normal((n, h, w, ldx)).astype(np.float32)
    x = xbuf[..., :c]
    wt = (rng.standard_normal((5, 5, c, 1)) / np.sqrt(25 * c)).astype(np.float32)
    d = ops.conv_desc(n, h, w, c, 1, 5, 5, 1, pad, ldx=ldx)
    assert ops.conv2d_bwd_both_supported(d)
    dz = rng.standard_normal((n, d.ho, d.wo, 1)).astype(np.float32)
    x64, w64, dz64 = x.astype(np.float64), wt.astype(np.float64), dz.astype(np.float64)
    dw_ref, db_ref = T.conv2d_bwd_filter(x64, dz64, wt.shape, 1, pad)
    dx_ref = T.conv2d_bwd_data(dz64, w64, x.shape, 1, pad)
    xd, wd, dzd = dev(xbuf), dev(wt), dev(dz)
    for mask in (True, False):
        outs = []
        for _ in range(2):
            dw = torch.full_like(wd, float('nan'))
            db = torch.full((1,), float('nan'), device='cuda')
            dx = torch.full((n, h, w, lddx), 7.0, device='cuda', dtype=torch.bfloat16 if dx16 else torch.float32)
            ops.conv2d_bwd_both(d, xd, dzd, wd, dw, db, dx, relu_mask=mask)
            torch.cuda.synchronize()
            outs.append((dw.cpu().numpy(), db.cpu().numpy(), dx.float().cpu().numpy()))
        for a, b in zip(outs[0], outs[1]):
            np.testing.assert_array_equal(a, b)
        dw_g, db_g, dx_g = outs[0]
        assert rel_l2(dw_g, dw_ref) < RTOL_F32
        assert rel_l2(db_g, db_ref) < RTOL_F32
        want = dx_ref * (x > 0) if mask else dx_ref
        assert rel_l2(dx_g[..., :c], want) < (4e-3 if dx16 else RTOL_F32)
        assert (dx_g[..., c:] == 7.0).all()
    # a descriptor the fused kernel does not take is refused, not mis-run
    assert not ops.conv2d_bwd_both_supported(ops.conv_desc(n, h, w, c, 2, 5, 5, 1, pad, ldx=ldx))


BF16_CASES = [c for c in CONV_CASES if c[3] % 4 == 0 and c[4] % 4 == 0 and c[4] > 1]


@pytest.mark.parametrize('prec,tol', [('bf16x3', 5e-5), ('bf16', 2e-2)])
@pytest.mark.parametrize('n,h,w,c,k,ks,st,pad', BF16_CASES)
def test_conv2d_bf16_modes(ops, n, h, w, c, k, ks, st, pad, prec, tol):
    """The bf16-MFMA kernels (split-operand bf16x3 and plain bf16) against the float64 oracle: all three directions,
    fused bias/ReLU, fused ReluGrad mask and the fused BiasAddGrad."""
    rng = np.random.default_rng(300 + h * w + c + k)
    x = rng.standard_normal((n, h, w, c)).astype(np.float32)
    wt = (rng.standard_normal((ks, ks, c, k)) / np.sqrt(ks * ks * c)).astype(np.float32)
    b = rng.standard_normal(k).astype(np.float32)
    d = ops.conv_desc(n, h, w, c, k, ks, ks, st, pad, precision=prec)
    x64, w64, b64 = x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64)
    y_ref = T.conv2d_fwd(x64, w64, b64, st, pad, relu=True)
    xd, wd, bd = dev(x), dev(wt), dev(b)
    y = torch.empty(y_ref.shape, device='cuda')
    ops.conv2d_fwd(d, xd, wd, bd, y, 'relu')
    assert rel_l2(y.cpu().numpy(), y_ref) < tol
    dz = rng.standard_normal(y_ref.shape).astype(np.float32)
    dw_ref, db_ref = T.conv2d_bwd_filter(x64, dz.astype(np.float64), wt.shape, st, pad)
    dx_ref = T.conv2d_bwd_data(dz.astype(np.float64), w64, x.shape, st, pad)
    dzd = dev(dz)
    dw = torch.empty_like(wd)
    db = torch.empty_like(bd)
    ops.conv2d_bwd_filter(d, xd, dzd, dw, db)
    assert rel_l2(dw.cpu().numpy(), dw_ref) < tol
    assert rel_l2(db.cpu().numpy(), db_ref) < 1e-5          # bias gradient is summed in fp32 in every mode
    dx = torch.full_like(xd, float('nan'))
    ops.conv2d_bwd_data(d, dzd, wd, dx, relu_mask=xd)
    assert rel_l2(dx.cpu().numpy(), dx_ref * (x > 0)) < tol


def test_conv2d_strided_pixels(ops):
    """ldx / ldy pixel strides: the fine/second layer reads the 64-channel concat buffer and bwd-data writes it."""
    rng = np.random.default_rng(5)
    n, h, w, c, k = 2, 11, 14, 8, 12
    xbuf = rng.standard_normal((n, h, w, 16)).astype(np.float32)
    wt = rng.standard_normal((3, 3, c, k)).astype(np.float32)
    d = ops.conv_desc(n, h, w, c, k, 3, 3, 1, 'SAME', ldx=16, ldy=20)
    ybuf = torch.full((n, h, w, 20), 7.0, device='cuda')
    ops.conv2d_fwd(d, dev(xbuf), dev(wt), None, ybuf, None)
    y_ref = T.conv2d_fwd(xbuf[..., :c].astype(np.float64), wt.astype(np.float64), None, 1, 'SAME')
    got = ybuf.cpu().numpy()
    assert rel_l2(got[..., :k], y_ref) < RTOL_F32
    assert (got[..., k:] == 7.0).all()


GUARD_CASES = [
    # n, h, w, c, k, ksize, ld (output pixel stride)      rows of the last tile past M, columns past N, pitch wider than N
    (2, 21, 30, 64, 64, 5, 64),       # fine/second kind: one column tile exactly, 1260 rows = 9 tiles of 128 + 108
    (1, 27, 37, 96, 72, 5, 80),       # 72 of 128 (96) columns, pitch 80, 999 rows
    (3, 13, 18, 32, 200, 3, 208),     # two column tiles, the second with 72 columns
    (1, 9, 11, 16, 4, 3, 12),         # one tile, 99 rows x 4 columns of it
]


@pytest.mark.parametrize('n,h,w,c,k,ks,ld', GUARD_CASES)
def test_tile_epilogues_leave_guard_rows_and_columns_alone(ops, n, h, w, c, k, ks, ld):
    """The tile epilogues store through a buffer descriptor that ends with the tile's last valid row; columns past N are
    offsets past its end (igemm.h: store_tile_buf / store_tile_pool_buf).  Every output tensor here is a window of a
    larger allocation filled with a sentinel: pad columns, and the rows that follow the tensor, must keep it — forward
    (plain, bias + ReLU, fused pool with its argmax bytes) and bwd-data with the ReluGrad mask."""
    rng = np.random.default_rng(n * 1000 + k)
    x = rng.standard_normal((n, h, w, c)).astype(np.float32)
    wt = (rng.standard_normal((ks, ks, c, k)) / np.sqrt(ks * ks * c)).astype(np.float32)
    bias = rng.standard_normal(k).astype(np.float32)
    d = ops.conv_desc(n, h, w, c, k, ks, ks, 1, 'SAME', ldy=ld)
    rows = n * h * w
    big = torch.full((rows + 300, ld), 7.0, device='cuda')
    y = big[:rows].view(n, h, w, ld)
    y_ref = T.conv2d_fwd(x.astype(np.float64), wt.astype(np.float64), bias.astype(np.float64), 1, 'SAME')
    for act in (None, 'relu'):
        big.fill_(7.0)
        ops.conv2d_fwd(d, dev(x), dev(wt), dev(bias), y, act)
        got = big.cpu().numpy()
        ref = np.maximum(y_ref, 0) if act else y_ref
        assert rel_l2(got[:rows, :k], ref.reshape(rows, k)) < RTOL_F32
        assert (got[:rows, k:] == 7.0).all() and (got[rows:] == 7.0).all()
    # fused 2x2 max pool: the pooled tensor and its argmax bytes, each followed by guard rows
    ph, pw = h // 2, w // 2
    prow = n * ph * pw
    pbig = torch.full((prow + 100, ld), 7.0, device='cuda')
    abig = torch.full((prow + 100, k), 9, device='cuda', dtype=torch.uint8)
    ops.conv2d_pool_fwd(d, dev(x), dev(wt), dev(bias), pbig[:prow].view(n, ph, pw, ld), 'relu', abig[:prow].view(n, ph, pw, k))
    relu = np.maximum(y_ref, 0)[:, :2 * ph, :2 * pw].reshape(n, ph, 2, pw, 2, k).transpose(0, 1, 3, 5, 2, 4).reshape(prow, k, 4)
    got = pbig.cpu().numpy()
    assert rel_l2(got[:prow, :k], relu.max(-1)) < RTOL_F32
    assert (got[:prow, k:] == 7.0).all() and (got[prow:] == 7.0).all()
    a = abig.cpu().numpy()
    assert (a[:prow] < 4).all() and (a[prow:] == 9).all()
    # bwd-data into a window of a wider buffer (pixel stride ld2 > c), ReluGrad mask read with the same pitch
    ld2 = c + 8
    dd = ops.conv_desc(n, h, w, c, k, ks, ks, 1, 'SAME', ldx=ld2)
    dz = rng.standard_normal((n, h, w, k)).astype(np.float32)
    xb = torch.full((rows + 200, ld2), 0.0, device='cuda')
    xb[:rows, :c] = dev(x).view(rows, c)
    dxb = torch.full((rows + 200, ld2), 7.0, device='cuda')
    ops.conv2d_bwd_data(dd, dev(dz), dev(wt), dxb[:rows].view(n, h, w, ld2), relu_mask=xb[:rows].view(n, h, w, ld2))
    dx_ref = T.conv2d_bwd_data(dz.astype(np.float64), wt.astype(np.float64), (n, h, w, c), 1, 'SAME') * (x > 0)
    got = dxb.cpu().numpy()
    assert rel_l2(got[:rows, :c], dx_ref.reshape(rows, c)) < RTOL_F32
    assert (got[:rows, c:] == 7.0).all() and (got[rows:] == 7.0).all()


@pytest.mark.parametrize('n,h,w,k,ks,st,y16', [(2, 23, 32, 63, 9, 2, True), (3, 17, 20, 16, 5, 2, False), (1, 40, 36, 8, 11, 4, True)])
def test_conv_fwd_bf16_image_form(ops, n, h, w, k, ks, st, y16):
    """A 3-channel image stored as bf16 pixels of 4 channels (a3d_pad_channels_bf16) through the bf16 kernel's window-run
    form, against the fp32 kernel on the fp32 image: the arithmetic of BASELINE config 5 (bf16 operands, fp32 accumulate),
    so 2e-2; the 4th channel of image and filter never contributes."""
    rng = np.random.default_rng(n * 100 + k)
    x = rng.random((n, h, w, 3)).astype(np.float32)
    wt = (rng.standard_normal((ks, ks, 3, k)) * 0.1).astype(np.float32)
    bias = rng.standard_normal(k).astype(np.float32) * 0.1
    d32 = ops.conv_desc(n, h, w, 3, k, ks, ks, st, 'VALID')
    y32 = torch.empty((n, d32.ho, d32.wo, k), device='cuda')
    ops.conv2d_fwd(d32, dev(x), dev(wt), dev(bias), y32, 'relu')
    x4 = torch.empty((n, h, w, 4), device='cuda', dtype=torch.bfloat16)
    ops.pad_channels_bf16(dev(x), x4)
    np.testing.assert_array_equal(x4[..., 3].float().cpu().numpy(), 0)
    np.testing.assert_array_equal(x4[..., :3].float().cpu().numpy(), dev(x).to(torch.bfloat16).float().cpu().numpy())
    w4 = np.concatenate([wt, np.full((ks, ks, 1, k), 7.0, np.float32)], axis=2)         # channel 3 of the filter: anything
    ldy = (k + 7) // 8 * 8
    d = ops.with_storage(ops.conv_desc(n, h, w, 4, k, ks, ks, st, 'VALID', ldy=ldy, precision='bf16'),
                         ops.STORE_X | (ops.STORE_Y if y16 else 0))
    y = torch.full((n, d.ho, d.wo, ldy), -3.0, device='cuda', dtype=torch.bfloat16 if y16 else torch.float32)
    ops.conv2d_fwd(d, x4, dev(w4), dev(bias), y, 'relu')
    torch.cuda.synchronize()
    got = y[..., :k].float().cpu().numpy()
    assert rel_l2(got, y32.cpu().numpy()) < 2e-2
    assert (y[..., k:].float().cpu().numpy() == -3.0).all()                              # pad channels are not written
    # conv + ReLU + 2x2 max pool in one launch equals the two launches bit for bit (values are rounded to the output type
    # before they are compared, first maximum wins), odd output sizes drop their last row / column, and the recorded
    # position is the first maximum's
    ph, pw = d.ho // 2, d.wo // 2
    pooled = torch.full((n, ph, pw, ldy), -3.0, device='cuda', dtype=y.dtype)
    arg = torch.full((n, ph, pw, k), 9, device='cuda', dtype=torch.uint8)
    ops.conv2d_pool_fwd(d, x4, dev(w4), dev(bias), pooled, 'relu', arg)
    torch.cuda.synchronize()
    win = y[:, :2 * ph, :2 * pw, :k].float().reshape(n, ph, 2, pw, 2, k).permute(0, 1, 3, 5, 2, 4).reshape(n, ph, pw, k, 4)
    want, want_arg = win.max(-1).values, (win == win.max(-1, keepdim=True).values).float().argmax(-1)
    assert torch.equal(pooled[..., :k].float(), want)
    assert torch.equal(arg.long(), want_arg)
    assert (pooled[..., k:].float().cpu().numpy() == -3.0).all()


DENSE_CASES = [(32, 12288, 4096), (4, 512, 4070), (32, 4096, 4070), (7, 130, 66), (48, 12544, 128), (48, 128, 16),
               (48, 16, 1), (9, 1024, 1031),
               (64, 3, 4070), (2, 1, 8)]      # <= 4 inputs: still a GEMM (its workspace query once took the few-channel conv's)


@pytest.mark.parametrize('m,k,n', DENSE_CASES)
def test_dense_fwd_bwd(ops, m, k, n):
    rng = np.random.default_rng(m + k + n)
    x = rng.standard_normal((m, k)).astype(np.float32)
    w = (rng.standard_normal((k, n)) / np.sqrt(k)).astype(np.float32)
    b = rng.standard_normal(n).astype(np.float32)
    keep = rng.random((m, n)) >= 0.5
    x64, w64, b64 = x.astype(np.float64), w.astype(np.float64), b.astype(np.float64)
    xd, wd, bd = dev(x), dev(w), dev(b)
    y = torch.empty((m, n), device='cuda')
    ops.dense_fwd(xd, wd, bd, y, 'relu')
    y_ref = T.dense_fwd(x64, w64, b64, 'relu')
    assert rel_l2(y.cpu().numpy(), y_ref) < RTOL_F32
    ops.dense_fwd(xd, wd, bd, y, 'relu', drop_keep=dev(keep, torch.uint8))
    assert rel_l2(y.cpu().numpy(), T.dropout_fwd(y_ref, keep)) < RTOL_F32
    ops.dense_fwd(xd, wd, bd, y, 'sigmoid')
    assert rel_l2(y.cpu().numpy(), T.dense_fwd(x64, w64, b64, 'sigmoid')) < RTOL_F32
    dz = rng.standard_normal((m, n)).astype(np.float32)
    dx_ref, dw_ref, db_ref = T.dense_bwd(x64, w64, dz.astype(np.float64))
    dzd = dev(dz)
    dx = torch.empty_like(xd)
    ops.dense_bwd_data(dzd, wd, dx)
    assert rel_l2(dx.cpu().numpy(), dx_ref) < RTOL_F32
    ops.dense_bwd_data(dzd, wd, dx, mask=xd, scale=2.0)
    assert rel_l2(dx.cpu().numpy(), 2 * dx_ref * (x > 0)) < RTOL_F32
    dw = torch.empty_like(wd)
    db = torch.empty_like(bd)
    ops.dense_bwd_filter(xd, dzd, dw, db)
    assert rel_l2(dw.cpu().numpy(), dw_ref) < RTOL_F32
    assert rel_l2(db.cpu().numpy(), db_ref) < RTOL_F32


@pytest.mark.parametrize('n,h,w,c', [(2, 55, 74, 96), (2, 27, 37, 256), (1, 110, 148, 63), (3, 4, 5, 3), (2, 14, 14, 256)])
def test_maxpool_fwd_bwd_bitexact(ops, n, h, w, c):
    rng = np.random.default_rng(h * w + c)
    x = np.maximum(rng.standard_normal((n, h, w, c)), 0).astype(np.float32)      # post-ReLU: many exact ties at 0
    xd = dev(x)
    y = torch.empty((n, h // 2, w // 2, c), device='cuda')
    ops.maxpool2x2_fwd(xd, y)
    y_ref = T.maxpool2x2_fwd(x)
    np.testing.assert_array_equal(y.cpu().numpy(), y_ref)
    dy = rng.standard_normal(y_ref.shape).astype(np.float32)
    dx = torch.full_like(xd, float('nan'))
    ops.maxpool2x2_bwd(xd, dev(dy), dx, relu_mask=False)
    np.testing.assert_array_equal(dx.cpu().numpy(), T.maxpool2x2_bwd(x, dy))
    ops.maxpool2x2_bwd(xd, dev(dy), dx, relu_mask=True)
    np.testing.assert_array_equal(dx.cpu().numpy(), T.relu_grad(T.maxpool2x2_bwd(x, dy), x))


def test_maxpool_concat_and_strided_grad(ops):
    """fine/first pool writes channels 0..62 of the 64-channel buffer, coarse goes to channel 63 (src/models.py:246);
    the gradient reads the first 63 channels of a 64-channel pixel stride."""
    rng = np.random.default_rng(9)
    x = np.maximum(rng.standard_normal((2, 22, 30, 63)), 0).astype(np.float32)
    coarse = rng.standard_normal((2, 11, 15, 1)).astype(np.float32)
    cat = torch.empty((2, 11, 15, 64), device='cuda')
    ops.maxpool2x2_fwd(dev(x), cat, extra=dev(coarse))
    ref = np.concatenate([T.maxpool2x2_fwd(x), coarse], axis=-1)
    np.testing.assert_array_equal(cat.cpu().numpy(), ref)
    dcat = rng.standard_normal((2, 11, 15, 64)).astype(np.float32)
    dx = torch.empty((2, 22, 30, 63), device='cuda')
    ops.maxpool2x2_bwd(dev(x), dev(dcat), dx, relu_mask=True)
    np.testing.assert_array_equal(dx.cpu().numpy(), T.relu_grad(T.maxpool2x2_bwd(x, dcat[..., :63]), x))


@pytest.mark.parametrize('n,h,w,c,oh,ow', [(2, 480, 640, 3, 228, 304), (2, 480, 640, 1, 55, 74), (1, 48, 64, 3, 228, 304),
                                           (1, 6, 8, 1, 55, 74), (2, 480, 640, 3, 240, 320), (1, 7, 5, 2, 7, 5)])
def test_resize_bitexact(ops, n, h, w, c, oh, ow):
    rng = np.random.default_rng(h + w + oh)
    x = (rng.integers(0, 256, (n, h, w, c)) / 255).astype(np.float32)
    y = torch.empty((n, oh, ow, c), device='cuda')
    ops.resize_bilinear_tf1(dev(x), y)
    np.testing.assert_array_equal(y.cpu().numpy(), T.resize_bilinear_tf1(x, oh, ow))


def test_resize_pair_is_the_two_resizes_in_one_launch(ops):
    """Image and depth map of a step (src/models.py:282-283) in one launch, different output sizes and channel counts."""
    rng = np.random.default_rng(3)
    img = (rng.integers(0, 256, (3, 48, 64, 3)) / 255).astype(np.float32)
    dep = (rng.integers(0, 256, (3, 48, 64, 1)) / 255).astype(np.float32)
    y0, y1 = torch.empty((3, 23, 31, 3), device='cuda'), torch.empty((3, 55, 74, 1), device='cuda')
    ops.resize_bilinear_tf1_pair(dev(img), y0, dev(dep), y1)
    np.testing.assert_array_equal(y0.cpu().numpy(), T.resize_bilinear_tf1(img, 23, 31))
    np.testing.assert_array_equal(y1.cpu().numpy(), T.resize_bilinear_tf1(dep, 55, 74))


def test_extract_patches_bitexact(ops):
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 240, 320, 3)).astype(np.float32)
    ref = T.extract_patches(x, 100, 40, 'SAME')
    y = torch.empty((2 * 48, 100, 100, 3), device='cuda')
    ops.extract_patches(dev(x), 100, 40, y)
    np.testing.assert_array_equal(y.cpu().numpy(), ref.reshape(96, 100, 100, 3))


def test_silog_loss(ops):
    rng = np.random.default_rng(11)
    b, npix = 32, 4070
    o = (rng.standard_normal((b, npix)) * 0.05).astype(np.float32)          # about half negative -> NaN-masked logs
    t = (rng.integers(0, 256, (b, npix)) / 255).astype(np.float32)          # contains exact zeros
    o[0, 0] = -1e-8                                                          # log(0) = -inf stays: non-finite loss
    od, td = dev(o), dev(t)
    loss = torch.empty(1, device='cuda')
    ws = ops.silog_ws(b, 'cuda')                    # per-sample sums + the arrival ticket (zero before the first call) + partials
    dout = torch.empty_like(od)
    ops.silog_loss_fwd(od, td, loss, ws)
    assert not np.isfinite(loss.item())
    with np.errstate(invalid='ignore', divide='ignore'):
        assert not np.isfinite(T.silog_loss_fwd(o, t))
    o[0, 0] = 0.3
    od = dev(o)
    ops.silog_loss_fwd(od, td, loss, ws)
    ref = T.silog_loss_fwd(o.astype(np.float64), t.astype(np.float64))
    assert abs(loss.item() - ref) < 2e-6 * abs(ref)
    ops.silog_loss_bwd(od, td, ws, dout)
    g_ref = T.silog_loss_bwd(o.astype(np.float64), t.astype(np.float64))
    g = dout.cpu().numpy()
    assert rel_l2(g, g_ref) < 1e-5
    assert (g[o < -1e-8] == 0).all()
    for _ in range(3):                              # the ticket wraps back to zero: repeated calls give the same loss
        ops.silog_loss_fwd(od, td, loss, ws)
        assert abs(loss.item() - ref) < 2e-6 * abs(ref)
    assert ws[2 * b].view(torch.int32).item() == 0


@pytest.mark.parametrize('beta2', [1.0, 0.999])
def test_adam_bitexact(ops, beta2):
    rng = np.random.default_rng(17)
    count = 4 * 1000 + 3          # vector body + scalar tail
    var = rng.standard_normal(count).astype(np.float32)
    opt = T.AdamTF1(0.1, 0.9, beta2)
    vd = dev(var)
    md = torch.zeros_like(vd)
    sd = torch.zeros_like(vd)
    ref = {'w': var.copy()}
    b1p, b2p = np.float32(0.9), np.float32(beta2)
    for step in range(3):
        g = rng.standard_normal(count).astype(np.float32)
        opt.apply(ref, {'w': g})
        ops.adam_apply_tf1(vd, md, sd, dev(g), 0.1, 0.9, beta2, 1e-8, float(b1p), float(b2p))
        b1p, b2p = b1p * np.float32(0.9), b2p * np.float32(beta2)
        np.testing.assert_array_equal(md.cpu().numpy(), opt.m['w'])
        np.testing.assert_array_equal(sd.cpu().numpy(), opt.v['w'])
        np.testing.assert_array_equal(vd.cpu().numpy(), ref['w'])
    if beta2 == 1.0:
        np.testing.assert_array_equal(ref['w'], var)       # the reference's optimizer never moves the weights


def test_adam_frozen_path_poison_semantics(ops):
    """beta2 = 1 takes the 3-stream kernel; non-finite gradients must poison v / var exactly like the full formula
    (inf * 0 = NaN), and a non-zero finite v (foreign checkpoint) must be left untouched."""
    rng = np.random.default_rng(23)
    count = 4 * 64 + 2
    var = rng.standard_normal(count).astype(np.float32)
    v0 = (rng.random(count) * 0.01).astype(np.float32)
    g = rng.standard_normal((3, count)).astype(np.float32)
    g[0, 5] = np.inf; g[0, 77] = np.nan; g[1, 130] = -np.inf; g[1, 200] = 3e19; g[2, count - 1] = np.inf
    opt = T.AdamTF1(0.1, 0.9, 1.0)
    opt.v['w'] = v0.copy()
    opt.m['w'] = np.zeros(count, np.float32)
    ref = {'w': var.copy()}
    vd, md, sd = dev(var), torch.zeros(count, device='cuda'), dev(v0)
    b1p = np.float32(0.9)
    for step in range(3):
        with np.errstate(invalid='ignore', over='ignore'):
            opt.apply(ref, {'w': g[step]})
        ops.adam_apply_tf1(vd, md, sd, dev(g[step]), 0.1, 0.9, 1.0, 1e-8, float(b1p), 1.0)
        b1p = b1p * np.float32(0.9)
        np.testing.assert_array_equal(md.cpu().numpy(), opt.m['w'])
        np.testing.assert_array_equal(sd.cpu().numpy(), opt.v['w'])
        np.testing.assert_array_equal(vd.cpu().numpy(), ref['w'])
    assert np.isnan(ref['w']).sum() == 5 and np.isnan(opt.v['w']).sum() == 5


# the last two are large enough for the streaming kernel to walk several row groups per block (16- and 8-byte rows)
@pytest.mark.parametrize('m,k,n', [(32, 384, 520), (5, 130, 4070), (64, 512, 256), (1, 512, 257), (32, 4488, 1540),
                                   (17, 4488, 1538), (64, 4488, 1030), (47, 1028, 772)])
def test_dense_bwd_filter_adam_fused_equals_two_passes(ops, m, k, n):
    """a3d_dense_bwd_filter_adam_tf1 (gradient never written) against a3d_dense_bwd_filter + a3d_adam_apply_tf1 and the
    oracle's ApplyAdam, over two steps, with non-finite gradients in both the kernel and the bias: m, v and var must agree
    bit for bit (same MFMA sum order, same separate fp32 operations), the gradient with torch float64 to fp32 accuracy."""
    rng = np.random.default_rng(m * 1000 + n)
    x = rng.standard_normal((m, k)).astype(np.float32)
    dz = rng.standard_normal((2, m, n)).astype(np.float32)
    dz[1, 0, 3] = np.inf                       # poisons column 3 of the kernel gradient and bias 3
    dz[1, m - 1, n - 1] = 2e19                 # finite, but its square is not: poisons through g*g
    x[m - 1, 0] = 3e19
    var_w = rng.standard_normal((k, n)).astype(np.float32)
    var_b = rng.standard_normal(n).astype(np.float32)
    v_w0 = (rng.random((k, n)) * 0.01).astype(np.float32)
    scale = 0.5
    fused = [dev(var_w), torch.zeros((k, n), device='cuda'), dev(v_w0), dev(var_b), torch.zeros(n, device='cuda'),
             torch.zeros(n, device='cuda')]
    plain = [t.clone() for t in fused]
    opt = T.AdamTF1(0.1, 0.9, 1.0)
    opt.m = {'w': np.zeros((k, n), np.float32), 'b': np.zeros(n, np.float32)}
    opt.v = {'w': v_w0.copy(), 'b': np.zeros(n, np.float32)}
    ref = {'w': var_w.copy(), 'b': var_b.copy()}
    b1p = np.float32(0.9)
    xd = dev(x)
    for step in range(2):
        dzd = dev(dz[step])
        ops.dense_bwd_filter_adam_tf1(xd, dzd, *fused, 0.1, 0.9, 1.0, float(b1p), 1.0, scale)
        dw, db = torch.empty((k, n), device='cuda'), torch.empty(n, device='cuda')
        ops.dense_bwd_filter(xd, dzd, dw, db)
        ops.adam_apply_tf1(plain[0], plain[1], plain[2], dw, 0.1, 0.9, 1.0, 1e-8, float(b1p), 1.0, scale)
        ops.adam_apply_tf1(plain[3], plain[4], plain[5], db, 0.1, 0.9, 1.0, 1e-8, float(b1p), 1.0, scale)
        with np.errstate(invalid='ignore', over='ignore'):
            opt.apply(ref, {'w': dw.cpu().numpy() * np.float32(scale), 'b': db.cpu().numpy() * np.float32(scale)})
        b1p = b1p * np.float32(0.9)
        for a, b in zip(fused, plain):
            np.testing.assert_array_equal(a.cpu().numpy(), b.cpu().numpy())
        np.testing.assert_array_equal(fused[1].cpu().numpy(), opt.m['w'])
        np.testing.assert_array_equal(fused[2].cpu().numpy(), opt.v['w'])
        np.testing.assert_array_equal(fused[0].cpu().numpy(), ref['w'])
        np.testing.assert_array_equal(fused[4].cpu().numpy(), opt.m['b'])
        np.testing.assert_array_equal(fused[3].cpu().numpy(), ref['b'])
        if step == 0:
            with np.errstate(invalid='ignore', over='ignore'):
                want = x.astype(np.float64).T @ dz[0].astype(np.float64)
                norm = np.abs(x).astype(np.float64).T @ np.abs(dz[0]).astype(np.float64)
            err = np.abs(dw.cpu().numpy() - want) / norm
            assert np.nanmax(err[np.isfinite(norm)]) < 1e-6
    assert np.isnan(ref['w']).any() and np.isnan(ref['b']).any()
    with pytest.raises(Exception):             # not the frozen optimizer: the fused entry refuses
        ops.dense_bwd_filter_adam_tf1(xd, dzd, *fused, 0.1, 0.9, 0.999, float(b1p), 0.999, scale)


@pytest.mark.parametrize('m,k,n', [(64, 1024, 1024), (64, 392, 4070), (48, 512, 520), (33, 256, 512)])
def test_dense_bwd_filter_adam_with_bf16_arithmetic(ops, m, k, n):
    """a3d_dense_bwd_filter_adam_tf1_ex(precision bf16), BASELINE config 5's batch: the contraction x^T dz on the bf16 matrix
    cores (operands rounded to bf16 once, fp32 accumulation), ApplyAdam as before.  m = 0 + (g - 0)(1 - beta1) against the
    float64 product of the ROUNDED operands at fp32 accumulation tolerance; against the unrounded product only at bf16's."""
    rng = np.random.default_rng(m + k + n)
    x = rng.standard_normal((m, k)).astype(np.float32)
    dz = rng.standard_normal((m, n)).astype(np.float32)
    xd, dzd = dev(x), dev(dz)
    slots = [torch.zeros((k, n), device='cuda') for _ in range(3)] + [torch.zeros(n, device='cuda') for _ in range(3)]
    slots[0].fill_(0.25)
    ops.dense_bwd_filter_adam_tf1(xd, dzd, *slots, 0.1, 0.9, 1.0, 0.9, 1.0, 1.0, precision='bf16')
    xr = xd.to(torch.bfloat16).float().cpu().numpy().astype(np.float64)
    zr = dzd.to(torch.bfloat16).float().cpu().numpy().astype(np.float64)
    omb1 = float(np.float32(1) - np.float32(0.9))
    got = slots[1].cpu().numpy()
    assert rel_l2(got, (xr.T @ zr) * omb1) < 1e-5
    assert rel_l2(got, (x.astype(np.float64).T @ dz.astype(np.float64)) * omb1) < 1e-2
    assert rel_l2(slots[4].cpu().numpy(), dz.astype(np.float64).sum(0) * omb1) < 1e-6       # BiasAddGrad stays fp32
    assert (slots[0] == 0.25).all() and (slots[2] == 0).all()                                # alpha = 0: var, v untouched


@pytest.mark.parametrize('m,k,n', [(64, 12288, 4096), (64, 4096, 2048), (37, 1024, 1032), (2, 12288, 4096)])
def test_dense_layers_on_bf16_tensors_through_the_lds_dma_kernel(ops, m, k, n):
    """a3d_dense_fwd_ex / a3d_dense_bwd_data_ex with x / dz / dx stored as bf16 beside the bf16 weight copy (BASELINE config
    5's dense_0: c4 and dc4 are bf16 activations): a small batch against a [k, n] weight stream on igemm_ring.h's 64-row
    tiles, K split over several blocks per column tile, ReLU + dropout (forward) and the ReluGrad mask (bwd-data) applied by
    the split-K reduction.  Against float64 on the rounded operands."""
    rng = np.random.default_rng(m + n)
    bf = torch.bfloat16
    x = torch.from_numpy(rng.standard_normal((m, k)).astype(np.float32)).cuda().to(bf)
    w = torch.from_numpy((rng.standard_normal((k, n)) / np.sqrt(k)).astype(np.float32)).cuda().to(bf)
    b = dev(rng.standard_normal(n).astype(np.float32))
    keep = dev(rng.random((m, n)) >= 0.5, torch.uint8)
    x64, w64 = x.float().cpu().numpy().astype(np.float64), w.float().cpu().numpy().astype(np.float64)
    y = torch.full((m, n), float('nan'), device='cuda')
    ops.dense_fwd_ex(x, w, b, y, 'relu', drop_keep=keep, precision='bf16', storage=ops.STORE_W | ops.STORE_X)
    ref = np.maximum(x64 @ w64 + b.cpu().numpy().astype(np.float64), 0) * 2.0 * keep.cpu().numpy()
    assert rel_l2(y.cpu().numpy(), ref) < 2e-5
    assert (y.cpu().numpy()[keep.cpu().numpy() == 0] == 0).all()
    dz = torch.from_numpy(rng.standard_normal((m, n)).astype(np.float32)).cuda().to(bf)
    dx = torch.full((m, k), float('nan'), device='cuda', dtype=bf)
    ops.dense_bwd_data_ex(dz, w, dx, mask=x, scale=1.0, precision='bf16', storage=ops.STORE_W | ops.STORE_X | ops.STORE_Y)
    dref = (dz.float().cpu().numpy().astype(np.float64) @ w64.T) * (x64 > 0)
    got = dx.float().cpu().numpy()
    assert np.isfinite(got).all() and rel_l2(got, dref) < 4e-3 and (got[x64 <= 0] == 0).all()


def test_large_problem_plans_without_split(ops):
    """An output larger than the split-K slab budget (DCNF conv2d at batch 16: 1.6 GB) must still get a plan."""
    n = 96
    d = ops.conv_desc(n, 100, 100, 3, 64, 11, 11, 1, 'VALID')
    x = torch.randn((n, 100, 100, 3), device='cuda')
    w = torch.randn((11, 11, 3, 64), device='cuda') * 0.05
    y = torch.empty((n, 90, 90, 64), device='cuda')
    ops.conv2d_fwd(d, x, w, None, y, None)
    ref = T.conv2d_fwd(x[:2].cpu().numpy().astype(np.float64), w.cpu().numpy().astype(np.float64), None, 1, 'VALID')
    assert rel_l2(y[:2].cpu().numpy(), ref) < RTOL_F32
    ref = T.conv2d_fwd(x[-1:].cpu().numpy().astype(np.float64), w.cpu().numpy().astype(np.float64), None, 1, 'VALID')
    assert rel_l2(y[-1:].cpu().numpy(), ref) < RTOL_F32


def test_bad_arguments_fail_loudly(ops):
    from ann3depth_amd._lib import A3dError
    x = torch.zeros((1, 8, 8, 4), device='cuda')
    w = torch.zeros((3, 3, 4, 4), device='cuda')
    y = torch.zeros((1, 8, 8, 4), device='cuda')
    d = ops.conv_desc(1, 8, 8, 4, 4, 3, 3, 1, 'SAME')
    d.stride = 3
    with pytest.raises(A3dError, match='stride'):
        ops.conv2d_fwd(d, x, w, None, y)


def test_tensorflow_published_vectors_through_the_c_abi(ops):
    """The HIP kernels on the known-answer vectors of TensorFlow's own unit tests (exact small integers: every fp32
    sum is exact, so equality is bit-exact whatever the accumulation order)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import tf13_published_vectors as V

    def seq(shape):
        return np.arange(1, int(np.prod(shape)) + 1, dtype=np.float32).reshape(shape)

    from ann3depth_amd._lib import A3dError
    for name, xs, ws, stride, pad, want in V.CONV2D_FWD:
        d = ops.conv_desc(xs[0], xs[1], xs[2], xs[3], ws[3], ws[0], ws[1], stride, pad)
        y = torch.empty((xs[0], d.ho, d.wo, ws[3]), device='cuda')
        if stride == 3:            # the library serves the strides the reference uses (1, 2, 4) and says so
            with pytest.raises(A3dError, match='stride 3 unsupported'):
                ops.conv2d_fwd(d, dev(seq(xs)), dev(seq(ws)), None, y, None)
            continue
        ops.conv2d_fwd(d, dev(seq(xs)), dev(seq(ws)), None, y, None)
        np.testing.assert_array_equal(y.cpu().numpy().ravel(), np.array(want, np.float32), err_msg=name)
    for name, xs, ws, os_, stride, pad, want in V.CONV2D_BACKPROP_INPUT:
        d = ops.conv_desc(xs[0], xs[1], xs[2], xs[3], ws[3], ws[0], ws[1], stride, pad)
        dx = torch.full(xs, float('nan'), device='cuda')
        ops.conv2d_bwd_data(d, dev(seq(os_)), dev(seq(ws)), dx)
        np.testing.assert_array_equal(dx.cpu().numpy().ravel(), np.array(want, np.float32), err_msg=name)
    for name, xs, ws, os_, stride, pad, want in V.CONV2D_BACKPROP_FILTER:
        d = ops.conv_desc(xs[0], xs[1], xs[2], xs[3], ws[3], ws[0], ws[1], stride, pad)
        dw = torch.full(ws, float('nan'), device='cuda')
        db = torch.full((ws[3],), float('nan'), device='cuda')
        ops.conv2d_bwd_filter(d, dev(seq(xs)), dev(seq(os_)), dw, db)
        np.testing.assert_array_equal(dw.cpu().numpy().ravel(), np.array(want, np.float32), err_msg=name)
        np.testing.assert_array_equal(db.cpu().numpy(), seq(os_).sum(axis=(0, 1, 2)), err_msg=name)
    xs, want = V.MAXPOOL_VALID
    y = torch.empty((1, 1, 1, 3), device='cuda')
    ops.maxpool2x2_fwd(dev(seq(xs)), y)
    np.testing.assert_array_equal(y.cpu().numpy().ravel(), np.array(want, np.float32))
    for name, xs, data, h, w, want in V.RESIZE_BILINEAR:
        y = torch.empty((xs[0], h, w, xs[3]), device='cuda')
        ops.resize_bilinear_tf1(dev(np.array(data, np.float32).reshape(xs)), y)
        np.testing.assert_array_equal(y.cpu().numpy().ravel(), np.array(want, np.float32), err_msg=name)
    for pad, want in V.EXTRACT_PATCHES_2X2:
        if pad != 'SAME':
            continue                                   # the C ABI serves the reference's call (SAME, src/models.py:50-59)
        want = np.array(want, np.float32)
        y = torch.empty((want.shape[1] * want.shape[2], 2, 2, 1), device='cuda')
        ops.extract_patches(dev(np.array([1, 2, 3, 4], np.float32).reshape(1, 2, 2, 1)), 2, 1, y)
        np.testing.assert_array_equal(y.cpu().numpy().reshape(want.shape), want)
    # AdamOptimizerTest.testBasic: three ApplyAdam steps of two 2-element variables, against the test's float64 recurrence
    c = V.ADAM_TEST_BASIC
    for tag in ('0', '1'):
        var, m, v = dev(np.array(c['var' + tag], np.float32)), torch.zeros(2, device='cuda'), torch.zeros(2, device='cuda')
        g = dev(np.array(c['grads' + tag], np.float32))
        p64, m64, v64 = np.array(c['var' + tag], np.float64), 0.0, 0.0
        b1p, b2p = np.float32(c['beta1']), np.float32(c['beta2'])
        for t in range(1, c['steps'] + 1):
            ops.adam_apply_tf1(var, m, v, g, c['lr'], c['beta1'], c['beta2'], c['epsilon'], float(b1p), float(b2p))
            b1p, b2p = b1p * np.float32(c['beta1']), b2p * np.float32(c['beta2'])
            alpha_t = c['lr'] * np.sqrt(1 - c['beta2'] ** t) / (1 - c['beta1'] ** t)
            g64 = np.array(c['grads' + tag], np.float64)
            m64 = c['beta1'] * m64 + (1 - c['beta1']) * g64
            v64 = c['beta2'] * v64 + (1 - c['beta2']) * g64 * g64
            p64 = p64 - alpha_t * m64 / (np.sqrt(v64) + c['epsilon'])
            np.testing.assert_allclose(var.cpu().numpy(), p64, rtol=1e-6, atol=1e-6)
    # _testMaxPoolGradDirect1's rule (all ties -> first element of the window) at the path's 2x2 / stride 2, both kernels
    ones = dev(np.ones((1, 4, 4, 4), np.float32))
    dyp = dev(np.arange(1, 17, dtype=np.float32).reshape(1, 2, 2, 4))
    dx = torch.full((1, 4, 4, 4), float('nan'), device='cuda')
    ops.maxpool2x2_bwd(ones, dyp, dx, relu_mask=False)
    want = np.zeros((1, 4, 4, 4), np.float32)
    want[:, ::2, ::2, :] = dyp.cpu().numpy()
    np.testing.assert_array_equal(dx.cpu().numpy(), want)
    d = ops.conv_desc(1, 4, 4, 4, 4, 1, 1, 1, 'VALID')         # 1x1 identity conv + pool of an all-ones image: argmax = 0
    yp = torch.empty((1, 2, 2, 4), device='cuda')
    am = torch.full((1, 2, 2, 4), 9, dtype=torch.uint8, device='cuda')
    ops.conv2d_pool_fwd(d, ones, dev(np.eye(4, dtype=np.float32).reshape(1, 1, 4, 4)), torch.zeros(4, device='cuda'), yp, 'relu', am)
    assert int(am.max()) == 0 and float(yp.min()) == 1.0
    dx.fill_(float('nan'))
    ops.maxpool2x2_bwd_idx(am, yp, dyp, dx, relu_mask=True)
    np.testing.assert_array_equal(dx.cpu().numpy(), want)
    # histogram_fixed_width's clipping (below the range -> first bin, >= upper edge -> last bin) in the colour histogram:
    # the published values scaled onto color_histogram's [0, 2^24) / 256 bins land in bins 0, 0, 76, 102, 255, 255
    c = V.HISTOGRAM_FIXED_WIDTH
    img = np.zeros((1, 40, 40, 3), np.float32)
    img[0, 0, :6, 0] = np.array(c['new_values'], np.float32) / np.float32(5.0)       # red channel carries 2^24 * value
    hist = ops.superpixel_hist(dev(img), 40).cpu().numpy().reshape(256)
    from oracle import dcnf as OD
    np.testing.assert_array_equal(hist, OD.color_histogram(OD.superpixels(np.pad(img, ((0, 0), (0, 200), (0, 280), (0, 0))))[:, :1])[0, 0])
    assert hist[76] == 1 and hist[102] == 1 and hist[255] == 2 and hist[0] == 1600 - 4
    # nn.dropout: kept units come out as x / keep_prob = 2 x (src/models.py:230), dropped ones as 0
    x1 = dev(np.ones((4, 8), np.float32))
    keep = dev((np.arange(32).reshape(4, 8) % 3 == 0).astype(np.uint8), torch.uint8)
    y = torch.empty((4, 16), device='cuda')
    ops.dense_fwd(x1, dev(np.full((8, 16), 0.125, np.float32)), torch.zeros(16, device='cuda'), y, 'relu',
                  drop_keep=dev(np.repeat((np.arange(4) % 2 == 0)[:, None], 16, 1).astype(np.uint8), torch.uint8))
    np.testing.assert_array_equal(y.cpu().numpy(), np.repeat(np.array([2, 0, 2, 0], np.float32)[:, None], 16, 1))


@pytest.mark.parametrize('n,h,w,c,k,ks,st,pad', [
    (3, 228, 304, 3, 63, 9, 2, 'VALID'),      # fine/first: 110x148 -> 55x74, window-run form, Cout = 63
    (2, 228, 304, 3, 96, 11, 4, 'VALID'),     # conv2d_0: 55x74 -> 27x37, the odd last row is never computed
    (4, 27, 37, 96, 256, 5, 1, 'SAME'),       # conv2d_1: 27x37 -> 13x18, odd row and column dropped
    (1, 9, 8, 5, 7, 3, 1, 'SAME'),            # scalar operands, tiny
    (2, 2, 2, 4, 4, 1, 1, 'VALID'),           # exactly one window per image
    (3, 33, 31, 3, 64, 11, 1, 'VALID'),       # DCNF's first conv kind (conv3.hip, stride 1): 23x21 -> 11x10, odd row and column
    (2, 21, 18, 1, 40, 4, 1, 'VALID'),        # one channel, 40 filters
])
def test_conv2d_pool_fwd_equals_conv_then_pool(ops, n, h, w, c, k, ks, st, pad):
    """The fused conv + ReLU + 2x2 max pool writes what conv2d_fwd followed by maxpool2x2_fwd writes (to fp32 summation
    order: the unfused conv may run split-K or the LDS-DMA kernel) and stays within tolerance of the float64 oracle."""
    rng = np.random.default_rng(7 * h + w + c + k)
    x = rng.standard_normal((n, h, w, c)).astype(np.float32)
    wt = (rng.standard_normal((ks, ks, c, k)) / np.sqrt(ks * ks * c)).astype(np.float32)
    b = rng.standard_normal(k).astype(np.float32)
    d = ops.conv_desc(n, h, w, c, k, ks, ks, st, pad)
    xd, wd, bd = dev(x), dev(wt), dev(b)
    y = torch.empty((n, d.ho, d.wo, k), device='cuda')
    ops.conv2d_fwd(d, xd, wd, bd, y, 'relu')
    ref = torch.empty((n, d.ho // 2, d.wo // 2, k), device='cuda')
    ops.maxpool2x2_fwd(y, ref)
    for ld in (k, k + 1):                      # dense output, and a concat buffer with one extra channel
        out = torch.full((n, d.ho // 2, d.wo // 2, ld), -7.0, device='cuda')
        ops.conv2d_pool_fwd(d, xd, wd, bd, out, 'relu')
        assert rel_l2(out[..., :k].cpu().numpy(), ref.cpu().numpy()) < 2e-6
        fused = out[..., :k].clone()
        if ld > k:
            assert bool((out[..., k] == -7.0).all())                       # the extra channel is left alone
            src = torch.rand((n, d.ho // 2, d.wo // 2, 1), device='cuda')
            ops.copy_channel(src, 0, out, k)
            assert torch.equal(out[..., k], src[..., 0]) and torch.equal(out[..., :k], fused)
    y64 = T.maxpool2x2_fwd(T.conv2d_fwd(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64), st, pad, True))
    assert rel_l2(ref.cpu().numpy(), y64) < RTOL_F32
    no_act = torch.empty_like(ref)
    ops.conv2d_pool_fwd(d, xd, wd, None, no_act, None)
    y_lin = T.maxpool2x2_fwd(T.conv2d_fwd(x.astype(np.float64), wt.astype(np.float64), np.zeros(k), st, pad, False))
    assert rel_l2(no_act.cpu().numpy(), y_lin) < RTOL_F32


@pytest.mark.parametrize('n,h,w,c,k,ks,st,pad,ld', [(3, 40, 52, 3, 63, 9, 2, 'VALID', 64), (2, 27, 37, 96, 256, 5, 1, 'SAME', 256),
                                                   (1, 9, 8, 5, 7, 3, 1, 'SAME', 7), (2, 35, 48, 3, 96, 11, 4, 'VALID', 96),
                                                   (3, 33, 31, 3, 64, 11, 1, 'VALID', 64)])
def test_maxpool_bwd_from_recorded_argmax(ops, n, h, w, c, k, ks, st, pad, ld):
    """conv2d_pool_fwd's argmax bytes + pooled values give maxpool2x2_bwd_idx everything MaxPoolGrad + ReluGrad need:
    same dx as the unfused conv -> pool -> maxpool2x2_bwd path (odd last row / column zero), and as the oracle."""
    rng = np.random.default_rng(h * w + k)
    x = rng.standard_normal((n, h, w, c)).astype(np.float32)
    wt = (rng.standard_normal((ks, ks, c, k)) / np.sqrt(ks * ks * c)).astype(np.float32)
    b = rng.standard_normal(k).astype(np.float32)
    d = ops.conv_desc(n, h, w, c, k, ks, ks, st, pad)
    ph, pw = d.ho // 2, d.wo // 2
    xd, wd, bd = dev(x), dev(wt), dev(b)
    pooled = torch.zeros((n, ph, pw, ld), device='cuda')
    arg = torch.full((n, ph, pw, k), 9, dtype=torch.uint8, device='cuda')
    ops.conv2d_pool_fwd(d, xd, wd, bd, pooled, 'relu', arg)
    assert int(arg.max()) <= 3
    dy = rng.standard_normal((n, ph, pw, ld)).astype(np.float32)          # only the first k channels are read
    dyd = dev(dy)
    dx = torch.full((n, d.ho, d.wo, k), float('nan'), device='cuda')
    ops.maxpool2x2_bwd_idx(arg, pooled, dyd, dx, relu_mask=True)
    y = torch.empty((n, d.ho, d.wo, k), device='cuda')
    ops.conv2d_fwd(d, xd, wd, bd, y, 'relu')
    dx_ref = torch.full_like(dx, float('nan'))
    ops.maxpool2x2_bwd(y, dyd, dx_ref, relu_mask=True)
    assert rel_l2(dx.cpu().numpy(), dx_ref.cpu().numpy()) < 1e-5
    y64 = T.conv2d_fwd(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64), st, pad, True)
    want = T.relu_grad(T.maxpool2x2_bwd(y64, dy[..., :k].astype(np.float64)), y64)
    assert rel_l2(dx.cpu().numpy(), want) < 1e-5
    ops.maxpool2x2_bwd_idx(arg, pooled, dyd, dx, relu_mask=False)
    assert rel_l2(dx.cpu().numpy(), T.maxpool2x2_bwd(y64, dy[..., :k].astype(np.float64))) < 1e-5


def test_fuzz_of_forced_plans_in_a_tuning_process():
    """tools/fuzz_ops.py for a few seconds in its own process (A3D_TUNING=1: the A3D_FORCE_* switches are read per launch):
    random conv / dense problems with pinned tile configurations, split-K factors and stream-K grids — tile-major and
    K-sliced shares, more blocks than iterations, up to 700 contributors per tile — against torch float64.  The planner's
    own picks cover only a few of the combinations the kernels support."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'fuzz_ops.py'), '12', '77'], capture_output=True, text=True,
                       timeout=300, cwd=root)
    assert r.returncode == 0 and 'FAIL' not in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    summary = [ln for ln in r.stdout.splitlines() if ' cases, worst rel-L2 ' in ln]
    assert len(summary) == 1 and int(summary[0].split()[0]) > 200, r.stdout[-2000:]
    assert 'round-5 kernels exercised' in r.stdout


@pytest.mark.parametrize('n,h,w,c,k,ks,st,pad', [(32, 13, 18, 384, 256, 3, 2, 'VALID'), (20, 31, 33, 64, 48, 3, 2, 'SAME'),
                                                  (3, 9, 10, 8, 8, 1, 2, 'VALID')])
def test_strided_bwd_data_on_bf16_tensors_is_one_launch(ops, n, h, w, c, k, ks, st, pad):
    """Stride-2 bwd-data with dz, the filter copy and dx all bf16 (BASELINE config 5's conv2d_4): the parity classes run as
    ONE launch of the bf16 kernel (igemm_bf16_multi_kernel) — here unequal classes, SAME padding, a 1 x 1 filter whose odd
    classes receive no tap at all (zeros) — against the float64 oracle on the bf16-rounded operands: what is left is the
    rounding of the bf16 output (2^-9 per element)."""
    rng = np.random.default_rng(11 + h + c)
    bf = torch.bfloat16
    d = ops.conv_desc(n, h, w, c, k, ks, ks, st, pad, precision='bf16')
    xb = dev(rng.standard_normal((n, h, w, c)).astype(np.float32)).to(bf)
    wb = dev((rng.standard_normal((ks, ks, c, k)) / np.sqrt(ks * ks * c)).astype(np.float32)).to(bf)
    dzb = dev(rng.standard_normal((n, d.ho, d.wo, k)).astype(np.float32)).to(bf)
    x64, w64, dz64 = (t.float().cpu().numpy().astype(np.float64) for t in (xb, wb, dzb))
    dd = ops.with_storage(d, ops.STORE_X | ops.STORE_W | ops.STORE_Y)
    dx_ref = T.conv2d_bwd_data(dz64, w64, xb.shape, st, pad)
    for mask in (None, xb):
        dx = torch.full(xb.shape, float('nan'), device='cuda', dtype=bf)
        ops.conv2d_bwd_data(dd, dzb, wb, dx, relu_mask=mask)
        want = dx_ref if mask is None else dx_ref * (x64 > 0)
        got = dx.float().cpu().numpy()
        assert np.isfinite(got).all()
        assert rel_l2(got, want) < 4e-3
        assert (got[want == 0] == 0).all()                 # pixels no tap reaches, masked pixels: exact zeros


@pytest.mark.parametrize('n,h,w,c,k,ks,st,pad', [(4, 60, 76, 3, 63, 9, 2, 'VALID'), (3, 47, 64, 3, 96, 11, 4, 'VALID'),
                                                  (2, 23, 29, 1, 40, 5, 1, 'VALID'), (4, 55, 74, 64, 64, 5, 1, 'SAME'),
                                                  (2, 13, 18, 256, 384, 3, 1, 'SAME')])
def test_share_cu_hint_changes_the_launch_not_the_result(ops, n, h, w, c, k, ks, st, pad):
    """A3D_HINT_SHARE_CU (include/a3d.h): the forward GEMM leaves half of every CU to the other stream's bandwidth-bound
    kernels — the few-channel kernel with two wavefronts per SIMD and three chunks in flight (conv3.hip), the generic
    one with one block per CU — and returns the same bits, with and without the fused pool."""
    rng = np.random.default_rng(h + k)
    x = dev(rng.standard_normal((n, h, w, c)).astype(np.float32))
    wt = dev((rng.standard_normal((ks, ks, c, k)) / np.sqrt(ks * ks * c)).astype(np.float32))
    b = dev(rng.standard_normal(k).astype(np.float32))
    plain = ops.conv_desc(n, h, w, c, k, ks, ks, st, pad)
    hinted = ops.conv_desc(n, h, w, c, k, ks, ks, st, pad, hints=ops.HINT_SHARE_CU)
    y0 = torch.full((n, plain.ho, plain.wo, k), -7.0, device='cuda')
    y1 = torch.full_like(y0, -7.0)
    ops.conv2d_fwd(plain, x, wt, b, y0, 'relu')
    ops.conv2d_fwd(hinted, x, wt, b, y1, 'relu')
    assert torch.equal(y0, y1)
    ph, pw = plain.ho // 2, plain.wo // 2
    p0, p1 = torch.full((n, ph, pw, k), -7.0, device='cuda'), torch.full((n, ph, pw, k), -7.0, device='cuda')
    a0 = torch.full((n, ph, pw, k), 9, device='cuda', dtype=torch.uint8)
    a1 = torch.full_like(a0, 9)
    ops.conv2d_pool_fwd(plain, x, wt, b, p0, 'relu', a0)
    ops.conv2d_pool_fwd(hinted, x, wt, b, p1, 'relu', a1)
    assert torch.equal(p0, p1) and torch.equal(a0, a1) and int(a0.max()) <= 3


@pytest.mark.parametrize('n,h,w,c,k,ks,st,pad,px', [
    (2, 47, 64, 3, 96, 11, 4, 'VALID', (5, 15, 0)),     # conv2d_0 kind: pixel 15 = the pad floats 33..35 of window ox = 1's run
    (2, 40, 52, 3, 63, 9, 2, 'VALID', (7, 11, 2)),      # fine/first kind: run of 27 -> 28 floats, pad = pixel 2 ox + 9, channel 0
    (2, 40, 52, 3, 63, 9, 2, 'VALID', (7, 11, 0)),
    (2, 30, 35, 3, 64, 11, 1, 'VALID', (3, 12, 1)),     # DCNF's first conv kind (stride 1)
    (2, 23, 29, 1, 40, 5, 1, 'VALID', (4, 6, 0)),       # one channel: run of 5 -> 8 floats, three pad pixels
    (2, 21, 30, 64, 64, 5, 1, 'SAME', (10, 10, 7)),     # the generic kernel (out-of-image taps read as zeros)
])
def test_a_non_finite_pixel_reaches_only_the_windows_that_contain_it(ops, n, h, w, c, k, ks, st, pad, px):
    """VERDICT r3 (weak 4) / ADVICE r3: the few-channel forward reads the pad positions of a window run from the
    NEIGHBOURING pixels; they must not reach the matrix cores (0 * inf = NaN), or an image with one non-finite pixel
    poisons windows that do not contain it, where TensorFlow's Conv2D would not (the reference treats non-finite values as
    semantics: src/models.py:262-264).  One +inf, placed where a neighbouring window's pad reads it: the output is
    non-finite exactly where the oracle's is — the windows that contain the pixel — with the oracle's sign (one infinite
    product per such output: inf * w, no NaN), and unchanged elsewhere; with and without A3D_HINT_SHARE_CU (the two
    schedules of conv3.hip) and through the fused pool."""
    rng = np.random.default_rng(7 + h + k)
    x = rng.standard_normal((n, h, w, c)).astype(np.float32)
    wt = (rng.standard_normal((ks, ks, c, k)) / np.sqrt(ks * ks * c)).astype(np.float32)
    wt[wt == 0] = 1e-3
    b = rng.standard_normal(k).astype(np.float32)
    clean = T.conv2d_fwd(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64), st, pad, relu=False)
    x[1, px[0], px[1], px[2]] = np.inf
    with np.errstate(invalid='ignore', over='ignore'):
        ref = T.conv2d_fwd(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64), st, pad, relu=False)
    hit = ~np.isfinite(ref)
    assert hit.any() and not hit[0].any() and not np.isnan(ref).any()
    assert 0 < hit[1, :, :, 0].sum() <= (-(-ks // st)) ** 2                 # a handful of windows, not the image
    np.testing.assert_array_equal(ref[~hit], clean[~hit])
    xd, wd, bd = dev(x), dev(wt), dev(b)
    for hints in (0, ops.HINT_SHARE_CU):
        d = ops.conv_desc(n, h, w, c, k, ks, ks, st, pad, hints=hints)
        y = torch.empty((n, d.ho, d.wo, k), device='cuda')
        ops.conv2d_fwd(d, xd, wd, bd, y, None)
        got = y.cpu().numpy()
        np.testing.assert_array_equal(~np.isfinite(got), hit)
        np.testing.assert_array_equal(got[hit], ref[hit])                   # +-inf, the sign of the weight it met
        assert rel_l2(got[~hit], ref[~hit]) < RTOL_F32
        if pad == 'VALID' and c <= 4:                                       # conv + ReLU + max pool in one launch
            ph, pw = d.ho // 2, d.wo // 2
            yp = torch.empty((n, ph, pw, k), device='cuda')
            ops.conv2d_pool_fwd(d, xd, wd, bd, yp, 'relu')
            with np.errstate(invalid='ignore'):
                want = np.maximum(ref, 0)[:, :2 * ph, :2 * pw].reshape(n, ph, 2, pw, 2, k).max(axis=(2, 4))
            gp = yp.cpu().numpy()
            np.testing.assert_array_equal(np.isinf(gp), np.isinf(want))
            assert not np.isnan(gp).any()
            assert rel_l2(gp[np.isfinite(want)], want[np.isfinite(want)]) < RTOL_F32


RING_CASES = [
    # n, h, w, c, k, ks, stride, pad, y bf16?, the tile the planner must pick (igemm_host.hip: kRingCfgs)
    (28, 13, 18, 256, 384, 3, 1, 'SAME', True, (128, 128)),     # conv2d_2 kind, 128-row tiles (two blocks per CU), M tail
    (106, 13, 18, 64, 200, 3, 1, 'SAME', True, (256, 128)),      # 256 x 128 tiles, N tail (200 = 128 + 72), K = 576 = 9 k-tiles
    (26, 27, 37, 96, 128, 5, 1, 'SAME', True, (128, 128)),      # conv2d_1's 96 channels: a k-tile straddles two taps; K tail (2400)
    (8, 55, 74, 64, 64, 5, 1, 'SAME', False, (256, 64)),        # fine/second kind: 64 columns, fp32 output
    (50, 27, 37, 8, 256, 3, 1, 'SAME', True, (256, 256)),       # 256 x 256 tiles; 8 channels: every 16-byte piece is its own tap; K = 72
    (64, 27, 37, 32, 128, 3, 2, 'VALID', True, (128, 128)),     # stride 2 forward (bwd-data of a strided conv is not a ring launch)
    (26, 27, 37, 96, 64, 5, 1, 'SAME', True, (256, 64)),        # bwd-data: N = 96 input channels -> the 96-column tile
]


@pytest.mark.parametrize('n,h,w,c,k,ks,st,pad,y16,tile', RING_CASES)
def test_lds_dma_kernel_on_bf16_stored_operands(ops, n, h, w, c, k, ks, st, pad, y16, tile):
    """igemm_ring.h: forward and stride-1 bwd-data of bf16-STORED tensors (BASELINE config 5's
    activations and weight copies) staged by LDS-DMA — swizzled images, per-piece tap decode, out-of-range pieces as zeros.
    Against the float64 oracle on the rounded operands: a bf16 output within bf16 rounding (4e-3 relative L2), an fp32
    output within fp32 accumulation (2e-5); the launch must have been the ring kernel with the expected tile (timing
    record), so a silent fallback to igemm_bf16 cannot pass for it."""
    from ann3depth_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(c * 1000 + k + ks)
    bf = torch.bfloat16
    x = torch.from_numpy(rng.standard_normal((n, h, w, c)).astype(np.float32)).cuda().to(bf)
    wt = torch.from_numpy((rng.standard_normal((ks, ks, c, k)) / np.sqrt(ks * ks * c)).astype(np.float32)).cuda().to(bf)
    b = dev(rng.standard_normal(k).astype(np.float32))
    X, W, Y = ops.STORE_X, ops.STORE_W, ops.STORE_Y
    d = ops.conv_desc(n, h, w, c, k, ks, ks, st, pad, precision='bf16')
    x64, w64 = x.float().cpu().numpy().astype(np.float64), wt.float().cpu().numpy().astype(np.float64)
    ydt = bf if y16 else torch.float32
    tol = 4e-3 if y16 else 2e-5

    def launched(fn):
        from bench import collect_timing
        lib.a3d_timing_select(None)
        lib.a3d_timing_enable(1)
        try:
            fn()
            torch.cuda.synchronize()
        finally:
            lib.a3d_timing_enable(0)
        recs = collect_timing(lib)
        assert len(recs) == 1
        return recs[0]

    y = torch.full((n, d.ho, d.wo, k), float('nan'), device='cuda', dtype=ydt)
    r = launched(lambda: ops.conv2d_fwd(ops.with_storage(d, X | W | (Y if y16 else 0)), x, wt, b, y, 'relu'))
    assert r.lds_dma == 3 and (r.bm, r.bn) == tile, (r.lds_dma, r.bm, r.bn)
    ref = T.conv2d_fwd(x64, w64, b.cpu().numpy().astype(np.float64), st, pad, relu=True)
    got = y.float().cpu().numpy()
    assert np.isfinite(got).all() and rel_l2(got, ref) < tol
    if st != 1:
        return
    dz = torch.from_numpy(rng.standard_normal((n, d.ho, d.wo, k)).astype(np.float32)).cuda().to(bf)
    dx = torch.full((n, h, w, c), float('nan'), device='cuda', dtype=bf)
    r = launched(lambda: ops.conv2d_bwd_data(ops.with_storage(d, X | W | Y), dz, wt, dx, relu_mask=x))
    assert r.lds_dma == 3, (r.lds_dma, r.bm, r.bn)
    if c == 96:
        assert (r.bm, r.bn) == (256, 96)
    dref = T.conv2d_bwd_data(dz.float().cpu().numpy().astype(np.float64), w64, x64.shape, st, pad) * (x64 > 0)
    got = dx.float().cpu().numpy()
    assert np.isfinite(got).all() and rel_l2(got, dref) < 4e-3
    assert (got[x64 <= 0] == 0).all()                        # the fused ReluGrad of the layer below: exact zeros
    # filter gradient: bf16 x and dz through the pixel-major LDS images (both operands by the transposing read), float32 dw
    # by split-K slabs, BiasAddGrad by the column-sum kernels; the old kernel where the GEMM has too few rows / columns
    dw = torch.full((ks, ks, c, k), float('nan'), device='cuda')
    db = torch.full((k,), float('nan'), device='cuda')
    r = launched(lambda: ops.conv2d_bwd_filter(ops.with_storage(d, X | Y), x, dz, dw, db))
    assert (r.lds_dma == 3) == (ks * ks * c >= 512 and k >= 64), (r.lds_dma, r.bm, r.bn)
    wref, bref = T.conv2d_bwd_filter(x64, dz.float().cpu().numpy().astype(np.float64), wt.shape, st, pad)
    assert rel_l2(dw.cpu().numpy(), wref) < 1e-4
    assert rel_l2(db.cpu().numpy(), bref) < 1e-5


@pytest.mark.parametrize('n,h,w,c,k,ks,pad,same_kernel', [
    (26, 27, 37, 96, 256, 5, 'SAME', True),     # conv2d_1: odd height and width (the last row / column has no window), 256 x 256 tiles
    (122, 14, 18, 64, 128, 3, 'SAME', True),    # 128 x 128 tiles, M = 122 * 7 * 9 * 4 rows: a tail tile
    (100, 21, 30, 32, 64, 3, 'VALID', True),    # 64 columns: 256 x 64 tiles
    (9, 21, 30, 32, 64, 3, 'VALID', False),     # few tiles: the unpooled conv is igemm_bf16's (another summation order), the
                                                # pooled one has no other kernel than this one
])
def test_lds_dma_kernel_with_the_max_pool_in_its_epilogue(ops, n, h, w, c, k, ks, pad, same_kernel):
    """Config 5's conv2d_1 (src/models.py:214-215): conv + ReLU + 2x2 max pool of bf16 x, w into a bf16 pooled map and argmax
    bytes in ONE launch of the LDS-DMA kernel (a3d_conv2d_pool_fwd with all three storage bits).  The pooled values are bit
    for bit those of the two launches (conv to a bf16 tensor, a3d_maxpool2x2_fwd_bf16); the argmax bytes name a position
    that holds the window's maximum — the first one in scan order; and a3d_maxpool2x2_bwd_idx_bf16s routes a bf16 gradient
    exactly as a3d_maxpool2x2_bwd_bf16 does from the unpooled activations."""
    rng = np.random.default_rng(k + ks + n)
    bf = torch.bfloat16
    x = torch.from_numpy(rng.standard_normal((n, h, w, c)).astype(np.float32)).cuda().to(bf)
    wt = torch.from_numpy((rng.standard_normal((ks, ks, c, k)) / np.sqrt(ks * ks * c)).astype(np.float32)).cuda().to(bf)
    b = dev(rng.standard_normal(k).astype(np.float32) * 0.1)
    X, W, Y = ops.STORE_X, ops.STORE_W, ops.STORE_Y
    d = ops.with_storage(ops.conv_desc(n, h, w, c, k, ks, ks, 1, pad, precision='bf16'), X | W | Y)
    y = torch.empty((n, d.ho, d.wo, k), device='cuda', dtype=bf)
    ops.conv2d_fwd(d, x, wt, b, y, 'relu')
    ph, pw = d.ho // 2, d.wo // 2
    two = torch.empty((n, ph, pw, k), device='cuda', dtype=bf)
    ops.maxpool2x2_fwd_bf16(y, two)
    one = torch.full((n, ph, pw, k), float('nan'), device='cuda', dtype=bf)
    arg = torch.full((n, ph, pw, k), 9, device='cuda', dtype=torch.uint8)
    ops.conv2d_pool_fwd(d, x, wt, b, one, 'relu', arg)
    assert int(arg.max()) <= 3
    if not same_kernel:        # values within a bf16 rounding step of the two launches; routing checked on the kernel's own values below
        assert rel_l2(one.float().cpu().numpy(), two.float().cpu().numpy()) < 4e-3
        assert (one.view(torch.int16) != two.view(torch.int16)).float().mean() < 1e-3
        return
    assert torch.equal(one.view(torch.int16), two.view(torch.int16))
    yw = y[:, :2 * ph, :2 * pw, :].reshape(n, ph, 2, pw, 2, k).permute(0, 1, 3, 5, 2, 4).reshape(n, ph, pw, k, 4).float()
    vmax = yw.max(dim=-1, keepdim=True).values
    first = (yw == vmax).float().argmax(dim=-1)                # the first maximal position, as MaxPoolGrad scans a window
    assert torch.equal(arg.long(), first)
    ops.conv2d_pool_fwd(d, x, wt, b, one.fill_(float('nan')), 'relu', None)          # without argmax: the same map
    assert torch.equal(one.view(torch.int16), two.view(torch.int16))
    dy = torch.from_numpy(rng.standard_normal((n, ph, pw, k)).astype(np.float32)).cuda().to(bf)
    dx_ref = torch.full((n, d.ho, d.wo, k), float('nan'), device='cuda', dtype=bf)
    ops.maxpool2x2_bwd_bf16(y, dy, dx_ref, relu_mask=True)
    dx = torch.full((n, d.ho, d.wo, k), float('nan'), device='cuda', dtype=bf)
    ops.maxpool2x2_bwd_idx(arg, one, dy, dx, relu_mask=True)
    assert torch.equal(dx.view(torch.int16), dx_ref.view(torch.int16))


@pytest.mark.parametrize('kind,n', [('conv2d_1', 8), ('conv2d_0', 4)])
def test_pooled_epilogues_of_config5_against_the_oracle(ops, kind, n):
    """config 5's first two layers (conv2d_0: float32 image, bf16 pooled map; conv2d_1 at its own
    geometry: bf16 x / w / pooled map in the LDS-DMA kernel's epilogue) meet the ORACLE at layer tolerance, not only through
    the network test: pooled map vs maxpool2x2(relu(conv2d(x, w) + b)) in float64 on the operands as stored (x, w rounded to
    bf16 where they are bf16 tensors) at 4e-3 (one bf16 rounding of the output), and the argmax bytes vs the oracle's first
    maximum wherever the window's two largest values differ by more than a bf16 rounding step of the larger."""
    rng = np.random.default_rng(len(kind) + n)
    bf = torch.bfloat16
    if kind == 'conv2d_1':
        h, w, c, k, ks, st, pad = 27, 37, 96, 256, 5, 1, 'SAME'
        store = ops.STORE_X | ops.STORE_W | ops.STORE_Y
        x = torch.from_numpy(rng.standard_normal((n, h, w, c)).astype(np.float32)).cuda().to(bf)
        wt = torch.from_numpy((rng.standard_normal((ks, ks, c, k)) / np.sqrt(ks * ks * c)).astype(np.float32)).cuda().to(bf)
    else:
        h, w, c, k, ks, st, pad = 228, 304, 3, 96, 11, 4, 'VALID'
        store = ops.STORE_Y
        x = dev(rng.random((n, h, w, c)).astype(np.float32))
        wt = dev((rng.standard_normal((ks, ks, c, k)) / np.sqrt(ks * ks * c)).astype(np.float32))
    b = dev(rng.standard_normal(k).astype(np.float32) * 0.1)
    d = ops.with_storage(ops.conv_desc(n, h, w, c, k, ks, ks, st, pad, precision='bf16'), store)
    ph, pw = d.ho // 2, d.wo // 2
    one = torch.full((n, ph, pw, k), float('nan'), device='cuda', dtype=bf)
    arg = torch.full((n, ph, pw, k), 9, device='cuda', dtype=torch.uint8)
    ops.conv2d_pool_fwd(d, x, wt, b, one, 'relu', arg)
    torch.cuda.synchronize()
    # the oracle on what the kernel multiplies: bf16 operands (conv2d_0 rounds its float32 image and filter to bf16 on the way in)
    xr = x.to(bf).double().cpu().numpy()
    wr = wt.to(bf).double().cpu().numpy()
    y = T.conv2d_fwd(xr, wr, b.double().cpu().numpy(), st, pad, relu=True)
    yw = y[:, :2 * ph, :2 * pw, :].reshape(n, ph, 2, pw, 2, k).transpose(0, 1, 3, 5, 2, 4).reshape(n, ph, pw, k, 4)
    pooled = yw.max(-1)
    got = one.float().cpu().numpy()
    assert rel_l2(got, pooled) < 4e-3
    first = (yw == pooled[..., None]).argmax(-1)
    srt = np.sort(yw, axis=-1)
    clear = (srt[..., 3] - srt[..., 2]) > np.abs(srt[..., 3]) * 2.0 ** -7          # the runner-up is more than a bf16 step away
    assert clear.mean() > 0.5
    a = arg.cpu().numpy()
    assert int(a.max()) <= 3
    assert (a[clear] == first[clear]).all()
    # and everywhere the recorded position holds the window's maximum to a bf16 rounding step
    at = np.take_along_axis(yw, a[..., None].astype(np.int64), -1)[..., 0]
    assert rel_l2(at, pooled) < 4e-3


@pytest.mark.parametrize('n,h,w,k,ks,st', [(6, 100, 132, 96, 11, 4), (5, 61, 80, 63, 9, 2)])
def test_few_channel_conv_with_bf16_pooled_map_from_a_float32_image(ops, n, h, w, k, ks, st):
    """Config 5's conv2d_0 (src/models.py:211-213): float32 image and filter, bf16 arithmetic, conv + ReLU + 2x2 max pool in one
    launch with a bf16 pooled map and argmax bytes (a3d_conv2d_pool_fwd, storage A3D_STORE_Y_BF16).  Against the two launches
    (the same conv to a bf16 tensor, then a3d_maxpool2x2_fwd_bf16): equal to a bf16 rounding step (the unpooled conv may add
    its K range in another order), the recorded position holds the window's maximum to that step, and the by-index
    MaxPoolGrad to a bf16 gradient routes dy exactly where the bytes say."""
    rng = np.random.default_rng(k + ks)
    bf = torch.bfloat16
    x = dev(rng.standard_normal((n, h, w, 3)).astype(np.float32))
    wt = dev((rng.standard_normal((ks, ks, 3, k)) / np.sqrt(ks * ks * 3)).astype(np.float32))
    b = dev(rng.standard_normal(k).astype(np.float32) * 0.1)
    d = ops.with_storage(ops.conv_desc(n, h, w, 3, k, ks, ks, st, 'VALID', precision='bf16'), ops.STORE_Y)
    y = torch.empty((n, d.ho, d.wo, k), device='cuda', dtype=bf)
    ops.conv2d_fwd(d, x, wt, b, y, 'relu')
    ph, pw = d.ho // 2, d.wo // 2
    two = torch.empty((n, ph, pw, k), device='cuda', dtype=bf)
    ops.maxpool2x2_fwd_bf16(y, two)
    one = torch.full((n, ph, pw, k), float('nan'), device='cuda', dtype=bf)
    arg = torch.full((n, ph, pw, k), 9, device='cuda', dtype=torch.uint8)
    ops.conv2d_pool_fwd(d, x, wt, b, one, 'relu', arg)
    assert int(arg.max()) <= 3 and bool(torch.isfinite(one.float()).all())
    assert rel_l2(one.float().cpu().numpy(), two.float().cpu().numpy()) < 4e-3
    yw = y[:, :2 * ph, :2 * pw, :].reshape(n, ph, 2, pw, 2, k).permute(0, 1, 3, 5, 2, 4).reshape(n, ph, pw, k, 4).float()
    at_arg = yw.gather(-1, arg.long().unsqueeze(-1)).squeeze(-1)
    assert rel_l2(at_arg.cpu().numpy(), yw.max(-1).values.cpu().numpy()) < 4e-3
    if k % 8 == 0:
        dy = torch.from_numpy(rng.standard_normal((n, ph, pw, k)).astype(np.float32)).cuda().to(bf)
        dx = torch.full((n, d.ho, d.wo, k), float('nan'), device='cuda', dtype=bf)
        ops.maxpool2x2_bwd_idx(arg, one, dy, dx, relu_mask=True)
        routed = torch.where(one.float() > 0, dy.float(), torch.zeros_like(dy.float()))
        exp = torch.zeros((n, ph, pw, k, 4), device='cuda')
        exp.scatter_(-1, arg.long().unsqueeze(-1), routed.unsqueeze(-1))
        full = torch.zeros((n, d.ho, d.wo, k), device='cuda')
        full[:, :2 * ph, :2 * pw] = exp.reshape(n, ph, pw, k, 2, 2).permute(0, 1, 4, 2, 5, 3).reshape(n, 2 * ph, 2 * pw, k)
        assert torch.equal(dx.float(), full)


def test_cast_rows_between_row_pitches(ops):
    """a3d_cast_rows: float32 / bf16 either way, the first `cols` columns of every row, pad columns of the destination zero."""
    rng = np.random.default_rng(3)
    a = torch.from_numpy(rng.standard_normal((37, 4070)).astype(np.float32)).cuda()
    p = torch.full((37, 4072), float('nan'), device='cuda', dtype=torch.bfloat16)
    ops.cast_rows(a, p)
    assert torch.equal(p[:, :4070], a.to(torch.bfloat16)) and not p[:, 4070:].any()
    back = torch.full((37, 4070), float('nan'), device='cuda')
    ops.cast_rows(p, back)
    assert torch.equal(back, a.to(torch.bfloat16).float())
    f = torch.full((37, 4072), float('nan'), device='cuda')
    ops.cast_rows(a, f)
    assert torch.equal(f[:, :4070], a) and not f[:, 4070:].any()
    q = torch.full((37, 100), float('nan'), device='cuda', dtype=torch.bfloat16)
    ops.cast_rows(p, q, cols=90)
    assert torch.equal(q[:, :90], p[:, :90]) and not q[:, 90:].any()


def test_timing_brackets_every_launch_or_one_kernel(ops):
    """a3d_timing_enable / a3d_timing_select (include/a3d.h): bench.py learns the dominant kernel with every launch
    bracketed, then brackets only that kernel inside its timed region."""
    from ann3depth_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(5)
    small = ops.conv_desc(2, 13, 18, 64, 64, 3, 3, 1, 'SAME')          # a 64 x 64 tile configuration
    big = ops.conv_desc(16, 27, 37, 96, 256, 5, 5, 1, 'SAME')          # a 128 x 128 one
    args = {}
    for name, d in (('small', small), ('big', big)):
        x = dev(rng.standard_normal((d.n, d.h, d.w, d.c)).astype(np.float32))
        w = dev(rng.standard_normal((d.r, d.s, d.c, d.k)).astype(np.float32))
        args[name] = (d, x, w, dev(np.zeros(d.k, np.float32)), torch.empty((d.n, d.ho, d.wo, d.k), device='cuda'))

    def collect():
        arr = (_lib.TimingRecord * 64)()
        return [arr[i] for i in range(lib.a3d_timing_collect(arr, 64))]

    def run():
        lib.a3d_timing_enable(1)
        for name in ('small', 'big', 'small'):
            ops.conv2d_fwd(*args[name], None)
        torch.cuda.synchronize()
        lib.a3d_timing_enable(0)
        return collect()

    try:
        every = run()
        assert len(every) == 3 and every[0].bm * every[0].bn < every[1].bm * every[1].bn
        assert all(r.ms > 0 for r in every)
        lib.a3d_timing_select(every[1])
        only = run()
        assert len(only) == 1 and (only[0].bm, only[0].bn, only[0].m) == (every[1].bm, every[1].bn, every[1].m) and only[0].ms > 0
        lib.a3d_timing_select(None)
        assert len(run()) == 3
    finally:
        lib.a3d_timing_enable(0)
        lib.a3d_timing_select(None)
        collect()


def test_second_generation_kernel_with_pinned_plans():
    """csrc/igemm2.h (tile configuration 11: LDS-DMA staging, four waves of 64 x 64, pinned instruction order) is not what the
    planner picks (measured 2-15 % behind the first-generation kernels), so its parity is checked here with the
    plan pinned in a tuning process: eleven layer shapes x {plain, split-K 2 / 3 / 5, stream-K grids of 3 ... 700 blocks} in all
    three directions and the fused max pool, against torch float64 (tools/gen2_check.py)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'gen2_check.py')], capture_output=True, text=True,
                       timeout=600, cwd=root)
    assert r.returncode == 0 and 'FAIL' not in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    assert '113 gen-2 checks' in r.stdout and ' 0 failed' in r.stdout, r.stdout[-2000:]
