"""The oracle's TF-1.3 op restatements against an INDEPENDENT implementation (torch-CPU float64 with
explicit padding) and hand-computed known answers.  CPU only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import tf13_ops as T

RNG = np.random.default_rng(7)


def _t(x):  # NHWC numpy -> NCHW torch f64
    return torch.from_numpy(np.ascontiguousarray(x)).double().permute(0, 3, 1, 2)


def _n(t):  # NCHW torch -> NHWC numpy
    return t.permute(0, 2, 3, 1).contiguous().numpy()


CONV_CASES = [
    # H, W, Cin, Cout, k, stride, padding
    (23, 31, 3, 8, 11, 4, 'VALID'),     # conv2d_0-like
    (9, 13, 6, 10, 5, 1, 'SAME'),       # conv2d_1-like
    (13, 18, 5, 7, 3, 1, 'SAME'),
    (13, 18, 4, 6, 3, 2, 'VALID'),      # conv2d_4-like: 13x18 -> 6x8
    (24, 30, 3, 63, 9, 2, 'VALID'),     # fine/first-like (Cout=63)
    (7, 9, 64, 1, 5, 1, 'SAME'),        # fine/third-like (Cout=1)
    (10, 11, 3, 4, 4, 2, 'SAME'),       # asymmetric SAME padding (1 before, 2 after)
    (8, 8, 2, 3, 3, 2, 'SAME'),         # SAME stride 2, pad 0 before / 1 after
]


@pytest.mark.parametrize('H,W,C,K,k,s,pad', CONV_CASES)
def test_conv2d_fwd_bwd_vs_torch(H, W, C, K, k, s, pad):
    x = RNG.standard_normal((2, H, W, C))
    w = RNG.standard_normal((k, k, C, K))
    b = RNG.standard_normal(K)
    Ho, pt, pb = T.conv_out_size(H, k, s, pad)
    Wo, pl, pr = T.conv_out_size(W, k, s, pad)
    assert T.conv2d_fwd(x, w, b, s, pad).shape == (2, Ho, Wo, K)
    xt = _t(x).requires_grad_(True)
    wt = torch.from_numpy(w).permute(3, 2, 0, 1).contiguous().requires_grad_(True)
    bt = torch.from_numpy(b).requires_grad_(True)
    yt = F.conv2d(F.pad(xt, (pl, pr, pt, pb)), wt, bt, stride=s)
    y = T.conv2d_fwd(x, w, b, s, pad, relu=False)
    np.testing.assert_allclose(y, _n(yt.detach()), rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(T.conv2d_fwd(x, w, b, s, pad, relu=True), np.maximum(y, 0))
    dz = RNG.standard_normal(y.shape)
    yt.backward(_t(dz))
    dw, db = T.conv2d_bwd_filter(x, dz, w.shape, s, pad)
    dx = T.conv2d_bwd_data(dz, w, x.shape, s, pad)
    np.testing.assert_allclose(dw, wt.grad.permute(2, 3, 1, 0).numpy(), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(db, bt.grad.numpy(), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(dx, _n(xt.grad), rtol=1e-9, atol=1e-9)


def test_same_padding_known_answers():
    # TF: out = ceil(in/s); pad_total = max((out-1)*s + k - in, 0); before = total//2
    assert T.conv_out_size(27, 5, 1, 'SAME') == (27, 2, 2)
    assert T.conv_out_size(13, 3, 1, 'SAME') == (13, 1, 1)
    assert T.conv_out_size(240, 100, 40, 'SAME') == (6, 30, 30)     # dcnf patches rows
    assert T.conv_out_size(320, 100, 40, 'SAME') == (8, 30, 30)     # dcnf patches cols
    assert T.conv_out_size(10, 4, 2, 'SAME') == (5, 1, 1)
    assert T.conv_out_size(11, 4, 2, 'SAME') == (6, 1, 2)
    assert T.conv_out_size(228, 11, 4, 'VALID') == (55, 0, 0)
    assert T.conv_out_size(304, 11, 4, 'VALID') == (74, 0, 0)
    assert T.conv_out_size(13, 3, 2, 'VALID') == (6, 0, 0)
    assert T.conv_out_size(18, 3, 2, 'VALID') == (8, 0, 0)
    assert T.conv_out_size(228, 9, 2, 'VALID') == (110, 0, 0)
    assert T.conv_out_size(304, 9, 2, 'VALID') == (148, 0, 0)


@pytest.mark.parametrize('H,W', [(55, 74), (27, 37), (110, 148), (4, 5)])
def test_maxpool_vs_torch(H, W):
    x = RNG.standard_normal((2, H, W, 3))
    xt = _t(x).requires_grad_(True)
    yt = F.max_pool2d(xt, 2, 2)
    y = T.maxpool2x2_fwd(x)
    assert y.shape == (2, H // 2, W // 2, 3)
    np.testing.assert_array_equal(y, _n(yt.detach()))
    dy = RNG.standard_normal(y.shape)
    yt.backward(_t(dy))
    np.testing.assert_array_equal(T.maxpool2x2_bwd(x, dy), _n(xt.grad))


def test_maxpool_bwd_tie_goes_to_first():
    x = np.zeros((1, 2, 2, 1))
    dx = T.maxpool2x2_bwd(x, np.full((1, 1, 1, 1), 3.0))
    assert dx[0, 0, 0, 0] == 3.0 and dx.sum() == 3.0
    x[0, 1, 0, 0] = x[0, 1, 1, 0] = 5.0          # tie between (1,0) and (1,1): first in scan order wins
    dx = T.maxpool2x2_bwd(x, np.full((1, 1, 1, 1), 3.0))
    assert dx[0, 1, 0, 0] == 3.0 and dx.sum() == 3.0


def test_dense_and_dropout():
    x = RNG.standard_normal((4, 9))
    w = RNG.standard_normal((9, 5))
    b = RNG.standard_normal(5)
    np.testing.assert_allclose(T.dense_fwd(x, w, b, 'relu'), np.maximum(x @ w + b, 0))
    np.testing.assert_allclose(T.dense_fwd(x, w, b, 'sigmoid'), 1 / (1 + np.exp(-(x @ w + b))))
    dz = RNG.standard_normal((4, 5))
    dx, dw, db = T.dense_bwd(x, w, dz)
    xt = torch.from_numpy(x).requires_grad_(True)
    wt = torch.from_numpy(w).requires_grad_(True)
    (xt @ wt).backward(torch.from_numpy(dz))
    np.testing.assert_allclose(dx, xt.grad.numpy(), rtol=1e-12)
    np.testing.assert_allclose(dw, wt.grad.numpy(), rtol=1e-12)
    np.testing.assert_allclose(db, dz.sum(0))
    m = RNG.random((4, 9)) >= 0.5
    y = T.dropout_fwd(x, m)
    np.testing.assert_array_equal(y, np.where(m, 2 * x, 0))
    np.testing.assert_array_equal(T.dropout_bwd(x, m), np.where(m, 2 * x, 0))


def test_resize_known_answers():
    # exact 2x decimation (480 -> 240): lerp == 0 everywhere, picks even rows/cols
    x = RNG.random((1, 8, 12, 2)).astype(np.float32)
    np.testing.assert_array_equal(T.resize_bilinear_tf1(x, 4, 6), x[:, ::2, ::2])
    # legacy mapping src = dst * in/out (NO half pixel): 4 -> 3, scale 4/3: src = 0, 1.333, 2.667
    row = np.array([0., 10., 20., 30.], np.float32).reshape(1, 1, 4, 1)
    out = T.resize_bilinear_tf1(row, 1, 3).ravel()
    np.testing.assert_allclose(out, [0., 13.333333, 26.666666], rtol=1e-6)
    # upscale 2 -> 4: src = 0, .5, 1, 1.5 ; hi clamps to in-1 so the last sample equals the edge
    row = np.array([0., 10.], np.float32).reshape(1, 1, 2, 1)
    np.testing.assert_allclose(T.resize_bilinear_tf1(row, 1, 4).ravel(), [0., 5., 10., 10.])
    # identity
    np.testing.assert_array_equal(T.resize_bilinear_tf1(x, 8, 12), x)
    # torch's align_corners=False uses half-pixel centres: must DIFFER (guards against using it as oracle)
    big = RNG.random((1, 48, 64, 1)).astype(np.float32)
    tor = F.interpolate(torch.from_numpy(big).permute(0, 3, 1, 2), size=(11, 15), mode='bilinear',
                        align_corners=False).permute(0, 2, 3, 1).numpy()
    assert np.abs(T.resize_bilinear_tf1(big, 11, 15) - tor).max() > 1e-2


def test_resize_matches_explicit_loops():
    x = RNG.random((2, 48, 64, 3)).astype(np.float32)
    out = T.resize_bilinear_tf1(x, 23, 30)
    sy, sx = np.float32(48) / np.float32(23), np.float32(64) / np.float32(30)
    for (b, i, j, c) in [(0, 0, 0, 0), (1, 22, 29, 2), (0, 7, 13, 1), (1, 15, 3, 0)]:
        fy, fx = np.float32(i) * sy, np.float32(j) * sx
        y0, x0 = int(fy), int(fx)
        y1, x1 = min(y0 + 1, 47), min(x0 + 1, 63)
        ly, lx = fy - np.float32(y0), fx - np.float32(x0)
        top = x[b, y0, x0, c] + (x[b, y0, x1, c] - x[b, y0, x0, c]) * lx
        bot = x[b, y1, x0, c] + (x[b, y1, x1, c] - x[b, y1, x0, c]) * lx
        assert out[b, i, j, c] == np.float32(top + (bot - top) * ly)


def test_extract_patches_vs_torch_unfold():
    x = RNG.standard_normal((2, 24, 32, 3))
    p = T.extract_patches(x, 10, 4, 'SAME')
    Ho, pt, pb = T.conv_out_size(24, 10, 4, 'SAME')
    Wo, pl, pr = T.conv_out_size(32, 10, 4, 'SAME')
    assert p.shape == (2, Ho * Wo, 10, 10, 3)
    u = F.unfold(F.pad(_t(x), (pl, pr, pt, pb)), 10, stride=4)          # [B, C*100, L]
    u = u.reshape(2, 3, 10, 10, Ho * Wo).permute(0, 4, 2, 3, 1).numpy()
    np.testing.assert_array_equal(p, u)


def test_silog_loss_known_answer_and_grad():
    o = np.array([[1.0, 2.0, 4.0]], np.float64)
    t = np.array([[2.0, 2.0, 1.0]], np.float64)
    d = np.log(o + 1e-8) - np.log(t + 1e-8)
    expect = (d ** 2).sum() - 0.5 / 4070 * d.sum() ** 2
    assert abs(T.silog_loss_fwd(o, t) - expect) < 1e-12
    # autograd check incl. the NaN-masked branch (o < -eps -> log NaN -> 0, gradient 0)
    o = RNG.standard_normal((3, 50)) * 0.5
    t = RNG.random((3, 50))
    ot = torch.from_numpy(o).requires_grad_(True)
    tt = torch.from_numpy(t)
    lo = torch.log(ot + 1e-8)
    lo = torch.where(torch.isnan(lo), torch.zeros_like(lo), lo)
    dd = lo - torch.log(tt + 1e-8)
    loss = ((dd ** 2).sum(1) - 0.5 / (74 * 55) * dd.sum(1) ** 2).mean()
    loss.backward()
    assert abs(T.silog_loss_fwd(o, t) - loss.item()) < 1e-9 * abs(loss.item())
    g = T.silog_loss_bwd(o, t)
    np.testing.assert_allclose(g, ot.grad.numpy(), rtol=1e-9, atol=1e-12)
    assert (g[o < -1e-8] == 0).all() and (o < -1e-8).any()


def test_silog_loss_edge_semantics():
    # o == -eps exactly -> log(0) = -inf is KEPT (only NaN is replaced) -> sum d^2 = inf, (sum d)^2 = inf,
    # inf - c*inf = NaN, exactly as the TF graph would produce ; t == 0 -> log(1e-8) finite
    o = np.array([[-1e-8, 1.0]], np.float64)
    t = np.array([[0.0, 1.0]], np.float64)
    with np.errstate(invalid='ignore'):
        assert np.isnan(T.silog_loss_fwd(o, t))
    o = np.array([[-0.5, 1.0]], np.float64)        # NaN log -> 0 ; d0 = 0 - log(1e-8)
    l = T.silog_loss_fwd(o, t)
    d0 = -np.log(1e-8)
    d1 = np.log(1 + 1e-8) - np.log(1 + 1e-8)
    assert abs(l - ((d0 ** 2 + d1 ** 2) - 0.5 / 4070 * (d0 + d1) ** 2)) < 1e-9


def test_adam_tf1_formula_and_beta2_one():
    var = {'w': np.array([1.0, -2.0, 3.0], np.float32)}
    g = {'w': np.array([0.5, 0.25, -1.0], np.float32)}
    opt = T.AdamTF1(0.1, 0.9, 0.999)
    opt.apply(var, g)
    # step 1 of textbook Adam: m_hat = g, v_hat = g^2 -> var -= lr * g/(|g| + eps') ~ lr * sign(g)
    np.testing.assert_allclose(var['w'], [0.9, -2.1, 3.1], rtol=1e-5)
    np.testing.assert_allclose(opt.m['w'], 0.1 * g['w'], rtol=1e-6)
    np.testing.assert_allclose(opt.v['w'], 0.001 * g['w'] ** 2, rtol=1e-4)
    assert np.isclose(opt.beta1_power, 0.81) and np.isclose(opt.beta2_power, 0.999 ** 2)
    # the reference's AdamOptimizer(rate, 0.9, 1): alpha == 0, v stays 0, var never moves, m evolves
    var = {'w': np.array([1.0, -2.0, 3.0], np.float32)}
    ref = var['w'].copy()
    opt = T.AdamTF1(0.1, 0.9, 1.0)
    for _ in range(3):
        opt.apply(var, g)
    assert opt.alpha() == 0
    np.testing.assert_array_equal(var['w'], ref)
    np.testing.assert_array_equal(opt.v['w'], 0)
    np.testing.assert_allclose(opt.m['w'], g['w'] * (1 - 0.9 ** 3), rtol=1e-5)


def test_glorot_limits():
    w = T.glorot_uniform(np.random.default_rng(0), (11, 11, 3, 96))
    lim = np.sqrt(6 / (121 * 3 + 121 * 96))
    assert abs(w).max() <= lim * (1 + 1e-6) and abs(w).max() > 0.98 * lim
    w = T.glorot_uniform(np.random.default_rng(0), (12288, 4096))
    lim = np.sqrt(6 / (12288 + 4096))
    assert abs(w).max() <= lim * (1 + 1e-6) and abs(w).max() > 0.999 * lim


def test_dcnf_crf_oracle_against_autograd():
    """oracle.dcnf pairwise + CRF loss: structure checks and d loss / d z against torch float64 autograd of the same
    expression with A detached (the assumed TF-1.3 behaviour, see oracle/dcnf.py)."""
    from oracle import dcnf as D
    left, right = D.pair_indices()
    assert len(left) == 48 and D.N_SP == 48
    assert set(left) == {9, 11, 13, 18, 20, 22, 25, 27, 29, 34, 36, 38}          # interior checkerboard
    assert len({tuple(sorted(e)) for e in zip(left, right)}) == 48                # no duplicated edge
    rng = np.random.default_rng(1)
    B = 2
    img = (rng.integers(0, 256, (B, 240, 320, 3)) / 255)
    dep = (rng.integers(0, 256, (B, 240, 320, 1)) / 255)
    sp = D.superpixels(img)
    np.testing.assert_array_equal(sp[1, 9, 41, :], img[1, 40 + 1, 40 + 1, :])      # superpixel 9 = (row 1, col 1)
    hist = D.color_histogram(sp)
    assert hist.shape == (B, 48, 256) and (hist.sum(-1) == 1600).all()
    p = D.pairwise_init()
    p = {k: v.astype(np.float64) for k, v in p.items()}
    r, sims = D.pairwise_forward(p, img)
    assert r.shape == (B, 48, 1) and (sims > 0).all() and (sims <= 1).all()
    A = D.crf_matrix(r[0, :, 0])
    np.testing.assert_allclose(A, A.T)
    np.testing.assert_allclose(A.sum(axis=1), 1.0, atol=1e-12)                     # I + D - R: rows sum to 1
    z = rng.random((B, 48, 1)) * 0.2
    # keep the energies small so that the gradient is not lost in the + 1e-7 of the reference's log
    y = D.superpixels(dep).mean(axis=2)
    z = y + 0.05 * rng.standard_normal(y.shape)
    loss, losses, dz = D.crf_loss(dep, z, r)
    zt = torch.from_numpy(z).requires_grad_(True)
    tot = 0
    for b in range(B):
        At = torch.from_numpy(D.crf_matrix(r[b, :, 0]))
        zb, yb = zt[b, :, 0], torch.from_numpy(y[b, :, 0])
        E = yb @ At @ yb - 2 * zb @ yb + zb @ zb
        fac = np.pi ** 24 / (torch.sqrt(torch.det(At)) + 1e-7)
        g = zb @ (torch.inverse(At) + 1e-7) @ zb - zb @ zb
        Z = fac * torch.exp(g) + 1e-7
        tot = tot + -torch.log(torch.exp(-E) / Z + 1e-7)
    (tot / B).backward()
    assert abs(loss - (tot / B).item()) < 1e-9
    np.testing.assert_allclose(dz, zt.grad.numpy(), rtol=1e-6, atol=1e-18)
    assert abs(loss - 16.118) < 0.01         # exp(-E)/Z ~ 1e-13 drowns in the + epsilon: loss ~ -log(1e-7)
