"""The oracle's whole MSDN graph and DCNF unary stack against an INDEPENDENT restatement of the reference's model
functions (src/models.py:203-296, :61-83) in torch float64 with autograd: pins the wiring — which tensor feeds which
layer, the dropout mask, the concat, the loss formula with its tf.where masking, and that the hand-written backward
passes are the gradients of exactly those losses with respect to exactly the reference's variable lists."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import dcnf as OD
from oracle import msdn as O
from oracle import tf13_ops as T


def t64(a):
    return torch.from_numpy(np.asarray(a, np.float64))


def conv(x, w, b, stride, same, relu):
    """tf.layers.conv2d on NHWC / HWIO data (odd kernels, stride 1 when `same`)."""
    pad = w.shape[0] // 2 if same else 0
    y = F.conv2d(x.permute(0, 3, 1, 2).contiguous(), w.permute(3, 2, 0, 1).contiguous(), b, stride=stride,
                 padding=pad).permute(0, 2, 3, 1)
    return torch.relu(y) if relu else y


def pool(x):
    return F.max_pool2d(x.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)


def silog(outputs, targets):
    """src/models.py:255-275."""
    o = outputs.reshape(outputs.shape[0], -1)
    t = targets.reshape(targets.shape[0], -1)
    lo = torch.log(o + 1e-8)
    lo = torch.where(torch.isnan(lo), torch.zeros_like(lo), lo)
    lt = torch.log(t + 1e-8)
    lt = torch.where(torch.isnan(lt), torch.zeros_like(lt), lt)
    l2 = ((lo - lt) ** 2).sum(1)
    si = (lo - lt).sum(1) ** 2
    return (l2 - 0.5 / (74 * 55) * si).mean()


def test_msdn_graph_and_both_backward_passes():
    rng = np.random.default_rng(7)
    B = 1
    img = (rng.integers(0, 256, (B, 60, 80, 3)) / 255).astype(np.float64)
    dep = (rng.integers(1, 256, (B, 60, 80, 1)) / 255).astype(np.float64)
    keep = rng.random((B, 4096)) >= 0.5
    params = {k: v.astype(np.float64) for k, v in O.init_params(11).items()}
    for k in params:                                   # non-zero biases so that their wiring is visible too
        if k.endswith('/bias'):
            params[k] = 0.01 * rng.standard_normal(params[k].shape)
    # slightly positive final biases: most outputs on the differentiable side of log(), some not (tf.where masking)
    params['coarse/dense/dense_1/bias'] += 0.03
    params['fine/third/bias'] += 0.03
    a = O.forward(params, img, dep, keep)
    gc = O.backward_coarse(params, a)
    gf = O.backward_fine(params, a)

    P = {k: t64(v).requires_grad_(True) for k, v in params.items()}
    x = t64(T.resize_bilinear_tf1(img, 228, 304))      # the resize itself is pinned in test_oracle_ops / TF vectors
    tgt = t64(T.resize_bilinear_tf1(dep, 55, 74))

    def c(name, inp, stride=1, same=False, relu=True):
        return conv(inp, P[name + '/kernel'], P[name + '/bias'], stride, same, relu)
    t = pool(c('coarse/conv/conv2d_0', x, 4))
    t = pool(c('coarse/conv/conv2d_1', t, same=True))
    t = c('coarse/conv/conv2d_2', t, same=True)
    t = c('coarse/conv/conv2d_3', t, same=True)
    t = c('coarse/conv/conv2d_4', t, 2)
    t = t.reshape(B, -1)
    t = torch.relu(t @ P['coarse/dense/dense_0/kernel'] + P['coarse/dense/dense_0/bias'])
    t = t * t64(keep.astype(np.float64)) * 2.0          # tf.layers.dropout(rate 0.5, training): kept units / keep_prob
    coarse = (t @ P['coarse/dense/dense_1/kernel'] + P['coarse/dense/dense_1/bias']).reshape(B, 55, 74, 1)
    f = pool(c('fine/first/conv2d', x, 2))
    f = torch.cat([f, coarse], dim=-1)
    f = c('fine/second/conv2d', f, same=True)
    fine = c('fine/third', f, same=True, relu=False)
    loss_c, loss_f = silog(coarse, tgt), silog(fine, tgt)

    np.testing.assert_allclose(a['coarse'], coarse.detach().numpy(), rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(a['fine'], fine.detach().numpy(), rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(a['loss_coarse'], loss_c.item(), rtol=1e-10)
    np.testing.assert_allclose(a['loss_fine'], loss_f.item(), rtol=1e-10)
    assert (a['coarse'] <= 0).any() and (a['fine'] <= 0).any()        # the tf.where masking is exercised

    coarse_vars = [k for k in params if k.startswith('coarse/')]
    fine_vars = [k for k in params if k.startswith('fine/')]
    assert set(gc) == set(coarse_vars) and set(gf) == set(fine_vars)  # the reference's var lists (src/models.py:318-338)
    for names, loss, got in ((coarse_vars, loss_c, gc), (fine_vars, loss_f, gf)):
        grads = torch.autograd.grad(loss, [P[n] for n in names], retain_graph=True)
        for n, g in zip(names, grads):
            ref = g.numpy()
            assert np.abs(ref).max() > 0, n
            np.testing.assert_allclose(got[n], ref, rtol=1e-7, atol=1e-9 * np.abs(ref).max(), err_msg=n)


def test_dcnf_unary_stack_backward():
    rng = np.random.default_rng(3)
    patches = rng.random((2, 100, 100, 3))
    params = {k: v.astype(np.float64) for k, v in OD.init_params(5).items()}
    for k in params:
        if k.endswith('/bias'):
            params[k] = 0.01 * rng.standard_normal(params[k].shape)
    a = OD.unary_forward(params, patches)
    dz = rng.standard_normal((2, 1))
    g = OD.unary_backward(params, a, dz)
    P = {k: t64(v).requires_grad_(True) for k, v in params.items()}
    t = t64(patches)
    for n, _ in OD.CONVS:                                               # src/models.py:64-73
        t = conv(t, P[OD.PREFIX + n + '/kernel'], P[OD.PREFIX + n + '/bias'], 1, False, True)
        if n in OD.POOL_AFTER:
            t = pool(t)
    t = t.reshape(2, -1)
    t = torch.relu(t @ P[OD.PREFIX + 'dense/kernel'] + P[OD.PREFIX + 'dense/bias'])                 # :80-82
    t = torch.sigmoid(t @ P[OD.PREFIX + 'dense_1/kernel'] + P[OD.PREFIX + 'dense_1/bias'])
    z = t @ P[OD.PREFIX + 'dense_2/kernel'] + P[OD.PREFIX + 'dense_2/bias']
    np.testing.assert_allclose(a['z'], z.detach().numpy(), rtol=1e-10)
    names = list(params)
    grads = torch.autograd.grad((z * t64(dz)).sum(), [P[n] for n in names])
    for n, gr in zip(names, grads):
        np.testing.assert_allclose(g[n], gr.numpy(), rtol=1e-7, atol=1e-10 * max(np.abs(gr.numpy()).max(), 1e-30), err_msg=n)
