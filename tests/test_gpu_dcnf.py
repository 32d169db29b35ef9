"""DCNF pairwise part, CRF loss and the whole dcnf train step on the GPU (through the C ABI) against oracle/dcnf.py."""
import os

import numpy as np
import pytest
import torch

from oracle import dcnf as OD

pytestmark = pytest.mark.gpu


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def pairs_dev():
    left, right = OD.pair_indices()
    return (torch.tensor(left, dtype=torch.int32, device='cuda'), torch.tensor(right, dtype=torch.int32, device='cuda'))


def test_pair_indices_match_oracle():
    from ann3depth_amd import models
    left, right = models.dcnf_pair_indices(6, 8)
    ol, orr = OD.pair_indices()
    assert left == ol.tolist() and right == orr.tolist() and len(left) == 48


def test_superpixel_statistics_and_pairwise_match_oracle():
    from ann3depth_amd import ops
    rng = np.random.default_rng(11)
    # image 0: every superpixel is the same random tile except for a few dozen changed pixels, so both similarities
    # are well inside (0, 1); image 1: smooth (colour similarity in range, histogram similarity underflows to 0, as
    # on real photographs); image 2: white noise
    img = np.empty((3, 240, 320, 3), np.float32)
    img[0] = np.tile(rng.random((40, 40, 3)).astype(np.float32), (6, 8, 1))
    for _ in range(60):
        img[0, rng.integers(240), rng.integers(320)] = rng.random(3).astype(np.float32)
    base = rng.random((6, 8, 3)).astype(np.float32)
    img[1] = np.kron(base, np.ones((40, 40, 1), np.float32)) * np.float32(0.02) + np.float32(0.4)
    img[1] += (rng.random(img[1].shape).astype(np.float32) - np.float32(0.5)) * np.float32(0.004)
    img[2] = rng.random((240, 320, 3)).astype(np.float32)
    x = torch.from_numpy(img).cuda()
    hist = ops.superpixel_hist(x, 40)
    sp = OD.superpixels(img)
    np.testing.assert_array_equal(hist.cpu().numpy(), OD.color_histogram(sp))       # integer counts: exact
    assert float(hist.sum()) == 3 * 48 * 1600
    mean = ops.superpixel_mean(x, 40)
    np.testing.assert_allclose(mean.cpu().numpy(), sp.mean(axis=2), rtol=2e-6, atol=1e-7)
    p = OD.pairwise_init(7)
    left, right = pairs_dev()
    sims, r = ops.pair_similarity(x, 40, hist, left, right, torch.from_numpy(p[OD.PAIR_PREFIX + 'kernel']).cuda(),
                                  torch.from_numpy(p[OD.PAIR_PREFIX + 'bias']).cuda(), 1.0)
    r_ref, sims_ref = OD.pairwise_forward(p, img.astype(np.float64))
    assert sims_ref[0].min() > 1e-3 and sims_ref[0].max() < 1 and sims_ref[1, :, 0].min() > 1e-3
    np.testing.assert_allclose(sims.cpu().numpy(), sims_ref, rtol=2e-4, atol=1e-30)
    np.testing.assert_allclose(r.cpu().numpy(), r_ref[..., 0], rtol=2e-4, atol=1e-7)


@pytest.mark.parametrize('regime', ['reference', 'unsaturated'])
def test_crf_loss_and_gradient_match_oracle(regime):
    """'reference': pair weights as the pairwise layer produces them — exp(-E)/Z << epsilon, the loss sits at
    -log(epsilon) = 16.118 and the gradient is tiny (what the reference computes).  'unsaturated': large pair
    weights make det(A) big enough that every term of the loss matters."""
    from ann3depth_amd import ops
    rng = np.random.default_rng(3)
    B, n = 4, 48
    y = rng.random((B, n)).astype(np.float32)
    z = (y + 0.05 * rng.standard_normal((B, n))).astype(np.float32)
    if regime == 'reference':
        r = (rng.random((B, 48)) * 0.8 - 0.1).astype(np.float32)
    else:
        r = (2.0 + 0.3 * rng.random((B, 48))).astype(np.float32)
    left, right = pairs_dev()
    mean, per, dz = ops.crf_loss(torch.from_numpy(z).cuda(), torch.from_numpy(y).cuda(), torch.from_numpy(r).cuda(),
                                 left, right, OD.EPSILON)
    depths = np.kron(y.reshape(B, 6, 8, 1).astype(np.float64), np.ones((1, 40, 40, 1)))
    m_ref, per_ref, dz_ref = OD.crf_loss(depths, z.astype(np.float64)[..., None], r.astype(np.float64)[..., None])
    np.testing.assert_allclose(per.cpu().numpy(), per_ref, rtol=2e-5)
    np.testing.assert_allclose(float(mean), m_ref, rtol=2e-5)
    if regime == 'unsaturated':
        assert np.all(per_ref < 15.5) and per_ref.min() < 12.0      # really away from -log(eps) = 16.118
    else:
        assert np.allclose(per_ref, -np.log(OD.EPSILON), atol=0.05)
    assert rel(dz.cpu().numpy(), dz_ref[..., 0]) < 2e-3
    # the 32-bit oracle (what TF would run) agrees too
    m32, per32, dz32 = OD.crf_loss(depths.astype(np.float32), z[..., None], r[..., None])
    assert rel(dz.cpu().numpy(), dz32[..., 0]) < 5e-3


def test_sgd_apply_on_tensorflows_published_vector():
    """GradientDescentOptimizerTest.testBasic (tests/golden/tf13_published_vectors.py) through a3d_sgd_apply."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import tf13_published_vectors as V
    from ann3depth_amd import ops
    c = V.SGD_TEST_BASIC
    for k in ('0', '1'):
        var = torch.tensor(c['var' + k], device='cuda')
        ops.sgd_apply(var, torch.tensor(c['grads' + k], device='cuda'), c['learning_rate'])
        np.testing.assert_allclose(var.cpu().numpy(), c['expected' + k], rtol=1e-6)


def test_sgd_apply_is_exact():
    from ann3depth_amd import ops
    rng = np.random.default_rng(0)
    v = rng.standard_normal(100003).astype(np.float32)
    g = rng.standard_normal(100003).astype(np.float32)
    tv = torch.from_numpy(v.copy()).cuda()
    ops.sgd_apply(tv, torch.from_numpy(g).cuda(), 0.1)
    np.testing.assert_array_equal(tv.cpu().numpy(), v - np.float32(0.1) * g)


def test_dcnf_train_step_matches_oracle():
    """models.dcnf's whole step at B=1: resize, unary z, pairwise r, CRF loss, unary backward, gradient descent."""
    from ann3depth_amd import models
    from oracle import tf13_ops as T
    rng = np.random.default_rng(21)
    img = (rng.integers(0, 256, (1, 480, 640, 3)) / 255).astype(np.float32)
    dep = rng.random((1, 55, 74, 1)).astype(np.float32)
    params = OD.init_params(3000)
    params.update(OD.pairwise_init(3001))
    rep = models.DCNFReplica(1, params=params)
    before = {n: rep.unary.group.view(rep.unary.group.var, n).clone() for n in rep.unary.shapes}
    out = rep.step(torch.from_numpy(img).cuda(), torch.from_numpy(dep).cuda())
    torch.cuda.synchronize()
    assert rep.global_step == 1
    img240 = T.resize_bilinear_tf1(img, 240, 320)
    dep240 = T.resize_bilinear_tf1(dep, 240, 320)
    np.testing.assert_array_equal(rep.depths240.cpu().numpy(), dep240)
    r_ref, sims_ref = OD.pairwise_forward(params, img240)
    np.testing.assert_allclose(rep.r.cpu().numpy(), r_ref[..., 0], rtol=1e-3, atol=1e-7)
    z_gpu = rep.unary.z.cpu().numpy().reshape(1, 48, 1)
    m_ref, per_ref, dz_ref = OD.crf_loss(dep240.astype(np.float64), z_gpu.astype(np.float64),
                                         rep.r.cpu().numpy().astype(np.float64)[..., None])
    np.testing.assert_allclose(float(out['mean_loss']), m_ref, rtol=2e-5)
    assert rel(rep.dz.cpu().numpy(), dz_ref[..., 0]) < 2e-3
    # backward chain on the GPU activations, then var -= 0.1 * grad
    a_gpu = rep.unary.activations()
    a_gpu['flat'] = a_gpu['conv2d_4/pool'].reshape(48, -1)
    g = OD.unary_backward(params, a_gpu, rep.dz.cpu().numpy().reshape(48, 1))
    for n, gref in g.items():
        ggpu = rep.unary.group.view(rep.unary.group.grad, n)
        if np.linalg.norm(gref) > 0:
            assert rel(ggpu.cpu().numpy(), gref) < 1e-4, n
        np.testing.assert_array_equal(rep.unary.group.view(rep.unary.group.var, n).cpu().numpy(),
                                      (before[n] - np.float32(0.1) * ggpu).cpu().numpy())
    # the pairwise layer receives no gradient (A is a constant for the optimizer)
    np.testing.assert_array_equal(rep.pair_var('kernel').cpu().numpy(), params[OD.PAIR_PREFIX + 'kernel'])
    sd = rep.state_dict()
    assert 'pairwise/pairwise_layers/dense/kernel' in sd and 'unary/unary_layers/conv2d/kernel' in sd
    rep2 = models.DCNFReplica(1, seed=1)
    rep2.load_state_dict({k: v.clone() for k, v in sd.items()})
    assert rep2.global_step == 1
    assert torch.equal(rep2.unary.group.var, rep.unary.group.var)


def test_make_train_dcnf(tmp_path):
    """`--model dcnf` through the driver: summaries carry loss/mean_loss, a checkpoint is written."""
    import json
    from ann3depth_amd import ann3depth, tfrecord
    rng = np.random.default_rng(1)
    root = str(tmp_path / 'data')
    os.makedirs(os.path.join(root, 'nyu'))
    with tfrecord.TFRecordWriter(os.path.join(root, 'nyu', 'train.tfrecords')) as w:
        for _ in range(8):
            w.write_example(rng.random((48, 64, 3)).astype(np.float32), rng.random((6, 8, 1)).astype(np.float32))
    ck = str(tmp_path / 'ckpt')
    rc = ann3depth.main(['nyu', '--model', 'dcnf', '--steps', '3', '--batchsize', '2', '--datadir', root,
                         '--ckptdir', ck, '--sumfreq', '1', '--ckptfreq', '0'])
    assert rc == 0
    d = os.path.join(ck, 'dcnf')
    recs = [json.loads(l) for l in open(os.path.join(d, 'summaries.jsonl'))]
    assert [r['global_step'] for r in recs] == [1, 2, 3]
    assert all(abs(r['loss/mean_loss'] + np.log(1e-7)) < 1e-2 for r in recs)
    assert os.path.exists(os.path.join(d, 'model.ckpt-3.pt'))


def test_dcnf_unary_at_baseline_size_matches_oracle():
    """BASELINE config 4 itself: batch 16 -> 768 patches (the planner picks other tiles / split-K / stream-K shares at
    M = 768 * 8100 than at the 48 patches of the B = 1 tests).  Patches are independent (shared weights, tf.map_fn over
    the batch, src/models.py:85-89), so the numpy oracle runs on 2 of the 16 images — the first and the last — and
    must reproduce their slice of every activation; the backward is checked through the slice of every activation
    gradient and, for the filter gradients (sums over all 768 patches), through linearity: the gradient of the batch is
    the sum of the gradients of the 16 images taken one at a time through the B = 1 path that
    test_dcnf_unary_matches_oracle pins to the oracle."""
    from ann3depth_amd import models
    B = 16
    rng = np.random.default_rng(16)
    img = (rng.integers(0, 256, (B, 480, 640, 3)) / 255).astype(np.float32)
    params = OD.init_params(3000)
    net = models.DCNFUnary(B, params=params)
    assert net.P == 768
    timg = torch.from_numpy(img).cuda()
    z = net.forward(timg)
    dz = rng.standard_normal((768, 1)).astype(np.float32)
    net.backward(torch.from_numpy(dz).cuda())
    torch.cuda.synchronize()
    zg = z.cpu().numpy().reshape(768, 1)
    for i in (0, B - 1):
        sl = slice(48 * i, 48 * (i + 1))
        patches = OD.patches(img[i:i + 1])
        np.testing.assert_array_equal(net.act['x'][sl].cpu().numpy(), patches)
        a = OD.unary_forward(params, patches)
        assert rel(zg[sl], a['z']) < 1e-3
        for n in ('conv2d/pool', 'conv2d_1/pool', 'conv2d_2', 'conv2d_3', 'conv2d_4/pool', 'dense', 'dense_1'):
            assert rel(net.act[n][sl].cpu().numpy(), a[n]) < 1e-4, (i, n)
    # backward: one image at a time through the B = 1 replica, same weights, same dz slice — and the batch's own
    # activations and pool positions (a max pool window whose two largest values agree to the last bit may resolve
    # differently under another tile plan's summation order; that moves one gradient element, rel-L2 ~2e-3 downstream,
    # and says nothing about the kernels under test)
    one = models.DCNFUnary(1, params=params)
    acc = {n: torch.zeros_like(net.group.view(net.group.grad, n), dtype=torch.float64) for n in net.shapes}
    for i in range(B):
        sl = slice(48 * i, 48 * (i + 1))
        for k in one.act:
            one.act[k].copy_(net.act[k][sl])
        for k in one.argmax:
            one.argmax[k].copy_(net.argmax[k][sl])
        one.backward(torch.from_numpy(dz[sl]).cuda())
        if i in (0, B - 1):                      # activation gradients are per patch: slices must agree
            # (the first conv's pre-pool gradient is not materialised when its filter gradient is taken straight from the pooled
            # map's gradient — a3d_conv2d_bwd_filter_pooled; the kernel / bias gradients below cover it)
            keys = ['dense_1', 'dense', 'flat', 'conv2d_4', 'in:conv2d_4', 'in:conv2d_2', 'conv2d_1', 'in:conv2d_1']
            for key in keys + ([] if net.few_pooled else ['conv2d']):
                assert rel(net.dact[key][sl].cpu().numpy(), one.dact[key].cpu().numpy()) < 1e-4, (i, key)
        for n in net.shapes:
            acc[n] += one.group.view(one.group.grad, n).double()
    for n in net.shapes:
        g = net.group.view(net.group.grad, n).cpu().numpy()
        ref = acc[n].cpu().numpy()
        if np.linalg.norm(ref) > 0:
            assert rel(g, ref) < 1e-4, n
