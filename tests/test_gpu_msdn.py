"""End-to-end parity of the MSDN train step (HIP path, through the C ABI) against the numpy oracle."""
import numpy as np
import pytest
import torch

from oracle import msdn as O

pytestmark = pytest.mark.gpu

DEPTH_TOL = 1e-3      # north-star: depth maps within 1e-3 rel-L2 of the CPU reference (fp32)
GRAD_TOL = 1e-4       # backward chain given the same activations: fp32 accumulation order only
# d loss / d o_i = (...) / (o_i + 1e-8): outputs that happen to lie within ~1e-4 of zero dominate the gradient norm
# and turn a 1e-7 forward difference into a 1e-3 gradient difference.  That is a property of the reference's
# loss, not of either implementation, so the end-to-end gradient check is loose and the tight check feeds the
# oracle's backward with the activations the GPU produced.
GRAD_TOL_END_TO_END = 3e-2
LOSS_TOL = 2e-3


def gpu_activations(net):
    names = {'images': 'x', 'depths': 't', 'c0': 'c0', 'p0': 'p0', 'c1': 'c1', 'p1': 'p1', 'c2': 'c2', 'c3': 'c3',
             'c4': 'c4', 'drop': 'drop', 'coarse': 'coarse', 'f1': 'f1', 'cat': 'cat', 'f2': 'f2', 'fine': 'fine'}
    a = {k: getattr(net, v).float().cpu().numpy() for k, v in names.items() if getattr(net, v) is not None}
    for k in ('c0', 'c1', 'f1'):         # inside step() these are never written: what the backward sees of them
        a[k] = net.prepool_equivalent(k).float().cpu().numpy()
    a['flat'] = a['c4'].reshape(a['c4'].shape[0], -1)
    a['d0'] = a['drop']          # only its sign is used (ReluGrad); drop > 0 <=> d0 > 0 on kept units
    return a


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def synth(B, seed, h=480, w=640):
    rng = np.random.default_rng(seed)
    img = (rng.integers(0, 256, (B, h, w, 3)) / 255).astype(np.float32)
    dep = (rng.integers(0, 256, (B, h, w, 1)) / 255).astype(np.float32)
    keep = rng.random((B, 4096)) >= 0.5
    return img, dep, keep


@pytest.fixture(scope='module')
def models():
    from ann3depth_amd import models
    return models


@pytest.mark.parametrize('phase,global_step', [(1, 0), (2, 2000000 // 2), (3, 3500000 // 2)])
def test_msdn_step_matches_oracle(models, phase, global_step):
    B = 2
    img, dep, keep = synth(B, 1000)
    params = O.init_params(3000)
    net = models.MSDNReplica(B, params=params, global_step=global_step)
    out = net.step(torch.from_numpy(img).cuda(), torch.from_numpy(dep).cuda(), torch.from_numpy(keep).cuda())
    torch.cuda.synchronize()
    tr = O.Trainer({k: v.copy() for k, v in params.items()}, B, global_step=global_step)
    a, g, ph = tr.step(img, dep, keep)
    assert ph == phase == out['phase']
    assert net.global_step == tr.global_step == global_step + 1
    np.testing.assert_array_equal(net.x.cpu().numpy(), a['images'])          # resize is bit-exact
    np.testing.assert_array_equal(net.t.cpu().numpy(), a['depths'])
    assert rel(net.coarse.cpu().numpy(), a['coarse']) < DEPTH_TOL
    assert rel(net.fine.cpu().numpy(), a['fine']) < DEPTH_TOL
    # conv + pool run fused inside step(): c0 / c1 / f1 are never written, the trained network records the window
    # position of each maximum instead
    for name in ('p0', 'p1', 'c4', 'drop', 'cat', 'f2'):
        assert rel(getattr(net, name).cpu().numpy(), a[name]) < 1e-4, name
    for name, pooled in {1: (('c0', 'p0'), ('c1', 'p1')), 2: (('f1', 'cat'),), 3: ()}[phase]:
        eq = net.prepool_equivalent(name).cpu().numpy()                  # zeros + each window's maximum in place
        ref = a[name]
        c = eq.shape[-1]
        hit = eq > 0
        assert hit.sum() > 0.2 * a[pooled][..., :c].size                  # most windows have a positive maximum
        np.testing.assert_allclose(eq[hit], ref[hit], rtol=1e-3, atol=1e-5)   # ... found where the oracle has it
    # log(o + 1e-8) of outputs that land within ~1e-7 of zero moves by O(1) under last-bit changes of o
    assert abs(out['coarse_loss'].item() - a['loss_coarse']) < LOSS_TOL * abs(a['loss_coarse'])
    assert abs(out['fine_loss'].item() - a['loss_fine']) < LOSS_TOL * abs(a['loss_fine'])
    for n, gref in g.items():
        assert rel(net.grad(n).cpu().numpy(), gref) < GRAD_TOL_END_TO_END, n
    if phase in (1, 2):
        a_gpu = gpu_activations(net)
        a_gpu['keep_mask'] = keep
        g_chain = (O.backward_coarse if phase == 1 else O.backward_fine)(params, a_gpu)
        assert set(g_chain) == set(g)
        for n, gref in g_chain.items():
            assert rel(net.grad(n).cpu().numpy(), gref) < GRAD_TOL, n
    # optimizer state: the reference's beta2 = 1 leaves every weight untouched, m follows (1-beta1)*g
    for n in params:
        np.testing.assert_array_equal(net.var(n).cpu().numpy(), params[n])
        opt = tr.opt[net.group_of[n]]
        if n in opt.m:
            # m = 0 + (g - 0) * (1 - beta1) in separate fp32 operations (ApplyAdam, training_ops.cc): bit for bit from
            # the gradient the chain test above pinned at 1e-4; against the oracle's own m only as loosely as its g
            g32 = net.grad(n).cpu().numpy()
            np.testing.assert_array_equal(net.slot(n, 'm').cpu().numpy(), g32 * (np.float32(1) - np.float32(0.9)))
            assert rel(net.slot(n, 'm').cpu().numpy(), opt.m[n]) < GRAD_TOL_END_TO_END, n
            assert (net.slot(n, 'v').cpu().numpy() == 0).all()


def test_msdn_learning_mode_multi_step(models):
    """beta2 = 0.999 (flagged non-reference mode): weights move, and keep tracking the oracle over several steps.
    Adam's m/sqrt(v) is ~ +-lr whatever |g| is, so an element whose tiny gradient changes sign under fp32 reordering
    moves the other way, and from the second step on that feeds back into every gradient: the first step must match
    element by element, later steps are held to the direction of the accumulated update."""
    B = 2
    params = O.init_params(3000)
    net = models.MSDNReplica(B, params=params, beta2=0.999)
    tr = O.Trainer({k: v.copy() for k, v in params.items()}, B, beta2=0.999)
    for step in range(3):
        img, dep, keep = synth(B, 1000 + step, 96, 128)
        net.step(torch.from_numpy(img).cuda(), torch.from_numpy(dep).cuda(), torch.from_numpy(keep).cuda())
        tr.step(img, dep, keep)
        if step not in (0, 2):
            continue
        moved = 0
        for n in params:
            w, ref, w0 = net.var(n).cpu().numpy(), tr.p[n], params[n]
            if not n.startswith('coarse'):
                np.testing.assert_array_equal(w, w0)                        # fine/* frozen in the coarse phase
                continue
            moved += int(np.abs(w - w0).max() > 0)
            close = (np.abs(w - ref) <= 0.05 * np.abs(ref - w0) + 1e-7).mean()
            if n.endswith('/kernel'):
                corr = float(np.corrcoef((w - w0).ravel().astype(np.float64), (ref - w0).ravel().astype(np.float64))[0, 1])
            else:
                corr = 1.0                                                   # biases: too few elements for a correlation
            if step == 0:
                assert close > 0.99 and corr > 0.995, (n, close, corr)
            else:
                assert close > 0.6 and corr > 0.75, (n, close, corr)
        assert moved >= 12


def test_msdn_full_batch_properties(models):
    """B=32 (BASELINE config 2), size-independent properties:
    per-sample independence (sample i of the batch == the same sample run in a batch of 2) and determinism."""
    B = 32
    img, dep, keep = synth(B, 1234, 120, 160)
    params = O.init_params(3000)
    net = models.MSDNReplica(B, params=params)
    ti, td, tk = torch.from_numpy(img).cuda(), torch.from_numpy(dep).cuda(), torch.from_numpy(keep).cuda()
    net.step(ti, td, tk)
    coarse = net.coarse.clone(); fine = net.fine.clone()
    g0 = net.grad('coarse/conv/conv2d_1/kernel').clone()
    net2 = models.MSDNReplica(B, params=params)
    net2.step(ti, td, tk)
    assert torch.equal(net2.coarse, coarse) and torch.equal(net2.fine, fine)
    assert torch.equal(net2.grad('coarse/conv/conv2d_1/kernel'), g0)          # split-K slabs, no atomics: reproducible
    small = models.MSDNReplica(2, params=params)
    small.step(ti[5:7].contiguous(), td[5:7].contiguous(), tk[5:7].contiguous())
    torch.cuda.synchronize()
    assert rel(small.coarse.cpu().numpy(), coarse[5:7].cpu().numpy()) < 1e-5
    assert rel(small.fine.cpu().numpy(), fine[5:7].cpu().numpy()) < 1e-5


@pytest.mark.parametrize('phase,global_step', [(1, 0), (2, 2000000 // 32)])
def test_msdn_full_batch_matches_oracle(models, phase, global_step):
    """BASELINE config 2 itself (B = 32, 640x480 stored): the tile configs, LDS-DMA kernels and split-K factors the
    planner picks at this size against the oracle at the same size — depth maps within the north-star tolerance, and
    the whole backward chain fed with the GPU's activations.  ~10 s of numpy on the GPU box's host cores."""
    B = 32
    img, dep, keep = synth(B, 4321)
    params = O.init_params(3000)
    net = models.MSDNReplica(B, params=params, global_step=global_step)
    out = net.step(torch.from_numpy(img).cuda(), torch.from_numpy(dep).cuda(), torch.from_numpy(keep).cuda())
    torch.cuda.synchronize()
    assert out['phase'] == phase
    a = O.forward(params, img, dep, keep)
    np.testing.assert_array_equal(net.x.cpu().numpy(), a['images'])
    assert rel(net.coarse.cpu().numpy(), a['coarse']) < DEPTH_TOL
    assert rel(net.fine.cpu().numpy(), a['fine']) < DEPTH_TOL
    for name in ('p0', 'p1', 'c2', 'c3', 'c4', 'drop', 'cat', 'f2'):     # c0 / c1 / f1 are fused away inside step()
        assert rel(getattr(net, name).cpu().numpy(), a[name]) < 1e-4, name
    assert abs(out['coarse_loss'].item() - a['loss_coarse']) < LOSS_TOL * abs(a['loss_coarse'])
    assert abs(out['fine_loss'].item() - a['loss_fine']) < LOSS_TOL * abs(a['loss_fine'])
    a_gpu = gpu_activations(net)
    a_gpu['keep_mask'] = keep
    g_chain = (O.backward_coarse if phase == 1 else O.backward_fine)(params, a_gpu)
    for n, gref in g_chain.items():
        assert rel(net.grad(n).cpu().numpy(), gref) < GRAD_TOL, n


def shifted_params(seed=3000, shift=1.0):
    """The seed weights with both output biases at +shift: the depth maps then lie in [0.7, 1.3] instead of straddling
    zero, d loss / d o_i = (...) / (o_i + 1e-8) is well conditioned, and a gradient can be compared END TO END at the
    tolerance of the fp32 accumulation order instead of GRAD_TOL_END_TO_END."""
    params = O.init_params(seed)
    params['coarse/dense/dense_1/bias'] = np.full_like(params['coarse/dense/dense_1/bias'], shift)
    params['fine/third/bias'] = np.full_like(params['fine/third/bias'], shift)
    return params


def _window_argmax(x):
    """[B,H,W,C] -> (first-maximum position 0..3 of every 2x2 / stride-2 window, its value): MaxPoolGrad's routing."""
    B, H, W, C = x.shape
    ho, wo = H // 2, W // 2
    v = x[:, :2 * ho, :2 * wo, :].reshape(B, ho, 2, wo, 2, C).transpose(0, 1, 3, 5, 2, 4).reshape(B, ho, wo, C, 4)
    return v.argmax(axis=-1), v.max(axis=-1)


def decisions_taken_differently(net, a, keep, phase):
    """The discontinuous decisions of the network under training — ReluGrad's `y > 0`, MaxPoolGrad's argmax — that the
    GPU's forward took differently from the oracle's.  Returns (count, elements that carry a decision, a copy of the
    oracle's activations in which exactly those elements / windows are replaced by what the GPU saw)."""
    fix = dict(a)
    flipped = total = 0
    relu_layers = ('c2', 'c3', 'c4') if phase == 1 else ('f2',)
    for k in relu_layers:
        got = getattr(net, k).float().cpu().numpy()
        diff = (got > 0) != (a[k] > 0)
        flipped += int(diff.sum()); total += diff.size
        if diff.any():
            fix[k] = np.where(diff, got, a[k])
    if phase == 1:
        got = net.drop.cpu().numpy()                      # drop > 0  <=>  kept and dense_0's ReLU active
        diff = (got > 0) != ((a['d0'] > 0) & keep)
        flipped += int(diff.sum()); total += diff.size
        if diff.any():
            fix['d0'] = np.where(diff, np.where(got > 0, got, 0), a['d0'])
            fix['drop'] = np.where(diff, got, a['drop'])
        fix['flat'] = fix['c4'].reshape(fix['c4'].shape[0], -1)
    pools = (('c0', 'a0', 'p0'), ('c1', 'a1', 'p1')) if phase == 1 else (('f1', 'af1', 'cat'),)
    for full, argname, pooled in pools:
        arg_ref, max_ref = _window_argmax(a[full])
        arg_gpu = getattr(net, argname).cpu().numpy().astype(np.int64)
        c = arg_gpu.shape[-1]
        max_gpu = getattr(net, pooled).float().cpu().numpy()[..., :c]
        # a window's decision: is its maximum positive (ReluGrad), and if so, where is it (MaxPoolGrad)
        diff = ((max_gpu > 0) != (max_ref > 0)) | ((max_ref > 0) & (arg_gpu != arg_ref))
        flipped += int(diff.sum()); total += diff.size
        if diff.any():
            eq = net.prepool_equivalent(full).cpu().numpy()          # zeros + the window's maximum where the GPU found it
            B, H, W, C = a[full].shape
            ho, wo = H // 2, W // 2
            m6 = np.broadcast_to(diff[:, :, None, :, None, :], (B, ho, 2, wo, 2, C)).reshape(B, 2 * ho, 2 * wo, C)
            out = a[full].copy()
            out[:, :2 * ho, :2 * wo, :] = np.where(m6, eq[:, :2 * ho, :2 * wo, :], a[full][:, :2 * ho, :2 * wo, :])
            fix[full] = out
    return flipped, total, fix


@pytest.mark.parametrize('phase,global_step', [(1, 0), (2, 2000000 // 32)])
def test_msdn_step_the_bench_times_matches_oracle_end_to_end(models, phase, global_step):
    """The replica bench.py and `make train` build on one GPU — keep_dense_grads=False: dense dW goes from the matrix
    cores straight into ApplyAdam's m slot — at BASELINE config 2 (B = 32, 480x640 stored), against the oracle END TO
    END (its own forward, its own backward, its own ApplyAdam): depth maps, losses, every gradient that exists and every
    m slot at 1e-4; weights untouched (beta2 = 1).  Outputs are shifted away from zero (shifted_params) so that the loss
    gradient does not amplify last-bit differences of the forward.  What is left end to end are the network's own
    discontinuities: a ReLU or max-pool decision taken the other way by a last-bit difference of the forward moves a
    gradient by more than accumulation order does — 3.6e-4 on conv2d_3's kernel at this size — hence 1e-3 on the raw
    comparison.  That explanation is SHOWN, not assumed: the decisions the two forwards took
    differently are counted (a handful among ~10^7), and against the oracle's own end-to-end backward in which exactly
    those elements / pool windows see what the GPU saw — every other activation, the loss gradient and all arithmetic
    stay the oracle's — every gradient and m slot holds GRAD_TOL = 1e-4."""
    E2E = 1e-3
    B = 32
    img, dep, keep = synth(B, 4321)
    params = shifted_params()
    net = models.MSDNReplica(B, params=params, global_step=global_step, keep_dense_grads=False)
    assert net._fused_dense_adam()
    out = net.step(torch.from_numpy(img).cuda(), torch.from_numpy(dep).cuda(), torch.from_numpy(keep).cuda())
    torch.cuda.synchronize()
    tr = O.Trainer({k: v.copy() for k, v in params.items()}, B, global_step=global_step)
    a, g, ph = tr.step(img, dep, keep)
    assert ph == phase == out['phase']
    assert a['coarse'].min() > 0.5 and a['fine'].min() > 0.5                 # the point of the shift
    assert rel(net.coarse.cpu().numpy(), a['coarse']) < 1e-5
    assert rel(net.fine.cpu().numpy(), a['fine']) < 1e-5
    assert abs(out['coarse_loss'].item() - a['loss_coarse']) < 1e-5 * abs(a['loss_coarse'])
    assert abs(out['fine_loss'].item() - a['loss_fine']) < 1e-5 * abs(a['loss_fine'])
    omb1 = np.float32(1) - np.float32(0.9)
    fused = ('coarse/dense/dense_0/kernel', 'coarse/dense/dense_0/bias', 'coarse/dense/dense_1/kernel',
             'coarse/dense/dense_1/bias')
    for n, gref in g.items():
        if n not in fused:                                                   # the fused layers never write their gradient
            assert rel(net.grad(n).cpu().numpy(), gref) < E2E, n
        opt = tr.opt[net.group_of[n]]
        assert rel(net.slot(n, 'm').cpu().numpy(), opt.m[n]) < E2E, n       # m = (1 - beta1) g, the oracle's ApplyAdam
        np.testing.assert_array_equal(opt.m[n], gref * omb1)
        assert (net.slot(n, 'v').cpu().numpy() == 0).all()
        np.testing.assert_array_equal(net.var(n).cpu().numpy(), params[n])
    if phase == 1:
        assert set(fused) <= set(g)
    flipped, total, a_fix = decisions_taken_differently(net, a, keep, phase)
    print(f'phase {phase}: {flipped} of {total} ReLU / max-pool decisions differ between the GPU forward and the oracle')
    assert flipped <= 1e-5 * total                                           # 1e-7-sized forward differences flip very few
    a_fix['keep_mask'] = keep
    g_fix = (O.backward_coarse if phase == 1 else O.backward_fine)(params, a_fix)
    worst_raw = max(rel(net.slot(n, 'm').cpu().numpy(), g[n] * omb1) for n in g)
    for n, gref in g_fix.items():
        assert rel(net.slot(n, 'm').cpu().numpy(), gref * omb1) < GRAD_TOL, (n, flipped)
        if n not in fused:
            assert rel(net.grad(n).cpu().numpy(), gref) < GRAD_TOL, (n, flipped)
    if flipped == 0:
        assert worst_raw < GRAD_TOL                                          # nothing flipped: nothing to explain


def test_msdn_learning_mode_tracks_the_oracles_adam_slots(models):
    """beta2 = 0.999 (flagged non-reference mode), three steps on well-conditioned outputs (shifted_params).
    (1) The optimizer arithmetic, isolated from the forward: the oracle's ApplyAdam (oracle.tf13_ops.AdamTF1) is fed the
        GPU's OWN gradient of each step; m, v and the weights must follow it at 1e-6 over all three steps — a wrong beta,
        epsilon, learning rate or bias-correction power cannot hide (the old test accepted corr > 0.75 on the weights).
    (2) The gradients themselves against the oracle's end-to-end step: 1e-5 for the dense layers at the first step; the
        conv layers sit behind ReLU / max-pool decisions that a last-bit difference of the forward takes the other way
        (3.9e-3 on conv2d_0's kernel at this size), and from the second step on alpha * m / sqrt(v) turns the sign of a
        near-zero gradient element into a +-lr move of its weight, so those are held to 2e-2."""
    from oracle.tf13_ops import AdamTF1
    B = 2
    params = shifted_params()
    net = models.MSDNReplica(B, params=params, beta2=0.999)
    tr = O.Trainer({k: v.copy() for k, v in params.items()}, B, beta2=0.999)
    lrs = {'CoarseConv': 0.001, 'CoarseDense': 0.1}
    twin = {g: AdamTF1(lr, 0.9, 0.999) for g, lr in lrs.items()}          # fed with the GPU's gradients
    twin_p = {n: v.copy() for n, v in params.items()}
    for step in range(3):
        img, dep, keep = synth(B, 1000 + step, 96, 128)
        net.step(torch.from_numpy(img).cuda(), torch.from_numpy(dep).cuda(), torch.from_numpy(keep).cuda())
        tr.step(img, dep, keep)
        for gname, opt in twin.items():
            names = [n for n in params if net.group_of[n] == gname]
            opt.apply(twin_p, {n: net.grad(n).cpu().numpy() for n in names})
            for n in names:
                assert rel(net.slot(n, 'm').cpu().numpy(), opt.m[n]) < 1e-6, (step, n, 'm')
                assert rel(net.slot(n, 'v').cpu().numpy(), opt.v[n]) < 1e-6, (step, n, 'v')
                upd, ref = net.var(n).cpu().numpy() - params[n], twin_p[n] - params[n]
                assert rel(upd, ref) < 1e-5, (step, n, 'var')
                assert np.abs(ref).max() > 0
        for n in params:
            if not n.startswith('coarse'):
                np.testing.assert_array_equal(net.var(n).cpu().numpy(), params[n])          # fine/* frozen in the coarse phase
                continue
            opt = tr.opt[net.group_of[n]]
            tol = 1e-5 if (step == 0 and '/dense/' in n) else 2e-2
            assert rel(net.slot(n, 'm').cpu().numpy(), opt.m[n]) < tol, (step, n, 'oracle m')


@pytest.mark.parametrize('prec,tol', [('bf16x3', 1e-4), ('bf16', 5e-2)])
def test_msdn_bf16_modes_keep_the_depth_tolerance(models, prec, tol):
    """The bf16-MFMA modes end to end: bf16x3 stays far inside the north-star 1e-3 depth tolerance; plain bf16
    (BASELINE config 5's arithmetic) is reported with its own, looser tolerance."""
    B = 2
    img, dep, keep = synth(B, 1000)
    params = O.init_params(3000)
    net = models.MSDNReplica(B, params=params, precision=prec)
    net.step(torch.from_numpy(img).cuda(), torch.from_numpy(dep).cuda(), torch.from_numpy(keep).cuda())
    torch.cuda.synchronize()
    a = O.forward(params, img, dep, keep)
    assert rel(net.coarse.cpu().numpy(), a['coarse']) < tol
    assert rel(net.fine.cpu().numpy(), a['fine']) < tol
    if prec == 'bf16x3':
        assert tol <= DEPTH_TOL
        a_gpu = gpu_activations(net)
        a_gpu['keep_mask'] = keep
        g = O.backward_coarse(params, a_gpu)
        for n, gref in g.items():
            assert rel(net.grad(n).cpu().numpy(), gref) < 2e-4, n


def test_msdn_without_dropout_when_the_plugin_is_called_with_train_false(models):
    """models.msdn(images, depths, train=False) (src/models.py:277 -> :230): tf.layers.dropout is the identity, the
    optimizers still run.  keep_mask None is that mode of the replica."""
    B = 2
    img, dep, _ = synth(B, 1003)
    params = O.init_params(3000)
    net = models.MSDNReplica(B, params=params)
    out = net.step(torch.from_numpy(img).cuda(), torch.from_numpy(dep).cuda(), None)
    torch.cuda.synchronize()
    a = O.forward(params, img, dep, None)
    assert rel(net.drop.cpu().numpy(), a['d0']) < 1e-4 and rel(net.coarse.cpu().numpy(), a['coarse']) < DEPTH_TOL
    assert abs(out['coarse_loss'].item() - a['loss_coarse']) < LOSS_TOL * abs(a['loss_coarse'])
    a_gpu = gpu_activations(net)
    a_gpu['keep_mask'] = None
    for n, gref in O.backward_coarse(params, a_gpu).items():
        assert rel(net.grad(n).cpu().numpy(), gref) < GRAD_TOL, n


@pytest.mark.parametrize('B', [1, 5])
def test_msdn_odd_batch_sizes(models, B):
    """Batch sizes that leave ragged M tiles in every layer (B = 1: M = 4070, 999, 234, 48 ...)."""
    img, dep, keep = synth(B, 77, 96, 128)
    params = O.init_params(3000)
    net = models.MSDNReplica(B, params=params)
    net.step(torch.from_numpy(img).cuda(), torch.from_numpy(dep).cuda(), torch.from_numpy(keep).cuda())
    torch.cuda.synchronize()
    a = O.forward(params, img, dep, keep)
    assert rel(net.coarse.cpu().numpy(), a['coarse']) < DEPTH_TOL
    assert rel(net.fine.cpu().numpy(), a['fine']) < DEPTH_TOL
    a_gpu = gpu_activations(net)
    a_gpu['keep_mask'] = keep
    for n, gref in O.backward_coarse(params, a_gpu).items():
        assert rel(net.grad(n).cpu().numpy(), gref) < GRAD_TOL, n


@pytest.mark.parametrize('global_step', [0, 2000000 // 2])
def test_two_stream_schedule_is_bit_identical(models, monkeypatch, global_step):
    """A3D_OVERLAP=1 runs the fine forward and every backward-filter GEMM on a second stream: same kernels, same
    operands, so every output, gradient and slot must equal the one-stream schedule bit for bit (3 steps, beta2 < 1 so
    that the weights move and a race would propagate)."""
    B = 2
    params = O.init_params(3000)
    state = []
    for overlap in ('0', '1'):
        monkeypatch.setenv('A3D_OVERLAP', overlap)
        net = models.MSDNReplica(B, params=params, global_step=global_step, beta2=0.999)
        assert (net.side is not None) == (overlap == '1')
        for s in range(3):
            img, dep, keep = synth(B, 1000 + s)
            net.step(torch.from_numpy(img).cuda(), torch.from_numpy(dep).cuda(), torch.from_numpy(keep).cuda())
        torch.cuda.synchronize()
        state.append({k: v.clone() for k, v in net.state_dict().items()} | {'fine': net.fine.clone(),
                                                                             'coarse': net.coarse.clone()})
    for k in state[0]:
        assert torch.equal(state[0][k], state[1][k]), k


def test_msdn_fused_dense_adam_equals_kept_gradients(models):
    """models.msdn's replica (keep_dense_grads=False: the dense kernels' gradient goes from the matrix cores straight into
    ApplyAdam's m slot) against the replica that materialises every gradient: two coarse-phase steps, every variable and
    both Adam slots of every group bit for bit, same losses, same beta powers (src/models.py:319-330)."""
    B = 2
    img, dep, keep = synth(B, 1234)
    params = O.init_params(3000)
    args = [torch.from_numpy(a).cuda() for a in (img, dep, keep)]
    nets = [models.MSDNReplica(B, params=params, keep_dense_grads=k) for k in (True, False)]
    assert nets[0].keep_dense_grads and not nets[1].keep_dense_grads
    for _ in range(2):
        outs = [n.step(*args) for n in nets]
        torch.cuda.synchronize()
        assert float(outs[0]['coarse_loss']) == float(outs[1]['coarse_loss'])
    for gname in nets[0].groups:
        ga, gb = nets[0].groups[gname], nets[1].groups[gname]
        assert ga.beta1_power == gb.beta1_power and ga.beta2_power == gb.beta2_power
        for buf in ('var', 'm', 'v'):
            np.testing.assert_array_equal(getattr(ga, buf).cpu().numpy(), getattr(gb, buf).cpu().numpy(), err_msg=f'{gname}.{buf}')
    assert float(nets[1].groups['CoarseDense'].m.abs().sum()) > 0          # the fused path did update m
    # a non-reference beta2 keeps the two-pass path whatever the flag says
    assert not models.MSDNReplica(B, params=params, beta2=0.999, keep_dense_grads=False)._fused_dense_adam()


import bf16s_tol      # noqa: E402  (tests/bf16s_tol.py: per-tensor-class tolerances of precision 'bf16s', 1.5 x the observed error)


def test_msdn_bf16_storage_at_config5_batch(models):
    """BASELINE config 5's per-GPU workload: batch 64, bf16 arithmetic with bf16 activations / weight copies in HBM and
    fp32 masters, accumulators, gradients and Adam slots (precision 'bf16s').  Stated tolerance: depth maps within 5e-2
    rel-L2 of the fp32 oracle (the plain-bf16 mode's own bound, tests above; the oracle runs on 4 of the 64 images —
    samples are independent), within 2e-2 of the fp32-storage bf16 mode (what storing activations as bf16 adds), and every
    filter / bias gradient within 6e-2 of the oracle's fp32 backward of the stored activations, both trained phases.
    Round 5: the blanket 5e-2 / 6e-2 are replaced by per-tensor-class bounds of 1.5 x what the error
    actually is (tests/bf16s_tol.py; the values are printed: run with -s)."""
    B = 64
    img, dep, keep = synth(B, 6464)
    params = O.init_params(3000)
    args = [torch.from_numpy(a).cuda() for a in (img, dep, keep)]
    net = models.MSDNReplica(B, params=params, precision='bf16s')
    out = net.step(*args)
    torch.cuda.synchronize()
    assert net.c2.dtype == torch.bfloat16 and net.dc3.dtype == torch.bfloat16 and net.cat.dtype == torch.bfloat16
    assert net.wcopy['coarse/dense/dense_0'].dtype == torch.bfloat16 and net.groups['CoarseDense'].var.dtype == torch.float32
    sl = [0, 1, 62, 63]
    a = O.forward(params, img[sl], dep[sl], keep[sl])
    seen = {}
    for k in ('coarse', 'fine'):
        seen['depth/' + k] = e = rel(getattr(net, k)[sl].cpu().numpy(), a[k])
        assert e < bf16s_tol.DEPTH[k], (k, e)
    assert np.isfinite(float(out['coarse_loss'])) and np.isfinite(float(out['fine_loss']))
    ref = models.MSDNReplica(B, params=params, precision='bf16')
    ref.step(*args)
    torch.cuda.synchronize()
    assert rel(net.coarse.cpu().numpy(), ref.coarse.cpu().numpy()) < 2e-2
    assert rel(net.fine.cpu().numpy(), ref.fine.cpu().numpy()) < 2e-2
    # backward chain: the oracle's fp32 backward fed with the activations this replica stored (an end-to-end gradient
    # comparison would measure the loss gradient's conditioning, see GRAD_TOL_END_TO_END above)
    a_gpu = gpu_activations(net)
    a_gpu['keep_mask'] = keep
    for n, gref in O.backward_coarse(params, a_gpu).items():
        seen['grad/' + n] = e = rel(net.grad(n).cpu().numpy(), gref)
        assert e < bf16s_tol.grad_tol(n), (n, e)
    del net, ref
    # the fine phase at the same batch (round 5: fine/first's forward on the bf16 image form records the window positions, its
    # filter gradient comes straight from the pooled map's gradient, fine/second's backward reads a bf16 df2)
    net2 = models.MSDNReplica(B, params=params, precision='bf16s', global_step=2000000 // B)
    net2.step(*args)
    torch.cuda.synchronize()
    a2 = O.forward(params, img[sl], dep[sl], keep[sl])
    seen['depth/fine (fine phase)'] = e = rel(net2.fine[sl].cpu().numpy(), a2['fine'])
    assert e < bf16s_tol.DEPTH['fine'], e
    a_gpu = gpu_activations(net2)
    for n, gref in O.backward_fine(params, a_gpu).items():
        seen['grad/' + n] = e = rel(net2.grad(n).cpu().numpy(), gref)
        assert e < bf16s_tol.grad_tol(n), (n, e)
    print('\nprecision bf16s, B = 64, rel-L2 against the fp32 oracle:')
    for k, v in seen.items():
        print(f'  {k:45s} {v:.3e}')


def test_bf16_storage_fused_casts_are_the_same_step(models, monkeypatch):
    """Round 5: under precision 'bf16s' the five tensors that cross between the bf16 conv stack and the fp32 dense side are
    written by the reductions that produce their sources (a3d_second_output) instead of five cast / copy launches, and
    conv2d_0 / fine/first run from the 4-channel bf16 image: the fused casts change no bit of the step."""
    B = 64
    img, dep, keep = synth(B, 4242, 240, 320)
    params = O.init_params(3000)
    args = [torch.from_numpy(a).cuda() for a in (img, dep, keep)]
    nets = []
    for flag in ('1', '0'):
        monkeypatch.setenv('A3D_BF16S_FUSE_CASTS', flag)
        nets.append(models.MSDNReplica(B, params=params, precision='bf16s'))
    assert nets[0].fuse_casts and not nets[1].fuse_casts
    for _ in range(2):
        outs = [n.step(*args) for n in nets]
        torch.cuda.synchronize()
        for k in ('coarse_loss', 'fine_loss'):
            assert float(outs[0][k]) == float(outs[1][k])
    for name in ('coarse', 'fine', 'cat', 'drop', 'dz0', 'dc4', 'dc0' if nets[0].dc0 is not None else 'dp0'):
        assert torch.equal(getattr(nets[0], name), getattr(nets[1], name)), name
    for gname in nets[0].groups:
        for buf in ('grad', 'm'):
            assert torch.equal(getattr(nets[0].groups[gname], buf), getattr(nets[1].groups[gname], buf)), (gname, buf)


def test_bf16_storage_above_64_rows_per_batch(models):
    """precision 'bf16s' at a batch of 65..383 — the dense layers' bf16 x / dz form is the LDS-DMA kernel's
    64-row weight stream only; above that the replica keeps the dense layers' small side float32 (weight copies bf16).  B = 96,
    both trained phases, at the mode's tolerances."""
    B = 96
    img, dep, keep = synth(B, 9696, 240, 320)
    params = O.init_params(3000)
    args = [torch.from_numpy(a).cuda() for a in (img, dep, keep)]
    sl = [0, 47, 95]
    a = O.forward(params, img[sl], dep[sl], keep[sl])
    for gs in (0, 2000000 // B):
        net = models.MSDNReplica(B, params=params, precision='bf16s', global_step=gs)
        assert not net.dense_bf16_x and net.keep_dense_grads
        out = net.step(*args)
        torch.cuda.synchronize()
        assert out['phase'] == (1 if gs == 0 else 2)
        for k in ('coarse', 'fine'):
            assert rel(getattr(net, k)[sl].cpu().numpy(), a[k]) < bf16s_tol.DEPTH[k], k
        a_gpu = gpu_activations(net)
        a_gpu['keep_mask'] = keep
        for n, gref in (O.backward_coarse if gs == 0 else O.backward_fine)(params, a_gpu).items():
            assert rel(net.grad(n).cpu().numpy(), gref) < bf16s_tol.grad_tol(n), n
        del net


@pytest.mark.parametrize('global_step', [0, 2000000 // 2])
def test_bf16_storage_weight_copies_follow_the_masters(models, global_step):
    """under precision 'bf16s' every conv / dense_0 kernel has a bf16 copy beside its fp32 master.  With
    an optimizer that moves the weights (--beta2 < 1) the copies must be refreshed after every ApplyAdam, and after a
    checkpoint restore — otherwise forward and bwd-data keep computing with the initial weights."""
    B = 2
    params = shifted_params()
    net = models.MSDNReplica(B, params=params, beta2=0.999, precision='bf16s', global_step=global_step)
    assert net.wcopy
    for step in range(2):
        img, dep, keep = synth(B, 1000 + step, 96, 128)
        net.step(torch.from_numpy(img).cuda(), torch.from_numpy(dep).cuda(), torch.from_numpy(keep).cuda())
    moved = 0
    for n, c in net.wcopy.items():
        master = net.var(n + '/kernel')
        moved += int(not torch.equal(master.cpu(), torch.from_numpy(params[n + '/kernel'])))
        assert torch.equal(c, master.to(torch.bfloat16)), n                 # round-to-nearest-even, like a3d_cast_bf16
    assert moved >= (4 if global_step == 0 else 1)
    assert torch.equal(net.w4[:, :, :3, :], net.var('fine/first/conv2d/kernel')) and not net.w4[:, :, 3, :].any()
    # dense_1's copy has rows of 4072 elements (whole 16-byte pieces): 4070 rounded weights and two zeros; its bias alike
    k1, b1 = net.var('coarse/dense/dense_1/kernel'), net.var('coarse/dense/dense_1/bias')
    assert net.w1pad.shape == (4096, 4072) and torch.equal(net.w1pad[:, :4070], k1.to(torch.bfloat16)) and not net.w1pad[:, 4070:].any()
    assert torch.equal(net.b1pad[0, :4070], b1) and not net.b1pad[0, 4070:].any()
    if global_step == 0:
        assert not torch.equal(k1.cpu(), torch.from_numpy(params['coarse/dense/dense_1/kernel']))      # it did move
    # restore into a replica that was initialised with OTHER weights: copies follow, and the next step is the same step
    other = models.MSDNReplica(B, seed=1, beta2=0.999, precision='bf16s')
    other.load_state_dict(net.state_dict())
    for n, c in other.wcopy.items():
        assert torch.equal(c, net.wcopy[n]), n
    assert torch.equal(other.w4, net.w4) and torch.equal(other.w1pad, net.w1pad) and torch.equal(other.b1pad, net.b1pad)
    img, dep, keep = synth(B, 77, 96, 128)
    args = (torch.from_numpy(img).cuda(), torch.from_numpy(dep).cuda(), torch.from_numpy(keep).cuda())
    net.step(*args)
    other.step(*args)
    assert torch.equal(other.coarse, net.coarse) and torch.equal(other.fine, net.fine)
