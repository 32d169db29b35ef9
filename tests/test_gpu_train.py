"""BASELINE config 1 on the GPU: `make train`'s path end to end — TFRecord shard (48x64x3 / 6x8x1 records, B=4)
-> dataset plugin -> model plugin -> driver loop with summaries, checkpoint, resume and signal stop; plus the
committed golden vectors and the DCNF unary stack against the oracle."""
import json
import os
import signal
import threading
import time

import numpy as np
import pytest
import torch

from oracle import dcnf as OD
from oracle import msdn as O
from oracle import tf13_ops as T

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def write_shard(root, n=40, seed=0):
    from ann3depth_amd import tfrecord
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(root, 'nyu'), exist_ok=True)
    with tfrecord.TFRecordWriter(os.path.join(root, 'nyu', 'train.tfrecords')) as w:
        for _ in range(n):
            img = rng.integers(0, 256, (48, 64, 3)).astype(np.float32) / np.float32(255) - np.float32(.5)
            dep = rng.integers(0, 256, (6, 8, 1)).astype(np.float32) / np.float32(255) - np.float32(.5)
            w.write_example(img, dep)


@pytest.mark.parametrize('u8_records', [True, False])
def test_train_op_matches_oracle_on_a_dequeued_batch(tmp_path, monkeypatch, u8_records):
    """One dequeued batch through the model plugin against the oracle — on BOTH transfer paths: converter-written records
    staged and DMA'd as uint8 pixel values and rebuilt by the resize kernel (data.py), and the plain float32 path
    (A3D_NO_U8_RECORDS=1).  The oracle always sees the float32 the reference's loader would have produced."""
    from ann3depth_amd import data, models
    if not u8_records:
        monkeypatch.setenv('A3D_NO_U8_RECORDS', '1')
    write_shard(str(tmp_path))
    inputs, targets = data.inputs(str(tmp_path), 'nyu', 4, seed=3)
    op = models.msdn(inputs, targets)
    op.copied[0].synchronize()                       # batch 0 has landed in device buffer 0 (prefetched)
    assert (op.cur[0][0].dtype == torch.uint8) == u8_records and (op.cur[0][1].dtype == torch.uint8) == u8_records
    img, dep = (t.cpu().numpy() for t in op.cur[0])
    if u8_records:
        img, dep = data.expand_u8(img), data.expand_u8(dep)          # the loader's float32, from the pixel values
    out = op.run()                                   # consumes it; the buffer is refilled with batch 2 afterwards
    torch.cuda.synchronize()
    assert img.shape == (4, 48, 64, 3) and dep.shape == (4, 6, 8, 1) and img.dtype == np.float32
    assert img.min() >= 0 and img.max() <= 1                               # '+0.5' applied: k/255
    keep = op.keep.cpu().numpy().astype(bool)
    assert 0.45 < keep.mean() < 0.55
    a = O.forward(O.init_params(3000), img, dep, keep)
    np.testing.assert_array_equal(op.replica.x.cpu().numpy(), a['images'])   # the resized inputs: bit for bit on both paths
    np.testing.assert_array_equal(op.replica.t.cpu().numpy(), a['depths'])
    assert rel(op.replica.coarse.cpu().numpy(), a['coarse']) < 1e-3
    assert rel(op.replica.fine.cpu().numpy(), a['fine']) < 1e-3
    assert abs(float(out['coarse_loss']) - a['loss_coarse']) < 2e-3 * abs(a['loss_coarse'])
    assert op.global_step == 1
    op.pipeline.close()


def test_mixed_batches_fall_back_to_float32_per_feature(tmp_path):
    """A shard whose depth maps are NOT of the converter's form (arbitrary floats) while its images are: images travel as
    uint8, depths as float32; and a shard with ONE odd image among converter-written ones: the batches that hold it go
    as float32 (the uint8-staged records of such a batch are expanded on the host), all values exact."""
    from ann3depth_amd import data, models, tfrecord
    rng = np.random.default_rng(5)
    root = str(tmp_path)
    os.makedirs(os.path.join(root, 'nyu'))
    stored = []
    with tfrecord.TFRecordWriter(os.path.join(root, 'nyu', 'train.tfrecords')) as w:
        for i in range(24):
            img = rng.integers(0, 256, (48, 64, 3)).astype(np.float32) / np.float32(255) - np.float32(.5)
            if i == 7:
                img[3, 4, 1] = np.nextafter(img[3, 4, 1], np.float32(1))      # one float that is no pixel value
            img[0, 0, 0] = np.float32(i) / np.float32(255) - np.float32(.5)    # tag
            dep = (rng.random((6, 8, 1)) - 0.5).astype(np.float32)
            w.write_example(img, dep)
            stored.append((img, dep))
    inputs, targets = data.inputs(root, 'nyu', 4, epochs=1, seed=1, num_threads=2)
    op = models.msdn(inputs, targets)
    kinds = set()
    for _ in range(6):                               # all 24 records
        i = op.k & 1
        op.copied[i].synchronize()
        img_t, dep_t = op.cur[i]
        assert dep_t.dtype == torch.float32
        img = img_t.cpu().numpy()
        tags = [int(t) for t in (img[:, 0, 0, 0] if img.dtype == np.uint8 else np.rint(img[:, 0, 0, 0] * 255))]
        kinds.add((img.dtype == np.uint8, 7 in tags))
        want = np.stack([stored[t][0] for t in tags]) + np.float32(.5)
        np.testing.assert_array_equal(data.expand_u8(img) if img.dtype == np.uint8 else img, want)
        np.testing.assert_array_equal(dep_t.cpu().numpy(), np.stack([stored[t][1] for t in tags]) + np.float32(.5))
        op.run()
    assert (False, True) in kinds and not any(u8 and odd for u8, odd in kinds)    # the odd record's batch went as float32
    assert any(u8 for u8, _ in kinds)                                              # ... the others as uint8
    op.pipeline.close()


def test_resize_from_uint8_equals_resize_from_the_loaders_float32():
    """a3d_resize_bilinear_tf1_ex: every one of the 256 pixel values, image and depth map of a pair and the single-tensor
    form, against the float32 path bit for bit (BASELINE config 2's sizes)."""
    from ann3depth_amd import data, ops
    rng = np.random.default_rng(9)
    k_img = rng.integers(0, 256, (3, 480, 640, 3), dtype=np.uint8)
    k_img.reshape(-1)[:256] = np.arange(256, dtype=np.uint8)
    k_dep = rng.integers(0, 256, (3, 480, 640, 1), dtype=np.uint8)
    f_img, f_dep = data.expand_u8(k_img), data.expand_u8(k_dep)
    assert f_img.dtype == np.float32
    np.testing.assert_array_equal(f_img, (k_img.astype(np.float32) / np.float32(255.) - np.float32(.5)) + np.float32(.5))
    dev = lambda a: torch.from_numpy(a).cuda()
    outs = []
    for xi, xd in ((dev(f_img), dev(f_dep)), (dev(k_img), dev(k_dep)), (dev(k_img), dev(f_dep)), (dev(f_img), dev(k_dep))):
        y0, y1 = torch.empty((3, 228, 304, 3), device='cuda'), torch.empty((3, 55, 74, 1), device='cuda')
        ops.resize_bilinear_tf1_pair(xi, y0, xd, y1)
        outs.append((y0, y1))
    for y0, y1 in outs[1:]:
        assert torch.equal(y0, outs[0][0]) and torch.equal(y1, outs[0][1])
    np.testing.assert_array_equal(outs[0][0].cpu().numpy(), T.resize_bilinear_tf1(f_img, 228, 304))
    single = torch.empty((3, 240, 320, 3), device='cuda')
    ref = torch.empty_like(single)
    ops.resize_bilinear_tf1(dev(k_img), single)
    ops.resize_bilinear_tf1(dev(f_img), ref)
    assert torch.equal(single, ref)


def test_make_train_drop_in(tmp_path):
    from ann3depth_amd import ann3depth
    write_shard(str(tmp_path))
    ck = str(tmp_path / 'ckpt')
    base = ['--model', 'msdn', '--batchsize', '4', '--ckptdir', ck, '--datadir', str(tmp_path), '--sumfreq', '2',
            '--id', 'r1']
    assert ann3depth.main(base + ['--steps', '6', 'nyu']) == 0
    d = os.path.join(ck, 'msdn_r1')                                       # <ckptdir>/<model>_<id>, src/ann3depth.py:73-75
    assert ann3depth.latest_checkpoint(d).endswith('model.ckpt-6.pt')
    trace = json.load(open(os.path.join(d, 'trace-1.json')))                # TraceHook: first step after start
    assert trace['global_step'] == 1
    assert all(r['ms'] > 0 for r in trace['launches'])
    # one record per GEMM-class launch of a coarse-phase step (src/models.py:211-251 forward, the backward of coarse/*), named by
    # direction and GEMM extents — not a count: fusions remove launches, new kernels add records
    got = {(r['mode'], r['n'], r['k']) if r['mode'] != 'bwd_filter' else (r['mode'], r['m'], r['n']) for r in trace['launches']}
    want = {('fwd', 96, 363), ('fwd', 256, 2400), ('fwd', 384, 2304), ('fwd', 384, 3456), ('fwd', 256, 3456),      # conv2d_0 .. conv2d_4
            ('fwd', 63, 243), ('fwd', 64, 1600),                                                                       # fine/first, fine/second
            ('bwd_data', 12288, 4096), ('bwd_data', 4096, 4070),                                                       # dense_0, dense_1
            ('bwd_data', 384, 3456), ('bwd_data', 256, 3456), ('bwd_data', 96, 6400),                                  # conv2d_3 .. conv2d_1
            ('bwd_filter', 3456, 256), ('bwd_filter', 3456, 384), ('bwd_filter', 2304, 384), ('bwd_filter', 2400, 256),
            ('bwd_filter', 363, 96)}
    assert want <= got, sorted(want - got)
    # conv2d_4's stride-2 bwd-data: its four output-parity classes as one launch (K = all 9 taps x 256) or, at this small batch,
    # one launch per class (4 / 2 / 2 / 1 taps)
    assert ('bwd_data', 384, 2304) in got or {('bwd_data', 384, 1024), ('bwd_data', 384, 512), ('bwd_data', 384, 256)} <= got
    sums = [json.loads(l) for l in open(os.path.join(d, 'summaries.jsonl'))]
    assert [s['global_step'] for s in sums] == [2, 4, 6]
    assert all(np.isfinite(s['loss/coarse_loss']) and s['optimizers/Phase'] == 1 for s in sums)
    assert any(f.startswith('events.out.tfevents.') for f in os.listdir(d))
    sd = torch.load(ann3depth.latest_checkpoint(d))
    assert int(sd['global_step']) == 6
    assert 'coarse/conv/conv2d_0/kernel' in sd and 'coarse/conv/conv2d_0/kernel/CoarseConv' in sd
    w0 = O.init_params(3000)['coarse/dense/dense_1/kernel']
    np.testing.assert_array_equal(sd['coarse/dense/dense_1/kernel'].numpy(), w0)       # beta2 = 1: weights frozen
    assert float(sd['coarse/dense/dense_1/kernel/CoarseDense'].abs().max()) > 0       # ... but m evolves
    # resume from the checkpoint
    assert ann3depth.main(base + ['--steps', '8', 'nyu']) == 0
    assert ann3depth.latest_checkpoint(d).endswith('model.ckpt-8.pt')
    assert int(torch.load(ann3depth.latest_checkpoint(d))['global_step']) == 8
    # StopAtSignalHook: stop after the current step, save, exit code = signal number (src/ann3depth.py:129)
    threading.Thread(target=lambda: (time.sleep(1.0), os.kill(os.getpid(), signal.SIGUSR1)), daemon=True).start()
    rc = ann3depth.main(base + ['--steps', '100000000', 'nyu'])
    assert rc == signal.SIGUSR1
    assert int(torch.load(ann3depth.latest_checkpoint(d))['global_step']) > 8
    for s in (signal.SIGUSR1, signal.SIGUSR2, signal.SIGALRM, signal.SIGINT, signal.SIGTERM):
        signal.signal(s, signal.SIG_DFL)
    # parameter-server jobs do not exist any more
    assert ann3depth.main(base + ['--job-name', 'ps', 'nyu']) == 0


def test_continue_from_a_tensorflow_checkpoint(tmp_path):
    """A checkpoint directory as tf.train.Saver leaves it (`checkpoint` naming a bundle prefix, .index + .data files,
    model variables and Adam slots under their TF names) is picked up by `make train`; --tf-checkpoints writes one."""
    from ann3depth_amd import ann3depth, tfckpt
    write_shard(str(tmp_path))
    ck = str(tmp_path / 'ckpt')
    d = os.path.join(ck, 'msdn')
    os.makedirs(d)
    rng = np.random.default_rng(9)
    params = O.init_params(77)
    tensors = dict(params)
    tensors['coarse/dense/dense_1/kernel/CoarseDense'] = rng.standard_normal(params['coarse/dense/dense_1/kernel'].shape
                                                                             ).astype(np.float32)
    tensors['global_step'] = np.int64(40)
    tfckpt.write_bundle(os.path.join(d, 'model.ckpt-40'), tensors)
    with open(os.path.join(d, 'checkpoint'), 'w') as f:
        f.write('model_checkpoint_path: "model.ckpt-40"\nall_model_checkpoint_paths: "model.ckpt-40"\n')
    assert ann3depth.latest_checkpoint(d).endswith('model.ckpt-40')
    base = ['--model', 'msdn', '--batchsize', '4', '--ckptdir', ck, '--datadir', str(tmp_path), '--sumfreq', '1',
            '--tf-checkpoints']
    # nothing left to do at --steps 40: the session restores, and saves what it restored
    assert ann3depth.main(base + ['--steps', '40', 'nyu']) == 0
    sd = torch.load(os.path.join(d, 'model.ckpt-40.pt'))
    for n, a in params.items():
        np.testing.assert_array_equal(sd[n].numpy(), a)
    np.testing.assert_array_equal(sd['coarse/dense/dense_1/kernel/CoarseDense'].numpy(),
                                  tensors['coarse/dense/dense_1/kernel/CoarseDense'])
    assert float(sd['coarse/dense/dense_0/kernel/CoarseDense'].abs().max()) == 0          # absent slot: stays zero
    assert abs(float(sd['CoarseConv/beta1_power']) - 0.9 ** 41) < 1e-6 and float(sd['FineA/beta1_power']) == np.float32(0.9)
    # back to the TensorFlow-only directory, and train on
    os.remove(os.path.join(d, 'model.ckpt-40.pt'))
    with open(os.path.join(d, 'checkpoint'), 'w') as f:
        f.write('model_checkpoint_path: "model.ckpt-40"\n')
    os.remove(os.path.join(d, 'summaries.jsonl'))
    assert ann3depth.main(base + ['--steps', '42', 'nyu']) == 0
    sums = [json.loads(l) for l in open(os.path.join(d, 'summaries.jsonl'))]
    assert [s['global_step'] for s in sums] == [41, 42]                                   # continued from step 40
    sd = torch.load(os.path.join(d, 'model.ckpt-42.pt'))
    np.testing.assert_array_equal(sd['fine/third/kernel'].numpy(), params['fine/third/kernel'])
    # beta powers rebuilt from global_step: beta1^(40 + 1) restored, two more applies since
    assert abs(float(sd['CoarseConv/beta1_power']) - 0.9 ** 43) < 1e-6
    # and the bundle written next to the .pt file holds the same state
    back = tfckpt.read_bundle(os.path.join(d, 'model.ckpt-42'))
    assert int(back['global_step']) == 42 and back['global_step'].dtype == np.int64
    for k, v in sd.items():
        np.testing.assert_array_equal(back[k], v.numpy())
    for s in (signal.SIGUSR1, signal.SIGUSR2, signal.SIGALRM, signal.SIGINT, signal.SIGTERM):
        signal.signal(s, signal.SIG_DFL)


def test_golden_vectors_on_gpu():
    from ann3depth_amd import models, ops
    g = np.load(os.path.join(GOLD, 'msdn_b2.npz'))
    net = models.MSDNReplica(2, params=O.init_params(int(g['seed_params'])))
    cu = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dt).cuda()
    net.forward(cu(g['images']), cu(g['depths']), cu(g['keep'], torch.uint8))
    assert rel(net.coarse.cpu().numpy(), g['coarse']) < 1e-3 and rel(net.fine.cpu().numpy(), g['fine']) < 1e-3
    assert abs(net.loss_coarse.item() - g['loss_coarse']) < 2e-3 * abs(g['loss_coarse'])
    k = np.load(os.path.join(GOLD, 'op_kats.npz'))
    for name in ('same5', 's2valid', 'cout63', 'cout1', 'cin3s4'):
        st, same = (int(v) for v in k[f'conv_{name}_geom'])
        x, w, b, dz = (k[f'conv_{name}_{q}'] for q in ('x', 'w', 'b', 'dz'))
        n, h, wd, c = x.shape
        d = ops.conv_desc(n, h, wd, c, w.shape[3], w.shape[0], w.shape[1], st, 'SAME' if same else 'VALID')
        y = torch.empty(k[f'conv_{name}_y'].shape, device='cuda')
        ops.conv2d_fwd(d, cu(x), cu(w), cu(b), y, 'relu')
        assert rel(y.cpu().numpy(), k[f'conv_{name}_y']) < 1e-5
        dw = torch.empty(w.shape, device='cuda'); db = torch.empty(b.shape, device='cuda')
        ops.conv2d_bwd_filter(d, cu(x), cu(dz), dw, db)
        assert rel(dw.cpu().numpy(), k[f'conv_{name}_dw']) < 1e-5 and rel(db.cpu().numpy(), k[f'conv_{name}_db']) < 1e-5
        dx = torch.empty(x.shape, device='cuda')
        ops.conv2d_bwd_data(d, cu(dz), cu(w), dx)
        assert rel(dx.cpu().numpy(), k[f'conv_{name}_dx']) < 1e-5
    y = torch.empty(k['pool_y'].shape, device='cuda'); dx = torch.empty(k['pool_x'].shape, device='cuda')
    ops.maxpool2x2_fwd(cu(k['pool_x']), y)
    ops.maxpool2x2_bwd(cu(k['pool_x']), cu(k['pool_dy']), dx, relu_mask=False)
    np.testing.assert_array_equal(y.cpu().numpy(), k['pool_y'])
    np.testing.assert_array_equal(dx.cpu().numpy(), k['pool_dx'])
    for tag, oh, ow in (('up', 55, 74), ('dn', 23, 30)):
        x = k[f'resize_{tag}_x']
        y = torch.empty((x.shape[0], oh, ow, x.shape[3]), device='cuda')
        ops.resize_bilinear_tf1(cu(x), y)
        np.testing.assert_array_equal(y.cpu().numpy(), k[f'resize_{tag}_y'])
    for tag, b2 in (('ref', 1.0), ('learn', 0.999)):
        v = cu(k['adam_var0']); m = torch.zeros_like(v); s = torch.zeros_like(v)
        b1p, b2p = np.float32(0.9), np.float32(b2)
        for i in range(3):
            ops.adam_apply_tf1(v, m, s, cu(k['adam_g'][i]), 0.1, 0.9, b2, 1e-8, float(b1p), float(b2p))
            b1p, b2p = b1p * np.float32(0.9), b2p * np.float32(b2)
        np.testing.assert_array_equal(v.cpu().numpy(), k[f'adam_{tag}_var'])
        np.testing.assert_array_equal(m.cpu().numpy(), k[f'adam_{tag}_m'])


def test_dcnf_unary_matches_oracle():
    """BASELINE config 4 at B=1 (48 patches): forward z and the unary backward from a synthetic dz."""
    from ann3depth_amd import models
    rng = np.random.default_rng(5)
    img = (rng.integers(0, 256, (1, 480, 640, 3)) / 255).astype(np.float32)
    params = OD.init_params(3000)
    net = models.DCNFUnary(1, params=params)
    assert (net.rows, net.cols, net.P) == (6, 8, 48)
    z = net.forward(torch.from_numpy(img).cuda())
    patches = OD.patches(img)
    np.testing.assert_array_equal(net.act['x'].cpu().numpy(), patches)                    # resize + patches bit-exact
    a = OD.unary_forward(params, patches)
    assert rel(z.cpu().numpy().reshape(48, 1), a['z']) < 1e-3
    for n in ('conv2d/pool', 'conv2d_1/pool', 'conv2d_2', 'conv2d_4/pool', 'dense', 'dense_1'):
        assert rel(net.act[n].cpu().numpy(), a[n]) < 1e-4, n
    dz = rng.standard_normal((48, 1)).astype(np.float32)
    net.backward(torch.from_numpy(dz).cuda())
    torch.cuda.synchronize()
    a_gpu = net.activations()          # conv + pool run fused: pre-pool tensors come back as "maximum in place" images
    hit = a_gpu['conv2d'] > 0
    np.testing.assert_allclose(a_gpu['conv2d'][hit], a['conv2d'][hit], rtol=1e-3, atol=1e-5)
    a_gpu['flat'] = a_gpu['conv2d_4/pool'].reshape(48, -1)
    g = OD.unary_backward(params, a_gpu, dz)
    for n, gref in g.items():
        assert rel(net.group.view(net.group.grad, n).cpu().numpy(), gref) < 1e-4, n


def test_tracehook_as_a_rocprofv3_capture(tmp_path):
    """--profiler rocprofv3: the driver runs itself under rocprofv3 as a child (this test's process has long initialised the
    GPU: it only starts the driver, which starts the profiler); the kernel trace then holds the traced steps — the
    first and every `--trace-every`-th — and no others."""
    import csv
    import glob
    import shutil
    import subprocess
    import sys
    if not shutil.which('rocprofv3') and not os.path.exists('/opt/rocm/bin/rocprofv3'):
        pytest.skip('no rocprofv3')
    write_shard(str(tmp_path))
    ck = str(tmp_path / 'ckpt')
    env = dict(os.environ, TMPDIR=str(tmp_path))
    env.pop('A3D_UNDER_ROCPROF', None)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, '-m', 'ann3depth_amd.ann3depth', '--model', 'msdn', '--batchsize', '4', '--ckptdir', ck,
           '--datadir', str(tmp_path), '--steps', '3', '--trace-every', '2', '--profiler', 'rocprofv3', 'nyu']
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = os.path.join(ck, 'msdn')
    assert ann3depth_latest(d).endswith('model.ckpt-3.pt')
    traces = glob.glob(os.path.join(d, 'rocprof', '**', '*kernel_trace.csv'), recursive=True)
    assert traces, os.listdir(os.path.join(d, 'rocprof'))
    names = [row['Kernel_Name'] for row in csv.DictReader(open(traces[0]))]
    # one loss gradient per training step: steps 1 and 2 are traced ((step + 1) % 2 == 0 after step 1), step 3 is not
    assert sum('silog_bwd' in n for n in names) == 2
    assert any('igemm' in n for n in names)
    markers = glob.glob(os.path.join(d, 'rocprof', '**', '*marker*trace.csv'), recursive=True)
    assert markers
    text = open(markers[0]).read()
    assert 'global_step 1' in text and 'global_step 2' in text and 'global_step 3' not in text


def ann3depth_latest(d):
    from ann3depth_amd import ann3depth
    return ann3depth.latest_checkpoint(d)
