"""Data-parallel step through the real HIP path: two ranks on the one GPU of the test box (gloo transport)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_two_ranks_one_gpu(tmp_path):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / 'ok.txt')
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), A3D_DIST_BACKEND='gloo')
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, 'dp_gpu_worker.py'), out], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    assert open(out).read() == '1'


@pytest.mark.parametrize('model,batch,steps', [('msdn', 4, 5), ('dcnf', 1, 3)])
def test_make_train_two_replicas(tmp_path, model, batch, steps):
    """`make train GPUS=2` as the driver runs it (two processes of ann3depth_amd.ann3depth; gloo because both ranks share
    the one GPU of the test box): rank-sharded input, chief-only checkpoints and summaries."""
    import json

    import numpy as np
    import torch
    from ann3depth_amd import tfrecord
    root = str(tmp_path / 'data')
    rng = np.random.default_rng(0)
    os.makedirs(os.path.join(root, 'nyu'))
    with tfrecord.TFRecordWriter(os.path.join(root, 'nyu', 'train.tfrecords')) as w:
        for _ in range(48):
            w.write_example(rng.random((48, 64, 3)).astype(np.float32) - np.float32(.5),
                            rng.random((6, 8, 1)).astype(np.float32) - np.float32(.5))
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ck = str(tmp_path / 'ckpt')
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), A3D_DIST_BACKEND='gloo', PYTHONPATH=os.path.dirname(HERE))
        procs.append(subprocess.Popen(
            [sys.executable, '-m', 'ann3depth_amd.ann3depth', '--model', model, '--batchsize', str(batch), '--steps',
             str(steps),
             '--ckptdir', ck if rank == 0 else str(tmp_path / 'unused'), '--datadir', root, '--sumfreq', '1',
             '--beta2', '0.999', '--job-name', 'worker', '--timeout', '600', 'nyu'],
            env=env, cwd=os.path.dirname(HERE)))
    for p in procs:
        assert p.wait(timeout=600) == 0
    d = os.path.join(ck, model)
    sums = [json.loads(l) for l in open(os.path.join(d, 'summaries.jsonl'))]
    assert [s_['global_step'] for s_ in sums] == list(range(1, steps + 1))
    assert all(np.isfinite(s_['loss/coarse_loss' if model == 'msdn' else 'loss/mean_loss']) for s_ in sums)
    sd = torch.load(os.path.join(d, f'model.ckpt-{steps}.pt'))
    assert int(sd['global_step']) == steps
    if model == 'msdn':
        assert float(sd['coarse/dense/dense_1/kernel/CoarseDense'].abs().max()) > 0
    else:
        assert 'unary/unary_layers/conv2d/kernel' in sd and 'pairwise/pairwise_layers/dense/kernel' in sd
    assert not os.path.exists(str(tmp_path / 'unused'))                  # only the chief writes
