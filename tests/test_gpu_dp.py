"""Data-parallel step through the real HIP path: two ranks on the one GPU of the test box (gloo transport)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize('world,dense', [(2, 'scatter'), (4, 'scatter'), (2, 'allreduce')])
def test_ranks_share_one_gpu(tmp_path, world, dense):
    """tests/dp_gpu_worker.py: the data-parallel replica through the HIP kernels, the ranks sharing the box's one GPU over
    gloo — two ranks (sums of two addends: bit for bit), and four (slices, padding and the gather of m at a world size
    the scaling bench runs; sums to 1e-6); and the exchange a backend that fails the in-place self-check falls back to
    (A3D_DP_DENSE=allreduce: the dense bucket all-reduced, ApplyAdam replicated)."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / 'ok.txt')
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), A3D_DIST_BACKEND='gloo', A3D_DP_DENSE=dense)
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, 'dp_gpu_worker.py'), out], env=env))
    for p in procs:
        assert p.wait(timeout=400) == 0
    assert open(out).read() == '1'


def test_two_ranks_at_batch_32_each_match_the_oracle_on_all_64(tmp_path):
    """the step a data-parallel rank actually runs, at BASELINE config 3's per-rank batch (32, stored
    480x640), both trained phases, against the oracle on the concatenated batch (tests/dp_gpu_worker.py::main_b32)."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / 'ok.txt')
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), A3D_DIST_BACKEND='gloo')
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, 'dp_gpu_worker.py'), out, 'b32'], env=env))
    for p in procs:
        assert p.wait(timeout=900) == 0
    assert open(out).read() == '1'


def test_a_non_finite_gradient_on_one_rank_is_repaired_on_all(tmp_path):
    """_watch_poison / _poll_poison / _resync / gather_state of the rank-sharded dense optimizer state, at
    replica level, two ranks (tests/dp_gpu_worker.py::main_poison)."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / 'ok.txt')
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), A3D_DIST_BACKEND='gloo')
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, 'dp_gpu_worker.py'), out, 'poison'], env=env))
    for p in procs:
        assert p.wait(timeout=400) == 0
    assert open(out).read() == '1'


def test_two_bf16_storage_ranks_at_batch_64_each_match_the_oracle_on_all_128(tmp_path):
    """BASELINE config 5's data-parallel rank — precision 'bf16s' with a reducer, B = 64 per rank — both
    trained phases, against the oracle on the concatenated 128 samples (tests/dp_gpu_worker.py::main_bf16s)."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / 'ok.txt')
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), A3D_DIST_BACKEND='gloo')
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, 'dp_gpu_worker.py'), out, 'bf16s'], env=env))
    for p in procs:
        assert p.wait(timeout=1100) == 0
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


def test_signal_to_one_replica_stops_both(tmp_path):
    """SIGUSR1 delivered to ONE rank of a two-replica `make train`.  The stop decision is collective
    (dp.GradReducer.agree), so both ranks leave after the same global step, the chief writes the final checkpoint, and both
    exit with the signal number (src/ann3depth.py:129) instead of hanging in the next step's all-reduce."""
    import signal
    import time

    import numpy as np
    import torch
    from ann3depth_amd import tfrecord
    root = str(tmp_path / 'data')
    rng = np.random.default_rng(1)
    os.makedirs(os.path.join(root, 'nyu'))
    with tfrecord.TFRecordWriter(os.path.join(root, 'nyu', 'train.tfrecords')) as w:
        for _ in range(32):
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
            [sys.executable, '-m', 'ann3depth_amd.ann3depth', '--model', 'msdn', '--batchsize', '2', '--steps', '100000000',
             '--ckptdir', ck if rank == 0 else str(tmp_path / 'unused'), '--datadir', root, '--sumfreq', '1',
             '--job-name', 'worker', '--timeout', '600', 'nyu'], env=env, cwd=os.path.dirname(HERE)))
    d = os.path.join(ck, 'msdn')
    deadline = time.time() + 240
    while time.time() < deadline:                   # wait until training is under way (summaries appear every step)
        if os.path.exists(os.path.join(d, 'summaries.jsonl')) and os.path.getsize(os.path.join(d, 'summaries.jsonl')) > 0:
            break
        assert all(p.poll() is None for p in procs), 'a replica died before training started'
        time.sleep(0.5)
    time.sleep(1.0)
    procs[1].send_signal(signal.SIGUSR1)            # the NON-chief replica only
    for p in procs:
        assert p.wait(timeout=120) == signal.SIGUSR1
    last = [l for l in open(os.path.join(d, 'checkpoint')) if l.startswith('model_checkpoint_path')][0].split('"')[1]
    sd = torch.load(os.path.join(d, last))
    assert int(sd['global_step']) >= 1


def test_bench_two_ranks_reports_comm_fields():
    """bench.py --gpus 2 the way the driver launches it, here with both ranks on the one GPU over gloo: the line carries the
    rank count, the per-bucket all-reduce times and the exposed communication time (tests/test_gpu_rccl_multi.py runs the
    same over RCCL when two devices exist)."""
    import json
    root = os.path.dirname(HERE)
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, A3D_DIST_BACKEND='gloo')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
                        '127.0.0.1', '--master-port', str(port), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3',
                        '--warmup', '1', '--batch', '4', '--no-fine', '--also', ''], capture_output=True, text=True, env=env,
                       cwd=root, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['rccl_ranks'] == 2 and line['config']['per_gpu_batch'] == 4
    assert 'batch 4 per GPU' in line['config']['workload']
    assert set(line['allreduce_ms']) == {'dense_1', 'dense_0_piece', 'conv_tail', 'conv_head'}
    assert all(v['ms'] > 0 for v in line['allreduce_ms'].values())
    assert line['exposed_comm_ms'] is not None and line['ms_per_step_without_comm'] > 0
