"""Data-parallel path on CPU: gloo ranks through ann3depth_amd.dp; the all-reduced, 1/world-scaled per-rank gradients
must equal the gradient of the concatenated batch (loss mean over the GLOBAL batch), and the reduce-scattered
dense bucket with its rank-sharded Adam slot must equal one rank's — at the world sizes the driver's scaling bench runs."""
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_allreduce_equals_big_batch(tmp_path):
    port = free_port()
    out = str(tmp_path / 'result.txt')
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), OMP_NUM_THREADS='2', OPENBLAS_NUM_THREADS='2')
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, 'dp_worker.py'), out], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    e_dense, e_conv = (float(v) for v in open(out).read().split())
    assert e_dense < 1e-5 and e_conv < 1e-5
    # the stop decision was collective: both ranks left after step 3 although only rank 1 was signalled
    import signal
    assert open(out + '.stop0').read() == open(out + '.stop1').read() == f'3 {int(signal.SIGUSR1)}\n'


@pytest.mark.parametrize('world', [4, 8])
def test_sharded_dense_bucket_at_the_scaling_benchs_world_sizes(tmp_path, world):
    """tests/dp_world_worker.py: padded bucket, pieces in production order, in-place reduce-scatter, ApplyAdam of each
    rank's slice, gather of m — bit for bit what ONE rank computes from the summed gradient, at 4 and 8 ranks."""
    port = free_port()
    out = str(tmp_path / 'result.txt')
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1')
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, 'dp_world_worker.py'), out], env=env))
    for p in procs:
        assert p.wait(timeout=900) == 0          # (a cold container pages torch in once per process: minutes, not seconds)
    assert open(out).read() == f'{world} ok\n'
