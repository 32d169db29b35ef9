"""Data-parallel path on CPU: two gloo ranks through ann3depth_amd.dp; the all-reduced, 1/world-scaled per-rank
gradients must equal the gradient of the concatenated batch (loss mean over the GLOBAL batch)."""
import os
import socket
import subprocess
import sys

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
