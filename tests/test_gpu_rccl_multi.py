"""Data-parallel step over RCCL with one rank per device — needs at least two GPUs, skips cleanly on the one-GPU test
box (where tests/test_gpu_dp.py runs the same rank body over gloo).  N ranks == one rank on the concatenated batch, both
trained phases, the chunked dense bucket deferred across the step boundary (replaces src/ann3depth.py:77-92)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


needs_two = pytest.mark.skipif(torch.cuda.device_count() < 2, reason='RCCL with more than one rank needs two GPUs')


@needs_two
def test_two_ranks_two_gpus_rccl(tmp_path):
    out = str(tmp_path / 'ok.txt')
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), A3D_DIST_BACKEND='nccl', HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, 'dp_gpu_worker.py'), out], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    assert open(out).read() == '1'


@needs_two
def test_bench_two_gpus_reports_comm_fields(tmp_path):
    """bench.py --gpus 2 as the driver launches it: the line must carry the RCCL rank count, the per-bucket all-reduce
    times and the exposed (non-overlapped) communication time."""
    import json
    root = os.path.dirname(HERE)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
                        '127.0.0.1', '--master-port', str(_free_port()), os.path.join(root, 'bench.py'), '--gpus', '2',
                        '--steps', '5', '--warmup', '2', '--no-fine', '--also', ''], capture_output=True, text=True, env=env,
                       cwd=root, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['rccl_ranks'] == 2 and line['scaling'] == 'weak'
    assert set(line['allreduce_ms']) >= {'dense_1', 'dense_0_piece', 'conv_tail', 'conv_head'}
    assert line['exposed_comm_ms'] is not None and line['ms_per_step'] > 0
