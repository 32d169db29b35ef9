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
