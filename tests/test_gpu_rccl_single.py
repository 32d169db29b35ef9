"""The RCCL code path on the one GPU available to the tests: a world_size-1 'nccl' process group still goes through
RCCL's communicator setup, the asynchronous all_reduce on the process group's stream and the stream-level wait that
MSDNReplica.step relies on for compute/all-reduce overlap."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def test_step_with_rccl_reducer_world1():
    from ann3depth_amd import dp, models
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        red = dp.GradReducer()
        assert red.world_size == 1
        B = 2
        img = torch.rand((B, 96, 128, 3), device='cuda')
        dep = torch.rand((B, 12, 16, 1), device='cuda')
        keep = (torch.rand((B, 4096), device='cuda') >= 0.5).to(torch.uint8)
        solo = models.MSDNReplica(B, seed=3000)
        solo.step(img, dep, keep)
        net = models.MSDNReplica(B, seed=3000, reducer=red)
        for _ in range(3):                                   # several steps: buckets are reused, works are drained
            net.global_step = 0
            for g in net.groups.values():
                g.m.zero_()
                g.beta1_power = g.beta1_power * 0 + 0.9
            net.step(img, dep, keep)
        torch.cuda.synchronize()
        assert red.pending == []
        for gn in ('CoarseDense', 'CoarseConv'):
            assert torch.equal(net.groups[gn].grad, solo.groups[gn].grad)        # sum over one rank = identity
            assert torch.equal(net.groups[gn].m, solo.groups[gn].m)
    finally:
        dist.destroy_process_group()
