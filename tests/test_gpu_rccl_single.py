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
            net.settle()                                     # the dense bucket of the previous step is still in flight
            net.global_step = 0
            for g in net.groups.values():
                g.m.zero_()
                g.beta1_power = g.beta1_power * 0 + 0.9
            net.step(img, dep, keep)
            assert len(red.pending) == 4                     # CoarseDense (dense_1 + three pieces of dense_0) rides across the step boundary
        net.settle()
        torch.cuda.synchronize()
        assert red.pending == []
        for gn in ('CoarseDense', 'CoarseConv'):
            assert torch.equal(net.groups[gn].grad, solo.groups[gn].grad)        # sum over one rank = identity
            assert torch.equal(net.groups[gn].m, solo.groups[gn].m)
    finally:
        dist.destroy_process_group()


def test_deferred_dense_bucket_equals_immediate_update():
    """Several consecutive steps with the dense bucket settled late (before the next dense_0) must leave exactly the
    state of a replica without reducer: same kernels, same operands, only the enqueue order differs (beta2 < 1 so that
    the weights move and a late or missing ApplyAdam would show)."""
    from ann3depth_amd import dp, models
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        B = 2
        torch.manual_seed(0)
        batches = [(torch.rand((B, 96, 128, 3), device='cuda'), torch.rand((B, 12, 16, 1), device='cuda'),
                    (torch.rand((B, 4096), device='cuda') >= 0.5).to(torch.uint8)) for _ in range(4)]
        solo = models.MSDNReplica(B, seed=3000, beta2=0.999)
        net = models.MSDNReplica(B, seed=3000, beta2=0.999, reducer=dp.GradReducer())
        for b in batches:
            solo.step(*b)
            net.step(*b)
        sd_solo, sd_net = solo.state_dict(), net.state_dict()              # state_dict() settles
        assert net._deferred is None
        for k in sd_solo:
            assert torch.equal(sd_solo[k], sd_net[k]), k
        assert torch.equal(solo.coarse, net.coarse) and torch.equal(solo.fine, net.fine)
    finally:
        dist.destroy_process_group()
