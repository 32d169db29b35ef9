"""Rank body of tests/test_gpu_dp.py: two ranks share cuda:0 and reduce over gloo (RCCL refuses two ranks on one
device; the reducer code path — async bucket all-reduce, finish, 1/world folded into ApplyAdam — is the same)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ann3depth_amd import dp, models          # noqa: E402


def main(out_path):
    rank, local_rank, world = dp.init_from_env()
    assert world == 2 and dist.get_backend() == 'gloo'
    B = 2
    rng = np.random.default_rng(99)
    img = (rng.integers(0, 256, (world * B, 96, 128, 3)) / 255).astype(np.float32)
    dep = (rng.integers(0, 256, (world * B, 12, 16, 1)) / 255).astype(np.float32)
    keep = rng.random((world * B, 4096)) >= 0.5
    sl = slice(rank * B, (rank + 1) * B)
    cu = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dt).cuda()
    ti, td, tk = cu(img[sl]), cu(dep[sl]), cu(keep[sl], torch.uint8)
    ok = True
    for gstep, gnames in ((0, ('CoarseDense', 'CoarseConv')), (models.SAMPLES_COARSE // B, ('FineA', 'FineB'))):
        solo = models.MSDNReplica(B, seed=3000, global_step=gstep)
        solo.step(ti, td, tk)
        net = models.MSDNReplica(B, seed=3000, global_step=gstep, reducer=dp.GradReducer())
        net.step(ti, td, tk)
        if gstep == 0:
            assert net._deferred is not None          # the dense bucket rides across the step boundary ...
        net.settle()                                   # ... until someone needs it
        assert net._deferred is None
        torch.cuda.synchronize()
        for gn in gnames:
            local = solo.groups[gn].grad.clone()
            dist.all_reduce(local)                                        # sum of both ranks' local gradients
            ok &= bool(torch.equal(net.groups[gn].grad, local))           # the bucket holds exactly that sum
            m_expect = local * (1.0 / world) * np.float32(1 - np.float32(0.9))
            err = (net.groups[gn].m - m_expect).abs().max() / m_expect.abs().max()
            ok &= bool(err < 1e-6)                                         # ApplyAdam saw the mean gradient
            ok &= bool(torch.equal(net.groups[gn].var, solo.groups[gn].var))   # beta2 = 1: weights frozen
            other = net.groups[gn].m.clone()
            dist.broadcast(other, 0)
            ok &= bool(torch.equal(other, net.groups[gn].m))              # replicas stay bit-identical
    flag = torch.tensor([int(ok)])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        open(out_path, 'w').write(str(int(flag.item())))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main(sys.argv[1])
