"""Rank body of tests/test_gpu_dp.py and tests/test_gpu_rccl_multi.py.  On the one-GPU test box two ranks share cuda:0 and
reduce over gloo (RCCL refuses two ranks on one device; the reducer code path — async bucket all-reduces in production
order, the dense bucket deferred across the step boundary, 1/world folded into ApplyAdam — is the same); on a box with at
least two GPUs the same body runs one rank per device over RCCL (A3D_DIST_BACKEND=nccl)."""
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
    assert world == 2 and dist.get_backend() == os.environ.get('A3D_DIST_BACKEND', 'gloo')
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
            if not ok:
                print('bucket sum check failed', gn, gstep, flush=True)
            m_expect = local * (1.0 / world) * np.float32(1 - np.float32(0.9))
            err = (net.groups[gn].m - m_expect).abs().max() / m_expect.abs().max()
            ok &= bool(err < 1e-6)                                         # ApplyAdam saw the mean gradient
            ok &= bool(torch.equal(net.groups[gn].var, solo.groups[gn].var))   # beta2 = 1: weights frozen
            other = net.groups[gn].m.clone()
            dist.broadcast(other, 0)
            ok &= bool(torch.equal(other, net.groups[gn].m))              # replicas stay bit-identical
    # N ranks == one rank on the concatenated batch: the activations of the big replica's forward, sliced
    # per rank, are what each rank computed (same kernels, same per-sample arithmetic), and its gradient of the GLOBAL
    # batch mean is the all-reduced sum / world.  The loss gradient ~ 1/(o + 1e-8) makes this an fp32-conditioning
    # check, not a bit-level one (tests/test_gpu_msdn.py), hence the loose bound.
    big = models.MSDNReplica(world * B, seed=3000, global_step=0)
    big.step(cu(img), cu(dep), cu(keep, torch.uint8))
    net = models.MSDNReplica(B, seed=3000, global_step=0, reducer=dp.GradReducer())
    net.step(ti, td, tk)
    net.settle()
    torch.cuda.synchronize()
    ok &= bool(((net.coarse - big.coarse[sl]).norm() / big.coarse[sl].norm()) < 1e-5)   # (tile plans differ with the batch)
    for gn in ('CoarseDense', 'CoarseConv'):
        a, b = net.groups[gn].grad * (1.0 / world), big.groups[gn].grad
        ok &= bool(((a - b).norm() / b.norm()) < 3e-2)
        if not ok:
            print('concat-batch check failed', gn, float((a - b).norm() / b.norm()), flush=True)
    flag = torch.tensor([int(ok)], device='cuda' if dist.get_backend() == 'nccl' else 'cpu')
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        open(out_path, 'w').write(str(int(flag.item())))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main(sys.argv[1])
