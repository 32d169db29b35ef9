"""Rank body of tests/test_gpu_dp.py and tests/test_gpu_rccl_multi.py.  On the one-GPU test box two ranks share cuda:0 and
reduce over gloo (RCCL refuses two ranks on one device; the reducer code path — async conv buckets all-reduced in
production order, the dense bucket reduce-scattered and deferred across the step boundary, ApplyAdam of the rank's own
slices with 1/world folded in, the m slot gathered on demand — is the same); on a box with at least two GPUs the same body
runs one rank per device over RCCL (A3D_DIST_BACKEND=nccl).

    dp_gpu_worker.py OUT          small batch: bucket contents, replicas bit-identical, == one replica on the big batch
    dp_gpu_worker.py OUT b32      BASELINE config 3's per-rank workload (B = 32 per rank, 480x640 stored), both trained
                                  phases, against the oracle on the concatenated 64-sample batch"""
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
    assert world in (2, 4) and dist.get_backend() == os.environ.get('A3D_DIST_BACKEND', 'gloo')
    # two addends sum to the same bits in any order; four do not (all-reduce and reduce-scatter may add in different orders)
    same = torch.equal if world == 2 else (lambda a, b: bool((a - b).norm() <= 1e-6 * b.norm()))
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
        sharded = net._m_sharded
        ok &= sharded == (gstep == 0)                  # the reference's optimizer: the dense m slot lives in slices
        net.gather_state()                             # collective: every rank has all of m again
        ok &= not net._m_sharded
        for gn in gnames:
            local = solo.groups[gn].grad.clone()
            n = local.numel()
            local = torch.nn.functional.pad(local, (0, net.groups[gn].count - n))    # the rank-padded flat length
            dist.all_reduce(local)                                        # sum of both ranks' local gradients
            if gn == 'CoarseDense':                                       # reduce-scattered: only my slices hold the sum
                early, late = net._dense_buckets()
                for a, b in early + late:
                    lo, hi = net._my_slice(a, b)
                    ok &= bool(same(net.groups[gn].grad[lo:hi], local[lo:hi]))
            else:
                ok &= bool(same(net.groups[gn].grad, local))              # the bucket holds exactly that sum
            if not ok:
                print('bucket sum check failed', gn, gstep, flush=True)
            # ApplyAdam saw the mean gradient: m = 0 + (g * 1/world - 0) * (1 - beta1), the kernel's fp32 operations
            m_expect = (local * np.float32(1.0 / world)) * (np.float32(1) - np.float32(0.9))
            ok &= bool(same(net.groups[gn].m, m_expect))
            if not ok:
                print('m slot check failed', gn, gstep, flush=True)
            ok &= bool(torch.equal(net.groups[gn].var[:solo.groups[gn].count], solo.groups[gn].var))   # beta2 = 1: weights frozen
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
    net.gather_state()
    for gn in ('CoarseDense', 'CoarseConv'):
        a, b = net.groups[gn].m[:big.groups[gn].count], big.groups[gn].m
        ok &= bool(((a - b).norm() / b.norm()) < 3e-2)
        if not ok:
            print('concat-batch check failed', gn, float((a - b).norm() / b.norm()), flush=True)
    flag = torch.tensor([int(ok)], device='cuda' if dist.get_backend() == 'nccl' else 'cpu')
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        open(out_path, 'w').write(str(int(flag.item())))
    dist.barrier()
    dist.destroy_process_group()


def main_b32(out_path):
    """BASELINE config 3's per-rank workload: B = 32 per rank at the stored 480x640, two ranks = one global batch of 64.
    Forward of every rank and the m slots every rank ends with are checked against the oracle on the CONCATENATED batch
    (the reference's loss is a mean over the batch, src/models.py:272: N ranks must equal one rank on all samples); the
    step under test is the one a data-parallel rank runs — dense gradients materialised, reduce-scattered, ApplyAdam of
    the rank's own slices — not the fused single-GPU step."""
    from oracle import msdn as O
    rank, local_rank, world = dp.init_from_env()
    assert world == 2
    B = 32
    rng = np.random.default_rng(4321)
    img = (rng.integers(0, 256, (world * B, 480, 640, 3)) / 255).astype(np.float32)
    dep = (rng.integers(0, 256, (world * B, 480, 640, 1)) / 255).astype(np.float32)
    keep = rng.random((world * B, 4096)) >= 0.5
    sl = slice(rank * B, (rank + 1) * B)
    cu = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dt).cuda()
    ti, td, tk = cu(img[sl]), cu(dep[sl]), cu(keep[sl], torch.uint8)
    params = O.init_params(3000)
    rel = lambda x, y: float(np.linalg.norm(np.asarray(x, np.float64) - y) / max(np.linalg.norm(y), 1e-30))
    omb1 = np.float32(1) - np.float32(0.9)
    problems = []
    names = {'images': 'x', 'depths': 't', 'p0': 'p0', 'p1': 'p1', 'c2': 'c2', 'c3': 'c3', 'c4': 'c4', 'drop': 'drop',
             'coarse': 'coarse', 'cat': 'cat', 'f2': 'f2', 'fine': 'fine'}
    for phase, gstep in ((1, 0), (2, models.SAMPLES_COARSE // B)):
        net = models.MSDNReplica(B, params=params, global_step=gstep, reducer=dp.GradReducer())
        out = net.step(ti, td, tk)
        assert out['phase'] == phase
        if phase == 1:
            assert net._deferred is not None and net._sharded_in_flight()
        net.gather_state()
        torch.cuda.synchronize()
        # what each rank computed, on every rank (gloo moves host tensors)
        acts = {k: getattr(net, v).cpu() for k, v in names.items()}
        for k in ('c0', 'c1', 'f1'):
            acts[k] = net.prepool_equivalent(k).cpu()
        both = {}
        for k, t in acts.items():
            parts = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(parts, t.contiguous())
            both[k] = torch.cat(parts).numpy()
        if rank == 0:
            a = O.forward(params, img, dep, keep)                       # one replica's forward of all 64 samples
            np.testing.assert_array_equal(both['images'], a['images'])
            for k in ('coarse', 'fine'):
                e = rel(both[k], a[k])
                if not e < 1e-3:
                    problems.append(f'phase {phase} {k} rel-L2 {e}')
            for k in ('p0', 'p1', 'c2', 'c3', 'c4', 'drop', 'cat', 'f2'):
                e = rel(both[k], a[k])
                if not e < 1e-4:
                    problems.append(f'phase {phase} activation {k} {e}')
            both['flat'] = both['c4'].reshape(world * B, -1)
            both['d0'] = both['drop']
            both['keep_mask'] = keep
            g_chain = (O.backward_coarse if phase == 1 else O.backward_fine)(params, both)   # mean over the 64 samples
            for n, gref in g_chain.items():
                e = rel(net.slot(n, 'm').cpu().numpy(), gref * omb1)     # m = (1 - beta1) * mean gradient
                if not e < 1e-4:
                    problems.append(f'phase {phase} m slot {n} {e}')
                if not torch.equal(net.var(n).cpu(), torch.from_numpy(params[n])):
                    problems.append(f'phase {phase} {n} moved under beta2 = 1')
        # the ranks end with the same m, bit for bit
        for gn in (('CoarseConv', 'CoarseDense') if phase == 1 else ('FineA', 'FineB')):
            other = net.groups[gn].m.clone()
            dist.broadcast(other, 0)
            if not torch.equal(other, net.groups[gn].m):
                problems.append(f'phase {phase} rank {rank}: m of {gn} differs from rank 0')
        del net
    flag = torch.tensor([int(not problems)])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    for pr in problems:
        print(f'rank {rank}:', pr, flush=True)
    if rank == 0:
        open(out_path, 'w').write(str(int(flag.item())))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    (main_b32 if len(sys.argv) > 2 and sys.argv[2] == 'b32' else main)(sys.argv[1])
