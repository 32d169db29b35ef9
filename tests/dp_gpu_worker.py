"""Rank body of tests/test_gpu_dp.py and tests/test_gpu_rccl_multi.py.  On the one-GPU test box two ranks share cuda:0 and
reduce over gloo (RCCL refuses two ranks on one device; the reducer code path — async conv buckets all-reduced in
production order, the dense bucket reduce-scattered and deferred across the step boundary, ApplyAdam of the rank's own
slices with 1/world folded in, the m slot gathered on demand — is the same); on a box with at least two GPUs the same body
runs one rank per device over RCCL (A3D_DIST_BACKEND=nccl).

    dp_gpu_worker.py OUT          small batch: bucket contents, replicas bit-identical, == one replica on the big batch
    dp_gpu_worker.py OUT b32      BASELINE config 3's per-rank workload (B = 32 per rank, 480x640 stored), both trained
                                  phases, against the oracle on the concatenated 64-sample batch
    dp_gpu_worker.py OUT poison   a non-finite gradient on ONE rank: the owner's flag, the all-rank MAX, the late host read,
                                  _resync() and gather_state(), against one optimizer fed the summed gradients
    dp_gpu_worker.py OUT bf16s    BASELINE config 5's per-rank workload (B = 64 per rank, bf16 storage), both trained
                                  phases, against the oracle on the concatenated 128-sample batch"""
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
        scatter = net.dense_exchange == 'reduce_scatter'
        assert scatter == (os.environ.get('A3D_DP_DENSE', 'scatter') == 'scatter')      # (gloo passes the in-place self-check)
        ok &= sharded == (gstep == 0 and scatter)      # the reference's optimizer: the dense m slot lives in slices
        net.gather_state()                             # collective: every rank has all of m again
        ok &= not net._m_sharded
        for gn in gnames:
            local = solo.groups[gn].grad.clone()
            n = local.numel()
            local = torch.nn.functional.pad(local, (0, net.groups[gn].count - n))    # the rank-padded flat length
            dist.all_reduce(local)                                        # sum of both ranks' local gradients
            if gn == 'CoarseDense' and scatter:                           # reduce-scattered: only my slices hold the sum
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


import bf16s_tol      # noqa: E402


def main_bf16s(out_path):
    """BASELINE config 5's data-parallel rank: precision 'bf16s' (bf16 arithmetic, bf16 activations and weight copies in
    HBM, fp32 masters / gradients / Adam slots) WITH a reducer, B = 64 per rank at the stored 480x640, two ranks = one
    global batch of 128, both trained phases.  The weight gradients this mode exchanges are the fp32 buffers (the
    reduce-scatter / all-reduce of src/ann3depth.py:77-92's replacement is dtype-blind).  Checked:
      * depth maps of all 128 samples against the fp32 oracle on the concatenated batch at the mode's per-class tolerances
        (tests/bf16s_tol.py: 1.5 x the observed error; the values are printed);
      * the m slots every rank ends with (the dense group's after gather_state()) against (1 - beta1) x the oracle's
        fp32 backward of the activations the ranks stored, mean over all 128 samples, at the same per-class tolerances;
      * the ranks bit-identical, the weights untouched, the bf16 weight copies still equal to the masters."""
    from oracle import msdn as O
    rank, local_rank, world = dp.init_from_env()
    assert world == 2
    B = 64
    rng = np.random.default_rng(6464)
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
    ref_fwd = O.forward(params, img, dep, keep) if rank == 0 else None     # the fp32 oracle on all 128 samples
    for phase, gstep in ((1, 0), (2, models.SAMPLES_COARSE // B)):
        net = models.MSDNReplica(B, params=params, global_step=gstep, reducer=dp.GradReducer(), precision='bf16s')
        assert net.bf16s and net.c2.dtype == torch.bfloat16 and net.groups['CoarseDense'].grad.dtype == torch.float32
        out = net.step(ti, td, tk)
        assert out['phase'] == phase
        if phase == 1:
            assert net._deferred is not None and net._sharded_in_flight()
        net.gather_state()
        torch.cuda.synchronize()
        acts = {k: getattr(net, v).float().cpu() for k, v in names.items()}
        for k in ('c0', 'c1', 'f1'):
            acts[k] = net.prepool_equivalent(k).float().cpu()
        both = {}
        for k, t in acts.items():
            parts = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(parts, t.contiguous())
            both[k] = torch.cat(parts).numpy()
            del parts
        del acts
        if rank == 0:
            np.testing.assert_array_equal(both['images'], ref_fwd['images'])
            for k in ('coarse', 'fine'):
                e = rel(both[k], ref_fwd[k])
                print(f'bf16s 2 x B=64 phase {phase} depth/{k} rel-L2 {e:.3e}', flush=True)
                if not e < bf16s_tol.DEPTH[k]:
                    problems.append(f'phase {phase} {k} rel-L2 {e}')
            both['flat'] = both['c4'].reshape(world * B, -1)
            both['d0'] = both['drop']
            both['keep_mask'] = keep
            g_chain = (O.backward_coarse if phase == 1 else O.backward_fine)(params, both)   # mean over the 128 samples
            for n, gref in g_chain.items():
                e = rel(net.slot(n, 'm').cpu().numpy(), gref * omb1)     # m = (1 - beta1) * mean gradient
                print(f'bf16s 2 x B=64 phase {phase} m slot {n} rel-L2 {e:.3e}', flush=True)
                if not e < bf16s_tol.grad_tol(n):
                    problems.append(f'phase {phase} m slot {n} {e}')
                if not torch.equal(net.var(n).cpu(), torch.from_numpy(params[n])):
                    problems.append(f'phase {phase} {n} moved under beta2 = 1')
        del both
        for gn in (('CoarseConv', 'CoarseDense') if phase == 1 else ('FineA', 'FineB')):
            other = net.groups[gn].m.clone()
            dist.broadcast(other, 0)
            if not torch.equal(other, net.groups[gn].m):
                problems.append(f'phase {phase} rank {rank}: m of {gn} differs from rank 0')
            if not float(net.groups[gn].m.abs().max()) > 0:
                problems.append(f'phase {phase} rank {rank}: m of {gn} is all zero')
        for n, c in net.wcopy.items():                                       # frozen optimizer: copies == masters, still
            if not torch.equal(c, net.var(n + '/kernel').to(torch.bfloat16)):
                problems.append(f'phase {phase} rank {rank}: bf16 copy of {n} differs from its master')
        del net
    flag = torch.tensor([int(not problems)])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    for pr in problems:
        print(f'rank {rank}:', pr, flush=True)
    if rank == 0:
        open(out_path, 'w').write(str(int(flag.item())))
    dist.barrier()
    dist.destroy_process_group()


def main_poison(out_path):
    """the divergence-repair machinery of the rank-sharded optimizer state at replica level.  The reference's
    optimizer (beta2 = 1, src/models.py:309) changes a weight in exactly one case — a non-finite gradient turns it (and v)
    into NaN — and under sharding only the slice's owner sees that: _watch_poison (device flag, MAX over ranks),
    _poll_poison (host read two steps late), _resync (all-gather of var / v / m), gather_state (blocking drain).
    Rank 0 gets +inf in ONE element of its local dense gradient at step 1 — a padding element of the flat buffer, owned by
    rank 1, so the forward stays finite and only the machinery is exercised — and five more steps run.  Expected at the
    end, on every rank: var / v / m of CoarseDense bit for bit those of ONE ParamGroup that was fed the rank-summed
    gradient of every step through the same ApplyAdam kernel; exactly one _resync on every rank; the flag cleared."""
    rank, local_rank, world = dp.init_from_env()
    assert world == 2
    B, steps, bad_step = 2, 6, 1
    net = models.MSDNReplica(B, seed=3000, reducer=dp.GradReducer())
    gd = net.groups['CoarseDense']
    off, shp = gd.offsets['coarse/dense/dense_1/bias']
    idx = off + int(np.prod(shp)) + 3                      # padding behind the last tensor: no kernel reads or writes it
    assert idx < gd.count
    shapes = {n: s for n, (o, s) in gd.offsets.items()}
    ref = models.ParamGroup('ref', gd.lr, shapes, net.device, beta2=1.0, multiple=world)
    assert ref.count == gd.count
    ref.var.copy_(gd.var)
    (early,), late = net._dense_buckets()
    c = early[0]
    snap = torch.zeros_like(gd.grad)
    orig = net._bwd_filter

    def spying(name, x, dz):
        orig(name, x, dz)
        if name == 'coarse/dense/dense_1':                 # ... and then its bucket leaves (after_dense1)
            gd.grad[idx] = float('inf') if (rank == 0 and net.global_step == bad_step) else 0.0
            snap[c:] = gd.grad[c:]
        elif name == 'coarse/dense/dense_0':
            snap[:c] = gd.grad[:c]
    net._bwd_filter = spying
    resyncs = []
    orig_resync = net._resync
    net._resync = lambda: (resyncs.append(net.global_step), orig_resync())[1]
    cu = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dt).cuda()
    problems = []
    for step in range(steps):
        rng = np.random.default_rng(500 + 10 * step + rank)
        img = (rng.integers(0, 256, (B, 96, 128, 3)) / 255).astype(np.float32)
        dep = (rng.integers(0, 256, (B, 12, 16, 1)) / 255).astype(np.float32)
        out = net.step(cu(img), cu(dep), cu(rng.random((B, 4096)) >= 0.5, torch.uint8))
        assert out['phase'] == 1
        total = snap.clone()
        dist.all_reduce(total)                              # what ONE optimizer would have been handed this step
        ref.grad.copy_(total)
        ref.apply(1.0 / world)
        if not np.isfinite(float(out['coarse_loss'])):
            problems.append(f'step {step}: the forward went non-finite (the poisoned element is padding)')
    net.gather_state()
    torch.cuda.synchronize()
    bits = lambda t: t.view(torch.int32)
    for name, a, b in (('var', gd.var, ref.var), ('v', gd.v, ref.v), ('m', gd.m, ref.m)):
        if not torch.equal(bits(a), bits(b)):
            bad = (bits(a) != bits(b)).nonzero().flatten()
            problems.append(f'rank {rank}: {name} differs from the one-optimizer reference at {bad.numel()} elements, first {int(bad[0])}')
        other = a.clone()
        dist.broadcast(other, 0)
        if not torch.equal(bits(other), bits(a)):
            problems.append(f'rank {rank}: {name} differs from rank 0')
    if not (torch.isnan(gd.var[idx]) and torch.isnan(gd.v[idx])):
        problems.append(f'rank {rank}: the poisoned element did not turn var / v into NaN: {float(gd.var[idx])}, {float(gd.v[idx])}')
    n_bad = int((~torch.isfinite(gd.var)).sum()) + int((~torch.isfinite(gd.v)).sum())
    if n_bad != 2:
        problems.append(f'rank {rank}: {n_bad} non-finite var / v elements, expected the poisoned one of each')
    if len(resyncs) != 1 or not bad_step < resyncs[0] <= bad_step + 4:
        problems.append(f'rank {rank}: _resync ran at steps {resyncs}, expected once within four steps of step {bad_step}')
    if int(net._poison.item()) != 0 or net._m_sharded or net._poison_seen:
        problems.append(f'rank {rank}: state after gather_state(): flag {int(net._poison.item())}, m sharded {net._m_sharded}, '
                        f'{len(net._poison_seen)} unread flags')
    flag = torch.tensor([int(not problems)])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    for pr in problems:
        print(pr, flush=True)
    if rank == 0:
        open(out_path, 'w').write(str(int(flag.item())))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    {'b32': main_b32, 'bf16s': main_bf16s, 'poison': main_poison}.get(sys.argv[2] if len(sys.argv) > 2 else '', main)(sys.argv[1])
