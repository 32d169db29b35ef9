"""Rank body of tests/test_dp_gloo.py::test_eight_ranks_* (gloo, CPU, no oracle forward: the gradients are synthetic and
exactly summable, so N ranks must equal ONE rank bit for bit whatever order the collective adds in).  The dense bucket
as the product sends it under the reference's frozen optimizer (ann3depth_amd/dp.py, models.MSDNReplica._dense_buckets):
pieces in production order, in-place reduce-scatter, ApplyAdam of the rank's own slice, m gathered on demand."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ann3depth_amd import dp            # noqa: E402
from oracle.tf13_ops import AdamTF1     # noqa: E402


def main(out_path):
    rank, _, world = dp.init_from_env('gloo')
    red = dp.GradReducer()
    assert (red.rank, red.world_size) == (rank, world)
    # the reducer's own check of the aliasing collectives (what the first RCCL job runs before it trusts reduce-scatter):
    # passes on gloo; a reducer whose reduce-scatter returns a wrong sum on ONE rank makes EVERY rank fall back
    assert red.inplace_ok(torch.device('cpu')) and red.inplace_ok(torch.device('cpu'))      # (cached)

    class Broken(dp.GradReducer):
        def reduce_scatter(self, flat):
            work, self.own = super().reduce_scatter(flat)
            return work, self.own

        def wait(self, work):
            super().wait(work)
            if self.rank == world - 1:
                self.own[3] += 1
    assert not Broken().inplace_ok(torch.device('cpu'))

    class Refusing(dp.GradReducer):
        def all_gather(self, flat):
            raise RuntimeError('input aliases output')
    assert not Refusing().inplace_ok(torch.device('cpu'))
    numel = 1_000_003                                       # not a multiple of anything: the bucket is padded
    q = world * 64
    padded = -(-numel // q) * q
    cut = -(-(padded // 3) // q) * q
    pieces = [(cut, padded), (0, cut)]                      # production order: the tail leaves first
    m_sharded = np.zeros(padded, np.float32)
    one, mine = AdamTF1(0.1, 0.9, 1.0), AdamTF1(0.1, 0.9, 1.0)
    var_one = {'w': np.zeros(padded, np.float32)}
    detached = dp.DetachedReducer(world, rank)
    for step in range(3):
        # multiples of 2^-6 below 2^10: any order of adding eight of them is exact in float32
        rng = np.random.default_rng(1000 * step + rank)
        gl = torch.from_numpy((rng.integers(-2 ** 15, 2 ** 15, padded) / 64.0).astype(np.float32))
        gl[numel:] = 0
        total = gl.clone()
        torch.distributed.all_reduce(total)
        ref = sum(torch.from_numpy((np.random.default_rng(1000 * step + r).integers(-2 ** 15, 2 ** 15, padded) / 64.0)
                                   .astype(np.float32)) for r in range(world))
        ref[numel:] = 0
        assert torch.equal(total, ref)                       # exact whatever the ring's order
        handles = [(red.reduce_scatter(gl[a:b]), a, b) for a, b in pieces]
        for (work, own), a, b in handles:
            red.wait(work)
            n = (b - a) // world
            lo = a + rank * n
            assert own.data_ptr() == gl[lo:lo + n].data_ptr() and torch.equal(own, total[lo:lo + n])
            _, det = detached.reduce_scatter(gl[a:b])
            assert det.data_ptr() == own.data_ptr() and det.numel() == n       # the one-GPU stand-in cuts the same slices
            var = {'s': np.zeros(n, np.float32)}
            mine.m['s'], mine.v['s'] = m_sharded[lo:lo + n], np.zeros(n, np.float32)
            p1, p2 = mine.beta1_power, mine.beta2_power
            mine.apply(var, {'s': own.numpy() * np.float32(1.0 / world)})
            mine.beta1_power, mine.beta2_power = p1, p2
            assert not var['s'].any()                        # alpha == 0: the weights do not move
        mine.beta1_power, mine.beta2_power = mine.beta1_power * mine.beta1, mine.beta2_power * mine.beta2
        assert red.pending == []
        one.apply(var_one, {'w': total.numpy() * np.float32(1.0 / world)})
    gathered = torch.from_numpy(m_sharded.copy())
    for a, b in pieces:
        red.all_gather(gathered[a:b])
    assert np.array_equal(gathered.numpy(), one.m['w']) and float(np.abs(one.m['w']).max()) > 0
    assert not gathered.numpy()[numel:].any()
    assert red.agree_all([rank, world - rank, 0]) == [world - 1, world, 0]
    flag = torch.tensor([1.0 if rank == world - 2 else 0.0])
    red.wait(red.any(flag))
    assert float(flag) == 1.0                                # one rank's poisoned slice reaches everybody
    if rank == 0:
        open(out_path, 'w').write(f'{world} ok\n')
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main(sys.argv[1])
