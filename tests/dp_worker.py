"""Rank body of tests/test_dp_gloo.py (world_size 2, gloo, CPU).  Exercises the product's process-group setup and
bucket reducer (ann3depth_amd/dp.py); gradients come from the numpy oracle because the HIP path needs a GPU."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ann3depth_amd import dp            # noqa: E402
from oracle import msdn as O            # noqa: E402


def main(out_path):
    rank, local_rank, world = dp.init_from_env('gloo')
    assert world == 2
    red = dp.GradReducer()
    assert (red.rank, red.world_size) == (rank, 2)
    params = O.init_params(3000)
    rng = np.random.default_rng(77)
    img = (rng.integers(0, 256, (2, 48, 64, 3)) / 255).astype(np.float32)
    dep = (rng.integers(0, 256, (2, 6, 8, 1)) / 255).astype(np.float32)
    keep = rng.random((2, 4096)) >= 0.5
    # replicas start from rank 0's weights
    w = torch.from_numpy(params['coarse/conv/conv2d_0/kernel'].copy())
    if rank == 1:
        w.zero_()
    red.broadcast(w)
    assert torch.equal(w, torch.from_numpy(params['coarse/conv/conv2d_0/kernel']))
    # each rank: its own sample (per-GPU batch 1); buckets: dense first (async), then conv, like MSDNReplica.step
    # activations are taken from ONE forward of both samples and sliced per rank: the loss gradient ~ 1/(o + 1e-8)
    # amplifies the last-bit differences BLAS produces between a batch-1 and a batch-2 forward (see test_gpu_msdn.py)
    a2 = O.forward(params, img, dep, keep)
    a = {k: (v[rank:rank + 1] if isinstance(v, np.ndarray) and v.ndim > 0 and v.shape[0] == 2 else v)
         for k, v in a2.items()}
    g = O.backward_coarse(params, a)
    dense = torch.cat([torch.from_numpy(g[n]).reshape(-1) for n in sorted(g) if n.startswith('coarse/dense')])
    conv = torch.cat([torch.from_numpy(g[n]).reshape(-1) for n in sorted(g) if n.startswith('coarse/conv')])
    red.start(dense)
    red.start(conv)
    red.finish()
    assert red.pending == []
    dense *= 1.0 / world
    conv *= 1.0 / world
    if rank == 0:
        g2 = O.backward_coarse(params, a2)                # the same two samples as ONE batch of 2
        d2 = np.concatenate([g2[n].reshape(-1) for n in sorted(g2) if n.startswith('coarse/dense')])
        c2 = np.concatenate([g2[n].reshape(-1) for n in sorted(g2) if n.startswith('coarse/conv')])
        rel = lambda x, y: float(np.linalg.norm(x.astype(np.float64) - y) / np.linalg.norm(y))
        with open(out_path, 'w') as f:
            f.write(f'{rel(dense.numpy(), d2)} {rel(conv.numpy(), c2)}\n')
    # ---- the dense bucket as the product sends it under the reference's frozen optimizer: reduce-scatter, ApplyAdam of
    # this rank's slice only, m gathered on demand (ann3depth_amd/dp.py, models.MSDNReplica._dense_buckets).  Two steps
    # with different gradients; N ranks must equal ONE rank applying the summed gradient — bit for bit on m.
    from oracle.tf13_ops import AdamTF1
    local = torch.cat([torch.from_numpy(g[n]).reshape(-1) for n in sorted(g) if n.startswith('coarse/dense')])
    q = world * 64
    padded = -(-local.numel() // q) * q
    cut = -(-(padded // 3) // q) * q
    pieces = [(cut, padded), (0, cut)]                        # production order: the tail leaves first
    m_sharded = np.zeros(padded, np.float32)
    one = AdamTF1(0.1, 0.9, 1.0)                              # the reference: AdamOptimizer(rate, 0.9, 1)
    var_one = {'w': np.zeros(padded, np.float32)}
    mine = AdamTF1(0.1, 0.9, 1.0)
    for step in range(2):
        gl = torch.nn.functional.pad(local * (1.0 + step) * (1 + rank), (0, padded - local.numel()))
        total = gl.clone()
        torch.distributed.all_reduce(total)                   # what ONE rank holding every sample's gradient would sum
        handles = [(red.reduce_scatter(gl[a:b]), a, b) for a, b in pieces]
        for (work, own), a, b in handles:
            red.wait(work)
            n = (b - a) // world
            lo = a + rank * n
            assert own.data_ptr() == gl[lo:lo + n].data_ptr()              # in place: my slice of the bucket itself
            assert torch.equal(own, total[lo:lo + n])
            var = {'s': np.zeros(n, np.float32)}
            mine.m['s'], mine.v['s'] = m_sharded[lo:lo + n], np.zeros(n, np.float32)
            p1, p2 = mine.beta1_power, mine.beta2_power
            mine.apply(var, {'s': own.numpy() * np.float32(1.0 / world)})
            mine.beta1_power, mine.beta2_power = p1, p2                      # powers advance once per step, not per slice
            assert not var['s'].any()                                        # alpha == 0: the weights do not move
        mine.beta1_power, mine.beta2_power = mine.beta1_power * mine.beta1, mine.beta2_power * mine.beta2
        assert red.pending == []
        one.apply(var_one, {'w': total.numpy() * np.float32(1.0 / world)})
    gathered = torch.from_numpy(m_sharded.copy())
    for a, b in pieces:
        red.all_gather(gathered[a:b])                          # every rank contributes its slice of every piece
    assert np.array_equal(gathered.numpy(), one.m['w']), 'sharded m differs from one rank on the summed gradient'
    assert float(np.abs(one.m['w']).max()) > 0
    assert red.agree_all([rank, 5 - rank, 0]) == [1, 5, 0]
    # ---- the second communicator for the buckets the step waits for at its end (A3D_DP_URGENT_GROUP=1; off by default,
    # DESIGN 5): a large bucket started first on the data group, the small one after it on the urgent group — the small one
    # can be waited for FIRST, both sums are right, and without the switch `urgent=True` falls back to the one group
    assert red.urgent_group is None
    os.environ['A3D_DP_URGENT_GROUP'] = '1'
    red2 = dp.GradReducer()                                   # (new_group is collective: both ranks construct it here)
    del os.environ['A3D_DP_URGENT_GROUP']
    assert red2.urgent_group is not None and red2.urgent_group is not red2.group
    big = torch.full((1 << 20,), float(rank + 1))
    small = torch.full((256,), float(10 * (rank + 1)))
    hb = red2.start(big)
    hs = red2.start(small, urgent=True)
    red2.wait(hs)
    assert red2.pending == [hb] and torch.equal(small, torch.full((256,), 30.0))
    red2.finish()
    assert red2.pending == [] and torch.equal(big, torch.full((1 << 20,), 3.0))
    again = torch.full((8,), float(rank))
    red.wait(red.start(again, urgent=True))                   # no second communicator: same result on the data group
    assert torch.equal(again, torch.full((8,), 1.0))
    # ---- the driver's per-step decision (ann3depth.Session._decide): an exhausted input on ONE rank stops both cleanly, a
    # reader FAILURE on one rank is raised on both (it used to look like a clean end of input), the chief's
    # checkpoint request reaches everybody
    import types
    from ann3depth_amd import ann3depth, data
    def session(end):
        rep = types.SimpleNamespace(reducer=red, global_step=0)
        op = types.SimpleNamespace(replica=rep, end=end, k=0)
        sess = ann3depth.Session(op, None, last_step=10, stop_at_signal=types.SimpleNamespace(signal_received=0),
                                 save_checkpoint_secs=0, save_summaries_steps=0, trace_every=0, logger=None, world=world)
        return sess, op
    sess, op = session((0, data.OutOfRangeError('dry')) if rank == 1 else None)
    assert sess._decide(op) is False and sess.stop and sess.sig.signal_received == 0           # both ranks stop, exit code 0
    sess, op = session((0, ValueError('tfrecord: corrupt payload')) if rank == 1 else None)
    try:
        sess._decide(op)
        raise AssertionError('a reader failure must not pass for a clean end of input')
    except ValueError as e:
        assert rank == 1 and 'corrupt' in str(e)
    except RuntimeError as e:
        assert rank == 0 and 'another replica' in str(e)
    sess, op = session(None)
    assert sess._decide(op, want_save=(rank == 0)) is True and not sess.stop                   # the chief's timer fired
    sess, op = session((5, data.OutOfRangeError('later')))                                    # not due yet: batch 5, k = 0
    assert sess._decide(op) is False and not sess.stop
    # collective stop decision (dp.GradReducer.agree): SIGUSR1 reaches rank 1 only, during its third "step"; both ranks
    # must leave the loop after the same step with the signal number as the agreed value
    import signal
    got = {'sig': 0}
    signal.signal(signal.SIGUSR1, lambda s, f: got.__setitem__('sig', s))
    steps = 0
    while True:
        t = torch.ones(4)
        torch.distributed.all_reduce(t)            # the step's gradient all-reduce: needs every rank
        steps += 1
        if rank == 1 and steps == 3:
            os.kill(os.getpid(), signal.SIGUSR1)
        agreed = red.agree(got['sig'])
        if agreed:
            break
        assert steps < 50
    assert agreed == signal.SIGUSR1 and steps == 3, (agreed, steps)
    with open(out_path + f'.stop{rank}', 'w') as f:
        f.write(f'{steps} {agreed}\n')
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main(sys.argv[1])
