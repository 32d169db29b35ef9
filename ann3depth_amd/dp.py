"""Synchronous data parallelism: one process per GPU, gradient buckets all-reduced with RCCL over xGMI.

Replaces the only data-parallel strategy of the reference, asynchronous between-graph replication through TF
parameter servers over gRPC (src/ann3depth.py:77-92).  Each optimizer group's gradients live in one
flat buffer (models.ParamGroup), so a bucket is one `all_reduce(sum)`; the 1/world mean is folded into the Adam
kernel's grad_scale.  `start()` is asynchronous: the process group's own stream waits for the kernels already
enqueued on the compute stream, so the dense-layer bucket (268 MB, produced first in backward) is reduced while
the conv backward kernels still run; `wait()` / `finish()` make the compute stream wait for one / all pending buckets.
MSDNReplica keeps the dense bucket in flight across the step boundary: its ApplyAdam is only due before the next
step's first dense layer, so the 268 MB reduction also overlaps the next step's conv forward.
On CPU (tests) the same code runs over the gloo backend.

The dense group under the reference's optimizer goes further (round 3).  AdamOptimizer(rate, 0.9, beta2 = 1)
(src/models.py:309) has alpha = 0: the weights never move and a variable's `m` slot is read by nothing but its own next
update.  No rank therefore needs the whole reduced gradient — `reduce_scatter()` leaves each rank with the sum of ONE
slice (half the xGMI bytes of an all-reduce: no all-gather phase), the rank updates `m` on that slice only (1/world of
the ApplyAdam traffic), and `all_gather()` reassembles `m` when somebody asks for it (checkpoints, tests).  This is the
parameter server's division of labour (src/ann3depth.py:77-92: every variable lives on ONE ps task which applies the
updates to it) without the server.
"""
import os

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, group=None, urgent_group=None):
        """urgent_group: a SECOND communicator over the same ranks for the small buckets the step waits for at its end (the
        conv group, 15 MB).  Collectives of one communicator run in the order they were started, so on the data group alone
        those 15 MB queue behind the dense group's 268 MB reduce-scatter that was started before them and is not due for a
        whole step.  OFF by default: measured with the one-GPU stand-in of round 5 (bench.py --standin-one-stream vs not) the
        queueing is not what a rank's step waits for — 3.17 ms on one stream, 3.21 on two, 2.89 without stand-ins: the cost is
        the CUs and the HBM bandwidth the exchange takes from the GEMMs beside it.  A3D_DP_URGENT_GROUP=1 creates the second
        communicator here (every rank constructs its reducer at the same point: new_group is collective) for the first job
        that can measure it over xGMI."""
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.pending = []
        if urgent_group is None and self.world_size > 1 and os.environ.get('A3D_DP_URGENT_GROUP', '0') == '1':
            urgent_group = dist.new_group(ranks=dist.get_process_group_ranks(group) if group is not None else None,
                                          backend=dist.get_backend(group))
        self.urgent_group = urgent_group

    def start(self, flat_grad, urgent=False):
        """Begin the all-reduce of one bucket; returns its handle (see wait()).  urgent: on the second communicator."""
        g = self.urgent_group if (urgent and self.urgent_group is not None) else self.group
        work = dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=g, async_op=True)
        self.pending.append(work)
        return work

    def wait(self, work):
        """Make the compute stream wait for ONE bucket (the others stay in flight)."""
        work.wait()
        self.pending = [w for w in self.pending if w is not work]

    def finish(self):
        for w in self.pending:
            w.wait()
        self.pending = []

    def reduce_scatter(self, flat):
        """Begin the in-place reduce-scatter of one bucket (numel a multiple of world_size); returns (handle, own): after
        wait(handle), `own` — this rank's 1/world slice of `flat` — holds the sum over ranks of that slice and the rest
        of `flat` is stale.  RCCL runs it in place (recvbuff == sendbuff + rank * count), so does gloo."""
        n = flat.numel() // self.world_size
        assert n * self.world_size == flat.numel(), 'bucket length must be a multiple of the world size'
        own = flat[self.rank * n:(self.rank + 1) * n]
        work = dist.reduce_scatter_tensor(own, flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.pending.append(work)
        return work, own

    def all_gather(self, flat):
        """In place and synchronous: every rank contributes its own 1/world slice of `flat`, all end with all slices."""
        n = flat.numel() // self.world_size
        assert n * self.world_size == flat.numel()
        dist.all_gather_into_tensor(flat, flat[self.rank * n:(self.rank + 1) * n], group=self.group)

    def inplace_ok(self, device):
        """Does THIS backend run the two aliasing collectives of the sharded dense path correctly?  reduce_scatter() writes
        its result into a slice of its own input and all_gather() reads its contribution out of its own output — the
        in-place forms NCCL / RCCL document (recvbuff == sendbuff + rank * count) and gloo accepts.  Every test of this
        repository ran them over gloo (no pipeline box has had two GPUs: ADVICE r3), so the first RCCL job checks them
        itself, once per reducer, on 4 KiB of known integers against a plain all-reduce: any wrong sum, or an exception,
        on any rank makes every rank answer False, and MSDNReplica then exchanges the dense bucket with all-reduce (the
        path without aliasing), saying so in its log line and in bench.py's JSON."""
        if getattr(self, '_inplace_ok', None) is None:
            ok = 1
            try:
                n = 256
                base = torch.arange(self.world_size * n, device=device, dtype=torch.float32)
                mine = base * (self.rank + 1)                        # rank r contributes (r + 1) * [0, 1, 2, ...]
                want = base * (self.world_size * (self.world_size + 1) // 2)
                work, own = self.reduce_scatter(mine)
                self.wait(work)
                lo = self.rank * n
                ok &= int(torch.equal(own, want[lo:lo + n])) and own.data_ptr() == mine[lo:lo + n].data_ptr()
                self.all_gather(mine)                                # every rank's slice of sums -> all of `want`
                ok &= int(torch.equal(mine, want))
            except Exception as e:                                   # noqa: BLE001 - a refusal of the aliasing form IS the answer
                print(f'dp: in-place collectives refused by backend {dist.get_backend(self.group)}: {e}', flush=True)
                ok = 0
            t = torch.tensor([ok], device=device, dtype=torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
            self._inplace_ok = bool(t.item())
        return self._inplace_ok

    def any(self, flag):
        """MAX of a small device tensor over the ranks (asynchronous; returns the handle)."""
        work = dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group, async_op=True)
        self.pending.append(work)
        return work

    def broadcast(self, tensor, src=0):
        dist.broadcast(tensor, src, group=self.group)

    def agree(self, value):
        """MAX of an integer over all ranks, on the host (a gloo control group beside the RCCL data group): the collective
        stop decision of the training loop.  Every step queues synchronous gradient all-reduces, so a rank that left the
        loop on its own (its signal handler fired, its input ran dry) would leave the others waiting in step k+1's
        collective until the watchdog kills them; with this, all ranks leave after the same global step."""
        return self.agree_all([value])[0]

    def agree_all(self, values):
        """agree() for several integers at once (element-wise MAX): one host collective per training step."""
        if self.world_size == 1:
            return [int(v) for v in values]
        if not hasattr(self, '_control'):
            backend = dist.get_backend(self.group)
            self._control = self.group if backend == 'gloo' else dist.new_group(backend='gloo')
        t = torch.tensor([int(v) for v in values], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self._control)
        return [int(v) for v in t.tolist()]


class DetachedReducer:
    """A reducer with the collectives taken out: a replica built with one runs exactly the kernels rank `rank` of a
    `world_size`-way data-parallel job runs — dense gradients materialised, ApplyAdam of its own slice only — and no
    communication.  bench.py times it on ONE GPU (`ms_per_step_dp_rank`): the local gradient stands in for the sum, so
    the numbers it trains are meaningless and nothing but a timer should look at them."""

    class _Done:
        def wait(self):
            return True

    def __init__(self, world_size, rank=0):
        self.world_size, self.rank, self.pending = world_size, rank, []

    def inplace_ok(self, device):
        return True

    def start(self, flat_grad, urgent=False):
        return self._Done()

    def reduce_scatter(self, flat):
        n = flat.numel() // self.world_size
        return self._Done(), flat[self.rank * n:(self.rank + 1) * n]

    def all_gather(self, flat):
        pass

    def any(self, flag):
        return self._Done()

    def wait(self, work):
        pass

    def finish(self):
        pass

    def broadcast(self, tensor, src=0):
        pass

    def agree(self, value):
        return int(value)

    def agree_all(self, values):
        return [int(v) for v in values]


class StandinReducer(DetachedReducer):
    """DetachedReducer plus the COST of the collectives: wherever a rank of a `world_size`-way job would
    start a collective, a stand-in kernel of the same size runs on a second HIP stream — a few workgroups reading the bucket
    and writing the rank's share of the sums at a capped rate (a3d_comm_standin) — ordered after the kernels that produced
    the bucket, and wait() makes the compute stream wait for it exactly as for an RCCL work handle.  One GPU, no
    communication: the numbers the replica trains are meaningless; what is measured is how much a reduce-scatter of 268 MB
    at xGMI-like rates slows the conv backward and the next forward it is meant to hide under.
    gbytes_per_s: the rate the launch is paced to (reads + writes); workgroups: how many CUs it occupies (RCCL uses 16-32)."""

    class _Work:
        def __init__(self, event):
            self.event = event

        def wait(self):
            torch.cuda.current_stream().wait_event(self.event)
            return True

    def __init__(self, world_size, rank=0, gbytes_per_s=200.0, workgroups=24, urgent_stream=True, inplace=True):
        super().__init__(world_size, rank)
        self.inplace = bool(inplace)            # False: answer inplace_ok() like a backend that refused the aliasing collectives —
        #                                         the replica then exchanges the dense bucket as all-reduces (the fallback path)
        self.gbytes_per_s, self.workgroups = float(gbytes_per_s), int(workgroups)
        self.streams = {}
        self.urgent_stream = urgent_stream      # False: every stand-in on ONE stream, as with a single communicator
        self.scratch = {}
        self.launched_bytes = 0

    def _standin(self, flat, write_fraction, lane):
        import ctypes
        from . import _lib
        from .ops import check
        if lane not in self.streams:
            self.streams[lane] = torch.cuda.Stream(device=flat.device)
        stream = self.streams[lane]
        nbytes = flat.numel() * flat.element_size() // 16 * 16
        wbytes = max(16, int(nbytes * write_fraction) // 16 * 16)
        if lane not in self.scratch or self.scratch[lane].numel() * 4 < wbytes:
            self.scratch[lane] = torch.empty(max(wbytes // 4, 1 << 20), device=flat.device)
        stream.wait_stream(torch.cuda.current_stream())               # the bucket is complete before the exchange reads it
        with torch.cuda.stream(stream):
            check(_lib.load().a3d_comm_standin(ctypes.c_void_p(flat.data_ptr()), nbytes, ctypes.c_void_p(self.scratch[lane].data_ptr()),
                                               wbytes, self.workgroups, self.gbytes_per_s,
                                               ctypes.c_void_p(stream.cuda_stream)), 'a3d_comm_standin')
            ev = torch.cuda.Event()
            ev.record(stream)
        self.launched_bytes += nbytes + wbytes
        return self._Work(ev)

    def inplace_ok(self, device):
        return self.inplace

    def start(self, flat_grad, urgent=False):
        # ring all-reduce: a rank reads and writes its bucket about twice (reduce-scatter + all-gather phases)
        return self._standin(flat_grad, 1.0, 'urgent' if (urgent and self.urgent_stream) else 'data')

    def reduce_scatter(self, flat):
        n = flat.numel() // self.world_size
        return self._standin(flat, 1.0 / self.world_size, 'data'), flat[self.rank * n:(self.rank + 1) * n]

    def wait(self, work):
        work.wait()

    def any(self, flag):
        return self._Done()


def init_from_env(backend=None):
    """Process-group setup from the launcher's RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* variables.
    Returns (rank, local_rank, world_size); world_size 1 needs no process group."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = os.environ.get('A3D_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if torch.cuda.is_available():                                    # 'nccl' is RCCL on ROCm
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world
