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
"""
import os

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, group=None):
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.pending = []

    def start(self, flat_grad):
        """Begin the all-reduce of one bucket; returns its handle (see wait())."""
        work = dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
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

    def broadcast(self, tensor, src=0):
        dist.broadcast(tensor, src, group=self.group)

    def agree(self, value):
        """MAX of an integer over all ranks, on the host (a gloo control group beside the RCCL data group): the collective
        stop decision of the training loop.  Every step queues synchronous gradient all-reduces, so a rank that left the
        loop on its own (its signal handler fired, its input ran dry) would leave the others waiting in step k+1's
        collective until the watchdog kills them; with this, all ranks leave after the same global step."""
        if self.world_size == 1:
            return int(value)
        if not hasattr(self, '_control'):
            backend = dist.get_backend(self.group)
            self._control = self.group if backend == 'gloo' else dist.new_group(backend='gloo')
        t = torch.tensor([int(value)], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self._control)
        return int(t.item())


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
