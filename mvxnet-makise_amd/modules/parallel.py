"""Data-parallel gradient exchange: one process per GPU, frames sharded across ranks, ONE
all-reduce (RCCL over xGMI, `nccl` backend; `gloo` on CPU for tests) of a flat fp32 gradient
bucket per step.  The reference has no distributed code; this is the
MI355X-native addition of SURVEY section 8e.  Ring all-reduce is per-link bound on xGMI, so the
whole trainable gradient (<= 29.5 MB) goes out as a single large bucket."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun).  Returns
    (rank, world, local_rank); a plain single-process run returns (0, 1, 0)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if 'MASTER_PORT' not in os.environ:
            # a fixed default port collides as soon as two jobs share a host: the launcher (torchrun) must name it
            raise RuntimeError('WORLD_SIZE > 1 needs MASTER_PORT (launch with torchrun / torch.distributed.run)')
        if backend is None:
            backend = os.environ.get('MVX_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_frames(n_frames_total, rank, world):
    """Frames {i : i mod world == rank}."""
    return list(range(rank, n_frames_total, world))


def global_count(n_local, device=None):
    """Sum of a per-rank count over all ranks (at least 1): the divisor of a step whose ranks contributed different
    numbers of frames (a short last chunk of an epoch, frames without voxels)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dev = device if (device is not None and dist.get_backend() == 'nccl') else torch.device('cpu')
        t = torch.tensor([float(n_local)], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        n_local = int(round(float(t)))
    return max(1, int(n_local))


class GradBucket:
    """Flat view over the gradients of the trainable parameters that took part in the step."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device('cpu')
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:                 # parameters' .grad become views of the bucket
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero(self):
        self.flat.zero_()

    def check_views(self):
        """Every parameter's .grad must still be its view of the flat buffer: ``zero_grad(set_to_none=True)`` or an
        assignment to .grad detaches it silently, and the all-reduce would then exchange stale memory."""
        base, end = self.flat.data_ptr(), self.flat.data_ptr() + 4 * self.flat.numel()
        for p in self.params:
            if p.grad is None or not (base <= p.grad.data_ptr() < end):
                raise RuntimeError('a parameter gradient is no longer a view of the GradBucket (use bucket.zero(), or '
                                   'optimizer.zero_grad(set_to_none=False))')

    def all_reduce_mean(self, frames_total):
        """Sum over ranks, then divide by the global number of frames."""
        self.check_views()
        if dist.is_initialized():             # also with one rank: the collective path is the same code at every world size
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        self.flat.mul_(1.0 / float(frames_total))


def assert_replicas_in_sync(params):
    """Every rank must hold bit-identical parameters after an optimizer step on the all-reduced gradient: compares a
    (sum, sum of squares) fingerprint in float64 across ranks; raises on divergence.  No-op in a single process."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return
    flat = torch.cat([p.detach().reshape(-1).double() for p in params])
    mine = torch.stack([flat.sum(), (flat * flat).sum()])
    all_ = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(all_, mine)
    if not all(torch.equal(all_[0], a) for a in all_):
        raise RuntimeError('data-parallel replicas diverged: %s' % [a.tolist() for a in all_])
