"""Data-parallel gradient exchange: one process per GPU, frames sharded across ranks, ONE
all-reduce (RCCL over xGMI, `nccl` backend; `gloo` on CPU for tests) of a flat fp32 gradient
bucket per step.  The reference has no distributed code; this is the
MI355X-native addition of SURVEY section 8e.  Ring all-reduce is per-link bound on xGMI, so the
whole trainable gradient (<= 29.5 MB) goes out as a single large bucket."""
import collections
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
    """Flat view over the gradients of the trainable parameters that took part in the step.

    Layout of the buffer: [early parameters | late parameters | one count slot].  ``late``: parameters whose gradient is the
    LAST thing a step produces (the hot step ends with 0.7-0.9 ms in which only the weight gradient of the fusion MLP's first
    layer runs): with more than one rank the early part is all-reduced on a communication stream while
    that kernel is still running and the small late part after it (``all_reduce_mean`` does both; with no late parameters or
    one rank it is the single call it always was).  The count slot carries this rank's number of frames through the same
    collective, so that a step whose ranks contributed different numbers of frames (short last chunk, frames without
    voxels) is divided by the global count ON THE DEVICE -- no second collective, no host read (``frames_local=``)."""

    def __init__(self, params, late=()):
        late_ids = {id(p) for p in late}
        ps = [p for p in params if p.requires_grad]
        self.params = [p for p in ps if id(p) not in late_ids] + [p for p in ps if id(p) in late_ids]
        n = sum(p.numel() for p in self.params)
        self.n_early = sum(p.numel() for p in self.params if id(p) not in late_ids)
        dev = self.params[0].device if self.params else torch.device('cpu')
        self._buf = torch.zeros(n + 1, dtype=torch.float32, device=dev)
        self.flat = self._buf[:n]                # the gradients (what tests and optimizers look at)
        self._count = self._buf[n:]
        self.times = []                           # (start, end) event pairs of the collectives, when timing is on
        self.timing = False
        self.two_part_on_cpu = False              # tests: take the two-part branch on CPU tensors too (gloo)
        self.calls = collections.deque(maxlen=16)   # which branch the last exchanges took (tests)
        off = 0
        for p in self.params:                 # parameters' .grad become views of the bucket
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero(self):
        self._buf.zero_()

    def check_views(self):
        """Every parameter's .grad must still be its view of the flat buffer: ``zero_grad(set_to_none=True)`` or an
        assignment to .grad detaches it silently, and the all-reduce would then exchange stale memory."""
        base, end = self.flat.data_ptr(), self.flat.data_ptr() + 4 * self.flat.numel()
        for p in self.params:
            if p.grad is None or not (base <= p.grad.data_ptr() < end):
                raise RuntimeError('a parameter gradient is no longer a view of the GradBucket (use bucket.zero(), or '
                                   'optimizer.zero_grad(set_to_none=False))')

    def _timed(self, stream):
        if not (self.timing and self.flat.is_cuda):
            return None
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(stream)
        return s, e

    def all_reduce_mean(self, frames_total=None, frames_local=None):
        """Sum over ranks, then divide by the global number of frames: ``frames_total`` when the caller knows it (every rank
        ran the same number of frames), else ``frames_local`` = this rank's count, summed through the bucket's count slot and
        applied on the device (at least 1)."""
        self.check_views()
        assert (frames_total is None) != (frames_local is None)
        n = self.flat.numel()
        if frames_local is not None:
            self._count.fill_(float(frames_local))
        if dist.is_initialized():             # also with one rank: the collective path is the same code at every world size
            two = self.n_early < n and (self.flat.is_cuda or self.two_part_on_cpu)
            if two and not self.flat.is_cuda:
                # CPU tensors (gloo; tests): the same two collectives over the same [early | late + count] split, no streams
                w_early = dist.all_reduce(self._buf[:self.n_early], op=dist.ReduceOp.SUM, async_op=True)
                dist.all_reduce(self._buf[self.n_early:], op=dist.ReduceOp.SUM)
                w_early.wait()
                self.calls.append('two-part')
            elif two:
                from modules import _hip
                dev = self.flat.device
                main = torch.cuda.current_stream(dev)
                comm = _hip.comm_stream(dev)
                evs = _hip.tail_events(dev)          # (side stream before the tail kernel, main stream at the end of the backward)
                if evs is None:
                    comm.wait_stream(main)
                else:
                    comm.wait_event(evs[0])
                    comm.wait_event(evs[1])
                self._buf.record_stream(comm)
                with torch.cuda.stream(comm):
                    t_early = self._timed(comm)
                    w_early = dist.all_reduce(self._buf[:self.n_early], op=dist.ReduceOp.SUM, async_op=True)
                t = self._timed(main)
                dist.all_reduce(self._buf[self.n_early:], op=dist.ReduceOp.SUM)      # late parameters + the count slot
                if t:
                    t[1].record(main)
                    self.times.append(('late', ) + t)
                with torch.cuda.stream(comm):
                    # the collective runs on the process group's own stream: `comm` is behind it only after wait(), so the end
                    # event is recorded here and not right after the async call
                    w_early.wait()
                    if t_early:
                        t_early[1].record(comm)
                        self.times.append(('early', ) + t_early)
                main.wait_stream(comm)
                self.calls.append('two-part' if evs is not None else 'two-part, after the main stream')
            else:
                t = self._timed(torch.cuda.current_stream(self.flat.device)) if self.flat.is_cuda else None
                dist.all_reduce(self._buf if frames_local is not None else self.flat, op=dist.ReduceOp.SUM)
                if t:
                    t[1].record(torch.cuda.current_stream(self.flat.device))
                    self.times.append(('all', ) + t)
                self.calls.append('single')
        if frames_local is not None:
            self.flat.div_(self._count.clamp_min(1.0))
        else:
            self.flat.mul_(1.0 / float(frames_total))

    def collective_ms(self):
        """Average duration of the collectives recorded since ``timing`` was switched on, by kind (call after a sync)."""
        out = {}
        for kind, s, e in self.times:
            out.setdefault(kind, []).append(s.elapsed_time(e))
        return {k: sum(v) / len(v) for k, v in out.items()}


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
