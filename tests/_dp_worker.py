"""Worker of the world_size-2 gloo tests (CPU): gradient exchange of the data-parallel step.

mode toy   : two small parameters, every rank adds the gradients of its frames, one all-reduce -> mean over all frames.
mode chunks: (the divisor travels in the bucket's count slot: one collective per step) train_like.py's --mode fast step schedule on an ODD frame count (7 and 9 frames, 2 per rank per step): every rank
             runs the same number of steps, a rank without a frame in the short last chunk still joins the all-reduce, the
             divisor is the global number of contributing frames, every frame is used exactly once.
mode model : (world 2 and world 8) the REAL flat bucket over MVXNet's hot-path parameters (same construction as bench.py / train_like.py): every
             rank fills the gradients of its own frames with a frame-dependent pattern, all-reduces, and compares with
             the single-process sum over all frames computed locally; then one AdamW step must leave both ranks with
             bit-identical parameters.  The HIP kernels are not involved (no GPU here): this covers the N > 1 path's
             sharding, bucket layout, reduction and scaling."""
import os
import sys

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd'))
sys.argv = sys.argv[:2]
mode = sys.argv[1] if len(sys.argv) > 1 else 'toy'
sys.argv = sys.argv[:1]
from modules import parallel  # noqa: E402

rank, world, _ = parallel.init_from_env('gloo')
torch.manual_seed(0)
if mode == 'toy':
    params = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7))]
    bucket = parallel.GradBucket(params)
    frames_total = 6
    mine = parallel.shard_frames(frames_total, rank, world)
    bucket.zero()
    for f in mine:                                   # "backward" of frame f: gradient = (f+1) * ones
        loss = sum(((f + 1.0) * p).sum() for p in params)
        loss.backward()
    assert params[0].grad.data_ptr() == bucket.flat.data_ptr(), 'gradients must stay views of the flat bucket'
    bucket.all_reduce_mean(frames_total)
    expect = sum(f + 1.0 for f in range(frames_total)) / frames_total
    assert torch.allclose(bucket.flat, torch.full_like(bucket.flat, expect)), (bucket.flat, expect)
elif mode == 'chunks':
    import train_like
    B = 2
    for n_frames in (7, 9, 3):
        params = [torch.nn.Parameter(torch.zeros(4))]
        bucket = parallel.GradBucket(params)
        chunks = train_like.fast_chunks(n_frames, B, world)
        assert chunks[0][0] == 0 and chunks[-1][1] == n_frames and all(a[1] == b[0] for a, b in zip(chunks, chunks[1:]))
        seen = []
        for lo, hi in chunks:                        # the loop of train_like.train (--mode fast), gradients = frame id + 1
            mine = list(range(lo + rank, hi, world))
            bucket.zero()
            for f in mine:
                params[0].grad.add_(float(f + 1))
            # this rank's frame count rides in the bucket's count slot: ONE collective, the divisor applied on the device
            bucket.all_reduce_mean(frames_local=len(mine))
            want = sum(f + 1.0 for f in range(lo, hi)) / (hi - lo)
            assert torch.allclose(bucket.flat, torch.full_like(bucket.flat, want)), (bucket.flat, want)
            assert int(bucket._count[0]) == hi - lo, (bucket._count, lo, hi)
            assert parallel.global_count(len(mine)) == hi - lo
            seen.extend(mine)
        cnt = torch.tensor([float(len(seen)), float(len(chunks))])
        both = [torch.empty_like(cnt) for _ in range(world)]
        dist.all_gather(both, cnt)
        assert sum(int(b[0]) for b in both) == n_frames and len({int(b[1]) for b in both}) == 1
else:
    from MVXNet import MVXNet
    model = MVXNet()                                 # CPU parameters: only the bucket / optimizer logic runs here
    hot = [(k, p) for k, p in model.named_parameters() if p.requires_grad and '.rpn.' not in k]
    # the two-part layout of bench.py / train_like.py: the first fusion layer's weight (the step's last gradient) goes last
    late = model.head.fusion.fcn1.fc.weight
    bucket = parallel.GradBucket([p for _, p in hot], late=[late])
    assert bucket.flat.numel() == sum(p.numel() for _, p in hot) == 1169440
    assert bucket.n_early == 1169440 - late.numel() and late.grad.data_ptr() == bucket.flat[bucket.n_early:].data_ptr()
    opt = torch.optim.AdamW([p for _, p in hot], lr=1e-3, eps=1e-6)
    # the TWO-PART exchange on CPU tensors (on a GPU: early part on the communication stream, late part + count slot after the
    # join; here the same two collectives over the same split)
    bucket.two_part_on_cpu = True
    frames_total = max(8, 2 * world)                 # frames {i : i mod world == rank}: 4 per rank at world 2, 2 at world 8

    def frame_grad(f, p, j):
        g = torch.Generator().manual_seed(1000 * f + j)
        return torch.randn(p.shape, generator=g)

    mine = parallel.shard_frames(frames_total, rank, world)
    assert mine == list(range(rank, frames_total, world))
    bucket.zero()
    for f in mine:                                   # what the reduction kernels do: ADD into the existing .grad views
        for j, (_, p) in enumerate(hot):
            p.grad.add_(frame_grad(f, p, j))
    bucket.all_reduce_mean(frames_local=len(mine))    # count slot: 4 + 4 frames
    assert list(bucket.calls) == ['two-part'] and int(bucket._count[0]) == frames_total
    for j, (k, p) in enumerate(hot):
        want = sum(frame_grad(f, p, j) for f in range(frames_total)) / frames_total
        assert torch.allclose(p.grad, want, rtol=1e-5, atol=1e-6), k
    opt.step()
    flat_params = torch.cat([p.detach().reshape(-1) for _, p in hot])
    gathered = [torch.empty_like(flat_params) for _ in range(world)]
    dist.all_gather(gathered, flat_params)
    assert all(torch.equal(gathered[0], g) for g in gathered), 'replicas diverged after the optimizer step'
    # a stray zero_grad(set_to_none=True) would silently detach the parameters from the bucket: it must be caught
    hot[0][1].grad = None
    try:
        bucket.all_reduce_mean(frames_total)
        raise SystemExit('detached gradient not detected')
    except RuntimeError:
        pass
dist.barrier()
dist.destroy_process_group()
print('DP_OK rank', rank)
