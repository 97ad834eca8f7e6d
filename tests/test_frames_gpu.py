"""Frame sets (modules/frames.py: every layer ONCE for all frames of a step) against the per-frame executor
(modules/tape.py) on the same frames: middle maps and every parameter gradient."""
import numpy as np
import pytest
import torch

import mvx_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def rel_err(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.fixture
def small_cfg():
    import modules.config as cfg
    old, old_r = list(cfg.config['voxelshape']), list(cfg.config['velorange'])
    cfg.config['voxelshape'] = [16, 24, 10]
    cfg.config['velorange'] = [0.0, -2.4, -3.0, 3.2, 2.4, 1.0]
    cfg.config['voxelsize'] = [0.2, 0.2, 0.4]
    yield cfg
    cfg.config['voxelshape'], cfg.config['velorange'] = old, old_r
    cfg.config['voxelsize'] = [(old_r[k + 3] - old_r[k]) / old[k] for k in range(3)]


def _small_batch(golden, B, with_empty=False):
    from modules.pipeline import FrameBatch
    g = golden('mvxnet_small')
    gp = golden('group_small')
    gen = torch.Generator().manual_seed(5)
    frames, n = [], []
    base = torch.from_numpy(gp['pcd'].copy())
    P = base.shape[0]
    for k in range(B):
        pts = base.clone()
        if k % 2:
            pts = pts.flip(0)
        pts[:, :3] += (torch.rand((P, 3), generator=gen) - 0.5) * 0.05 * k
        lo = torch.tensor([0.0, -2.4, -3.0]) + 1e-3
        hi = torch.tensor([3.2, 2.4, 1.0]) - 1e-3
        pts[:, :3] = torch.minimum(torch.maximum(pts[:, :3], lo), hi)
        pts[:, 4] = torch.rand(P, generator=gen) * 369
        pts[:, 5] = torch.rand(P, generator=gen) * 1223
        frames.append(pts)
        n.append(P - 37 * k)                       # different live point counts per frame
    if with_empty:
        n[1] = 0
    fpn = [[(torch.from_numpy(g[k]) * (1.0 + 0.1 * f))[None].to(DEV) for k in ('f0', 'f1', 'f2')] for f in range(B)]
    perm = torch.stack([torch.from_numpy(gp['perm'])] * B)
    batch = FrameBatch(torch.stack(frames).to(DEV).contiguous(), perm.to(DEV).contiguous(),
                       torch.tensor(n, dtype=torch.int32, device=DEV), fpn)
    return batch, torch.from_numpy(g['G'])[None].to(DEV)


@pytest.mark.parametrize('B,with_empty', [(1, False), (3, False), (4, True)])
def test_frame_set_equals_per_frame_execution(golden, small_cfg, B, with_empty):
    from MVXNet import MVXNet
    from modules import parallel
    from modules.pipeline import train_step_frame_set, train_step_frames
    torch.manual_seed(3)
    model = MVXNet().to(DEV)
    batch, G = _small_batch(golden, B, with_empty)
    # the permutation must be a permutation of the LIVE points of each frame
    for f in range(B):
        nlive = int(batch.n_points[f])
        batch.perms[f, :nlive] = torch.randperm(nlive, generator=torch.Generator().manual_seed(f)).to(DEV)
    hot = [(k, p) for k, p in model.named_parameters() if p.requires_grad and '.rpn.' not in k]
    bucket = parallel.GradBucket([p for _, p in hot])
    imsize = [370.0, 1224.0]
    bucket.zero()
    mids_ref = []
    nv_ref, st = train_step_frames(model, batch, G, imsize, keep_mid=mids_ref)
    torch.cuda.synchronize()
    ref = {k: p.grad.clone() for k, p in hot}
    bucket.zero()
    mids = []
    nv, st2 = train_step_frame_set(model, batch, G, imsize, keep_mid=mids)
    torch.cuda.synchronize()
    assert int(torch.stack([s.reshape(()) for s in st2]).max()) == 0
    assert list(nv) == list(nv_ref)
    assert len(mids) == len(mids_ref)
    for a, b in zip(mids, mids_ref):
        assert rel_err(a, b) < 2e-5
    for k, p in hot:
        assert rel_err(p.grad, ref[k]) < 2e-4, k
    # per-frame upstream gradients: (B,128,H,W) instead of one shared map
    if not with_empty:
        Gb = torch.stack([G[0] * (1.0 + 0.5 * f) for f in range(B)])
        bucket.zero()
        train_step_frame_set(model, batch, Gb, imsize)
        torch.cuda.synchronize()
        got = {k: p.grad.clone() for k, p in hot}
        bucket.zero()
        from modules.pipeline import FrameBatch
        for f in range(B):
            one = FrameBatch(batch.points6[f:f + 1], batch.perms[f:f + 1], batch.n_points[f:f + 1], [batch.fpn_levels[f]])
            train_step_frames(model, one, Gb[f:f + 1], imsize)
        torch.cuda.synchronize()
        for k, p in hot:
            assert rel_err(got[k], p.grad) < 2e-4, k


def test_frame_set_bf16x3_matches_per_frame_bf16x3_and_f32(golden, small_cfg):
    """convmath: bf16x3 on the frame-set executor (conv2 / conv3 forward, dgrad, wgrad on the split-MFMA kernels with a frame
    dimension): the same arithmetic as the per-frame executor in that mode (maps to summation order), and the maps
    within the split's 2e-5 of the exact-f32 frame set."""
    from MVXNet import MVXNet
    from modules import parallel
    from modules.pipeline import train_step_frame_set, train_step_frames
    torch.manual_seed(3)
    model = MVXNet().to(DEV)
    B = 3
    batch, G = _small_batch(golden, B)
    for f in range(B):
        nlive = int(batch.n_points[f])
        batch.perms[f, :nlive] = torch.randperm(nlive, generator=torch.Generator().manual_seed(f)).to(DEV)
    hot = [(k, p) for k, p in model.named_parameters() if p.requires_grad and '.rpn.' not in k]
    bucket = parallel.GradBucket([p for _, p in hot])
    imsize = [370.0, 1224.0]
    old = small_cfg.config.get('convmath', 'f32')
    small_cfg.config['convmath'] = 'f32'
    try:
        bucket.zero()
        mids_f32 = []
        train_step_frame_set(model, batch, G, imsize, keep_mid=mids_f32)
        torch.cuda.synchronize()
    finally:
        small_cfg.config['convmath'] = old
    small_cfg.config['convmath'] = 'bf16x3'
    try:
        bucket.zero()
        mids_ref = []
        train_step_frames(model, batch, G, imsize, keep_mid=mids_ref)
        torch.cuda.synchronize()
        ref = {k: p.grad.clone() for k, p in hot}
        bucket.zero()
        mids = []
        _, st = train_step_frame_set(model, batch, G, imsize, keep_mid=mids)
        torch.cuda.synchronize()
    finally:
        small_cfg.config['convmath'] = old
    assert int(torch.stack([s.reshape(()) for s in st]).max()) == 0
    for a, b, c in zip(mids, mids_ref, mids_f32):
        assert rel_err(a, b) < 2e-5                      # same split arithmetic, different launch shape
        assert rel_err(a, c) < 1e-4                      # against the exact-f32 mode
    for k, p in hot:
        assert rel_err(p.grad, ref[k]) < 2e-3, k


@pytest.mark.parametrize('B,with_empty', [(4, False), (4, True), (3, False)])
def test_frame_set_lanes_equal_one_frame_set(golden, small_cfg, B, with_empty):
    """MVX_SET_LANES = 2: the frames of a step as TWO frame sets on two streams (modules/pipeline.py) -- per-frame BatchNorm
    statistics, so every frame's map is that of the single frame set; the parameter gradients (second lane accumulated in
    its own buffer, added once) equal the single set's up to the order of the sums over the frames.  Repeated with the next
    batch prepared on the preparation stream, as bench.py runs it."""
    import modules.pipeline as pl
    from MVXNet import MVXNet
    from modules import parallel
    torch.manual_seed(3)
    model = MVXNet().to(DEV)
    batch, G = _small_batch(golden, B, with_empty)
    for f in range(B):
        nlive = int(batch.n_points[f])
        if nlive:
            batch.perms[f, :nlive] = torch.randperm(nlive, generator=torch.Generator().manual_seed(f)).to(DEV)
    hot = [(k, p) for k, p in model.named_parameters() if p.requires_grad and '.rpn.' not in k]
    bucket = parallel.GradBucket([p for _, p in hot])
    imsize = [370.0, 1224.0]
    Gb = torch.stack([G[0] * (1.0 + 0.5 * f) for f in range(B)])
    bucket.zero()
    mids_ref = []
    nv_ref, _ = pl.train_step_frame_set(model, batch, Gb, imsize, keep_mid=mids_ref)
    torch.cuda.synchronize()
    ref = bucket.flat.clone()
    old = pl.SET_LANES
    pl.SET_LANES = 2
    try:
        ready = None
        for rep in range(2):
            bucket.zero()
            mids = []
            nv, st, ready = pl.train_step_frame_set(model, batch, Gb, imsize, ready=ready, prepare_next=batch, keep_mid=mids)
            torch.cuda.synchronize()
            assert int(torch.stack([s.reshape(()) for s in st]).max()) == 0
            assert list(nv) == list(nv_ref) and len(mids) == len(mids_ref)
            for a, b in zip(mids, mids_ref):
                assert rel_err(a, b) < 1e-6
            assert rel_err(bucket.flat, ref) < 2e-4
    finally:
        pl.SET_LANES = old


def test_batch_prepared_a_step_ahead_with_its_fpn_sampling_gives_the_same_step(golden, small_cfg):
    """The next batch is voxelized, mapped AND its FPN features sampled on the preparation stream
    (pipeline.prepare_frame_set(sample=...), frames.sample_rows): the step that consumes it must equal the step that prepares
    its batch itself -- maps bit-identical, parameter gradients up to the order of the f64 atomics."""
    import modules.pipeline as pl
    from MVXNet import MVXNet
    from modules import parallel
    torch.manual_seed(5)
    model = MVXNet().to(DEV)
    B = 3
    batch, G = _small_batch(golden, B, False)
    hot = [(k, p) for k, p in model.named_parameters() if p.requires_grad and '.rpn.' not in k]
    bucket = parallel.GradBucket([p for _, p in hot])
    imsize = [370.0, 1224.0]
    Gb = torch.stack([G[0] * (1.0 + 0.5 * f) for f in range(B)])
    bucket.zero()
    mids_ref = []
    pl.train_step_frame_set(model, batch, Gb, imsize, keep_mid=mids_ref)
    torch.cuda.synchronize()
    ref = bucket.flat.clone()
    assert pl.PRESAMPLE and pl.PREP_STREAM
    ready = None
    for rep in range(3):                                  # rep 0 prepares its own batch, reps 1-2 consume a prepared one
        bucket.zero()
        mids = []
        _, st, ready = pl.train_step_frame_set(model, batch, Gb, imsize, ready=ready, prepare_next=batch, keep_mid=mids)
        assert ready[0].sampled is not None               # the prepared set carries its sampled rows
        torch.cuda.synchronize()
        assert int(torch.stack([s.reshape(()) for s in st]).max()) == 0
        for a, b in zip(mids, mids_ref):
            assert torch.equal(a, b)
        assert rel_err(bucket.flat, ref) < 1e-6


def test_sampler_forms_the_range_tag_of_the_image_features(golden, small_cfg):
    """fp16x3: frames.sample_rows hands the first fusion layer the sampled FPN features WITH their max |value| -- raised by the
    sampling kernel while it writes the rows (mvx_feature_sample_rows_frames, out_amax), not by a pass over them: the tag is
    bit-equal to the tensor's maximum."""
    from MVXNet import MVXNet
    from modules import _hip
    from modules.pipeline import prepare_frame_set
    torch.manual_seed(3)
    model = MVXNet().to(DEV)
    batch, _ = _small_batch(golden, 3)
    for f in range(3):
        nlive = int(batch.n_points[f])
        batch.perms[f, :nlive] = torch.randperm(nlive, generator=torch.Generator().manual_seed(f)).to(DEV)
    old = small_cfg.config.get('convmath', 'f32')
    try:
        for math, tagged in (('fp16x3', True), ('bf16x6', False)):
            small_cfg.config['convmath'] = math
            fs, live, counts, status = prepare_frame_set(batch, sample=(model.head, [370.0, 1224.0]))
            compact, st = fs.sampled
            torch.cuda.synchronize()
            am = _hip.amax_of(compact)
            assert (am is not None) == tagged
            if tagged:
                assert float(am) == float(compact.abs().max()) and float(am) > 0
    finally:
        small_cfg.config['convmath'] = old


def test_last_layer_writes_the_reference_layout_itself(golden, small_cfg, monkeypatch):
    """conv3's BatchNorm apply writing (F, C * D, H, W) directly (mvx_bn_apply_tiles_bev_frames) against the two-step form it
    replaces (channels-last output, then mvx_cl_to_bev_frames): the same middle maps bit for bit."""
    from MVXNet import MVXNet
    from modules import frames as fr
    from modules import parallel
    from modules.pipeline import train_step_frame_set
    torch.manual_seed(3)
    model = MVXNet().to(DEV)
    batch, G = _small_batch(golden, 3)
    for f in range(3):
        nlive = int(batch.n_points[f])
        batch.perms[f, :nlive] = torch.randperm(nlive, generator=torch.Generator().manual_seed(f)).to(DEV)
    bucket = parallel.GradBucket([p for k, p in model.named_parameters() if p.requires_grad and '.rpn.' not in k])
    out = {}
    for fused in (True, False):
        monkeypatch.setattr(fr, 'BEV_FUSED', fused)
        bucket.zero()
        mids = []
        train_step_frame_set(model, batch, G, [370.0, 1224.0], keep_mid=mids)
        torch.cuda.synchronize()
        out[fused] = (torch.cat(mids), bucket.flat.clone())
    assert torch.equal(out[True][0], out[False][0])
    assert torch.equal(out[True][1], out[False][1])
