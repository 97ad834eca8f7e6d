"""Frame pipeline of the hot path: resident point clouds -> GPU voxelizer -> MVXNet middle
(fusion sampling + fusion MLP + VFE stack + scatter + CML) forward and backward.

This reproduces the per-frame call sequence of the reference's training loop (train.py:31-44
CPU preprocessing, :113-131 forward, :161 backward) for a batch of independent frames: the
reference is strictly batch-1 (config.yml:18), so a batch of B frames is B forwards with
per-frame BatchNorm statistics and summed gradients."""
import torch

import modules.config as cfg
from modules import _hip
from modules import tape


import os as _os
ASYNC_WGRAD = _os.environ.get('MVX_ASYNC_WGRAD', '1') != '0'    # weight-gradient kernels on a second stream (see modules/_hip.py)
LANES = int(_os.environ.get('MVX_LANES', '2'))   # frames in flight: frame f runs on lane stream f % LANES (needs ASYNC_WGRAD for the
                        # single-writer gradient accumulation); 1 = all frames on the caller's stream
TAPE = _os.environ.get('MVX_TAPE', '1') != '0'     # frames run through modules/tape.py (no autograd engine); 0 = autograd
LANE_PRIORITY = int(_os.environ.get('MVX_LANE_PRIORITY', '0'))      # -1 = above the side (weight-gradient) stream
_LANE_STREAMS = {}


def lane_streams(device, n):
    key = device.index
    if key not in _LANE_STREAMS or len(_LANE_STREAMS[key]) < n:
        _LANE_STREAMS[key] = [torch.cuda.Stream(device=device, priority=LANE_PRIORITY) for _ in range(n)]
    return _LANE_STREAMS[key][:n]


class FrameBatch:
    """B frames resident in HBM with a fixed point capacity per frame.  Either the prepared clouds ``points6`` /
    ``n_points`` or the RAW clouds ``raw`` (B, capRaw, 4) + ``calib`` (dict of 4x4 matrices, modules/data/Load.py:24-41): the
    pipeline then runs crop + cropToSight + lidar2Img on the GPU first (cropdata.py:30-65, train.py:31-34)."""

    def __init__(self, points6, perms, n_points, fpn_levels, raw=None, calib=None, cap_points=None, n_raw=None):
        self.points6 = points6          # f32 (B, capP, 6)  x y z r row col
        self.perms = perms              # i32 (B, capP)
        self.n_points = n_points        # i32 (B,)
        self.fpn_levels = fpn_levels    # list over frames of [f0, f1, f2], each (1, C, H, W)
        self.raw, self.n_raw, self.calib = raw, n_raw, calib
        self.cap_points = cap_points if cap_points is not None else (points6.shape[1] if points6 is not None else None)
        self._mats = None
        self.created = None             # event after the last kernel / copy that WRITES this batch's tensors (mark_created)

    def mark_created(self):
        """Record that everything enqueued so far on the current stream produced this batch.  A batch that is handed to a
        step as ``prepare_next`` is read on the preparation stream, which does NOT wait for the training stream (it would
        queue behind the step just enqueued): it waits for this event instead.  Batches whose tensors are complete and
        synchronised (bench.py's resident batch) need no mark."""
        dev = self.device
        if dev.type == 'cuda':
            self.created = torch.cuda.Event()
            self.created.record(torch.cuda.current_stream(dev))
        return self

    @property
    def n_frames(self):
        return self.points6.shape[0] if self.points6 is not None else self.raw.shape[0]

    @property
    def device(self):
        return self.points6.device if self.points6 is not None else self.raw.device

    def prepared(self):
        """(points6, n_points): as given, or cropped + projected from the raw clouds on the current stream."""
        if self.raw is None:
            return self.points6, self.n_points
        if self._mats is None:
            import numpy as np
            c = self.calib
            r0, tr, p2 = (np.asarray(c[k], dtype=np.float64) for k in ('R0_rect', 'Tr_velo_to_cam', 'P2'))
            m32 = (torch.as_tensor(r0).float() @ torch.as_tensor(tr).float()).double().numpy()      # torch-path product (f32)
            self._mats = (r0 @ tr, p2, m32, torch.as_tensor(p2).float().double().numpy())
        m64, p64, m32, p32 = self._mats
        imsize_wh = (float(cfg.imsize[1]), float(cfg.imsize[0]))
        return _hip.crop_project(self.raw, self.n_raw, cfg.velorange, m64, p64, imsize_wh, m32, p32, self.cap_points)


def _wait_created(prep, nxt):
    """The preparation stream reads the next batch: wait for the event of its creation, if it has one (FrameBatch.mark_created)."""
    b = nxt[0] if isinstance(nxt, tuple) else nxt
    ev = getattr(b, 'created', None)
    if ev is not None:
        prep.wait_event(ev)


def voxelize_batch(batch, T=None, with_maps=False):
    """One batched voxelizer call for all frames; a single host read of the voxel counts.  ``with_maps``: also build every
    frame's compact-row map here (ONE more host read for all frames) and attach it to the frame's voxel tensor
    (``voxels._mvx_fs``): MVXNet.forward then finds it and needs no host read of its own, so a loop that calls the model
    one frame at a time (train.py:110-164) can enqueue frame k+1 while frame k is still running."""
    T = cfg.samplenum if T is None else T
    points6, n_points = batch.prepared()
    res = _hip.voxelize(points6, batch.perms, n_points, cfg.velorange[0:3], cfg.voxelsize, T, 9)
    counts = res.n_voxels.tolist()      # host sync: output sizes are data dependent
    frames = []
    for f, v in enumerate(counts):
        frames.append((res.voxels[f, :v].unsqueeze(0), res.coords[f, :v]))
    if with_maps:
        from modules import frames as fr
        sets = [fr.FrameSet(v[0], idx, [0, v.shape[1]], T) if v.shape[1] > 0 else None for v, idx in frames]
        live = [fs for fs in sets if fs is not None]
        if live:
            offs = torch.stack([fs.enqueue_map() for fs in live]).tolist()          # the padded rows are zeroed in place here
            for fs, off in zip(live, offs):
                fs.finish_map(off)
        for (v, _), fs in zip(frames, sets):
            if fs is not None:
                v._mvx_fs = fs
    return frames, res.status


def prepare_frames(batch, head):
    """Voxelize the batch and build every frame's compact-row map with TWO host reads per step (voxel
    counts, then real-row counts) instead of one per frame, so the host can enqueue a whole step
    ahead of the GPU."""
    frames, status = voxelize_batch(batch)
    # a frame without voxels (everything cropped away) takes no part in the step: the reference cannot run on one
    # either (BatchNorm over zero rows)
    maps = [head.compact_map(v) if v.shape[1] > 0 else None for v, _ in frames]
    live = [m for m in maps if m is not None]
    n_real = iter(torch.cat([m[2] for m in live]).tolist() if live else [])
    prepared = [(m[0], m[1], int(next(n_real))) if m is not None else None for m in maps]
    return frames, prepared, status


# ---- input pipelining: the voxelization of step k+1 runs on its own stream while step k computes --------------
# Output sizes are data dependent (voxel counts, real-row counts), so a step needs two host reads.  Done on the
# main stream they drain the whole queue twice per step; done one step ahead on a dedicated stream they only
# wait for a few small kernels that finished long ago.  A data loader handing over the next batch plays the same
# role in training (the reference prepares its batches on CPU workers, train.py:31-44).
_PREP_STREAMS = {}


def _prep_stream(device):
    if device.index not in _PREP_STREAMS:
        # high priority: the few small preparation kernels should not queue behind the step's long ones (the host
        # waits for their two results)
        _PREP_STREAMS[device.index] = torch.cuda.Stream(device=device, priority=int(_os.environ.get('MVX_PREP_PRIORITY', '-1')))
    return _PREP_STREAMS[device.index]


# End-of-step fence between the training stream and the preparation stream (FrameSet.hand_over(..., fenced=True)): a step
# function records an event on the training stream once everything that reads the prepared tensors is enqueued (after it joined
# the side stream), the tensors stay referenced until the step function returns, and every preparation first waits for the
# latest fence -- so a block of the preparation stream's pool is never rewritten before its last reader has run, without
# `record_stream`.  MVX_PREP_FENCE=0: the record_stream hand-over of rounds 2-4.
PREP_FENCE = _os.environ.get('MVX_PREP_FENCE', '1') != '0'
_FENCE = {}


def _fence_record(device):
    if PREP_FENCE and PREP_STREAM:
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        _FENCE[device.index] = ev


def _fence_wait(prep, device):
    ev = _FENCE.get(device.index)
    if PREP_FENCE and ev is not None:
        prep.wait_event(ev)


_PINNED = {}


def _pinned(tag, like, turn):
    """Reused page-locked host buffers (allocating one per step costs more than the step's syncs did)."""
    key = (tag, tuple(like.shape), like.dtype, turn)
    buf = _PINNED.get(key)
    if buf is None:
        buf = torch.empty(like.shape, dtype=like.dtype, pin_memory=True)
        _PINNED[key] = buf
    return buf


_TURN = [0]


def _spin(ev):
    """Wait for an event by polling: hipEventSynchronize parks the thread and wakes it late (measured: it alone made
    the pipelined step 2.5 ms slower than the unpipelined one)."""
    while not ev.query():
        pass


class PendingPrepare:
    __slots__ = ('batch', 'res', 'counts_host', 'ev_a', 'frames', 'maps', 'nreal_host', 'ev_b', 'turn')


def prepare_begin(batch, T=None):
    """Phase A (enqueue only): batched voxelization of ``batch`` on the preparation stream + async read of the counts."""
    T = cfg.samplenum if T is None else T
    dev = batch.points6.device
    prep = _prep_stream(dev)
    prep.wait_stream(torch.cuda.current_stream(dev))     # memory handed back by earlier steps is free by then
    h = PendingPrepare()
    h.batch = batch
    _TURN[0] ^= 1
    h.turn = _TURN[0]
    with torch.cuda.stream(prep):
        h.res = _hip.voxelize(batch.points6, batch.perms, batch.n_points, cfg.velorange[0:3], cfg.voxelsize, T, 9)
        h.counts_host = _pinned('counts', h.res.n_voxels, h.turn)
        h.counts_host.copy_(h.res.n_voxels, non_blocking=True)
        h.ev_a = torch.cuda.Event()
        h.ev_a.record(prep)
    h.frames = None
    return h


def prepare_mid(h, head):
    """Phase B (enqueue only; waits for phase A's counts, which are long there): compact-row maps of every frame."""
    _spin(h.ev_a)
    counts = h.counts_host.tolist()
    dev = h.batch.points6.device
    prep = _prep_stream(dev)
    with torch.cuda.stream(prep):
        h.frames = [(h.res.voxels[f, :v].unsqueeze(0), h.res.coords[f, :v]) for f, v in enumerate(counts)]
        h.maps = [head.compact_map(v) if v.shape[1] > 0 else None for v, _ in h.frames]
        live = [m[2] for m in h.maps if m is not None]
        nreal = torch.cat(live) if live else torch.zeros((0,), dtype=torch.int32, device=dev)
        h.nreal_host = _pinned('nreal', nreal, h.turn)
        h.nreal_host.copy_(nreal, non_blocking=True)
        h.ev_b = torch.cuda.Event()
        h.ev_b.record(prep)


def prepare_end(h, head):
    """Host side of a finished preparation: (frames, prepared, status, event the consumers must wait for)."""
    if h.frames is None:
        prepare_mid(h, head)
    _spin(h.ev_b)
    n_real = iter(h.nreal_host.tolist())
    prepared = [(m[0], m[1], int(next(n_real))) if m is not None else None for m in h.maps]
    return h.frames, prepared, h.res.status, h.ev_b


def train_step_frames(model, batch, grad_mid, imsize, ready=None, prepare_next=None, keep_mid=None):
    """Forward + backward of every frame of the batch through ``model.middle``; gradients
    accumulate in the parameters.  ``grad_mid`` is dL/d(middle output) (1,128,H,W), standing for
    the RPN + loss that follow the hot path.  ``ready``: result of ``prepare_end`` for this batch (input
    pipelining); None prepares it here.  ``prepare_next``: a batch to voxelize on the caller's stream right after this
    step's frames have been handed to the lane streams -- the caller's stream is idle then, so the two host reads only
    wait for those few kernels; the result is returned as a third value, to be passed as ``ready`` (with event None) next
    time.  ``keep_mid``: a list that receives every frame's middle map (tests).
    Returns (voxels per frame, list of device status words to be checked by the caller once per step)."""
    if ready is None:
        frames, prepared, status = prepare_frames(batch, model.head)
        ev_ready = None
    else:
        frames, prepared, status, ev_ready = ready
    statuses = [status]
    nvox = []
    # gradients are accumulated over the frames of the step anyway: let the reduction kernels add them
    # straight into the (pre-existing) .grad buffers, and clear all BatchNorm accumulators of a frame
    # with one fill
    old_sink, _hip.GRAD_SINK = _hip.GRAD_SINK, True
    old_async, _hip.ASYNC_WGRAD = _hip.ASYNC_WGRAD, ASYNC_WGRAD
    dev = batch.device
    main = torch.cuda.current_stream(dev)
    lanes = lane_streams(dev, LANES) if (LANES > 1 and ASYNC_WGRAD) else [main]
    if ev_ready is not None:
        main.wait_event(ev_ready)
    try:
        if len(lanes) > 1:
            model.prepack()
            for st in lanes:
                st.wait_stream(main)             # voxelization, maps, packed weights, zeroed gradients
        # flat gradient bucket (modules/parallel.py GradBucket) -> per-lane staging of the bias gradients
        some = next((p for p in model.parameters() if p.requires_grad and p.grad is not None), None)
        flat = some.grad._base if (some is not None and ASYNC_WGRAD) else None
        for f, (voxels, idx) in enumerate(frames):
            if prepared[f] is None or prepared[f][2] == 0:      # empty frame: nothing to learn from
                nvox.append(0)
                continue
            with torch.cuda.stream(lanes[f % len(lanes)]):
                _hip.arena_begin(dev)
                if flat is not None:
                    _hip.bias_stage_begin(dev, flat)
                if TAPE:
                    mid = tape.middle_train(model, voxels, batch.fpn_levels[f], idx, imsize, prepared[f], statuses, grad_mid)
                else:
                    mid = model.middle(voxels, batch.fpn_levels[f], idx, [None], imsize, prepared=prepared[f],
                                       status_sink=statuses)
                    mid.backward(grad_mid)
                if keep_mid is not None:
                    mid.record_stream(main)
                    keep_mid.append(mid.detach())
                if flat is not None:
                    _hip.bias_stage_flush(dev)
            nvox.append(voxels.shape[1])
        next_ready = None
        if prepare_next is not None:
            fr, pr, stt = prepare_frames(prepare_next, model.head)
            next_ready = (fr, pr, stt, None)
    finally:
        _hip.GRAD_SINK = old_sink
        _hip.ASYNC_WGRAD = old_async
        _hip.arena_end()
        _hip.bias_stage_end()
        for st in lanes:
            if st is not main:
                main.wait_stream(st)
        _hip.join_side_stream()          # the gradients are complete for whoever comes next on this stream
    if prepare_next is not None:
        return nvox, statuses, next_ready
    return nvox, statuses


# ---- frame sets: every layer ONCE for all frames of the step (modules/frames.py) --------------------------------------
BATCHED = _os.environ.get('MVX_FRAME_SETS', '1') != '0'


PREGRID = _os.environ.get('MVX_PREGRID', '1') != '0'          # ... and the coordinate-only bookkeeping of the CML (frames.grid_activity)


def prepare_frame_set(batch, T=None, sample=None, grid=None):
    """Voxelize the batch and build the frame set (voxels of all non-empty frames back to back + compact-row maps) with TWO
    host reads: the voxel counts, then the real-row offsets.  Returns (frame set or None, frame ids in it, voxel counts of
    every frame of the batch, status word).  ``sample`` = (fusion head, imsize): also sample the FPN features of the real
    rows here (frames.sample_rows) -- input preparation like the rest, no parameter involved."""
    from modules import frames as fr
    T = cfg.samplenum if T is None else T
    points6, n_points = batch.prepared()
    # the voxelizer writes the batch layout directly: voxels of all frames back to back, frame index in coords[:,0]
    voxels, coords, _, _, vox_off, status_v = _hip.voxelize_concat(points6, batch.perms, n_points, cfg.velorange[0:3],
                                                                   cfg.voxelsize, T, 9)
    offs = vox_off.tolist()                         # host read 1: output sizes are data dependent
    counts = [offs[f + 1] - offs[f] for f in range(batch.n_frames)]
    live = [f for f, v in enumerate(counts) if v > 0]
    if not live:
        return None, [], counts, status_v
    voxels, coords = voxels[:offs[-1]], coords[:offs[-1]]
    off = [0]
    for f in live:                                  # empty frames occupy no rows: the live frames stay back to back
        off.append(off[-1] + counts[f])
    fs = fr.FrameSet(voxels, coords, off, T)
    real_off = fs.enqueue_map().tolist()            # host read 2
    fs.finish_map(real_off)
    if sample is not None:
        fs.sampled = fr.sample_rows(sample[0], fs, [batch.fpn_levels[f] for f in live], sample[1])
    if grid is not None:
        fs.grid = fr.grid_activity(grid, fs)          # ``grid`` = the model (layer geometry only; no parameter is read)
    return fs, live, counts, status_v


# ... enqueued between this step's forward and backward instead of after the backward.  OFF: +0.6 % of the step (484.4 -> 487.5
# frames/s, the preparation no longer runs beside the step's last weight gradient), but the host is so far ahead that the
# preparation then lands beside conv2's forward gather, which goes from 0.58 to 0.90 ms (profiles/r05b_prep_early_*): the
# convolution -- the kernel the roofline is quoted on -- loses more than the tail gains once a step is not host-paced
PREP_EARLY = _os.environ.get('MVX_PREP_EARLY', '0') != '0'
PREP_STREAM = _os.environ.get('MVX_PREP_STREAM', '1') != '0'    # next batch prepared on its own stream (host reads return early)
PRESAMPLE = _os.environ.get('MVX_PRESAMPLE', '1') != '0'        # ... including the FPN feature sampling of its real rows (frames.sample_rows)
# Frame-set lanes: the frames of a step split into this many frame sets that run on their own streams, so that the small
# latency-bound kernels of one set (BatchNorm passes, VFE, list builders) execute beside the MFMA-bound kernels of the other.
# Every set keeps per-frame BatchNorm statistics, so the result per frame is unchanged; lane k > 0 accumulates its parameter
# gradients in a buffer of its own that is added to the bucket once per step (no two streams ever write one address).
SET_LANES = int(_os.environ.get('MVX_SET_LANES', '1'))
_LANE_FLAT = {}


def prepare_frame_sets(batch, lanes, T=None):
    """prepare_frame_set for ``lanes`` frame sets over contiguous groups of the batch's non-empty frames: ONE voxelizer call,
    one compact-row map per set, the same two host reads.  Returns ([(frame set, frame ids)], live ids, counts, status)."""
    from modules import frames as fr
    T = cfg.samplenum if T is None else T
    points6, n_points = batch.prepared()
    voxels, coords, _, _, vox_off, status_v = _hip.voxelize_concat(points6, batch.perms, n_points, cfg.velorange[0:3],
                                                                   cfg.voxelsize, T, 9)
    offs = vox_off.tolist()                         # host read 1
    counts = [offs[f + 1] - offs[f] for f in range(batch.n_frames)]
    live = [f for f, v in enumerate(counts) if v > 0]
    if not live:
        return [], [], counts, status_v
    lanes = max(1, min(lanes, len(live)))
    per = (len(live) + lanes - 1) // lanes
    sets = []
    for k in range(0, len(live), per):
        ids = live[k:k + per]
        off = [0]
        for f in ids:
            off.append(off[-1] + counts[f])
        # empty frames hold no voxels, so the voxels of a group of live frames are contiguous in the concatenated layout
        a = offs[ids[0]]
        sets.append((fr.FrameSet(voxels[a:a + off[-1]], coords[a:a + off[-1]], off, T), ids))
    devs = [fs.enqueue_map() for fs, _ in sets]
    for (fs, _), d in zip(sets, devs):
        fs.finish_map(d.tolist())                   # host read 2 (the later ones return at once)
    return sets, live, counts, status_v


def _lane_flat(flat, k):
    key = (flat.data_ptr(), flat.numel(), k)
    buf = _LANE_FLAT.get(key)
    if buf is None:
        buf = torch.zeros_like(flat)
        _LANE_FLAT[key] = buf
    return buf


def _train_step_frame_set_lanes(model, batch, grad_mid, imsize, ready, prepare_next, keep_mid):
    """train_step_frame_set with SET_LANES > 1 (see SET_LANES)."""
    from modules import frames as fr
    dev = batch.device
    main = torch.cuda.current_stream(dev)
    ev_ready = None
    if ready is None:
        ready = prepare_frame_sets(batch, SET_LANES)
    elif len(ready) == 5:
        ready, ev_ready = ready[:4], ready[4]
    sets, live, counts, status = ready
    if ev_ready is not None:
        main.wait_event(ev_ready)
        status.record_stream(main)
    statuses = [status]
    old_sink, _hip.GRAD_SINK = _hip.GRAD_SINK, True
    old_async, _hip.ASYNC_WGRAD = _hip.ASYNC_WGRAD, True
    next_ready = None
    streams = [main] + lane_streams(dev, max(0, len(sets) - 1))
    flat, done = None, False
    try:
        if sets:
            model.prepack()
            params = [p for p in model.parameters() if p.requires_grad and p.grad is not None]
            flat = params[0].grad._base
            if flat is None:
                raise _hip.X.MvxHipError('frame-set lanes need the flat gradient bucket (modules/parallel.py GradBucket)')
            lane_state = []
            for k, (fs, ids) in enumerate(sets):
                st = streams[k]
                if st is not main:
                    st.wait_stream(main)                        # packed weights, zeroed bucket, the preparation event
                    fs.hand_over(st)
                elif ev_ready is not None:
                    fs.hand_over(main)
                targets = None
                if k > 0:
                    lf = _lane_flat(flat, k)
                    targets = {}
                    for p in params:
                        o = (p.grad.data_ptr() - flat.data_ptr()) // 4
                        targets[id(p)] = lf[o:o + p.numel()].view_as(p)
                gm = grad_mid if grad_mid.shape[0] == 1 else grad_mid[ids]
                lane_state.append([st, fs, ids, targets, gm, None, None])
            with torch.no_grad():
                for ls in lane_state:                           # forward of every lane, then the backward of every lane
                    st, fs, ids, targets, gm, _, _ = ls
                    with torch.cuda.stream(st):
                        if targets is not None:
                            _lane_flat(flat, lane_state.index(ls)).zero_()
                        _hip.arena_begin(dev, doubles=1 << 21)
                        ls[5], ls[6] = fr.middle_forward(model, fs, [batch.fpn_levels[f] for f in ids], imsize, statuses)
                for ls in lane_state:
                    st, fs, ids, targets, gm, mid, saved = ls
                    with torch.cuda.stream(st), fr.grad_targets(targets):
                        if gm.device == dev and st is not main:
                            gm.record_stream(st)
                        fr.middle_backward(model, saved, gm)
                    ls[6] = None
            if keep_mid is not None:
                for ls in lane_state:
                    ls[5].record_stream(main)
                    for k in range(len(ls[2])):
                        keep_mid.append(ls[5][k:k + 1])
        if prepare_next is not None:
            if PREP_STREAM:
                prep = _prep_stream(dev)
                _wait_created(prep, prepare_next)
                with torch.cuda.stream(prep):
                    nr = prepare_frame_sets(prepare_next, SET_LANES)
                    ev = torch.cuda.Event()
                    ev.record(prep)
                next_ready = nr + (ev,)
            else:
                next_ready = prepare_frame_sets(prepare_next, SET_LANES)
        done = True
    finally:
        _hip.GRAD_SINK = old_sink
        _hip.ASYNC_WGRAD = old_async
        _hip.arena_end()
        for st in streams[1:]:
            main.wait_stream(st)
        _hip.join_side_stream()
        if done and sets and flat is not None:                  # success only: partial lane gradients never reach the bucket
            for k in range(1, len(sets)):                       # the other lanes' gradients join the bucket (main stream)
                flat.add_(_lane_flat(flat, k))
            if len(sets) > 1:
                _hip.drop_tail(dev)                             # written after the join: the early exchange waits for the main stream
    if prepare_next is not None:
        return counts, statuses, next_ready
    return counts, statuses


def train_step_frame_set(model, batch, grad_mid, imsize, ready=None, prepare_next=None, keep_mid=None):
    """Same contract as train_step_frames, executed through modules/frames.py: ONE launch per layer for all frames of the
    batch, weight-gradient kernels on the side stream.  ``grad_mid``: (1,128,H,W) for every frame or (B,128,H,W).
    ``prepare_next`` is voxelized and mapped on the preparation stream right after this step has been enqueued: its two
    host reads only wait for those few small kernels (which share the GPU with the step), so the host stays a step ahead
    of the GPU."""
    from modules import frames as fr
    if SET_LANES > 1:
        return _train_step_frame_set_lanes(model, batch, grad_mid, imsize, ready, prepare_next, keep_mid)
    dev = batch.device
    main = torch.cuda.current_stream(dev)
    ev_ready = None
    if ready is None:
        ready = prepare_frame_set(batch)
    elif len(ready) == 5:
        ready, ev_ready = ready[:4], ready[4]
    fs, live, counts, status = ready
    if ev_ready is not None:
        main.wait_event(ev_ready)
        status.record_stream(main)
        if fs is not None:
            fs.hand_over(main, fenced=PREP_FENCE)
    statuses = [status]
    old_sink, _hip.GRAD_SINK = _hip.GRAD_SINK, True
    next_ready = None
    try:
        if fs is not None:
            model.prepack()
            _hip.arena_begin(dev, doubles=1 << 21)
            gm = grad_mid if grad_mid.shape[0] == 1 or len(live) == batch.n_frames else grad_mid[live]
            with torch.no_grad():
                mid, saved = fr.middle_forward(model, fs, [batch.fpn_levels[f] for f in live], imsize, statuses)
                if prepare_next is not None and PREP_STREAM and PREP_EARLY:
                    # the next batch's preparation is enqueued BEFORE this step's backward: its HBM-bound kernels then run beside
                    # the backward's first convolutions instead of beside the step's last weight gradient, which nothing else
                    # hides (the step ended when the preparation did, 0.17 ms after the last gradient kernel)
                    prep = _prep_stream(dev)
                    _wait_created(prep, prepare_next)
                    _fence_wait(prep, dev)
                    with torch.cuda.stream(prep):
                        nr = prepare_frame_set(prepare_next, sample=(model.head, imsize) if PRESAMPLE else None,
                                               grid=model if PREGRID else None)
                        ev = torch.cuda.Event()
                        ev.record(prep)
                    next_ready = nr + (ev,)
                fr.middle_backward(model, saved, gm)
            if keep_mid is not None:
                for k in range(len(live)):
                    keep_mid.append(mid[k:k + 1])
        if prepare_next is not None and next_ready is None:
            if PREP_STREAM:
                prep = _prep_stream(dev)
                _wait_created(prep, prepare_next)
                _fence_wait(prep, dev)
                with torch.cuda.stream(prep):
                    nr = prepare_frame_set(prepare_next, sample=(model.head, imsize) if PRESAMPLE else None, grid=model if PREGRID else None)
                    ev = torch.cuda.Event()
                    ev.record(prep)
                next_ready = nr + (ev,)
            else:
                next_ready = prepare_frame_set(prepare_next)
    finally:
        _hip.GRAD_SINK = old_sink
        _hip.arena_end()
        _hip.join_side_stream()
        _fence_record(dev)
    if prepare_next is not None:
        return counts, statuses, next_ready
    return counts, statuses


def train_step_rows_only(model, batch, state, ready=None, prepare_next=None, with_fusion=False, imsize=None):
    """BASELINE.json config 2 ("VFE-only"): crop + projection, voxelizer, then the VFE stack -- SVFE + FCN + max
    (voxelnet/Pipe.py:5-29, VoxelNet.py:27-33) -- forward and backward for all frames as one frame set.  The 16 fused image
    channels of every row and dL/d(voxel features) are resident random tensors (``state`` caches them) standing for the
    fusion branch and for everything behind the VFE, which this configuration does not run.
    ``with_fusion`` (config 4, "PointFusion on: FPN feature sample + VFE"): the fusion branch is real -- bilinear sampling of
    the frames' FPN maps at the projected points + the fusion MLP (imhead/Pipe.py:23-104), forward and backward."""
    from modules import frames as fr
    dev = batch.device
    main = torch.cuda.current_stream(dev)
    ev_ready = None
    if ready is None:
        ready = prepare_frame_set(batch)
    elif len(ready) == 5:
        ready, ev_ready = ready[:4], ready[4]
    fs, live, counts, status = ready
    if ev_ready is not None:
        main.wait_event(ev_ready)
        status.record_stream(main)
        if fs is not None:
            fs.hand_over(main, fenced=PREP_FENCE)
    old_sink, _hip.GRAD_SINK = _hip.GRAD_SINK, True
    old_async, _hip.ASYNC_WGRAD = _hip.ASYNC_WGRAD, True
    next_ready = None
    try:
        if fs is not None:
            cap_rows = batch.n_frames * (batch.cap_points + 1)
            if 'imfeat' not in state or state['imfeat'].shape[0] < cap_rows:
                g = torch.Generator(device='cpu').manual_seed(5)
                state['imfeat'] = torch.randn((cap_rows, 16), generator=g).to(dev)
                state['dfeat'] = (torch.randn((batch.n_frames * batch.cap_points, 128), generator=g) * 1e-3).to(dev)
            _hip.arena_begin(dev, doubles=1 << 21)
            with torch.no_grad():
                if with_fusion:
                    statuses = [status]
                    feat, saved = fr.rows_forward(model, fs, [batch.fpn_levels[f] for f in live], imsize, statuses)
                    state['statuses'] = statuses
                else:
                    feat, saved = fr.rows_forward(model, fs, None, None, [], imfeat=state['imfeat'][:fs.Rt + fs.F])
                fr.rows_backward(model, saved, state['dfeat'][:fs.Vt])
        if prepare_next is not None:
            if PREP_STREAM:
                prep = _prep_stream(dev)
                _wait_created(prep, prepare_next)
                _fence_wait(prep, dev)
                with torch.cuda.stream(prep):
                    nr = prepare_frame_set(prepare_next)
                    ev = torch.cuda.Event()
                    ev.record(prep)
                next_ready = nr + (ev,)
            else:
                next_ready = prepare_frame_set(prepare_next)
    finally:
        _hip.GRAD_SINK = old_sink
        _hip.ASYNC_WGRAD = old_async
        _hip.arena_end()
        _hip.join_side_stream()
        _fence_record(dev)
    st_out = state.pop('statuses', [status])
    if prepare_next is not None:
        return counts, st_out, next_ready
    return counts, st_out


# ---- whole model on the fast path: frame sets up to the BEV map, RPN + VoxelLoss per frame ------------------------------
def batch_from_dataset(group, names, device, anchorBevs, fpn_fn, cap_points):
    """Frames of ``modules.data.Load.createDataset`` -> (FrameBatch resident on the GPU, per-frame targets).  Per frame, as
    train.py:26-49: lidar2Img on the torch path + (row, col) swap, the shuffle permutation drawn with np.random, and
    classifyAnchors for the 'Car' boxes; (pi, ni, gi, gt) or None when the frame has no box."""
    import numpy as np
    from modules import Calc
    from modules.data.Preprocessing import _calib_products
    B = len(group)
    pts6 = torch.zeros((B, cap_points, 6), dtype=torch.float32, device=device)
    perms = np.zeros((B, cap_points), np.int32)
    n = np.zeros((B,), np.int32)
    fpn = []
    for k, (velo, img, bbox2d, bbox3d, bev, calib) in enumerate(group):
        P = velo.shape[0]
        src = torch.from_numpy(np.ascontiguousarray(velo, dtype=np.float32)).to(device)
        pts6[k, :P, :4] = src
        m, p2 = _calib_products(calib, True)
        _hip.lidar2img(src, m, p2, math_f32=True, out=pts6[k, :P], col_offset=4, swap_rc=True)
        a = np.arange(P, dtype=np.int32)
        np.random.shuffle(a)
        perms[k, :P] = a
        n[k] = P
        fpn.append(fpn_fn(names[k], device))
    # target assignment of all frames in one kernel pass / one host read
    boxes = [(d[4], d[3][:, [0, 1]]) if (d[4] is not None and d[4].shape[0] != 0) else None for d in group]
    lists = Calc.classifyAnchorsFrames(boxes, anchorBevs, cfg.velorange, 0.45, 0.6)
    targets = [None if t is None else (t[0], t[1], t[2], d[3].to(device)) for t, d in zip(lists, group)]
    batch = FrameBatch(pts6, torch.from_numpy(perms).to(device), torch.from_numpy(n).to(device), fpn).mark_created()
    return batch, targets


RPN_HIP = _os.environ.get('MVX_RPN_HIP', '1') != '0'       # RPN frame sets on this library's kernels (modules/rpn_frames.py); 0 = torch modules (MIOpen)


def heads_loss(heads, F, h1, w1, targets, criterion, anchors):
    """VoxelLoss (voxelnet/Loss.py:15-45) of every frame on the channels-last head output (F*h1*w1, 16) = [cls logits | reg]
    and its gradient, without the autograd engine: score = sigmoid(logits) (voxelnet/Pipe.py:74), one loss call per frame
    writing d(loss)/dscore and d(loss)/dreg straight into the gradient of the heads, then the sigmoid's derivative.
    Returns (losses (F,2) on the device = (clsLoss, regLoss or 0), per-frame bool "has a regression loss", d_heads)."""
    from modules.voxelnet.Loss import _index_block
    dev = heads.device
    v = heads.view(F, h1, w1, 16)
    score = torch.sigmoid(v[..., :2])
    d = torch.zeros_like(v)
    dscore = torch.empty_like(score)
    anc = anchors.detach().float().contiguous()
    losses, has_reg = [], []
    for k in range(F):
        t = targets[k]
        pos = neg = gi = gts = None
        n_pos = n_neg = 0
        if t is not None:
            pos, n_pos = _index_block(t[0], dev)
            neg, n_neg = _index_block(t[1], dev)
            if n_pos > 0:
                gi = torch.as_tensor(t[2]).long().to(dev).contiguous()
                gts = t[3].detach().float().to(dev).contiguous()
        regress = n_pos > 0
        ls, _, _ = _hip.voxel_loss(score[k], v[k, :, :, 2:] if regress else None, pos, neg, gi, n_pos, n_neg, gts,
                                   anc if regress else None, 2, float(criterion.a), float(criterion.b), float(criterion.eps),
                                   dscore_out=dscore[k], dreg_out=d[k, :, :, 2:] if regress else None)
        losses.append(ls)
        has_reg.append(regress)
    d[..., :2] = dscore * score * (1.0 - score)
    return torch.stack(losses), has_reg, d.view(F * h1 * w1, 16)


def train_step_full(model, batch, targets, criterion, anchors, imsize, ready=None, prepare_next=None, rpn_hip=None, read=True,
                    keep=None):
    """One optimizer step's worth of forward + backward of the WHOLE model for the frames of ``batch``: frame sets up to
    the CML output (modules/frames.py), the RPN of all frames on this library's kernels (modules/rpn_frames.py; per-frame
    BatchNorm statistics, exactly B reference forwards, train.py:131-161), VoxelLoss per frame, and the whole way back.
    ``targets``: per frame (pi, ni, gi, gt boxes) or None.  Gradients are ADDED into the existing .grad buffers
    (GradBucket).  ``rpn_hip=False`` runs the RPN + loss through the torch modules under autograd instead (comparison).
    ``prepare_next`` = (next batch, callable returning its targets): voxelized, mapped and target-assigned on the
    preparation stream after this step has been enqueued; returned as out['next'] = (ready, targets) for the next call.
    ``read=False`` leaves the losses on the device (out['losses_dev'] (F,2)) and skips the status read: the caller reads
    them a step later, so the host never waits for the step it has just enqueued.
    ``keep`` (tests): a dict that receives the step's intermediate maps -- 'x3' the normalised CML output, channels-last
    [F*D3][H][W][C3] (BEV channel c*D3 + d, VoxelNet.py:36), 'heads' (F*h1*w1, 16) = [cls logits | reg], 'geom'."""
    from modules import frames as fr
    from modules import rpn_frames as rf
    rpn_hip = RPN_HIP if rpn_hip is None else rpn_hip
    dev = batch.device
    main = torch.cuda.current_stream(dev)
    ev_ready = None
    if ready is None:
        ready = prepare_frame_set(batch)
    elif len(ready) == 5:
        ready, ev_ready = ready[:4], ready[4]
    fs, live, counts, status = ready
    if ev_ready is not None:
        main.wait_event(ev_ready)
        status.record_stream(main)
        if fs is not None:
            fs.hand_over(main, fenced=PREP_FENCE)
        for t in targets:
            if t is not None:
                for x in tuple(t[0]) + tuple(t[1]) + (t[2], t[3]):
                    if isinstance(x, torch.Tensor) and x.is_cuda:
                        x.record_stream(main)
    out = {'loss': [], 'cls': [], 'reg': [], 'voxels': counts, 'live': live}
    hw = [float(imsize[0]), float(imsize[1])]
    old_sink, _hip.GRAD_SINK = _hip.GRAD_SINK, True
    old_async, _hip.ASYNC_WGRAD = _hip.ASYNC_WGRAD, True
    statuses = [status]
    try:
        if fs is not None:
            model.prepack()
            _hip.arena_begin(dev, doubles=1 << 22)
            F = len(live)
            tl = [targets[f] for f in live]
            with torch.no_grad():
                feat, saved = fr.rows_forward(model, fs, [batch.fpn_levels[f] for f in live], hw, statuses)
                if rpn_hip:
                    fr.cml_forward(model, fs, feat, saved, statuses, want_bev=False)
                    rpn = model.backbone.rpn
                    heads, rs = rf.rpn_forward(rpn, saved.x3, F, saved.D3, saved.H, saved.W, saved.C3)
                    losses, has_reg, d_heads = heads_loss(heads, F, rs['h1'], rs['w1'], tl, criterion, anchors)
                    if keep is not None:
                        keep.update(x3=saved.x3, heads=heads, geom=(F, saved.D3, saved.H, saved.W, saved.C3, rs['h1'], rs['w1']))
                    g_cl = rf.rpn_backward(rpn, rs, d_heads)
                    fr.rows_backward(model, saved, fr.cml_backward(model, saved, None, g_cl=g_cl))
                else:
                    mid = fr.cml_forward(model, fs, feat, saved, statuses)
            if not rpn_hip:
                leaf = mid.detach().requires_grad_(True)
                total, parts = None, []
                for k in range(F):
                    score, reg = model.backbone.rpn.forward_torch(leaf[k:k + 1])
                    score = score.squeeze(dim=0).permute(1, 2, 0)
                    reg = reg.squeeze(dim=0).permute(1, 2, 0)
                    t = tl[k]
                    if t is None:
                        cls_loss, reg_loss = criterion(None, None, None, None, score, None, anchors, 2)
                    else:
                        cls_loss, reg_loss = criterion(t[0], t[1], t[2], t[3], score, reg, anchors, 2)
                    loss = cls_loss if reg_loss is None else cls_loss + reg_loss
                    total = loss if total is None else total + loss
                    parts.append(torch.stack([cls_loss.detach(), reg_loss.detach() if reg_loss is not None else cls_loss.detach() * 0]))
                total.backward()
                has_reg = [t is not None and len(t[0][0]) > 0 for t in tl]
                losses = torch.stack(parts)
                with torch.no_grad():
                    fr.rows_backward(model, saved, fr.cml_backward(model, saved, leaf.grad))
            out['losses_dev'], out['has_reg'] = losses, has_reg
        if prepare_next is not None:
            nb, target_fn = prepare_next
            if PREP_STREAM:
                prep = _prep_stream(dev)
                _wait_created(prep, nb)
                _fence_wait(prep, dev)
                with torch.cuda.stream(prep):
                    nr = prepare_frame_set(nb, sample=(model.head, hw) if PRESAMPLE else None, grid=model if PREGRID else None)
                    nt = target_fn()
                    ev = torch.cuda.Event()
                    ev.record(prep)
                out['next'] = (nr + (ev,), nt)
            else:
                out['next'] = (prepare_frame_set(nb), target_fn())
        out['statuses'] = statuses
        if read and fs is not None:
            read_losses(out)
    finally:
        _hip.GRAD_SINK = old_sink
        _hip.ASYNC_WGRAD = old_async
        _hip.arena_end()
        _hip.join_side_stream()
        _fence_record(dev)
    return out


def read_losses(out):
    """Host side of a step enqueued with read=False: the status words and the per-frame losses (ONE device read each)."""
    if 'losses_dev' not in out:
        return out
    bad = int(torch.stack([s.reshape(()) for s in out['statuses']]).max())
    if bad:
        raise _hip.X.MvxHipError('a kernel reported a data-dependent error (status %d)' % bad)
    vals = out['losses_dev'].tolist()
    for (c, r), hr in zip(vals, out['has_reg']):
        out['cls'].append(c)
        if hr:
            out['reg'].append(r)
        out['loss'].append(c + r if hr else c)
    return out
