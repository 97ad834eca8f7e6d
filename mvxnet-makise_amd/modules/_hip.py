"""Typed torch-tensor wrappers over the C ABI (plumbing only: allocation, pointers, stream)."""
import collections

import os

import torch

from modules import Extension as X

VoxelizeResult = collections.namedtuple('VoxelizeResult', 'voxels coords counts n_voxels status')

_ws_cache = {}
STATS_REPLICAS = 32      # MVX_STATS_REPLICAS of include/mvx_hip.h
FLAG_RELU, FLAG_PREZEROED, FLAG_ACCUMULATE, FLAG_CONV2D = 1, 2, 4, 8      # MVX_FLAG_* of include/mvx_hip.h
FLAG_SPLIT = 64                                                          # MVX_FLAG_SPLIT: bf16x3 arithmetic of the wide row GEMMs
FLAG_SPLIT3 = 128                                                        # MVX_FLAG_SPLIT3: three bf16 pieces per operand (bf16x6, fp32-grade)
FLAG_AMAX_COARSE = 1024                                                  # MVX_FLAG_AMAX_COARSE: the bound x is a foreign forward input (8-binade scale steps)
FLAG_NO_BG_FILL = 2048                                                   # MVX_FLAG_NO_BG_FILL: voxel-free tiles of the sparse conv1 output are not written
FLAG_SPLIT_F16 = 512                                                     # MVX_FLAG_SPLIT_F16: two fp16 pieces per operand (fp16x3)


def split_flags(split, row=False):
    """``split``: 0 / False = exact-f32 MFMA, 2 / True = bf16x3, 3 = bf16x6 -> the flag bits of a call (row GEMMs also need
    MVX_FLAG_SPLIT; the *_split convolution entry points only look at MVX_FLAG_SPLIT3)."""
    if not split:
        return 0
    return (FLAG_SPLIT if row else 0) | (FLAG_SPLIT3 if int(split) == 3 else 0) | (FLAG_SPLIT_F16 if int(split) == 4 else 0)


def amax_of(t):
    """The 1-element float32 tensor holding max |t| that a producer kernel wrote (``tag_amax``), or None."""
    return getattr(t, '_mvx_amax', None)


def tag_amax(t, amax):
    """Attach the device-side max |t| (1-element float32 tensor, kept alive by the attribute) to ``t``: the fp16x3 kernels scale
    the operand by it (include/mvx_hip.h: mvx_split_operand_amax).  A ``.view()`` of t does not carry the tag: re-tag it."""
    if t is not None and amax is not None:
        t._mvx_amax = amax
    return t


def new_amax(device):
    """A 1-element float32 tensor for a producer kernel to write max |dz| into.  Its OWN allocation, not a slot of the
    accumulator arena: side-stream weight-gradient kernels read it after the arena of their lane has been cleared for the next
    frame."""
    return torch.empty((1,), dtype=torch.float32, device=device)


def tensor_amax(t):
    """Measure max |t| on the device (one pass over t) and tag t with it."""
    am = torch.empty((1,), dtype=torch.float32, device=t.device)
    X.check(X.lib.mvx_tensor_amax(X.ptr(t), t.numel(), X.ptr(am), 0, X.stream()), 'mvx_tensor_amax')
    return tag_amax(t, am)


def bind_amax(split, a=None, b=None):
    """fp16x3 (split code 4): bind the operand ranges of the NEXT split launch to the tags of tensors a / b (untagged: not
    scaled).  The entry point that follows consumes the binding."""
    if split and int(split) == 4:
        X.lib.mvx_split_operand_amax(X.ptr(amax_of(a)) if a is not None else None, X.ptr(amax_of(b)) if b is not None else None)


def foreign_split(split, x=None):
    """Arithmetic of a FORWARD row GEMM whose input was not produced by this library (sampled image features, the input of a
    stand-alone FCN).  fp16x3 needs the input's range there, and a scale taken over whatever tensor the executor happens to
    hold (one frame, a frame set) makes two executors round differently -- enough to flip ReLUs between them
    (tools/dbg_fp16_seeds.py: 3e-2 on 4 of 6 seeds).  So: a foreign input WITH a range tag runs fp16x3 with the COARSE scale
    (MVX_FLAG_AMAX_COARSE: 8-binade steps, the same for a frame and its set unless their maxima straddle a step); one without a
    tag runs bf16x6, which has the range of f32.  Returns (split code, extra flags)."""
    if split and int(split) == 4:
        return (4, FLAG_AMAX_COARSE) if (x is not None and amax_of(x) is not None) else (3, 0)
    return split, 0


class foreign_input_math:
    """Scope of ONE stand-alone block call (CRB3d / CRB2d / DeCRB2d through the nn.Module interface) on a tensor this library
    did not produce: under ``convmath: fp16x3`` the block runs in bf16x6 -- an input of unknown range has no place in fp16's
    five exponent bits (1e-3-sized inputs would keep ~15 bits, 1e5-sized ones overflow, and the BatchNorm behind the
    convolution would hide both), bf16 pieces have the range of f32.  Tensors produced by this library's blocks
    (outputs of a BatchNorm: bounded by sqrt(N)) carry ``_mvx_lib`` and keep fp16x3.  ``math``: re-enter the arithmetic a
    forward call chose (its backward)."""

    def __init__(self, x=None, math=None):
        self.x, self.math, self.old = x, math, None

    def __enter__(self):
        import modules.config as cfg
        cur = cfg.config.get('convmath', 'f32')
        want = self.math
        if want is None:
            want = 'bf16x6' if (cur == 'fp16x3' and not getattr(self.x, '_mvx_lib', False)) else cur
        if want != cur:
            self.old, cfg.config['convmath'] = cur, want
        return want

    def __exit__(self, *exc):
        if self.old is not None:
            import modules.config as cfg
            cfg.config['convmath'] = self.old
        return False


def mark_lib(t):
    """``t`` is the output of one of this library's BatchNorms (see foreign_input_math)."""
    t._mvx_lib = True
    return t


# The 1 % mutation hooks of the gradient tests (frames._MUTATE, rpn_frames._MUTATE) only act in a process started with
# MVX_ALLOW_MUTATION=1 (tests/conftest.py sets it): a product run cannot switch them on by accident.
MUTATION_ALLOWED = os.environ.get('MVX_ALLOW_MUTATION') == '1'


def mutate(table, name, t):
    """``t * table[name]`` when the test hook names this term (and keeps t's range tag, scaled: the mutated tensor then takes
    the same arithmetic path -- fp16x3 with its amax -- as the unmutated one); ``t`` otherwise."""
    f = table.get(name)
    if f is None or t is None:
        return t
    if not MUTATION_ALLOWED:
        raise X.MvxHipError('gradient mutation hooks are test-only: start the process with MVX_ALLOW_MUTATION=1')
    t2 = t * f
    am = amax_of(t)
    if am is not None:
        tag_amax(t2, am * abs(float(f)))
    return t2


# ---- fp16x3: the weight-range guard.  Weights are cut times 2^8 (csrc/split_common.h), so a weight at or above 255.9 would
# overflow its high fp16 piece into inf -- silently, behind a BatchNorm that hides the scale.  Every weight
# tensor an fp16x3 kernel is about to read is checked ON THE DEVICE once per parameter version (a tiny kernel; no host read) into
# a per-device status word that joins the step's other status words (modules/frames.py) and is decoded by raise_on_status:
# the step that used such a weight is reported, loudly, at its status check.
STATUS_F16_WEIGHT_RANGE = 8
_W16_STATUS = {}


def fp16_weight_status(device):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _W16_STATUS.get(idx)
    if st is None:
        st = _W16_STATUS[idx] = torch.zeros((1,), dtype=torch.int32, device=torch.device('cuda', idx))
    return st


def guard_fp16_weight(w):
    """Range check of a weight tensor for the fp16-piece arithmetic, once per version of the parameter it views."""
    base = w._base if w._base is not None else w
    seen = base.__dict__.setdefault('_mvx_w16', {})
    tag = (base._version, w.data_ptr())
    if seen.get(tuple(w.shape)) != tag:
        seen[tuple(w.shape)] = tag
        wc = w.detach()
        wc = wc if wc.is_contiguous() else wc.contiguous()
        X.check(X.lib.mvx_split_f16_weight_check(X.ptr(wc), wc.numel(), X.ptr(fp16_weight_status(w.device)), X.stream()),
                'mvx_split_f16_weight_check')


def raise_on_status(word):
    """Decode the OR of a step's device status words (one host read by the caller)."""
    word = int(word)
    if word & STATUS_F16_WEIGHT_RANGE:
        raise X.MvxHipError('convmath fp16x3: a weight reached |w| >= 255.9 (or is not finite): its fp16 pieces overflow.  The results '
                            'of this step are invalid; use convmath bf16x6 (three bf16 pieces: the exponent range of f32)')
    if word & 1:
        raise AssertionError('projected point outside the feature map (imhead/Pipe.py:71)')
    if word:
        raise X.MvxHipError('a kernel reported a data-dependent error (status %d)' % word)


def grad_split(split, dz):
    """Arithmetic of a row GEMM whose operand ``dz`` is a gradient: fp16x3 needs its range (a tag), else bf16x6 stands in (the
    row kernels split both operands in the kernel, so any arithmetic can run any call)."""
    if split and int(split) == 4 and amax_of(dz) is None:
        return 3
    return split


def split_pieces():
    """convmath of config.yml -> 0 (f32: exact-f32 MFMA), 2 (bf16x3) or 3 (bf16x6)."""
    import modules.config as cfg
    m = cfg.config.get('convmath', 'f32')
    if m not in ('f32', 'bf16x3', 'bf16x6', 'fp16x3'):
        raise X.MvxHipError('convmath must be f32, bf16x3, bf16x6 or fp16x3, not %r' % (m,))
    return {'f32': 0, 'bf16x3': 2, 'bf16x6': 3, 'fp16x3': 4}[m]
FLAG_BG_TAPS = 32

# When True, the backward of the hot-path layers adds weight / bias gradients straight into the existing
# ``.grad`` buffers inside the reduction kernels (and returns None to autograd), instead of producing a
# temporary that autograd adds afterwards.  Set by the training pipeline (modules/pipeline.py).
GRAD_SINK = False


# When True (and gradients go straight into .grad, see GRAD_SINK), weight-gradient kernels are enqueued
# on a second HIP stream: their results are only needed at the optimizer step, so they overlap with the
# rest of the backward pass (their MFMA work fills the CUs while BatchNorm / gather kernels wait on HBM).
ASYNC_WGRAD = False
_SIDE = {}


def side_stream(device):
    st = _SIDE.get(device.index)
    if st is None:
        st = torch.cuda.Stream(device=device, priority=int(os.environ.get('MVX_SIDE_PRIORITY', '0')))
        _SIDE[device.index] = st
    return st


if os.environ.get('MVX_SPLIT16_MIN_UNITS'):        # developer knob: launch-shape threshold of the split gather (csrc/conv3d_split.hip)
    X.check(X.lib.mvx_tuning_set(1, int(os.environ['MVX_SPLIT16_MIN_UNITS'])), 'mvx_tuning_set')
if os.environ.get('MVX_K128'):                     # developer knob (A/B runs): 0 = 128-deep row layers on linear_fwd_split again
    X.check(X.lib.mvx_tuning_set(3, int(os.environ['MVX_K128'])), 'mvx_tuning_set')
SIDE_KEEP = os.environ.get('MVX_SIDE_KEEP', '1') != '0'
_KEEP = {}          # device index -> tensors read by side-stream kernels since the last join
_COMM = {}
_TAIL = {}          # device index -> [event on the side stream before the step's last weight gradient, event at the end of the main stream's backward]


def comm_stream(device):
    """Stream of the data-parallel exchange (modules/parallel.py: the early part of the gradient bucket goes out on it while
    the step's last weight gradient is still running)."""
    st = _COMM.get(device.index)
    if st is None:
        st = _COMM[device.index] = torch.cuda.Stream(device)
    return st


def mark_tail(device):
    """Called by the backward right BEFORE it enqueues the step's last weight-gradient kernel on the side stream: every other
    parameter gradient of the side stream is complete when this event fires."""
    if device.index in _SIDE:
        ev = torch.cuda.Event()
        ev.record(_SIDE[device.index])
        _TAIL[device.index] = [ev, None]


def tail_events(device):
    """The pair of the step that has just been enqueued, consumed by the caller (None: no tail was marked)."""
    t = _TAIL.pop(device.index, None)
    return tuple(t) if t is not None and t[1] is not None else None


def drop_tail(device=None):
    """Forget the tail pair: a gradient has been (or will be) written on the main stream AFTER join_side_stream recorded the end
    of the backward -- lane gradients added into the bucket, autograd's AccumulateGrad of returned views -- so the early part
    of the exchange must wait for the whole main stream (parallel.py falls back to ``comm.wait_stream(main)``)."""
    if device is None:
        _TAIL.clear()
    else:
        _TAIL.pop(device.index, None)


def join_side_stream(device=None):
    """Make the current stream wait for every weight-gradient kernel enqueued on the side stream."""
    for idx, st in _SIDE.items():
        if device is None or device.index == idx:
            cur = torch.cuda.current_stream(st.device)
            t = _TAIL.get(idx)
            if t is not None:
                if t[1] is None:                    # the main stream's own gradient work ends here (before it waits for the tail)
                    ev = torch.cuda.Event()
                    ev.record(cur)
                    t[1] = ev
                else:                               # a mark of an earlier join: stale
                    del _TAIL[idx]
            cur.wait_stream(st)
            keep = _KEEP.pop(idx, None)             # dropped AFTER the wait has been enqueued (see _SideStream)
            del keep


class _SideStream:
    """Run the enclosed launches on the side stream, ordered after everything already enqueued on the
    current stream; the tensors they read are kept alive for the side stream (record_stream)."""

    def __init__(self, *tensors):
        self.tensors = [t for t in tensors if t is not None]
        # the side-stream kernels also read the range tags of their operands (fp16x3: amax_of): separate allocations
        self.tensors += [a for a in (amax_of(t) for t in self.tensors) if a is not None]

    def __enter__(self):
        dev = self.tensors[0].device
        self.side = side_stream(dev)
        self.side.wait_stream(torch.cuda.current_stream(dev))
        if SIDE_KEEP:
            # keep the tensors alive until the owning stream has JOINED the side stream (join_side_stream drops the list after
            # its wait): their blocks then return to the owning stream's pool in stream order.  record_stream instead parks a
            # block until an event of the side stream has been seen to complete, which is timing dependent: the hot loop held
            # 40 GB reserved for 8 GB allocated and its peak crept 6.6 % over 1,500 identical steps (profiles/r03_soak.txt)
            _KEEP.setdefault(dev.index, []).extend(self.tensors)
        else:
            for t in self.tensors:
                t.record_stream(self.side)
        self.ctx = torch.cuda.stream(self.side)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        return self.ctx.__exit__(*exc)


class _Inline:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def _wgrad_scope(accumulate_into, *tensors):
    if ASYNC_WGRAD and accumulate_into is not None:
        return _SideStream(*tensors)
    return _Inline()


class ZeroArena:
    """One f64 buffer cleared with ONE fill per frame, from which every BatchNorm accumulator of the
    frame is carved (the C entry points then skip their own memsets: MVX_FLAG_PREZEROED)."""

    def __init__(self, device, doubles=1 << 19):
        self.buf = torch.empty((doubles,), dtype=torch.float64, device=device)
        self.off = doubles                 # exhausted until begin()

    def begin(self):
        self.buf.zero_()
        self.off = 0

    def take(self, n):
        n_al = (n + 31) & ~31
        if self.off + n_al > self.buf.numel():
            return None
        view = self.buf[self.off:self.off + n]
        self.off += n_al
        return view


_ARENAS = {}          # one arena per (device, stream): frames in flight on different streams never share one
_ARENA_ON = False


def _arena_key(device):
    return (device.index, X.raw_stream(device.index))


def arena_begin(device, doubles=1 << 19):
    """Start a frame (or a frame set) on the current stream: clear its accumulator arena (created on first use; grown
    when a larger one is asked for)."""
    global _ARENA_ON
    key = _arena_key(device)
    if key not in _ARENAS or _ARENAS[key].buf.numel() < doubles:
        with torch.cuda.device(device):
            _ARENAS[key] = ZeroArena(device, doubles)
    _ARENAS[key].begin()
    _ARENA_ON = True


def arena_end():
    global _ARENA_ON
    _ARENA_ON = False
    for a in _ARENAS.values():
        a.off = a.buf.numel()


def _acc_f64(shape, device):
    """(accumulator tensor, flag): carved from the frame arena (pre-zeroed) or freshly allocated."""
    n = 1
    for d in shape:
        n *= int(d)
    if _ARENA_ON:
        a = _ARENAS.get(_arena_key(device))
        if a is not None:
            v = a.take(n)
            if v is not None:
                return v.view(shape), FLAG_PREZEROED
    return torch.empty(shape, dtype=torch.float64, device=device), 0

# Optional live kernel timing (bench.py): name -> list of (start_event, end_event, algorithmic_flops).
# Events are recorded on the current stream, the stream every kernel of this library is launched on.
KERNEL_TIMERS = None


_EVENT_POOL = []          # timing events are recycled: creating one costs tens of microseconds of host time


def _timing_event():
    return _EVENT_POOL.pop() if _EVENT_POOL else torch.cuda.Event(enable_timing=True)


def recycle_timing_events(timers):
    """Hand the events of a finished (and read) KERNEL_TIMERS dict back to the pool."""
    for evs in timers.values():
        for s_, e_, _ in evs:
            _EVENT_POOL.append(s_)
            _EVENT_POOL.append(e_)


_STREAM_OBJS = {}


def _current_stream_obj():
    """torch.cuda.current_stream() without its ~10 us of Python per call (Event.record() without a stream argument pays it
    too): the Stream object of the current raw stream, looked up once per (device, raw stream)."""
    idx = torch.cuda.current_device()
    key = (idx, X.raw_stream(idx))
    st = _STREAM_OBJS.get(key)
    if st is None:
        st = torch.cuda.current_stream()
        _STREAM_OBJS[key] = st
    return st


TIMER_NAMES = None     # bench.py: with KERNEL_TIMERS on, only these timer names record events (None = all).  Two event records per
                       # library call are ~180 markers in a hot step: 0.5 ms of an 8.2 ms step (tools/soak_plain.py runs 7.65)


class _Timed:
    def __init__(self, name, flops):
        self.name, self.flops = name, flops

    def __enter__(self):
        self.on = KERNEL_TIMERS is not None and (TIMER_NAMES is None or self.name in TIMER_NAMES)
        if self.on:
            self.s = _timing_event()
            self.e = _timing_event()
            self.s.record(_current_stream_obj())
        return self

    def __exit__(self, *exc):
        if self.on and KERNEL_TIMERS is not None:
            self.e.record(_current_stream_obj())
            KERNEL_TIMERS.setdefault(self.name, []).append((self.s, self.e, self.flops))
        return False


def _timed_bytes(name, nbytes):
    """Timer for an HBM-bound stage: the third tuple field carries ALGORITHMIC bytes (negative marks bytes)."""
    if KERNEL_TIMERS is None:
        return _Timed('hbm:' + name, 0)
    # a data-dependent byte count (e.g. bytes of the flagged tiles) is passed as a zero-argument CALLABLE that the
    # consumer of the timers evaluates after the run (timer_value): neither a host read nor an extra launch inside the step
    return _Timed('hbm:' + name, nbytes if (callable(nbytes) or isinstance(nbytes, torch.Tensor)) else float(nbytes))


def timer_value(v):
    """The FLOP / byte figure of a timer entry as a float: numbers as they are, device scalars read, callables evaluated
    (data-dependent counts are deferred until after the timed region)."""
    if callable(v):
        v = v()
    return float(v)


def require_plain_batchnorm():
    """The kernels normalise with (mean, 1/sqrt(var + eps)) of the batch only -- the reference's configuration
    (config.yml:19-20: bnaffine False, bntrack False; Blocks.py:10,25).  With affine parameters or running statistics
    switched on, bn.weight / bn.bias would enter the optimizer with permanently zero gradients and the running buffers
    would never move, silently: refuse instead."""
    import modules.config as cfg
    if bool(cfg.config.get('bnaffine', False)) or bool(cfg.config.get('bntrack', False)):
        raise X.MvxHipError('bnaffine / bntrack are not supported by the HIP BatchNorm kernels (batch statistics, no affine '
                            'parameters, no running buffers: the reference\'s config.yml:19-20); set both to False')


def conv_flops(d_out_planes, d_src_planes, H, W, cin, cout, sd, pd, dgrad=False):
    """Executed multiply-add FLOPs of one gather launch: depth taps that fall outside the source
    volume are skipped by the kernel and are NOT counted (in-plane zero padding is computed and
    is counted)."""
    taps = 0
    for d in range(d_out_planes):
        for kd in range(3):
            if not dgrad:
                s = d * sd - pd + kd
                taps += 0 <= s < d_src_planes
            else:
                t = d + pd - kd
                taps += t >= 0 and t % sd == 0 and t // sd < d_src_planes
    return 2.0 * taps * 9 * H * W * cin * cout


def workspace(nbytes, dev, tag):
    """Grow-only scratch buffer per (device, tag); stream-ordered reuse on the current stream."""
    key = (dev.index, tag, X.raw_stream(dev.index))    # one buffer per (tag, stream): no cross-stream reuse
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)
        _ws_cache[key] = buf
    return buf


def voxelize(pcd, perm, n_points, lo, size, T, out_channels, cap_voxels=None, ext_idx=None):
    """pcd f32 (F,capP,ncol) on the GPU; perm i32 (F,capP) or None; n_points i32 (F,) or None
    (= every frame full).  Returns capacity-sized outputs plus device-side counts."""
    assert pcd.dim() == 3 and pcd.dtype == torch.float32
    F, cap, ncol = pcd.shape
    dev = pcd.device
    if n_points is None:
        n_points = torch.full((F,), cap, dtype=torch.int32, device=dev)
    cap_voxels = cap if cap_voxels is None else int(cap_voxels)
    voxels = torch.empty((F, cap_voxels, T, out_channels), dtype=torch.float32, device=dev)
    coords = torch.empty((F, cap_voxels, 4), dtype=torch.int64, device=dev)
    counts = torch.empty((F, cap_voxels), dtype=torch.int32, device=dev)
    n_vox = torch.empty((F,), dtype=torch.int32, device=dev)
    status = torch.zeros((1,), dtype=torch.int32, device=dev)
    nbytes = X.lib.mvx_voxelize_workspace_bytes(F, cap)
    ws = workspace(nbytes, dev, 'voxelize')
    # algorithmic bytes: points + permutation read; the voxel payload is priced at ~V = P/4 voxels per frame
    with _timed_bytes('voxelize', F * cap * (ncol * 4 + 4) + F * (cap // 4) * (T * out_channels * 4 + 36)):
        rc = X.lib.mvx_voxelize(X.ptr(pcd), X.ptr(perm), X.ptr(n_points), X.ptr(ext_idx), F, cap, ncol,
                                float(lo[0]), float(lo[1]), float(lo[2]),
                                float(size[0]), float(size[1]), float(size[2]),
                                int(T), int(out_channels), cap_voxels,
                                X.ptr(voxels), X.ptr(coords), X.ptr(counts), X.ptr(n_vox), X.ptr(status),
                                X.ptr(ws), ws.numel(), X.stream())
    X.check(rc, 'mvx_voxelize')
    return VoxelizeResult(voxels, coords, counts, n_vox, status)


def voxelize_concat(pcd, perm, n_points, lo, size, T, out_channels):
    """The batch layout: voxels of all frames back to back.  Returns (voxels (F*cap,T,C), coords (F*cap,4) with the frame
    index in column 0, counts, n_voxels i32 (F,), vox_off i32 (F+1,), status) -- capacity-sized, offsets on the device."""
    assert pcd.dim() == 3 and pcd.dtype == torch.float32
    F, cap, ncol = pcd.shape
    dev = pcd.device
    if n_points is None:
        n_points = torch.full((F,), cap, dtype=torch.int32, device=dev)
    total = F * cap
    voxels = torch.empty((total, T, out_channels), dtype=torch.float32, device=dev)
    coords = torch.empty((total, 4), dtype=torch.int64, device=dev)
    counts = torch.empty((total,), dtype=torch.int32, device=dev)
    n_vox = torch.empty((F,), dtype=torch.int32, device=dev)
    vox_off = torch.empty((F + 1,), dtype=torch.int32, device=dev)
    status = torch.zeros((1,), dtype=torch.int32, device=dev)
    ws = workspace(X.lib.mvx_voxelize_workspace_bytes(F, cap), dev, 'voxelize')
    with _timed_bytes('voxelize', F * cap * (ncol * 4 + 4) + F * (cap // 4) * (T * out_channels * 4 + 36)):
        rc = X.lib.mvx_voxelize_frames(X.ptr(pcd), X.ptr(perm), X.ptr(n_points), None, F, cap, ncol,
                                       float(lo[0]), float(lo[1]), float(lo[2]), float(size[0]), float(size[1]), float(size[2]),
                                       int(T), int(out_channels), total, 1, X.ptr(voxels), X.ptr(coords), X.ptr(counts),
                                       X.ptr(n_vox), X.ptr(vox_off), X.ptr(status), X.ptr(ws), ws.numel(), X.stream())
    X.check(rc, 'mvx_voxelize_frames')
    return voxels, coords, counts, n_vox, vox_off, status


# ---------------------------------------------------------------------------------------------
# dense grid scatter / gather
# ---------------------------------------------------------------------------------------------
_TILE = None


def conv_tile_shape():
    global _TILE
    if _TILE is None:
        import ctypes
        th, tw = ctypes.c_int32(0), ctypes.c_int32(0)
        X.lib.mvx_conv3d_tile_shape(ctypes.byref(th), ctypes.byref(tw))
        _TILE = (th.value, tw.value)
    return _TILE


def scatter_voxels(feat, coords, dhw, grid=None, zero=True, want_occupancy=False):
    D, H, W = dhw
    V, C = feat.shape
    if grid is None:
        grid = torch.empty((D, H, W, C), dtype=torch.float32, device=feat.device)
    status = torch.zeros((1,), dtype=torch.int32, device=feat.device)
    th, tw = conv_tile_shape()
    occ = bits = None
    if want_occupancy:
        occ = torch.empty((D, -(-H // th), -(-W // tw)), dtype=torch.int32, device=feat.device)
        bits = torch.empty((D, H, -(-W // 32)), dtype=torch.int32, device=feat.device)
    X.check(X.lib.mvx_scatter_voxels(X.ptr(feat), X.ptr(coords), X.ptr(grid), V, C, D, H, W, int(zero),
                                     X.ptr(status), X.ptr(occ), th, tw, X.ptr(bits), X.stream()), 'mvx_scatter_voxels')
    if want_occupancy:
        return grid, status, (occ, bits)
    return grid, status


def gather_voxels(grid, coords, V):
    D, H, W, C = grid.shape
    feat = torch.empty((V, C), dtype=torch.float32, device=grid.device)
    X.check(X.lib.mvx_gather_voxels(X.ptr(grid), X.ptr(coords), X.ptr(feat), V, C, D, H, W, X.stream()),
            'mvx_gather_voxels')
    return feat


def cl_to_bev(cl):
    """channels-last (D,H,W,C) -> (C*D,H,W) contiguous, channel = c*D + d."""
    D, H, W, C = cl.shape
    bev = torch.empty((C * D, H, W), dtype=torch.float32, device=cl.device)
    with _timed_bytes('cl_bev_transpose', 2 * cl.numel() * 4):
        X.check(X.lib.mvx_cl_to_bev(X.ptr(cl), X.ptr(bev), D, H, W, C, 0, X.stream()), 'mvx_cl_to_bev')
    return bev


def bev_to_cl(bev, D):
    CD, H, W = bev.shape
    C = CD // D
    cl = torch.empty((D, H, W, C), dtype=torch.float32, device=bev.device)
    X.check(X.lib.mvx_cl_to_bev(X.ptr(cl), X.ptr(bev), D, H, W, C, 1, X.stream()), 'mvx_cl_to_bev')
    return cl


# ---------------------------------------------------------------------------------------------
# BatchNorm (batch statistics) + ReLU
# ---------------------------------------------------------------------------------------------
def row_stats(y2d):
    rows, C = y2d.shape
    stats = torch.empty((STATS_REPLICAS, 2, C), dtype=torch.float64, device=y2d.device)
    X.check(X.lib.mvx_row_stats(X.ptr(y2d), X.ptr(stats), rows, C, X.stream()), 'mvx_row_stats')
    return stats


def bn_finalize(stats, count, eps):
    C = stats.shape[-1]
    mi = torch.empty((2, C), dtype=torch.float32, device=stats.device)
    X.check(X.lib.mvx_bn_finalize(X.ptr(stats), float(count), float(eps), X.ptr(mi), C, X.stream()),
            'mvx_bn_finalize')
    return mi


def bn_apply(y, mi, out=None):
    C = mi.shape[1]
    rows = y.numel() // C
    if out is None:
        out = torch.empty_like(y)
    with _timed_bytes('bn_apply', 2 * y.numel() * 4):
        X.check(X.lib.mvx_bn_apply(X.ptr(y), X.ptr(mi), X.ptr(out), rows, C, X.stream()), 'mvx_bn_apply')
    return out


def bn_relu_backward(dyhat, y, mi, count, want_dbias=True, dz=None, row_w=None, dbias_out=None):
    """dbias_out: existing gradient buffer to ADD the bias gradient to (returns None for dbias then)."""
    C = mi.shape[1]
    rows = y.numel() // C
    if dz is None:
        dz = torch.empty_like(y)
    flags = 0
    if dbias_out is not None:
        dbias, flags = dbias_out, FLAG_ACCUMULATE
    else:
        dbias = torch.empty((C,), dtype=torch.float32, device=y.device) if want_dbias else None
    scratch, fz = _acc_f64((X.lib.mvx_bn_backward_scratch_bytes(C) // 8,), y.device)
    amax = new_amax(y.device)
    with _timed_bytes('bn_relu_backward', 5 * y.numel() * 4):      # reduce reads 2 tensors, apply reads 2 + writes 1
        X.check(X.lib.mvx_bn_relu_backward_frames(X.ptr(dyhat), X.ptr(y), X.ptr(mi), float(count), X.ptr(dz),
                                                  X.ptr(dbias), X.ptr(scratch), X.ptr(row_w), rows, C, flags | fz, None,
                                                  X.ROWS_SINGLE, X.ptr(amax), X.stream()), 'mvx_bn_relu_backward_frames')
    tag_amax(dz, amax)                                   # max |dz|: the range the fp16x3 kernels scale dz by
    return dz, (None if dbias_out is not None else dbias)


# ---------------------------------------------------------------------------------------------
# dense 3x3x3 convolution (channels-last, one frame)
# ---------------------------------------------------------------------------------------------
def conv3d_pack(weight, for_dgrad, split=False):
    """Kernel-layout weights; split=True: pre-split (hi, lo) bf16 pairs for the bf16x3 kernels.  A 2-D kernel
    (cout,cin,3,3) is packed as the middle depth slice of a 3-D one (exact-f32 kernels only): nn.Conv2d 3x3 / stride 1 /
    padding 1 then runs on the conv3d kernels with a depth-1 tensor and pad_d = 1."""
    cout, cin = weight.shape[0], weight.shape[1]
    two_d = tuple(weight.shape[2:]) == (3, 3)
    assert two_d or tuple(weight.shape[2:]) == (3, 3, 3)
    assert not (two_d and split)
    if split:
        sf = split_flags(split)
        if int(split) == 4:
            guard_fp16_weight(weight)
        wpk = torch.empty((X.lib.mvx_conv3d_packed_weight_bytes_split(cout, cin, sf) // 2,), dtype=torch.int16, device=weight.device)
        X.check(X.lib.mvx_conv3d_pack_weights_split(X.ptr(weight.contiguous()), X.ptr(wpk), cout, cin, int(for_dgrad), sf, X.stream()),
                'mvx_conv3d_pack_weights_split')
        return wpk
    wpk = torch.empty((27 * cout * cin,), dtype=torch.float32, device=weight.device)
    X.check(X.lib.mvx_conv3d_pack_weights(X.ptr(weight.contiguous()), X.ptr(wpk), cout, cin, int(for_dgrad) | (2 if two_d else 0),
                                          X.stream()), 'mvx_conv3d_pack_weights')
    return wpk


def conv_out_depth(din, sd, pd):
    return (din + 2 * pd - 3) // sd + 1


def conv3d_forward(x, wpk, bias, cout, sd, pd, relu=True, want_stats=True, split=False):
    """Dense forward of one frame; split=True: bf16x3 kernel (wpk from conv3d_pack(..., split=True))."""
    din, H, W, cin = x.shape
    dout = conv_out_depth(din, sd, pd)
    out = torch.empty((dout, H, W, cout), dtype=torch.float32, device=x.device)
    stats, fz = _acc_f64((STATS_REPLICAS, 2, cout), x.device) if want_stats else (None, 0)
    flags = (FLAG_RELU if relu else 0) | fz | split_flags(split)
    if split:
        with _Timed('conv3d_gather_split', conv_flops(dout, din, H, W, cin, cout, sd, pd) if KERNEL_TIMERS is not None else 0):
            X.check(X.lib.mvx_conv3d_forward_split(X.ptr(x), X.ptr(wpk), X.ptr(bias), X.ptr(out), X.ptr(stats), din, dout,
                                                   H, W, cin, cout, sd, pd, flags, X.stream()),
                    'mvx_conv3d_forward_split')
        return out, stats
    with _Timed('conv3d_gather', conv_flops(dout, din, H, W, cin, cout, sd, pd) if KERNEL_TIMERS is not None else 0):
        X.check(X.lib.mvx_conv3d_forward(X.ptr(x), X.ptr(wpk), X.ptr(bias), X.ptr(out), X.ptr(stats),
                                         din, dout, H, W, cin, cout, sd, pd, flags, X.ptr(_work_counter(x.device)),
                                         X.stream()), 'mvx_conv3d_forward')
    return out, stats


def conv3d_dgrad(dz, wpk_d, din, cin, sd, pd, split=False):
    dout, H, W, cout = dz.shape
    dx = torch.empty((din, H, W, cin), dtype=torch.float32, device=dz.device)
    if split:
        with _Timed('conv3d_gather_split', conv_flops(din, dout, H, W, cout, cin, sd, pd, True) if KERNEL_TIMERS is not None else 0):
            bind_amax(split, dz)
            X.check(X.lib.mvx_conv3d_dgrad_split(X.ptr(dz), X.ptr(wpk_d), X.ptr(dx), din, dout, H, W, cin, cout, sd, pd,
                                                 split_flags(split), X.stream()), 'mvx_conv3d_dgrad_split')
        return dx
    with _Timed('conv3d_gather', conv_flops(din, dout, H, W, cout, cin, sd, pd, True) if KERNEL_TIMERS is not None else 0):
        X.check(X.lib.mvx_conv3d_dgrad(X.ptr(dz), X.ptr(wpk_d), X.ptr(dx), din, dout, H, W, cin, cout, sd, pd,
                                       X.ptr(_work_counter(dz.device)), X.stream()), 'mvx_conv3d_dgrad')
    return dx


def conv3d_wgrad(x, dz, sd, pd, split=False, accumulate_into=None, two_d=False):
    """accumulate_into: existing (cout,cin,3,3,3) gradient buffer to ADD to (returns None then).
    two_d: the gradient of a 2-D kernel (cout,cin,3,3) (depth-1 tensors, pad_d = 1)."""
    din, H, W, cin = x.shape
    dout, _, _, cout = dz.shape
    if accumulate_into is not None:
        dw, flags = accumulate_into, FLAG_ACCUMULATE
    else:
        shape = (cout, cin, 3, 3) if two_d else (cout, cin, 3, 3, 3)
        dw, flags = torch.empty(shape, dtype=torch.float32, device=x.device), 0
    if two_d:
        assert not split and din == 1 and dout == 1 and pd == 1
        flags |= FLAG_CONV2D
    # split arithmetic: conv3d_wgrad4s behind the same entry point (cin a multiple of 64), else the older 9-wave split kernel
    new_split = bool(split) and cin % 64 == 0
    flags |= split_flags(split, new_split)
    nbytes = X.lib.mvx_conv3d_wgrad_workspace_bytes(H, W, cin, cout)
    fn, name = (X.lib.mvx_conv3d_wgrad_split, 'conv3d_wgrad_split') if (split and not new_split) else (X.lib.mvx_conv3d_wgrad, 'conv3d_wgrad')
    with _wgrad_scope(accumulate_into, x, dz) as scope:
        ws = workspace(nbytes, x.device, 'wgrad_side' if isinstance(scope, _SideStream) else 'wgrad')
        with _Timed(name, conv_flops(dout, din, H, W, cin, cout, sd, pd) if KERNEL_TIMERS is not None else 0):
            bind_amax(split, x, dz)
            X.check(fn(X.ptr(x), X.ptr(dz), X.ptr(dw), din, dout, H, W, cin, cout, sd, pd, flags, X.ptr(ws), ws.numel(),
                       X.stream()), 'mvx_' + name)
    return None if accumulate_into is not None else dw


# ---------------------------------------------------------------------------------------------
# background rewrite of the CML stack (csrc/activity.hip)
# ---------------------------------------------------------------------------------------------
class Background:
    """What a CML activation looks like away from the voxels: per (plane, channel) constant ``c`` f32 (D,C)
    (``y_bg``: its value before the BatchNorm), site mask u8 (D,H,W) of the sites that do NOT hold it, per-tile
    flags i32 (D,tiles): ``hflag`` = the tile's halo holds such a site, ``tflag`` = the tile itself does.
    ``back``: dict the CONSUMER's backward fills for the producer's backward ('plane_grad_sums'), or None when
    the producer cannot use it (then the consumer computes a dense input gradient)."""
    __slots__ = ('c', 'y_bg', 'mask', 'hflag', 'tflag', 'bflag', 'back')

    def __init__(self, c, mask, hflag, tflag=None, y_bg=None, back=None, bflag=None):
        self.c, self.mask, self.hflag, self.tflag, self.y_bg, self.back = c, mask, hflag, tflag, y_bg, back
        # tiles on which this tensor's GRADIENT is produced/consumed by the restricted backward (defaults to tflag)
        self.bflag = bflag if bflag is not None else tflag


EXEC_STAGES = None       # device u64 counter of executed gather stages while KERNEL_TIMERS is on (bench roofline)
STAGE_FLOP = 2.0 * 9 * 128 * 32 * 64


def n_tiles(H, W):
    return ((H + 7) // 8) * ((W + 15) // 16)


def activity_dilate(src, src_is_index, din, H, W, sd, pd, mark_border, want_tile_flags=False):
    """(mask u8 (dout,H,W), halo flags i32 (dout,tiles)[, tile flags]) of a layer output from its input's activity."""
    dout = conv_out_depth(din, sd, pd)
    mask = torch.empty((dout, H, W), dtype=torch.uint8, device=src.device)
    hflag = torch.empty((dout, n_tiles(H, W)), dtype=torch.int32, device=src.device)
    tflag = torch.empty_like(hflag) if want_tile_flags else None
    X.check(X.lib.mvx_activity_dilate(X.ptr(src), 1 if src_is_index else 0, din, dout, H, W, sd, pd,
                                      1 if mark_border else 0, X.ptr(mask), X.ptr(hflag), X.ptr(tflag), X.stream()),
            'mvx_activity_dilate')
    return (mask, hflag, tflag) if want_tile_flags else (mask, hflag)


def conv3d_background(w, c_in, din, sd, pd):
    cout, cin = w.shape[0], w.shape[1]
    dout = conv_out_depth(din, sd, pd)
    bg_pre = torch.empty((dout, cout), dtype=torch.float32, device=w.device)
    X.check(X.lib.mvx_conv3d_background(X.ptr(w.contiguous()), X.ptr(c_in), din, dout, cin, cout, sd, pd, X.ptr(bg_pre),
                                        X.stream()), 'mvx_conv3d_background')
    return bg_pre


def bn_background(bg_pre, bias, mi, planes, channels, relu=True, want_y=False):
    c_out = torch.empty((planes, channels), dtype=torch.float32, device=mi.device)
    y_bg = torch.empty_like(c_out) if want_y else None
    X.check(X.lib.mvx_bn_background(X.ptr(bg_pre), X.ptr(bias), X.ptr(mi), planes, channels, FLAG_RELU if relu else 0,
                                    X.ptr(y_bg), X.ptr(c_out), X.stream()), 'mvx_bn_background')
    return (c_out, y_bg) if want_y else c_out


def plane_tap_sums(dz, tile_flags=None, inactive_sums=None):
    """f32 (planes, 9, C): border-corrected per-tap sums of dz over each plane (mvx_plane_tap_sums); with
    ``tile_flags`` dz is only read on the flagged tiles and ``inactive_sums`` (planes, C) stands for the rest."""
    D, H, W, C = dz.shape
    T = torch.empty((D, 9, C), dtype=torch.float32, device=dz.device)
    ws = workspace(X.lib.mvx_plane_tap_sums_workspace_bytes(D, C), dz.device, 'tap_sums')
    X.check(X.lib.mvx_plane_tap_sums(X.ptr(dz), D, H, W, C, X.ptr(tile_flags), X.ptr(inactive_sums), X.ptr(T), X.ptr(ws),
                                     ws.numel(), X.stream()), 'mvx_plane_tap_sums')
    return T


def tile_dilate_flags(in_tflag, self_tflag, din, H, W, sd, pd):
    dout = conv_out_depth(din, sd, pd)
    out = torch.empty((dout, n_tiles(H, W)), dtype=torch.int32, device=in_tflag.device)
    X.check(X.lib.mvx_tile_dilate_flags(X.ptr(in_tflag), X.ptr(self_tflag), din, dout, H, W, sd, pd, X.ptr(out), X.stream()),
            'mvx_tile_dilate_flags')
    return out


def conv3d_input_grad_sums(w, T, din, sd, pd):
    cout, cin = w.shape[0], w.shape[1]
    A = torch.empty((din, cin), dtype=torch.float32, device=w.device)
    X.check(X.lib.mvx_conv3d_input_grad_sums(X.ptr(w.contiguous()), X.ptr(T), din, T.shape[0], cin, cout, sd, pd, X.ptr(A),
                                             X.stream()), 'mvx_conv3d_input_grad_sums')
    return A


def conv3d_dgrad_tiles(dz, wpk_d, din, cin, sd, pd, tflag, split=False):
    """Input gradient on the flagged tiles only; the rest of the returned tensor is NOT initialised."""
    dout, H, W, cout = dz.shape
    dx = torch.empty((din, H, W, cin), dtype=torch.float32, device=dz.device)
    if split:
        with _Timed('conv3d_gather_split', 0):
            bind_amax(split, dz)
            X.check(X.lib.mvx_conv3d_dgrad_tiles_split(X.ptr(dz), X.ptr(wpk_d), X.ptr(dx), din, dout, H, W, cin, cout, sd, pd,
                                                       split_flags(split), X.ptr(tflag), X.stream()), 'mvx_conv3d_dgrad_tiles_split')
        return dx
    global EXEC_STAGES
    counter = None
    if KERNEL_TIMERS is not None:
        if EXEC_STAGES is None:
            EXEC_STAGES = torch.zeros((1,), dtype=torch.int64, device=dz.device)
        counter = EXEC_STAGES
    with _Timed('conv3d_gather_tiles', conv_flops(din, dout, H, W, cout, cin, sd, pd, True) if KERNEL_TIMERS is not None else 0):   # dense-equivalent
        X.check(X.lib.mvx_conv3d_dgrad_tiles(X.ptr(dz), X.ptr(wpk_d), X.ptr(dx), din, dout, H, W, cin, cout, sd, pd,
                                             X.ptr(tflag), X.ptr(counter), X.ptr(_work_counter(dz.device)), X.stream()),
                'mvx_conv3d_dgrad_tiles')
    return dx


def bn_relu_backward_tiles(dyhat, y, mi, bg, plane_grad_sums, dbias_out=None, want_inactive_sums=False):
    """BatchNorm+ReLU backward of a layer output with Background ``bg``; dyhat valid on bg.bflag tiles only.
    Returns (dz valid on those tiles only, dbias[, sums of dz over the other tiles (planes, C)])."""
    D, H, W, C = y.shape
    dz = torch.empty_like(y)
    if dbias_out is not None:
        db, flags = dbias_out, FLAG_ACCUMULATE
    else:
        db, flags = torch.empty((C,), dtype=torch.float32, device=y.device), 0
    ws = workspace(X.lib.mvx_bn_relu_backward_tiles_workspace_bytes(D, H, W, C), y.device, 'bn_tiles')
    inact = torch.empty((D, C), dtype=torch.float32, device=y.device) if want_inactive_sums else None
    amax = new_amax(y.device)                            # max |dz| over the written tiles (zeroed by the call)
    with _timed_bytes('bn_relu_backward_tiles', 0):
        X.check(X.lib.mvx_bn_relu_backward_tiles_frames(X.ptr(dyhat), X.ptr(y), X.ptr(mi), X.ptr(bg.c), X.ptr(bg.y_bg),
                                                        X.ptr(plane_grad_sums), X.ptr(bg.bflag), D, H, W, C, X.ptr(dz), X.ptr(db),
                                                        X.ptr(inact), X.ptr(amax), flags, X.ptr(ws), ws.numel(), 1, X.stream()),
                'mvx_bn_relu_backward_tiles_frames')
    tag_amax(dz, amax)
    db = None if dbias_out is not None else db
    return (dz, db, inact) if want_inactive_sums else (dz, db)


def conv3d_forward_bg(x, wpk, bias, cout, sd, pd, bg_in, out_mask, bg_pre, relu=True, want_stats=True, finalize_eps=None,
                      split=False):
    """finalize_eps: also form the BatchNorm mean / inverse std in the kernel (returns (y, mean_inv) then).
    split=True: bf16x3 kernel (wpk from conv3d_pack(..., split=True)); the statistics are finalised separately."""
    global EXEC_STAGES
    din, H, W, cin = x.shape
    dout = conv_out_depth(din, sd, pd)
    out = torch.empty((dout, H, W, cout), dtype=torch.float32, device=x.device)
    stats, fz = _acc_f64((STATS_REPLICAS, 2, cout), x.device) if want_stats else (None, 0)
    flags = (FLAG_RELU if relu else 0) | fz | split_flags(split)
    if split:
        with _Timed('conv3d_gather_split', conv_flops(dout, din, H, W, cin, cout, sd, pd) if KERNEL_TIMERS is not None else 0):
            X.check(X.lib.mvx_conv3d_forward_bg_split(X.ptr(x), X.ptr(wpk), X.ptr(bias), X.ptr(out), X.ptr(stats), din, dout,
                                                      H, W, cin, cout, sd, pd, flags, X.ptr(bg_in.hflag), X.ptr(out_mask),
                                                      X.ptr(bg_pre), 1, X.stream()), 'mvx_conv3d_forward_bg_split')
        if finalize_eps is not None and want_stats:
            return out, bn_finalize(stats, float(dout * H * W), finalize_eps)
        return out, stats
    counter = None
    if KERNEL_TIMERS is not None:
        if EXEC_STAGES is None:
            EXEC_STAGES = torch.zeros((1,), dtype=torch.int64, device=x.device)
        counter = EXEC_STAGES
    fin, mi, npos = None, None, float(dout * H * W)
    if finalize_eps is not None and want_stats:
        fin = _fin_slot(x.device, fz)
        if fin is None:
            fin = torch.zeros((1,), dtype=torch.float64, device=x.device)
        mi = torch.empty((2, cout), dtype=torch.float32, device=x.device)
    with _Timed('conv3d_gather_bg', conv_flops(dout, din, H, W, cin, cout, sd, pd) if KERNEL_TIMERS is not None else 0):   # dense-equivalent
        X.check(X.lib.mvx_conv3d_forward_bg(X.ptr(x), X.ptr(wpk), X.ptr(bias), X.ptr(out), X.ptr(stats), din, dout, H, W,
                                            cin, cout, sd, pd, flags, X.ptr(bg_in.hflag), X.ptr(out_mask), X.ptr(bg_pre),
                                            1, X.ptr(counter), X.ptr(fin), npos, float(finalize_eps or 0.0), X.ptr(mi),
                                            X.ptr(_work_counter(x.device)), X.stream()), 'mvx_conv3d_forward_bg')
    return (out, mi) if mi is not None else (out, stats)


def conv3d_wgrad_bg(x, dz, sd, pd, bg_in, tap_sums=None, accumulate_into=None, split=False):
    din, H, W, cin = x.shape
    dout, _, _, cout = dz.shape
    if accumulate_into is not None:
        dw, flags = accumulate_into, FLAG_ACCUMULATE
    else:
        dw, flags = torch.empty((cout, cin, 3, 3, 3), dtype=torch.float32, device=x.device), 0
    flags |= split_flags(split, True)              # split arithmetic: conv3d_wgrad4s behind the same entry point
    fn_bytes = X.lib.mvx_conv3d_wgrad_bg_workspace_bytes
    fn = X.lib.mvx_conv3d_wgrad_bg
    nbytes = fn_bytes(dout, H, W, cin, cout)
    if tap_sums is None:
        tap_sums = plane_tap_sums(dz)
    # every tensor the side-stream kernels read must be kept alive for that stream (record_stream), not only x and dz:
    # the tap sums and the background description are temporaries of the frame's own stream
    with _wgrad_scope(accumulate_into, x, dz, tap_sums, bg_in.c, bg_in.hflag) as scope:
        ws = workspace(nbytes, x.device, 'wgrad_bg_side' if isinstance(scope, _SideStream) else 'wgrad_bg')
        with _Timed('conv3d_wgrad_bg', 0):
            bind_amax(split, None, dz)
            X.check(fn(X.ptr(x), X.ptr(dz), X.ptr(dw), din, dout, H, W, cin, cout, sd, pd, flags,
                                              X.ptr(bg_in.hflag), X.ptr(bg_in.c), X.ptr(tap_sums), X.ptr(ws), ws.numel(),
                                              X.stream()),
                    'mvx_conv3d_wgrad_bg')
    return None if accumulate_into is not None else dw


# ---------------------------------------------------------------------------------------------
# row-wise linear layers (2-D views with explicit leading dimensions)
# ---------------------------------------------------------------------------------------------
def _ld(t):
    assert t.dim() == 2 and t.stride(1) == 1, 'need a row-major 2-D view'
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


def _vptr(t):
    """Pointer of a (possibly column-sliced) row-major view."""
    if t is None:
        return None
    if not t.is_cuda:
        raise X.MvxHipError('libmvx_hip needs device tensors (no CPU fallback)')
    import ctypes
    return ctypes.c_void_p(t.data_ptr())


def _fin_slot(device, zeroed):
    """u32 counter for the in-kernel BatchNorm finalisation: from the frame arena when the statistics came from it
    (pre-zeroed), else a fresh buffer the C entry clears."""
    if zeroed:
        a = _ARENAS.get(_arena_key(device))
        v = a.take(1) if a is not None else None
        if v is not None:
            return v
    return None


def _work_counter(device):
    """Zeroed u32 slot for the persistent convolution launches: from the frame arena, else a fresh zero."""
    if _ARENA_ON:
        a = _ARENAS.get(_arena_key(device))
        v = a.take(1) if a is not None else None
        if v is not None:
            return v
    return torch.zeros((1,), dtype=torch.float64, device=device)


ROW_SPLIT = tuple(k for k in os.environ.get('MVX_ROW_SPLIT', 'dgrad,wgrad,rpn').split(',') if k)
# ... and under convmath: bf16x6 (fp32-grade arithmetic: every wide row GEMM may use it)
ROW_SPLIT6 = tuple(k for k in os.environ.get('MVX_ROW_SPLIT6', 'fusion,vfe,conv1,rpn,dgrad,wgrad').split(',') if k)


def row_split(tag):
    """Does the wide row GEMM ``tag`` run in bf16x3 arithmetic (csrc/linear_split.hip)?  Under ``convmath: bf16x3``, when an
    entry of ROW_SPLIT is a prefix of the tag.  Tags: 'fusion_<N>x<K>' (forward of a fusion MLP layer), 'vfe', 'conv1',
    'rpn' (forward of the wide VFE layer, of conv1's per-voxel GEMM, of the RPN's deconvolution GEMMs), 'dgrad' (every
    input-gradient row GEMM), 'wgrad' (every weight-gradient row GEMM).

    Default: the input and weight gradients and the RPN's forward GEMMs only.  A forward layer in split arithmetic carries ~5e-6 relative
    error (exact-f32 MFMA: ~8e-7) and the BatchNorm chain behind the FIRST layers of the network amplifies it 5-7x on the
    way to the BEV map.  Measured on one full-size frame against the float64 oracle and on bench.py --convmath bf16x3
    (tools/split_accuracy.py, tools/split_speed.sh -> profiles/r03_split_accuracy.json, r03_split_speed.txt):

        forward rows in bf16x3          BEV map   cls logits  hot frames/s  full frames/s   (input gradients split, weight gradients f32)
        none (convolutions only)        7.6e-6    1.36e-4     443           200
        rpn                             7.6e-6    1.37e-4       (within noise of the row above)
        conv1                           1.3e-5    1.49e-4
        vfe                             1.1e-5    1.82e-4
        fusion_768x768                  3.1e-5    4.5e-4      466           202
        fusion (all five layers)        4.4e-5    6.1e-4
        fusion,vfe,conv1,rpn            4.8e-5    6.3e-4      472           208

    i.e. the 768 -> 768 layer, the only one whose speed matters (81 % of the row-GEMM flops of a step, +5 % frames/s), is
    also the one that costs a factor 3-4 in accuracy of every later map: it stays on the exact-f32 kernel unless
    MVX_ROW_SPLIT asks otherwise.  The weight-gradient GEMMs ('wgrad', linear_wgrad_split: 0.955 -> 0.517 ms for the
    768 x 768 layer over 80 k rows) touch no forward map and are split by default: hot 443 -> 470, full 200 -> 210 frames/s.

    Returns the number of bf16 pieces (0 = exact-f32 kernel, 2 = bf16x3, 3 = bf16x6).  Under ``convmath: bf16x6`` the split is
    fp32-grade (three pieces = the whole f32 mantissa), so every tag of ROW_SPLIT6 -- by default all of them -- uses it."""
    np_ = split_pieces()
    if np_ == 2:
        return 2 if any(tag.startswith(k) for k in ROW_SPLIT) else 0
    if np_ in (3, 4):
        return np_ if any(tag.startswith(k) for k in ROW_SPLIT6) else 0
    return 0


def transposed_weight(w2):
    """Row-major W^T of a (N, K) weight view, cached ON the parameter it views (so it dies with the model) and keyed by the
    parameter's version counter and the view's address: an optimizer step, load_state_dict or .to() makes the copy stale
    and it is made again.  (Not a module-level cache keyed by address: a new model may reuse a freed model's addresses at
    version 0.)  The input-gradient GEMM then reads both operands with 16-byte loads along k -- the transposed-read form of
    the f32 kernel collects its weight tile with 4-byte loads and ran at 0.43 matrix-pipe busy against 0.61 for the forward
    form -- and the split row GEMM reads row-major weights only (csrc/linear_split.hip)."""
    base = w2._base if w2._base is not None else w2
    cache = base.__dict__.setdefault('_mvx_wt', {})
    tag = (base._version, w2.data_ptr())
    hit = cache.get(tuple(w2.shape))
    if hit is None or hit[0] != tag:
        hit = (tag, w2.detach().t().contiguous())
        cache[tuple(w2.shape)] = hit
    return hit[1]


def padded_weight(w2, k):
    """(N, k) copy of a (N, K <= k) weight view with zero columns appended, cached on the parameter like transposed_weight: the
    operand of a row layer whose input rows were written with a pitch of k floats (frames._vfe_forward: 23 -> 24)."""
    if w2.shape[1] == k:
        return w2
    base = w2._base if w2._base is not None else w2
    cache = base.__dict__.setdefault('_mvx_wpad', {})
    tag = (base._version, w2.data_ptr())
    hit = cache.get((tuple(w2.shape), k))
    if hit is None or hit[0] != tag:
        wp = torch.zeros((w2.shape[0], k), dtype=w2.dtype, device=w2.device)
        wp[:, :w2.shape[1]] = w2.detach()
        hit = (tag, wp)
        cache[(tuple(w2.shape), k)] = hit
    return hit[1]


# ---------------------------------------------------------------------------------------------
# Row GEMMs on pre-cut operands (csrc/rowgemm_pre.hip): the operands are PLANES of 16-bit pieces, int16 (pieces, rows, k),
# written by their producers; the GEMM moves them global -> LDS by DMA.  Used for the wide layers whose shape fills its
# 256 x 256 tiles (n a multiple of 256: the 768 -> 768 fusion layer, 70 % of the row-GEMM work of a step) in the bf16x6
# arithmetic; every other call keeps the in-kernel-cut kernels.  MVX_PRECUT=0 switches it off (A/B runs).
# ---------------------------------------------------------------------------------------------
PRECUT = os.environ.get('MVX_PRECUT', '1') != '0'
# The FORWARD on pre-cut operands: its products and their accumulation order are those of linear_fwd_split (y is bit-identical)
# and its BatchNorm sums are formed term by term in f64 from an LDS copy of the tile (csrc/rowgemm_pre.hip), equal to that
# kernel's to f64 rounding -- the four-term f32 partial sums of its first form (1e-9 off: enough to move a mean by an ulp between
# executors that tile the rows differently) are gone, and with them the reason it was opt-in.  MVX_PRECUT_FWD=0: linear_fwd_split.
PRECUT_FWD = os.environ.get('MVX_PRECUT_FWD', '1') != '0'
# Row ranges in which the step's last BatchNorm backward + weight gradient are enqueued (frames.rows_backward): the product of
# range p runs on the side stream beside the apply pass of range p + 1.  1 = one pass, one product: the default, because the
# ranges measured SLOWER (same box, bf16x6, 40 steps: 463.7 / 463.1 frames/s in one part, 460.7 / 460.9 in two, 455.4 / 455.3 in
# four): the main queue's idle end of the step is not an idle chip, the HBM-bound apply pass and the MFMA-bound product slow each
# other down by more than the overlap hides.
TAIL_PARTS = max(1, int(os.environ.get('MVX_TAIL_PARTS', '1')))
# MVX_FLAG_PRE_XCD_STRIPS of mvx_linear_wgrad_pre: every block of a row strip on the same XCD (developer knob, A/B runs)
PRE_XCD_STRIPS = 4096 if os.environ.get('MVX_PRE_XCD', '0') != '0' else 0


def precut_ok(split, rows, K, N):
    return bool(PRECUT and split and int(split) == 3 and N % 256 == 0 and K % 16 == 0 and rows >= 2048 and
                3 * rows * max(K, N) * 2 < (1 << 32))


def split_rows(x, split, scale=1.0):
    """f32 (rows, k) -> planes int16 (pieces, rows, k) (mvx_split_rows).  For tensors whose producer does not write planes."""
    rows, K = x.shape
    flags = split_flags(split, True)
    planes = torch.empty((3 if int(split) == 3 else 2, rows, K), dtype=torch.int16, device=x.device)
    X.check(X.lib.mvx_split_rows(_vptr(x), _ld(x), rows, K, X.ptr(planes), flags, float(scale), X.stream()), 'mvx_split_rows')
    return planes


def weight_planes(w2, split):
    """Planes of a (N, K) weight view, cached ON the parameter it views and keyed by its version counter (see transposed_weight):
    cut once per optimizer step instead of in every workgroup of every launch."""
    base = w2._base if w2._base is not None else w2
    cache = base.__dict__.setdefault('_mvx_wp', {})
    tag = (base._version, w2.data_ptr())
    key = (tuple(w2.shape), int(split))
    hit = cache.get(key)
    if hit is None or hit[0] != tag:
        hit = (tag, split_rows(w2.detach().contiguous(), split))
        cache[key] = hit
    return hit[1]


def linear_forward_pre(x_planes, w_planes, bias, y, stats, row_w, flags, counter, eps, mi, desc, kind):
    """y = [ReLU](x w^T + b) with per-frame BatchNorm sums / finalisation from operand planes (mvx_linear_forward_pre_frames)."""
    _, rows, K = x_planes.shape
    N = w_planes.shape[1]
    with _Timed('linear_fwd', 2.0 * rows * K * N if KERNEL_TIMERS is not None else 0):
        X.check(X.lib.mvx_linear_forward_pre_frames(X.ptr(x_planes), X.ptr(w_planes), X.ptr(bias), _vptr(y), _ld(y), X.ptr(stats),
                                                    X.ptr(row_w), rows, K, N, flags, 1.0, X.ptr(counter), float(eps), X.ptr(mi),
                                                    desc.ref() if desc is not None else None, kind, X.stream()),
                'mvx_linear_forward_pre_frames')


def linear_wgrad_pre(x_planes, dz_planes, accumulate_into=None, rows=None):
    """dW (N, K) (+)= dz^T x from operand planes (mvx_linear_wgrad_pre); on the side stream like linear_wgrad.  ``rows=(lo, hi)``:
    only those rows of the planes (mvx_linear_wgrad_pre_rows)."""
    pieces, plane_rows, K = x_planes.shape
    N = dz_planes.shape[2]
    lo, hi = rows if rows is not None else (0, plane_rows)
    if accumulate_into is not None:
        assert accumulate_into.is_contiguous() and accumulate_into.numel() == N * K
        dw, flags = accumulate_into, FLAG_ACCUMULATE
    else:
        dw, flags = torch.empty((N, K), dtype=torch.float32, device=x_planes.device), 0
    flags |= split_flags(3 if pieces == 3 else 4, True) | PRE_XCD_STRIPS
    nbytes = max(X.lib.mvx_linear_wgrad_pre_workspace_bytes(plane_rows, K, N), X.lib.mvx_linear_wgrad_pre_workspace_bytes(hi - lo, K, N))
    with _wgrad_scope(accumulate_into, x_planes, dz_planes) as scope:
        ws = workspace(nbytes, x_planes.device, 'lwgrad_pre_side' if isinstance(scope, _SideStream) else 'lwgrad_pre')
        with _Timed('linear_wgrad', 2.0 * (hi - lo) * K * N if KERNEL_TIMERS is not None else 0):
            X.check(X.lib.mvx_linear_wgrad_pre_rows(X.ptr(x_planes), X.ptr(dz_planes), X.ptr(dw), plane_rows, lo, hi, K, N, flags,
                                                    1.0, X.ptr(ws), ws.numel(), X.stream()), 'mvx_linear_wgrad_pre_rows')
    return None if accumulate_into is not None else dw


def rows_dgrad(dz, w2, label='linear_dgrad'):
    """dx = dz w2 of a row layer with weight (N, K): in bf16x3 arithmetic (``row_split('dgrad')``) through the cached
    transposed copy, otherwise through the f32 kernel's transposed-weight read."""
    if row_split('dgrad'):
        dx, _ = linear_forward(dz, transposed_weight(w2), None, relu=False, want_stats=False, label=label, split=row_split('dgrad'))
    else:
        dx, _ = linear_forward(dz, w2, None, relu=False, want_stats=False, w_transposed=True, label=label)
    return dx


def linear_forward(x, w, bias, relu=True, want_stats=True, w_transposed=False, row_w=None, out=None, finalize=None,
                   label=None, split=False, foreign=False):
    """x (R,K) view, w (N,K) [or (K,N) if w_transposed] -> y (R,N), stats f64 (2,N) or None.
    ``finalize=(count, eps)``: the BatchNorm mean / inverse std are formed by the kernel's last workgroup
    (mvx_linear_forward_bn); returns (y, mean_inv) then.  ``split``: bf16x3 arithmetic where the shape qualifies
    (MVX_FLAG_SPLIT); ``label``: timer name of this call (default linear_fwd / linear_dgrad by w_transposed)."""
    R, K = x.shape
    N = w.shape[1] if w_transposed else w.shape[0]
    if out is None:
        out = torch.empty((R, N), dtype=torch.float32, device=x.device)
    stats, fz = _acc_f64((STATS_REPLICAS, 2, N), x.device) if want_stats else (None, 0)
    if label == 'linear_dgrad':
        split = grad_split(split, x)
    xfl = 0
    if foreign:                                      # x comes from outside the library: foreign_split
        split, xfl = foreign_split(split, x)
    if split and int(split) == 4:
        guard_fp16_weight(w)
    if finalize is not None and want_stats and R > 0:
        counter = _fin_slot(x.device, fz)
        if counter is None:
            counter = torch.zeros((1,), dtype=torch.float64, device=x.device)
        mi = torch.empty((2, N), dtype=torch.float32, device=x.device)
        bind_amax(split, x)
        X.check(X.lib.mvx_linear_forward_bn(_vptr(x), _ld(x), _vptr(w), _ld(w), int(w_transposed), X.ptr(bias),
                                            _vptr(out), _ld(out), X.ptr(stats), X.ptr(row_w), R, K, N,
                                            (FLAG_RELU if relu else 0) | fz | split_flags(split, True) | xfl, X.ptr(counter), float(finalize[0]),
                                            float(finalize[1]), X.ptr(mi), X.stream()), 'mvx_linear_forward_bn')
        return out, mi
    ws = None
    if bias is None and not relu and not want_stats and K >= 256 and R * N <= (1 << 22):
        ws = workspace(X.lib.mvx_linear_splitk_workspace_bytes(R, N), x.device, 'splitk')
    with _Timed(label or ('linear_dgrad' if w_transposed else 'linear_fwd'), 2.0 * R * K * N if KERNEL_TIMERS is not None else 0):
        bind_amax(split, x)
        X.check(X.lib.mvx_linear_forward(_vptr(x), _ld(x), _vptr(w), _ld(w), int(w_transposed), X.ptr(bias),
                                         _vptr(out), _ld(out), X.ptr(stats), X.ptr(row_w), R, K, N,
                                         (FLAG_RELU if relu else 0) | fz | split_flags(split, True) | xfl,
                                         X.ptr(ws), ws.numel() if ws is not None else 0, X.stream()), 'mvx_linear_forward')
    if finalize is not None and want_stats:      # empty input: no launch happened, finalise the (zero) sums separately
        return out, bn_finalize(stats, finalize[0], finalize[1])
    return out, stats


def linear_wgrad(x, dz, accumulate_into=None, split=None, x_foreign=False):
    """dW (N,K) = dz^T x; accumulate_into: existing contiguous (N,K)-sized gradient buffer to ADD to.  ``split``: bf16x3
    arithmetic (MVX_FLAG_SPLIT, csrc/linear_split.hip linear_wgrad_split); None = as ``row_split('wgrad')`` says."""
    R, K = x.shape
    N = dz.shape[1]
    if accumulate_into is not None:
        assert accumulate_into.is_contiguous() and accumulate_into.numel() == N * K
        dw, flags = accumulate_into, FLAG_ACCUMULATE
    else:
        dw, flags = torch.empty((N, K), dtype=torch.float32, device=x.device), 0
    split = grad_split(row_split('wgrad') if split is None else split, dz)
    if x_foreign:                                    # a foreign input of unknown range: see foreign_split
        split = foreign_split(split, x)[0]
    flags |= split_flags(split, True)
    nbytes = X.lib.mvx_linear_wgrad_workspace_bytes(R, K, N)
    with _wgrad_scope(accumulate_into, x, dz) as scope:
        ws = workspace(nbytes, x.device, 'lwgrad_side' if isinstance(scope, _SideStream) else 'lwgrad')
        with _Timed('linear_wgrad', 2.0 * R * K * N if KERNEL_TIMERS is not None else 0):
            bind_amax(split, x, dz)
            X.check(X.lib.mvx_linear_wgrad(_vptr(x), _ld(x), _vptr(dz), _ld(dz), X.ptr(dw), R, K, N, flags, X.ptr(ws),
                                           ws.numel(), X.stream()), 'mvx_linear_wgrad')
    return None if accumulate_into is not None else dw


# ---------------------------------------------------------------------------------------------
# VFE glue (dense [V][T][C] rows, or compact rows described by a CompactRows)
# ---------------------------------------------------------------------------------------------
class CompactRows:
    """Row layout of one frame in compact form: n_real real rows followed by one padded row per
    voxel (see include/mvx_hip.h, 'Compact rows')."""

    def __init__(self, row_map, rows_sel, n_real, V, T):
        dev = row_map.device
        self.row_map, self.rows_sel = row_map, rows_sel
        self.n_real, self.V, self.T = int(n_real), int(V), int(T)
        self.voff = torch.empty((V,), dtype=torch.int32, device=dev)
        self.vcnt = torch.empty((V,), dtype=torch.int32, device=dev)
        self.row_w = torch.empty((self.n_real + V,), dtype=torch.float32, device=dev)
        X.check(X.lib.mvx_voxel_row_offsets(X.ptr(row_map), V, T, self.n_real, X.ptr(self.voff), X.ptr(self.vcnt),
                                            X.ptr(self.row_w), X.stream()), 'mvx_voxel_row_offsets')

    @property
    def rows(self):
        return self.n_real + self.V

    @property
    def count(self):                     # rows of the dense tensor this layout stands for
        return self.V * self.T


def _rows_args(cr):
    if cr is None:
        return None, None, 0
    return X.ptr(cr.voff), X.ptr(cr.vcnt), cr.n_real


def vfe_bn_max_concat(y, mi, V, T, cr=None):
    C = mi.shape[1]
    out = torch.empty((y.shape[0], 2 * C), dtype=torch.float32, device=y.device)
    am = torch.empty((V, C), dtype=torch.int32, device=y.device)
    vo, vc, nr = _rows_args(cr)
    X.check(X.lib.mvx_vfe_bn_max_concat(X.ptr(y), X.ptr(mi), X.ptr(out), X.ptr(am), V, T, C, vo, vc, nr, X.stream()),
            'mvx_vfe_bn_max_concat')
    return out, am


def vfe_max_concat_backward(g, am, V, T, cr=None):
    C = am.shape[1]
    dyh = torch.empty((g.shape[0], C), dtype=torch.float32, device=g.device)
    vo, vc, nr = _rows_args(cr)
    X.check(X.lib.mvx_vfe_max_concat_backward(X.ptr(g), X.ptr(am), X.ptr(dyh), V, T, C, vo, vc, nr, X.stream()),
            'mvx_vfe_max_concat_backward')
    return dyh


def bn_segment_max(y, mi, V, T, cr=None):
    C = mi.shape[1]
    out = torch.empty((V, C), dtype=torch.float32, device=y.device)
    am = torch.empty((V, C), dtype=torch.int32, device=y.device)
    vo, vc, nr = _rows_args(cr)
    X.check(X.lib.mvx_bn_segment_max(X.ptr(y), X.ptr(mi), X.ptr(out), X.ptr(am), V, T, C, vo, vc, nr, X.stream()),
            'mvx_bn_segment_max')
    return out, am


def segment_max_backward(dfeat, am, V, T, cr=None):
    C = am.shape[1]
    rows = cr.rows if cr is not None else V * T
    dyh = torch.empty((rows, C), dtype=torch.float32, device=dfeat.device)
    vo, vc, nr = _rows_args(cr)
    X.check(X.lib.mvx_segment_max_backward(X.ptr(dfeat), X.ptr(am), X.ptr(dyh), V, T, C, vo, vc, nr, X.stream()),
            'mvx_segment_max_backward')
    return dyh


def vfe_compact_input(vox2d, imfeat_c, cr):
    F = imfeat_c.shape[1]
    out = torch.empty((cr.rows, 7 + F), dtype=torch.float32, device=vox2d.device)
    X.check(X.lib.mvx_vfe_compact_input(X.ptr(vox2d), vox2d.shape[1], X.ptr(cr.rows_sel), X.ptr(imfeat_c), F,
                                        cr.n_real, cr.V, X.ptr(out), X.stream()), 'mvx_vfe_compact_input')
    return out


def vfe_compact_input_backward(g, F, cr):
    d = torch.empty((cr.n_real + 1, F), dtype=torch.float32, device=g.device)
    scratch = torch.empty((F,), dtype=torch.float64, device=g.device)
    X.check(X.lib.mvx_vfe_compact_input_backward(X.ptr(g), F, cr.n_real, cr.V, X.ptr(d), X.ptr(scratch), X.stream()),
            'mvx_vfe_compact_input_backward')
    return d


# ---------------------------------------------------------------------------------------------
# point <-> image fusion
# ---------------------------------------------------------------------------------------------
def row_compact_map(vox2d):
    """vox2d (R, vc) -> row_map i32 (R,), rows_sel i32 (R,), n_real i32 (1,) (all on the device)."""
    R, vc = vox2d.shape
    dev = vox2d.device
    row_map = torch.empty((R,), dtype=torch.int32, device=dev)
    rows_sel = torch.empty((R,), dtype=torch.int32, device=dev)
    n_real = torch.empty((1,), dtype=torch.int32, device=dev)
    nbytes = X.lib.mvx_row_compact_workspace_bytes(R)
    ws = workspace(nbytes, dev, 'compact')
    X.check(X.lib.mvx_row_compact_map(X.ptr(vox2d), vc, R, X.ptr(row_map), X.ptr(rows_sel), X.ptr(n_real),
                                      X.ptr(ws), ws.numel(), X.stream()), 'mvx_row_compact_map')
    return row_map, rows_sel, n_real


def feature_sample(vox2d, feats_cl, imsize_hw, eps, out, row_map=None, rows_sel=None, n_real=None):
    """vox2d (R, vc) modified in place; feats_cl: list of channels-last (H, W, C) maps.  With ``rows_sel`` and
    ``n_real`` (from row_compact_map, which already zeroed the padding rows) only the real rows are visited."""
    import ctypes
    R, vc = vox2d.shape
    L = len(feats_cl)
    C = feats_cl[0].shape[2]
    ptrs = (ctypes.c_void_p * L)(*[f.data_ptr() for f in feats_cl])
    hw = (ctypes.c_int32 * (2 * L))(*[int(v) for f in feats_cl for v in f.shape[:2]])
    status = torch.zeros((1,), dtype=torch.int32, device=vox2d.device)
    for f in feats_cl:
        assert f.is_contiguous() and f.dtype == torch.float32 and f.shape[2] == C
    # algorithmic bytes: 4 taps x L levels x C floats gathered + L*C floats written per sampled row, + the voxel rows
    nrows = out.shape[0]
    if rows_sel is not None and n_real is not None:
        with _timed_bytes('feature_sample', nrows * L * C * 4 * 5 + int(n_real) * vc * 4):
            X.check(X.lib.mvx_feature_sample_rows(X.ptr(vox2d), vc, X.ptr(rows_sel), int(n_real), ptrs, hw, L, C,
                                                  float(imsize_hw[0]), float(imsize_hw[1]), float(eps), X.ptr(out),
                                                  X.ptr(status), X.stream()), 'mvx_feature_sample_rows')
        return status
    with _timed_bytes('feature_sample', nrows * L * C * 4 * 5 + R * vc * 4):
        X.check(X.lib.mvx_feature_sample(X.ptr(vox2d), vc, R, X.ptr(row_map), ptrs, hw, L, C,
                                         float(imsize_hw[0]), float(imsize_hw[1]), float(eps), X.ptr(out),
                                         X.ptr(status), X.stream()), 'mvx_feature_sample')
    return status


def expand_rows(compact, row_map, pad_row):
    R = row_map.shape[0]
    C = compact.shape[1]
    out = torch.empty((R, C), dtype=torch.float32, device=compact.device)
    X.check(X.lib.mvx_expand_rows(X.ptr(compact), X.ptr(row_map), int(pad_row), X.ptr(out), R, C, X.stream()),
            'mvx_expand_rows')
    return out


def expand_rows_backward(g, row_map, pad_row, n_compact):
    R, C = g.shape
    dc = torch.empty((n_compact, C), dtype=torch.float32, device=g.device)
    scratch = torch.empty((C,), dtype=torch.float64, device=g.device)
    X.check(X.lib.mvx_expand_rows_backward(X.ptr(g), X.ptr(row_map), int(pad_row), X.ptr(dc), X.ptr(scratch),
                                           R, C, X.stream()), 'mvx_expand_rows_backward')
    return dc


# ---------------------------------------------------------------------------------------------
# crops and projection
# ---------------------------------------------------------------------------------------------
def _host_f64(a, n):
    import ctypes
    import numpy as np
    if a is None:
        return None, None
    arr = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))
    assert arr.size == n
    return arr, arr.ctypes.data_as(ctypes.c_void_p)


def crop_points(pcd, n_in=None, range6=None, bounds_f32=False, cam_from_velo=None, p2=None, imsize_wh=(0.0, 0.0),
                math_f32=False, want_index=False):
    """pcd f32 (F, cap, ncol) on the GPU -> (out (F, cap, ncol), n_out i32 (F,), src_index or None)."""
    F, cap, ncol = pcd.shape
    dev = pcd.device
    out = torch.empty_like(pcd)
    n_out = torch.empty((F,), dtype=torch.int32, device=dev)
    src = torch.empty((F, cap), dtype=torch.int32, device=dev) if want_index else None
    ws = workspace(X.lib.mvx_crop_workspace_bytes(F, cap), dev, 'crop')
    k1, r_ptr = _host_f64(range6, 6)
    k2, m_ptr = _host_f64(cam_from_velo, 16)
    k3, p_ptr = _host_f64(p2, 16)
    X.check(X.lib.mvx_crop_points(X.ptr(pcd), X.ptr(n_in), F, cap, ncol, r_ptr, int(bounds_f32), m_ptr, p_ptr,
                                  float(imsize_wh[0]), float(imsize_wh[1]), int(math_f32), X.ptr(out), X.ptr(n_out),
                                  X.ptr(src), X.ptr(ws), ws.numel(), X.stream()), 'mvx_crop_points')
    return out, n_out, src


def crop_project(pcd, n_in, range6, cam_from_velo64, p2_64, imsize_wh, cam_from_velo32, p2_32, cap_out):
    """Raw clouds f32 (F, cap, 4) -> (points6 (F, cap_out, 6) = [x y z r row col], n_out i32 (F,)): crop + cropToSight
    (numpy-path f64 masks) and the f32 projection of train.py:31-34 in one compaction pass."""
    F, cap, ncol = pcd.shape
    dev = pcd.device
    out = torch.empty((F, cap_out, ncol + 2), dtype=torch.float32, device=dev)
    n_out = torch.empty((F,), dtype=torch.int32, device=dev)
    ws = workspace(X.lib.mvx_crop_project_workspace_bytes(F, cap), dev, 'crop')
    k1, r_ptr = _host_f64(range6, 6)
    k2, m_ptr = _host_f64(cam_from_velo64, 16)
    k3, p_ptr = _host_f64(p2_64, 16)
    k4, m32_ptr = _host_f64(cam_from_velo32, 16)
    k5, p32_ptr = _host_f64(p2_32, 16)
    with _timed_bytes('crop_project', F * cap * ncol * 4 * 2 + F * cap_out * (ncol + 2) * 4):
        X.check(X.lib.mvx_crop_project_points(X.ptr(pcd), X.ptr(n_in), F, cap, ncol, r_ptr, m_ptr, p_ptr, float(imsize_wh[0]),
                                              float(imsize_wh[1]), m32_ptr, p32_ptr, X.ptr(out), int(cap_out), X.ptr(n_out),
                                              X.ptr(ws), ws.numel(), X.stream()), 'mvx_crop_project_points')
    return out, n_out


def lidar2img(pcd2d, cam_from_velo, p2, math_f32=True, out=None, col_offset=0, swap_rc=False, want_z=False):
    """pcd2d f32 (n, ncol) -> out (n, 2) = (u, v) [or written into columns col_offset.. of `out`]."""
    n, ncol = pcd2d.shape
    if out is None:
        out = torch.empty((n, 2), dtype=torch.float32, device=pcd2d.device)
    z = torch.empty((n,), dtype=torch.float32, device=pcd2d.device) if want_z else None
    k2, m_ptr = _host_f64(cam_from_velo, 16)
    k3, p_ptr = _host_f64(p2, 16)
    X.check(X.lib.mvx_lidar2img(X.ptr(pcd2d), ncol, n, m_ptr, p_ptr, int(math_f32), _vptr(out), _ld(out), col_offset,
                                int(swap_rc), X.ptr(z), X.stream()), 'mvx_lidar2img')
    return out, z


# ---------------------------------------------------------------------------------------------
# input-sparse convolution as voxel GEMMs + index-grid gathers
# ---------------------------------------------------------------------------------------------
def index_grid(coords, dhw):
    D, H, W = dhw
    buf = torch.empty((X.lib.mvx_index_grid_bytes(D, H, W) // 4,), dtype=torch.int32, device=coords.device)
    status = torch.zeros((1,), dtype=torch.int32, device=coords.device)
    X.check(X.lib.mvx_index_grid(X.ptr(coords), coords.shape[0], D, H, W, X.ptr(buf), X.ptr(status), X.stream()),
            'mvx_index_grid')
    return buf, status


def sparse_conv_output(P, idx_grid, dhw, bias, cout, sd, pd, relu=True, want_stats=True):
    din, H, W = dhw
    dout = conv_out_depth(din, sd, pd)
    out = torch.empty((dout, H, W, cout), dtype=torch.float32, device=P.device)
    stats, fz = _acc_f64((STATS_REPLICAS, 2, cout), P.device) if want_stats else (None, 0)
    with _timed_bytes('sparse_conv_output', out.numel() * 4 + din * H * W * 4):     # dense output written + index grid read
        X.check(X.lib.mvx_sparse_conv_output(X.ptr(P), X.ptr(idx_grid), X.ptr(bias), X.ptr(out), X.ptr(stats), din, dout,
                                             H, W, cout, sd, pd, (FLAG_RELU if relu else 0) | fz, X.stream()),
                'mvx_sparse_conv_output')
    return out, stats


def sparse_conv_gather_dz(dz, coords, din, sd, pd):
    dout, H, W, cout = dz.shape
    V = coords.shape[0]
    G = torch.empty((V, 27 * cout), dtype=torch.float32, device=dz.device)
    X.check(X.lib.mvx_sparse_conv_gather_dz(X.ptr(dz), X.ptr(coords), V, X.ptr(G), din, dout, H, W, cout, sd, pd,
                                            X.stream()), 'mvx_sparse_conv_gather_dz')
    return tag_amax(G, amax_of(dz))                     # G's rows are rows of dz


def sink_of(param):
    """The parameter's existing gradient buffer if direct accumulation is enabled, else None.  Used for
    WEIGHT gradients, whose kernels run on the side stream when ASYNC_WGRAD is on."""
    if GRAD_SINK and param is not None and param.grad is not None and param.grad.is_contiguous():
        return param.grad
    return None


class _BiasStage:
    """Per-lane staging copy of the flat gradient bucket for BIAS gradients.  With ASYNC_WGRAD every write to a
    .grad buffer must go through the one side stream (frames are in flight on several streams), but bias
    gradients are finished on the frame's own stream: they are accumulated here (single writer: the lane) and
    added to the bucket ONCE per frame on the side stream.  Two buffers alternate, so that clearing one for
    frame f+2 only waits for the flush of frame f."""

    def __init__(self, flat):
        self.flat = flat
        self.bufs = [torch.zeros_like(flat), torch.zeros_like(flat)]
        self.events = [None, None]
        self.touched = [{}, {}]            # per buffer: offset -> length of the slices handed out since its last clear
        self.turn = 0
        self.base = flat.data_ptr()

    def _slices(self, t, turn):
        return [t[o:o + n] for o, n in sorted(self.touched[turn].items())]

    def begin(self):
        self.turn ^= 1
        ev = self.events[self.turn]
        if ev is not None:
            torch.cuda.current_stream(self.flat.device).wait_event(ev)
        # only the bias slices are ever written: clear and flush those (one multi-tensor kernel each), not the whole
        # bucket (two full-bucket passes per frame to move a few KB)
        if self.touched[self.turn]:
            torch._foreach_zero_(self._slices(self.bufs[self.turn], self.turn))
            self.touched[self.turn] = {}

    def view_for(self, grad):
        off = (grad.data_ptr() - self.base) // 4
        n = grad.numel()
        if off < 0 or off + n > self.flat.numel():
            return None
        self.touched[self.turn][off] = n
        return self.bufs[self.turn][off:off + n]

    def flush(self):
        buf = self.bufs[self.turn]
        with _SideStream(buf) as sc:
            if self.touched[self.turn]:
                torch._foreach_add_(self._slices(self.flat, self.turn), self._slices(buf, self.turn))
            ev = torch.cuda.Event()
            ev.record(sc.side)
        self.events[self.turn] = ev


_BIAS_STAGES = {}        # (device, raw lane stream) -> _BiasStage
_BIAS_STAGE_ON = False


def bias_stage_begin(device, flat_grad):
    """Start a frame on the current stream with bias gradients staged (see _BiasStage)."""
    global _BIAS_STAGE_ON
    key = _arena_key(device)
    st = _BIAS_STAGES.get(key)
    if st is None or st.flat is not flat_grad:
        st = _BiasStage(flat_grad)
        _BIAS_STAGES[key] = st
    st.begin()
    _BIAS_STAGE_ON = True


def bias_stage_flush(device):
    st = _BIAS_STAGES.get(_arena_key(device))
    if st is not None:
        st.flush()


def bias_stage_end():
    global _BIAS_STAGE_ON
    _BIAS_STAGE_ON = False


def bias_sink_of(param):
    """Bias gradients are finished on the stream of the backward pass.  Without ASYNC_WGRAD they go straight
    into .grad; with it, into the lane's staging copy when the pipeline opened one (bias_stage_begin), else they
    are produced as a temporary and added by ``accumulate_grad`` on the side stream."""
    if not ASYNC_WGRAD:
        return sink_of(param)
    if _BIAS_STAGE_ON and GRAD_SINK and param is not None and param.grad is not None:
        st = _BIAS_STAGES.get(_arena_key(param.device))
        if st is not None:
            return st.view_for(param.grad)
    return None


def accumulate_grad(param, g):
    """Add ``g`` into ``param.grad`` without autograd when sinks are on (side stream if ASYNC_WGRAD);
    returns what the autograd node should return for this parameter (None, or g itself)."""
    if g is None:
        return None
    tgt = sink_of(param)
    if tgt is None:
        return g
    if ASYNC_WGRAD:
        with _SideStream(g):
            tgt.add_(g.view_as(tgt))
    else:
        tgt.add_(g.view_as(tgt))
    return None


# ---------------------------------------------------------------------------------------------
# target assignment and loss (csrc/anchors.hip, csrc/loss.hip)
# ---------------------------------------------------------------------------------------------
def bbox_pairwise(b1, b2, want_iou):
    """b1 (N,4,2), b2 (M,4,2) f32 BEV corners on the GPU -> (N,M) IoU or intersection area."""
    n, m = b1.shape[0], b2.shape[0]
    out = torch.empty((n, m), dtype=torch.float32, device=b1.device)
    X.check(X.lib.mvx_bbox_pairwise(X.ptr(b1), n, X.ptr(b2), m, int(want_iou), X.ptr(out), X.stream()), 'mvx_bbox_pairwise')
    return out


def classify_anchors(gts, anchor_bevs, nls, nws, neg_thr, pos_thr, radius, cap=None):
    """Device tensors in, capacity-sized device tensors out: (pos_idx i64 (3,cap), neg_idx i64 (3,cap), gi i64 (cap,),
    counts i32 (2,), status i32 (1,))."""
    G = gts.shape[0]
    L, W, A = anchor_bevs.shape[:3]
    dev = anchor_bevs.device
    wn = 2 * radius + 1
    cap = max(1, G * A * wn * wn) if cap is None else int(cap)
    pos = torch.empty((3, cap), dtype=torch.int64, device=dev)
    neg = torch.empty((3, cap), dtype=torch.int64, device=dev)
    gi = torch.empty((cap,), dtype=torch.int64, device=dev)
    counts = torch.empty((2,), dtype=torch.int32, device=dev)
    status = torch.zeros((1,), dtype=torch.int32, device=dev)
    ws = workspace(X.lib.mvx_classify_anchors_workspace_bytes(G, A, radius), dev, 'anchors')
    X.check(X.lib.mvx_classify_anchors(X.ptr(gts), G, X.ptr(anchor_bevs), L, W, A, X.ptr(nls), X.ptr(nws), float(neg_thr),
                                       float(pos_thr), int(radius), X.ptr(pos), X.ptr(neg), X.ptr(gi), cap, X.ptr(counts),
                                       X.ptr(status), X.ptr(ws), ws.numel(), X.stream()), 'mvx_classify_anchors')
    return pos, neg, gi, counts, status


def classify_anchors_frames(gts, gt_off, anchor_bevs, nls, nws, neg_thr, pos_thr, radius):
    """The frames of a step in one walk launch: gts (G_total,4,2) of all frames back to back, ``gt_off`` host list (F+1).
    Returns (pos_idx i64 (F,3,cap), neg_idx i64 (F,3,cap), gi i64 (F,cap), counts i32 (F,2), status i32 (1,))."""
    import ctypes
    F = len(gt_off) - 1
    L, W, A = anchor_bevs.shape[:3]
    dev = anchor_bevs.device
    wn = 2 * radius + 1
    gmax = max(gt_off[f + 1] - gt_off[f] for f in range(F))
    cap = max(1, gmax * A * wn * wn)
    pos = torch.empty((F, 3, cap), dtype=torch.int64, device=dev)
    neg = torch.empty((F, 3, cap), dtype=torch.int64, device=dev)
    gi = torch.empty((F, cap), dtype=torch.int64, device=dev)
    counts = torch.empty((F, 2), dtype=torch.int32, device=dev)
    status = torch.zeros((1,), dtype=torch.int32, device=dev)
    ws = workspace(X.lib.mvx_classify_anchors_workspace_bytes(int(gt_off[-1]), A, radius), dev, 'anchors')
    off = (ctypes.c_int32 * (F + 1))(*[int(v) for v in gt_off])
    X.check(X.lib.mvx_classify_anchors_frames(X.ptr(gts), off, F, X.ptr(anchor_bevs), L, W, A, X.ptr(nls), X.ptr(nws),
                                              float(neg_thr), float(pos_thr), int(radius), X.ptr(pos), X.ptr(neg), X.ptr(gi), cap,
                                              X.ptr(counts), X.ptr(status), X.ptr(ws), ws.numel(), X.stream()),
            'mvx_classify_anchors_frames')
    return pos, neg, gi, counts, status


def voxel_loss(score, reg, pos_idx, neg_idx, gi, n_pos, n_neg, gts, anchors, A, a, b, eps, want_grads=True,
               dscore_out=None, dreg_out=None):
    """score (L,W,A) / reg (L,W,7A) f32 views with arbitrary strides; pos_idx / neg_idx i64 (3,cap) or None.
    Returns (losses f32 (2,), dscore, dreg) -- the gradients share the inputs' memory layout, or are written into the
    given views ``dscore_out`` / ``dreg_out`` (dreg_out ZERO on entry)."""
    L, W = score.shape[0], score.shape[1]
    dev = score.device
    losses = torch.empty((2,), dtype=torch.float32, device=dev)
    scratch = torch.empty((4,), dtype=torch.float64, device=dev)
    dscore = dscore_out
    if want_grads and dscore is None:
        dscore = torch.empty_strided(score.shape, score.stride(), dtype=torch.float32, device=dev)
    dreg = None
    if want_grads and reg is not None:
        dreg = dreg_out
        if dreg is None:
            dreg = torch.empty_strided(reg.shape, reg.stride(), dtype=torch.float32, device=dev)
            dreg.zero_()
    ss = score.stride()
    rs = reg.stride() if reg is not None else (0, 0, 0)
    ds = dscore.stride() if dscore is not None else (0, 0, 0)
    dr = dreg.stride() if dreg is not None else (0, 0, 0)
    X.check(X.lib.mvx_voxel_loss(_vptr(score), ss[0], ss[1], ss[2], _vptr(reg), rs[0], rs[1], rs[2],
                                 X.ptr(pos_idx), pos_idx.shape[1] if pos_idx is not None else 0,
                                 X.ptr(neg_idx), neg_idx.shape[1] if neg_idx is not None else 0, X.ptr(gi), None,
                                 int(n_pos), int(n_neg), X.ptr(gts), gts.shape[1] if gts is not None else 7,
                                 X.ptr(anchors), L, W, int(A), float(a), float(b), float(eps),
                                 _vptr(dscore), ds[0], ds[1], ds[2], _vptr(dreg), dr[0], dr[1], dr[2],
                                 X.ptr(losses), X.ptr(scratch), X.stream()), 'mvx_voxel_loss')
    return losses, dscore, dreg
