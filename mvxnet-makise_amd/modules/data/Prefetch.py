"""Batch preparation ahead of the step: what the reference does with a ProcessPoolExecutor (train.py:26-49 run in worker
processes, train.py:185-187) done the MI355X way -- ONE worker thread with its own HIP stream prepares batch k+1 while the
GPU runs step k:

    worker, host:   points of the B frames -> page-locked staging buffers (a few sets, used in turn), the shuffle
                    permutations drawn with np.random (the reference's RNG, train.py -> Preprocessing.py:86);
    worker, stream: ONE async H2D copy per buffer (DMA engines: no compute unit involved) and classifyAnchors of all frames
                    (train.py:46: two kernels and the one host read of the list lengths, which waits for THIS stream only
                    instead of draining the training stream), an event;
    main:           waits for the event, then lidar2Img + (row, col) swap (train.py:31-34) and the FPN maps on the
                    training stream.

Only the copies and the target assignment run beside the step: a first version that also put the per-frame projection
and the FPN stand-in (about 25 small launches per batch) on the loader stream was SLOWER than no prefetching at all (82 vs
138 frames/s on the GPU box): every small kernel of a concurrent stream queues behind the persistent convolution launches
that hold all CUs for ~0.7 ms each, so the loader needed 49 ms per batch of a 28 ms step.

The worker is a thread, not a process: everything heavy it does (memcpy into pinned memory, HIP calls, numpy's shuffle)
releases the GIL, and the prepared tensors are device memory that a process pool would have to ship back through IPC.
``PrefetchLoader`` is an iterator over ``(FrameBatch, targets)``; ``depth`` batches are kept ready."""
import queue
import threading

import numpy as np
import torch

import modules.config as cfg
from modules import Calc, _hip


class _Staging:
    """Page-locked host buffers of one batch slot (reused: allocating pinned memory per step costs more than the copies)."""

    def __init__(self, B, cap):
        self.points = torch.zeros((B, cap, 6), dtype=torch.float32).pin_memory()      # x y z r | row col (filled on the GPU)
        self.perms = torch.zeros((B, cap), dtype=torch.int32).pin_memory()
        self.n = torch.zeros((B,), dtype=torch.int32).pin_memory()
        self.free = None            # event recorded after the H2D copies that read this slot


class PrefetchLoader:
    def __init__(self, groups, names_of, device, anchor_bevs, fpn_fn, cap_points, depth=2, priority=-1):
        """``groups``: iterable of lists of frames as ``modules.data.Load.createDataset`` returns them; ``names_of(frame)``:
        the frame's name (for ``fpn_fn``); ``fpn_fn(name, device)``: the frame's FPN maps (the frozen extractor or its
        stand-in); ``cap_points``: point capacity per frame of the resident batch."""
        self.groups = iter(groups)
        self.names_of, self.device, self.anchor_bevs, self.fpn_fn = names_of, device, anchor_bevs, fpn_fn
        self.cap = int(cap_points)
        self.stream = torch.cuda.Stream(device=device, priority=priority)
        self.q = queue.Queue(maxsize=max(1, depth))
        self.slots = {}
        self.turn = 0
        self.error = None
        self.stop = False
        self.stats = {'batches': 0, 'prepare_s': 0.0, 'host_fill_s': 0.0, 'targets_s': 0.0, 'consumer_wait_s': 0.0}
        self.thread = threading.Thread(target=self._work, name='mvx-prefetch', daemon=True)
        self.thread.start()

    # ---- worker thread
    def _slot(self, B, cap):
        key = (B, cap, self.turn)
        self.turn = (self.turn + 1) % (self.q.maxsize + 2)       # more slots than batches that can be in flight
        st = self.slots.get(key)
        if st is None:
            st = _Staging(B, cap)
            self.slots[key] = st
        elif st.free is not None:
            st.free.synchronize()                                # its previous copies have left the host buffers
        return st

    def _prepare(self, group):
        import time
        t0 = time.perf_counter()
        dev = self.device
        B = len(group)
        cap = max(self.cap, max(d[0].shape[0] for d in group))
        st = self._slot(B, cap)
        for k, d in enumerate(group):
            P = d[0].shape[0]
            st.points[k, :P, :4] = torch.from_numpy(np.ascontiguousarray(d[0], dtype=np.float32))
            a = np.arange(P, dtype=np.int32)
            np.random.shuffle(a)                                 # the reference's sampling RNG (Preprocessing.py:86)
            st.perms[k, :P] = torch.from_numpy(a)
            st.n[k] = P
        t1 = time.perf_counter()
        with torch.cuda.stream(self.stream):
            pts6 = st.points.to(dev, non_blocking=True)
            perms = st.perms.to(dev, non_blocking=True)
            n = st.n.to(dev, non_blocking=True)
            st.free = torch.cuda.Event()
            st.free.record(self.stream)
            boxes = [(d[4], d[3][:, [0, 1]]) if (d[4] is not None and d[4].shape[0] != 0) else None for d in group]
            lists = Calc.classifyAnchorsFrames(boxes, self.anchor_bevs, cfg.velorange, 0.45, 0.6)      # one pass, one host read
            targets = [None if t is None else (t[0], t[1], t[2], d[3].to(dev)) for t, d in zip(lists, group)]
            ev = torch.cuda.Event()
            ev.record(self.stream)
        t2 = time.perf_counter()
        self.stats['batches'] += 1
        self.stats['host_fill_s'] += t1 - t0
        self.stats['targets_s'] += t2 - t1
        self.stats['prepare_s'] += t2 - t0
        return (pts6, perms, n, group), targets, ev

    def _work(self):
        try:
            torch.cuda.set_device(self.device)
            for group in self.groups:
                if self.stop:
                    break
                self.q.put(self._prepare(group) if group else (None, None, None))
        except BaseException as e:                               # noqa: BLE001 -- handed to the consumer
            self.error = e
        finally:
            self.q.put(None)

    # ---- consumer
    def close(self):
        """Stop preparing (the consumer left the loop early): lets the worker run out and drops what it had ready."""
        self.stop = True
        while self.thread.is_alive():
            try:
                self.q.get(timeout=0.05)
            except queue.Empty:
                pass
        self.thread.join()

    def __iter__(self):
        return self

    def __next__(self):
        import time
        t0 = time.perf_counter()
        item = self.q.get()
        self.stats['consumer_wait_s'] += time.perf_counter() - t0
        if item is None:
            if self.error is not None:
                raise self.error
            raise StopIteration
        raw, targets, ev = item
        if raw is None:
            return None, None
        from modules.data.Preprocessing import _calib_products
        from modules.pipeline import FrameBatch
        pts6, perms, n, group = raw
        main = torch.cuda.current_stream(self.device)
        main.wait_event(ev)
        for t in (pts6, perms, n):
            t.record_stream(main)
        for t in targets:
            if t is not None:
                for x in tuple(t[0]) + tuple(t[1]) + (t[2], t[3]):
                    if isinstance(x, torch.Tensor) and x.is_cuda:
                        x.record_stream(main)
        fpn = []
        for k, d in enumerate(group):                            # on the training stream: projection and the FPN maps
            P = d[0].shape[0]
            m, p2 = _calib_products(d[5], True)
            _hip.lidar2img(pts6[k, :P], m, p2, math_f32=True, out=pts6[k, :P], col_offset=4, swap_rc=True)    # reads x y z, writes cols 4:6
            fpn.append(self.fpn_fn(self.names_of(d), self.device))
        return FrameBatch(pts6, perms, n, fpn), targets
