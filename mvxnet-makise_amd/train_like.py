"""A training loop with the call sequence of the reference's train.py (train.py:26-49 per-frame preparation, :51-180
loop, checkpoints and resume), running on this package's HIP modules.

    python train_like.py <dataroot> [-n EPOCHS] [-r LAST] [--steps K] [--mode module|fast] [--frames B] [--synthetic N]

What is the same as train.py: createDataset -> createAnchors / bbox3d2bev -> MVXNet, VoxelLoss, AdamW(lr 1e-3, eps) ->
per frame lidar2Img + (row, col) swap + group + classifyAnchors -> forward -> clsLoss (+ regLoss) -> backward -> step ->
running loss statistics -> per-epoch ``checkpoints/epoch{n}.pkl`` / ``epoch{n}_opt.pkl`` and ``-r`` resume.
What is not there: the GT-paste augmentation (modules/augment: needs the KINS-based gtdatabase, OpenCV and numba -- CPU
label preparation outside the hot path) and the frozen torchvision extractor when torchvision is not installed (FPN maps
are then deterministic synthetic tensors per frame; with torchvision the image goes through the real extractor).

--mode module : the reference's own interface, one frame at a time: ``model(voxel, img, idx, [calib], imsize)``.
--mode fast   : B frames per step through the frame-set executor (modules/frames.py: one launch per layer for all
                frames), RPN + VoxelLoss per frame (per-frame BatchNorm statistics, as B reference forwards), one AdamW
                step per B frames on the summed gradient / B; data-parallel over ranks when launched with torchrun.
"""
import argparse
import os
import random
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('dataroot', nargs='?', default=None)
    ap.add_argument('-n', '--numepochs', type=int, default=1)
    ap.add_argument('-r', '--resume', type=int, default=0, dest='lastiter')
    ap.add_argument('--steps', type=int, default=0, help='stop after this many optimizer steps (0 = whole epochs)')
    ap.add_argument('--mode', choices=['module', 'fast'], default='module')
    ap.add_argument('--frames', type=int, default=4, help='frames per step in --mode fast')
    ap.add_argument('--synthetic', type=int, default=0, help='write a synthetic KITTI tree with this many frames into dataroot first')
    ap.add_argument('--points', type=int, default=20000)
    ap.add_argument('--checkpoints', default='./checkpoints')
    ap.add_argument('--need-crop', action='store_true', help='read training/velodyne and crop on the GPU (cropdata.py on the fly)')
    ap.add_argument('--quiet', action='store_true')
    ap.add_argument('--prefetch', action='store_true',
                    help='--mode fast: prepare batch k+1 on a worker thread + stream (modules/data/Prefetch.py); off by default: with '
                         'the preparation on the GPU the loop already runs at the step time, and the thread costs more than it hides')
    ap.add_argument('--prefetch-depth', type=int, default=2)
    ap.add_argument('--prefetch-priority', type=int, default=-1, help='HIP stream priority of the loader stream (-1 high, 0 default)')
    return ap.parse_args(argv)


def fpn_maps_for(name, dev):
    """Stand-in for the frozen extractor when torchvision is absent: deterministic maps per frame name, generated on the
    device (47 MB per frame: drawing them on the host would make the stand-in the slowest part of the loop)."""
    g = torch.Generator(device=dev).manual_seed(3000 + int(name))
    return [torch.randn((1, 256, h, w), generator=g, device=dev).contiguous(memory_format=torch.channels_last)
            for h, w in ((104, 336), (52, 168), (26, 84))]


def have_torchvision():
    try:
        import torchvision  # noqa: F401
        return True
    except ImportError:
        return False


def cputask(data, anchorBevs, cfg):
    """train.py:26-49 without the GT-paste augmentation: projection, (row, col) swap, voxelization, target assignment."""
    from modules.Calc import classifyAnchors
    from modules.data import Preprocessing as pre
    from modules.utils import lidar2Img
    pcd, img, bbox2d, bbox3d, bev, calib = data
    pcd_t = torch.Tensor(pcd)
    proj = lidar2Img(pcd_t, calib, True)[:, [1, 0]]
    pcd6 = torch.concat([pcd_t, proj.cpu()], dim=1).numpy()
    voxel, idx = pre.group(pcd6, cfg.velorange, cfg.voxelsize, cfg.samplenum)
    if bev is not None and bev.shape[0] != 0:
        pi, ni, gi = classifyAnchors(bev, bbox3d[:, [0, 1]], anchorBevs, cfg.velorange, 0.45, 0.6)
    else:
        pi, ni, gi = None, None, None
    return voxel, idx, img, bbox3d, bev, pi, ni, gi, calib


def fast_chunks(n_frames, frames_per_rank, world):
    """[lo, hi) frame ranges of the global steps of one epoch in --mode fast: the same list on every rank."""
    stride = frames_per_rank * world
    return [(lo, min(n_frames, lo + stride)) for lo in range(0, n_frames, stride)]


def train(args):
    import modules.config as cfg
    from modules import parallel
    from modules.Calc import bbox3d2bev
    from modules.data import Load as load, Preprocessing as pre
    from modules.voxelnet import VoxelLoss
    from MVXNet import MVXNet

    rank, world, local = parallel.init_from_env()
    device = torch.device('cuda', local % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(device)
    say = (lambda *a, **k: None) if (args.quiet or rank != 0) else print

    if args.synthetic:
        from modules.data import Synthetic
        if rank == 0:
            Synthetic.write_kitti_tree(args.dataroot, list(range(args.synthetic)), points=args.points)
        if world > 1:
            torch.distributed.barrier()
    with open(os.path.join(args.dataroot, 'ImageSets/train.txt'), 'r') as f:
        trainSet = f.read().splitlines()
    t_ds = time.perf_counter()
    trainDataSet = load.createDataset(trainSet, needCrop=args.need_crop, root=args.dataroot)
    dataset_s = time.perf_counter() - t_ds           # file I/O: the whole split is read into RAM first, like train.py:53-57
    names = {id(d): n for d, n in zip(trainDataSet, trainSet)}

    anchors = pre.createAnchors(cfg.voxelshape[0] // 2, cfg.voxelshape[1] // 2, cfg.velorange, cfg.carsize)
    anchorBevs = bbox3d2bev(anchors.reshape(anchors.shape[:2] + (-1, 7))).to(device).contiguous()
    torch.manual_seed(0)
    model = MVXNet().to(device)
    criterion = VoxelLoss()
    params = [p for p in model.parameters() if p.requires_grad]
    # train.py:64; on the GPU as one fused multi-tensor kernel per step (same update, ~10 launches fewer; bench.py has the numbers)
    opt = torch.optim.AdamW(params, lr=0.001, eps=cfg.eps, fused=(device.type == 'cuda'))
    anchors = anchors.to(device)
    imsize = torch.Tensor(cfg.imsize).to(device)
    os.makedirs(args.checkpoints, exist_ok=True)
    if args.lastiter > 0:
        model.load_state_dict(torch.load(os.path.join(args.checkpoints, 'epoch%d.pkl' % args.lastiter), map_location=device))
        opt.load_state_dict(torch.load(os.path.join(args.checkpoints, 'epoch%d_opt.pkl' % args.lastiter), map_location=device))
    bucket = parallel.GradBucket(params, late=[model.head.fusion.fcn1.fc.weight]) if args.mode == 'fast' else None
    tv = have_torchvision()

    forwardTime = lossTime = backwardTime = 0.0
    steps_done, losses, loop_stats = 0, [], []
    for epoch in range(args.numepochs):
        random.Random(epoch + args.lastiter).shuffle(trainDataSet)
        mine = trainDataSet
        clsLossSum = regLossSum = 0.0
        clsCnt = regCnt = 0
        if args.mode == 'module':
            for i, data in enumerate(mine):
                voxel, idx, img, gt, gtbev, pi, ni, gi, calibCpu = cputask(data, anchorBevs, cfg)
                calib = {k: torch.Tensor(calibCpu[k]).to(device) for k in calibCpu}
                idx4 = np.concatenate([np.zeros((idx.shape[0], 1)), idx], axis=1)
                opt.zero_grad()
                st = time.perf_counter()
                voxel_t = torch.Tensor(voxel[None, :]).to(device)
                idx_t = torch.LongTensor(idx4).to(device)
                if tv:
                    x = (torch.Tensor(img.copy()).to(device).permute(2, 0, 1) / 255)[None]
                else:
                    x = fpn_maps_for(names[id(data)], device)
                score, reg = model(voxel_t, x, idx_t, [calib], imsize)
                score = score.squeeze(dim=0).permute(1, 2, 0)
                reg = reg.squeeze(dim=0).permute(1, 2, 0)
                forwardTime += time.perf_counter() - st
                st = time.perf_counter()
                clsLoss, regLoss = criterion(pi, ni, gi, gt.to(device) if gt is not None else None, score, reg, anchors, 2)
                loss = clsLoss if regLoss is None else clsLoss + regLoss
                lossTime += time.perf_counter() - st
                st = time.perf_counter()
                loss.backward()
                opt.step()
                backwardTime += time.perf_counter() - st
                c = float(clsLoss.detach())
                if c == c:
                    clsLossSum += c
                    clsCnt += 1
                if regLoss is not None:
                    regLossSum += float(regLoss.detach())
                    regCnt += 1
                losses.append(float(loss.detach()))
                steps_done += 1
                if (i + 1) % 50 == 0 or i + 1 == len(mine):
                    say('Epoch%d %d/%d  average classification loss %.6f, average regression loss %.6f'
                        % (epoch + args.lastiter + 1, i + 1, len(mine), clsLossSum / max(1, clsCnt), regLossSum / max(1, regCnt)))
                if args.steps and steps_done >= args.steps:
                    break
        else:
            from modules import pipeline as pl
            B = args.frames
            # ONE step count for every rank, derived from the global length (ranks that ran different numbers of steps would
            # issue different collective sequences): global step g covers frames [g*B*W, (g+1)*B*W) of the shuffled epoch,
            # rank r takes every W-th of them.  The last chunk may be short -- the reference trains on every frame
            # (train.py:110) -- so a rank may get fewer than B frames, or none: it still takes part in the all-reduce with
            # a zero gradient, and the divisor is the number of frames that contributed on all ranks.
            chunks = fast_chunks(len(trainDataSet), B, world)
            groups = [trainDataSet[lo + rank:hi:world] for lo, hi in chunks]
            cap = max(args.points, max(d[0].shape[0] for d in trainDataSet))
            if not args.prefetch:
                def batches():
                    for group in groups:
                        if not group:
                            yield None, None
                        else:
                            yield pl.batch_from_dataset(group, [names[id(d)] for d in group], device, anchorBevs, fpn_maps_for, cap_points=cap)
                loader = batches()
            else:
                # batch k+1 is prepared by a worker thread on its own stream while step k runs (the reference overlaps its CPU
                # preparation with a process pool, train.py:185-187)
                from modules.data.Prefetch import PrefetchLoader
                loader = PrefetchLoader(groups, lambda d: names[id(d)], device, anchorBevs, fpn_maps_for, cap,
                                        depth=args.prefetch_depth, priority=args.prefetch_priority)
            pending = None                        # the losses of a step are read one step later: the host never waits for
                                                  # the step it has just enqueued

            def account(p):
                nonlocal clsLossSum, clsCnt, regLossSum, regCnt
                if p is None:
                    return
                pl.read_losses(p)
                losses.extend(p['loss'])
                clsLossSum += sum(p['cls'])
                clsCnt += len(p['cls'])
                regLossSum += sum(p['reg'])
                regCnt += len(p['reg'])

            t_epoch = time.perf_counter()
            frames_epoch = 0
            for gstep, ((lo, hi), (batch, targets)) in enumerate(zip(chunks, loader)):
                st = time.perf_counter()
                bucket.zero()
                used = 0
                out = None
                if batch is not None:
                    out = pl.train_step_full(model, batch, targets, criterion, anchors, cfg.imsize, read=False)
                    used = len(out['live'])
                    if used < batch.n_frames:
                        say('Epoch%d step %d: %d frame(s) without a voxel skipped' % (epoch + args.lastiter + 1, gstep, batch.n_frames - used))
                forwardTime += time.perf_counter() - st
                # this rank's frame count rides in the bucket's count slot: the global divisor is applied on the device, with no
                # second collective and no host read (a blocking count exchange undid read=False)
                bucket.all_reduce_mean(frames_local=used)
                opt.step()
                account(pending)
                pending = out
                frames_epoch += hi - lo
                steps_done += 1
                say('Epoch%d %d/%d  average classification loss %.6f, average regression loss %.6f'
                    % (epoch + args.lastiter + 1, hi, len(trainDataSet), clsLossSum / max(1, clsCnt), regLossSum / max(1, regCnt)))
                if args.steps and steps_done >= args.steps:
                    break
            account(pending)
            if hasattr(loader, 'close'):
                loader.close()
            torch.cuda.synchronize(device)
            loop_s = time.perf_counter() - t_epoch
            loop_stats.append({'epoch': epoch + args.lastiter + 1, 'frames': frames_epoch, 'seconds': loop_s,
                               'frames_per_s': frames_epoch / loop_s if loop_s > 0 else 0.0,
                               'loader': dict(getattr(loader, 'stats', {}))})
            say('Epoch%d: %d frames in %.2f s = %.1f frames/s (all ranks, %s)'
                % (epoch + args.lastiter + 1, frames_epoch, loop_s, frames_epoch / max(loop_s, 1e-9),
                   'prefetch thread' if args.prefetch else 'batches prepared inside the step'))
        if rank == 0:
            n = epoch + args.lastiter + 1
            torch.save(model.state_dict(), os.path.join(args.checkpoints, 'epoch%d.pkl' % n))
            torch.save(opt.state_dict(), os.path.join(args.checkpoints, 'epoch%d_opt.pkl' % n))
        if args.steps and steps_done >= args.steps:
            break
    if args.mode == 'fast':
        parallel.assert_replicas_in_sync(params)
    say('forward %.2f s, loss %.2f s, backward %.2f s' % (forwardTime, lossTime, backwardTime))
    return {'losses': losses, 'steps': steps_done, 'model': model, 'opt': opt, 'loop': loop_stats, 'dataset_s': dataset_s,
            'dataset_frames': len(trainDataSet)}


if __name__ == '__main__':
    a = parse_args()
    if a.dataroot is None:
        raise SystemExit('usage: python train_like.py <dataroot> [--synthetic N] ...')
    sys.argv = sys.argv[:1]              # modules.config parses argv at import (reference modules/config/Parser.py:12)
    train(a)
