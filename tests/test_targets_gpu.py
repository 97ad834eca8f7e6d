"""Target assignment (classifyAnchors, bboxOverlap) and VoxelLoss on the GPU against the reference fixtures and the
oracle.  Integer results bit-exact and in the reference's order; loss values / gradients within 1e-5 relative."""
import numpy as np
import pytest
import torch

import mvx_oracle as O

pytestmark = pytest.mark.gpu


def _grid(g):
    anchors = O.create_anchors(176, 200, list(g['velorange']), list(g['carsize']))
    return anchors, O.bbox3d2bev(anchors.reshape(176, 200, 2, 7))


@pytest.mark.parametrize('tag', ['a', 'b', 'one', 'thr'])
def test_classify_anchors_matches_reference(golden, tag):
    from modules import Calc
    g = golden('classify_anchors_' + tag)
    _, bevs = _grid(g)
    gt = torch.from_numpy(g['gt'])
    bev = Calc.bbox3d2bev(gt)
    assert np.array_equal(bev.numpy(), g['bev'])
    pi, ni, gi = Calc.classifyAnchors(bev, gt[:, [0, 1]], bevs, list(g['velorange']), float(g['thr'][0]), float(g['thr'][1]))
    for got, key in zip(list(pi) + list(ni) + [gi], ('px', 'py', 'pz', 'nx', 'ny', 'nz', 'gi')):
        assert got.is_cuda and got.dtype == torch.int64
        assert np.array_equal(got.cpu().numpy(), g[key]), key


def test_extension_surface_matches_reference(golden):
    """The pybind11 module's other three exports (cpp/voxelutil.cpp:362-368) through modules.Extension.cpp."""
    from modules.Extension import cpp
    g = golden('classify_anchors_b')
    _, bevs = _grid(g)
    nls, nws = O.anchor_center_cells(torch.from_numpy(g['gt'])[:, :2], bevs.shape, list(g['velorange']))
    pi, ni, gi = cpp._classifyAnchors(g['bev'], bevs.numpy(), nls.numpy(), nws.numpy(), 0.45, 0.6)
    for got, key in zip(list(pi) + list(ni) + [gi], ('px', 'py', 'pz', 'nx', 'ny', 'nz', 'gi')):
        assert isinstance(got, np.ndarray) and got.dtype == np.int64 and np.array_equal(got, g[key]), key
    # IoU / intersection of box pairs: bit-equal to the oracle's restatement of the same f32 arithmetic
    rng = np.random.default_rng(3)
    gt = torch.from_numpy(g['gt'])
    other = gt.clone()
    other[:, :2] += torch.from_numpy(rng.normal(0, 1.0, (gt.shape[0], 2))).float()
    other[:, 6] += 0.4
    b1, b2 = g['bev'], O.bbox3d2bev(other).numpy()
    assert np.array_equal(cpp.bboxOverlap(b1, b2), O.bbox_pairwise(b1, b2, True))
    assert np.array_equal(cpp.bboxIntersection(b1, b2), O.bbox_pairwise(b1, b2, False))
    d = np.diag(cpp.bboxOverlap(b1, b1))
    assert np.all(np.abs(d - 1) < 1e-4)
    assert cpp.bboxOverlap(b1[:0], b2).shape == (0, b2.shape[0])


def test_classify_anchors_large_boxes_and_borders():
    """Boxes far larger than a car (the window is widened automatically) and ground truths whose walk runs into the
    grid border; against the oracle."""
    from modules import Calc
    anchors = O.create_anchors(176, 200)
    bevs = O.bbox3d2bev(anchors.reshape(176, 200, 2, 7))
    gt = torch.tensor([[35.0, 0.0, -1.0, 14.0, 3.0, 3.0, 0.2],        # a truck-sized box
                       [0.3, -39.7, -1.0, 3.9, 1.6, 1.5, 0.0],        # at the (0, 0) corner
                       [70.2, 39.8, -1.0, 3.9, 1.6, 1.5, 1.57],       # at the opposite corner
                       [20.0, 10.0, -1.0, 3.9, 1.6, 1.56, 0.0],       # exactly an anchor-shaped box
                       [20.0, 10.0, -1.0, 3.9, 1.6, 1.56, 0.0]])      # ... twice: repeated positives
    bev = Calc.bbox3d2bev(gt)
    ref = O.classify_anchors(bev, gt[:, :2], bevs, O.VELORANGE, 0.45, 0.6)
    pi, ni, gi = Calc.classifyAnchors(bev, gt[:, [0, 1]], bevs, O.VELORANGE, 0.45, 0.6)
    for got, want in zip(list(pi) + list(ni) + [gi], list(ref[0]) + list(ref[1]) + [ref[2]]):
        assert np.array_equal(got.cpu().numpy(), want)
    assert len(gi) > 10


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_voxel_loss_matches_reference(golden, tag):
    from modules.voxelnet import VoxelLoss
    g = golden('loss_' + tag)
    dev = torch.device('cuda')
    anchors = O.create_anchors(176, 200).to(dev)
    score0, reg0 = O.make_loss_inputs(176, 200, int(g['seed']))
    # the layout train.py hands over: (1,C,L,W) network outputs, squeezed and permuted (views, not copies)
    score_map = score0.permute(2, 0, 1)[None].contiguous().to(dev).requires_grad_(True)
    reg_map = reg0.permute(2, 0, 1)[None].contiguous().to(dev).requires_grad_(True)
    score = score_map.squeeze(0).permute(1, 2, 0)
    reg = reg_map.squeeze(0).permute(1, 2, 0)
    pi, ni = (g['px'], g['py'], g['pz']), (g['nx'], g['ny'], g['nz'])
    cls, rl = VoxelLoss()(pi, ni, g['gi'], torch.from_numpy(g['gt']).to(dev), score, reg, anchors, 2)
    (cls + rl).backward()
    assert abs(float(cls) - float(g['cls'])) < 1e-5 * abs(float(g['cls']))
    assert abs(float(rl) - float(g['regloss'])) < 1e-5 * abs(float(g['regloss']))
    ds = score_map.grad[0].permute(1, 2, 0).cpu()
    dr = reg_map.grad[0].permute(1, 2, 0).cpu()
    tp = tuple(torch.from_numpy(c) for c in pi)
    tn = tuple(torch.from_numpy(c) for c in ni)
    np.testing.assert_allclose(ds.reshape(-1)[torch.from_numpy(g['dscore_sel'])].numpy(), g['dscore_vals'], rtol=1e-5, atol=1e-10)
    np.testing.assert_allclose(ds[tp].numpy(), g['dscore_pos'], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(ds[tn].numpy(), g['dscore_neg'], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(dr.reshape(176, 200, 2, 7)[tp].numpy(), g['dreg_rows'], rtol=1e-5, atol=1e-9)
    assert abs(float(ds.double().sum()) - g['dscore_sums'][0]) < 1e-5 * g['dscore_sums'][1]
    assert abs(float(dr.abs().sum()) - float(g['dreg_abs_sum'])) < 1e-5 * float(g['dreg_abs_sum'])


def test_voxel_loss_edge_cases(golden):
    from modules.voxelnet import VoxelLoss
    g = golden('loss_edge')
    dev = torch.device('cuda')
    anchors = O.create_anchors(176, 200).to(dev)
    score0, _ = O.make_loss_inputs(176, 200, int(g['seed']))
    score = score0.to(dev).requires_grad_(True)
    cls, rl = VoxelLoss()(None, None, None, None, score, None, anchors, 2)
    assert rl is None
    cls.backward()
    assert abs(float(cls) - float(g['cls_none'])) < 1e-5 * float(g['cls_none'])
    np.testing.assert_allclose(score.grad.cpu().reshape(-1)[torch.from_numpy(g['sel_none'])].numpy(), g['vals_none'], rtol=1e-5)
    score2 = score0.to(dev).requires_grad_(True)
    e = (np.zeros(0, np.int64),) * 3
    ni = (g['nx'], g['ny'], g['nz'])
    cls2, rl2 = VoxelLoss()(e, ni, np.zeros(0, np.int64), None, score2, None, anchors, 2)
    assert rl2 is None
    cls2.backward()
    assert abs(float(cls2) - float(g['cls_nopos'])) < 1e-5 * float(g['cls_nopos'])
    tn = tuple(torch.from_numpy(c) for c in ni)
    np.testing.assert_allclose(score2.grad.cpu()[tn].numpy(), g['dscore_neg_nopos'], rtol=1e-5, atol=1e-10)
    np.testing.assert_allclose(score2.grad.cpu().reshape(-1)[torch.from_numpy(g['sel_nopos'])].numpy(), g['vals_nopos'], rtol=1e-5)


def test_classify_anchors_frames_equals_per_frame_calls(golden):
    """Calc.classifyAnchorsFrames: the frames of a step in ONE walk launch and one host read (mvx_classify_anchors_frames) give,
    per frame, exactly the lists of a classifyAnchors call on that frame alone -- the four reference fixtures as four frames
    of one batch (different box counts, ground-truth ids local to each frame), with a frame without boxes in between."""
    from modules import Calc
    tags = ('a', 'b', 'one', 'thr')
    gs = [golden('classify_anchors_' + t) for t in tags]
    _, bevs = _grid(gs[0])
    frames, want = [], []
    for g in gs[:3]:                                   # same thresholds (0.45 / 0.6) in the first three fixtures
        assert abs(float(g['thr'][0]) - 0.45) < 1e-6 and abs(float(g['thr'][1]) - 0.6) < 1e-6
        gt = torch.from_numpy(g['gt'])
        frames.append((Calc.bbox3d2bev(gt), gt[:, [0, 1]]))
        want.append(g)
    frames.insert(1, None)
    want.insert(1, None)
    res = Calc.classifyAnchorsFrames(frames, bevs.cuda().contiguous(), list(gs[0]['velorange']), 0.45, 0.6)
    assert len(res) == 4 and res[1] is None
    for r, g in zip(res, want):
        if g is None:
            continue
        pi, ni, gi = r
        for got, key in zip(list(pi) + list(ni) + [gi], ('px', 'py', 'pz', 'nx', 'ny', 'nz', 'gi')):
            assert got.is_cuda and got.dtype == torch.int64
            assert np.array_equal(got.cpu().numpy(), g[key]), key
