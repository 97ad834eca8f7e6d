"""Pins the oracle's target-assignment and loss restatements (oracle/anchors_c.c, mvx_oracle.voxel_loss) against the
fixtures produced by running the reference (oracle/gen_golden_r2.py).  CPU only."""
import numpy as np
import pytest
import torch

import mvx_oracle as O


def _anchor_grid(g):
    anchors = O.create_anchors(176, 200, list(g['velorange']), list(g['carsize']))
    return anchors, O.bbox3d2bev(anchors.reshape(176, 200, 2, 7))


@pytest.mark.parametrize('tag', ['a', 'b', 'one', 'thr'])
def test_classify_anchors_matches_reference(golden, tag):
    g = golden('classify_anchors_' + tag)
    _, bevs = _anchor_grid(g)
    gt = torch.from_numpy(g['gt'])
    assert np.array_equal(O.bbox3d2bev(gt).numpy(), g['bev'])
    before = O._oracle_c().oracle_anchor_stale_reads()
    pi, ni, gi = O.classify_anchors(torch.from_numpy(g['bev']), gt[:, :2], bevs, list(g['velorange']),
                                    float(g['thr'][0]), float(g['thr'][1]))
    assert O._oracle_c().oracle_anchor_stale_reads() == before
    for got, key in zip(list(pi) + list(ni) + [gi], ('px', 'py', 'pz', 'nx', 'ny', 'nz', 'gi')):
        assert np.array_equal(got, g[key]), key


def test_pairwise_iou_is_bracketed_by_reference_thresholds(golden):
    """The reference's bboxOverlap has no defined result (cpp/voxelutil.cpp:107-109); its IoU arithmetic is pinned
    through _classifyAnchors instead: at threshold t the reference lists exactly the visited cells with IoU >= t."""
    g = golden('iou_brackets')
    c = golden('classify_anchors_one')
    _, bevs = _anchor_grid(c)
    cells = g['cells']
    off = 0
    flat = bevs.reshape(-1, 4, 2).numpy()
    iou = O.bbox_pairwise(g['bev'], flat, True)[0].reshape(176, 200, 2)
    for t, n in zip(g['thresholds'], g['n_pos']):
        listed = cells[off:off + n]
        off += n
        vals = iou[listed[:, 0], listed[:, 1], listed[:, 2]]
        assert (vals >= t).all()
        # and nothing else reaches t: the cross-shaped walk visits every cell of the (convex) high-IoU region; only at
        # the walk's own stop level (0.1) a cell can lie off the visited cross
        assert int((iou >= t).sum()) == n or (t < 0.15 and int((iou >= t).sum()) >= n)


@pytest.mark.parametrize('tag', ['a', 'b'])
def test_voxel_loss_matches_reference(golden, tag):
    g = golden('loss_' + tag)
    anchors = O.create_anchors(176, 200)
    score, reg = O.make_loss_inputs(176, 200, int(g['seed']))
    score.requires_grad_(True)
    reg.requires_grad_(True)
    pi, ni = (g['px'], g['py'], g['pz']), (g['nx'], g['ny'], g['nz'])
    cls, rl = O.voxel_loss(pi, ni, g['gi'], torch.from_numpy(g['gt']), score, reg, anchors, 2)
    (cls + rl).backward()
    assert abs(float(cls) - float(g['cls'])) < 1e-6 * max(1.0, abs(float(g['cls'])))
    assert abs(float(rl) - float(g['regloss'])) < 1e-6 * max(1.0, abs(float(g['regloss'])))
    np.testing.assert_allclose(score.grad.reshape(-1)[torch.from_numpy(g['dscore_sel'])].numpy(), g['dscore_vals'], rtol=1e-5, atol=1e-9)
    tp = tuple(torch.from_numpy(c) for c in pi)
    np.testing.assert_allclose(score.grad[tp].numpy(), g['dscore_pos'], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(reg.grad.reshape(176, 200, 2, 7)[tp].numpy(), g['dreg_rows'], rtol=1e-5, atol=1e-9)


def test_voxel_loss_edge_cases_match_reference(golden):
    g = golden('loss_edge')
    anchors = O.create_anchors(176, 200)
    score, _ = O.make_loss_inputs(176, 200, int(g['seed']))
    cls, rl = O.voxel_loss(None, None, None, None, score, None, anchors, 2)
    assert rl is None and abs(float(cls) - float(g['cls_none'])) < 1e-6
    e = (np.zeros(0, np.int64),) * 3
    cls2, rl2 = O.voxel_loss(e, (g['nx'], g['ny'], g['nz']), np.zeros(0, np.int64), None, score, None, anchors, 2)
    assert rl2 is None and abs(float(cls2) - float(g['cls_nopos'])) < 1e-6
