"""Row layers with a 128-deep reduction on the weights-resident streaming kernel (csrc/rowgemm_k128.hip): nn.Linear(128, 128) of
the VFE head FCN (/root/reference modules/voxelnet/VoxelNet.py:28-33), the 128 -> 128 layer of the fusion MLP and the input
gradient of its 768 -> 128 layer (modules/imhead/Pipe.py:94-104).  Through the C ABI, against the kernel it stands in for
(linear_fwd_split, selected with mvx_tuning_set(MVX_TUNE_ROWGEMM_K128, 0)): y bit for bit, per-frame BatchNorm sums to f64
rounding, the finalised mean / inverse std; and against float64."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'
TUNE_K128 = 3


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-300))


def _run(on, x, w, b, rows, N, flags, row_w, desc, kind, F, stats=True):
    from modules import _hip
    from modules import Extension as X
    X.check(X.lib.mvx_tuning_set(TUNE_K128, 1 if on else 0), 'mvx_tuning_set')
    try:
        y = torch.full((rows, N), float('nan'), device=DEV)
        st = torch.zeros((F, _hip.STATS_REPLICAS, 2, N), dtype=torch.float64, device=DEV) if stats else None
        cnt = torch.zeros((1,), dtype=torch.float64, device=DEV) if stats else None
        mi = torch.empty((F, 2, N), device=DEV) if stats else None
        if stats:
            X.check(X.lib.mvx_linear_forward_bn_frames(_hip._vptr(x), x.stride(0), X.ptr(w), w.stride(0), 0, X.ptr(b), X.ptr(y), N,
                                                       X.ptr(st), X.ptr(row_w), rows, 128, N, flags, X.ptr(cnt), 1e-6, X.ptr(mi),
                                                       desc.ref(), kind, X.stream()), 'mvx_linear_forward_bn_frames')
        else:
            X.check(X.lib.mvx_linear_forward(_hip._vptr(x), x.stride(0), X.ptr(w), w.stride(0), 0, X.ptr(b), X.ptr(y), N, None, None, rows,
                                             128, N, flags, None, 0, X.stream()), 'mvx_linear_forward')
        torch.cuda.synchronize()
    finally:
        X.check(X.lib.mvx_tuning_set(TUNE_K128, 1), 'mvx_tuning_set')
    return y, st, mi


@pytest.mark.parametrize('split', [3, 4, 2])
@pytest.mark.parametrize('rows,N,frames', [(5000, 128, 1), (33, 128, 1), (100001, 128, 3), (4100, 768, 2), (31, 256, 1), (3001, 1728, 1), (2777, 192, 2)])
def test_equals_the_tiled_kernel(rows, N, frames, split):
    from modules import _hip
    from modules import Extension as X
    g = torch.Generator().manual_seed(11)
    xs = torch.randn((rows, 160), generator=g).to(DEV)          # a row stride that is not K
    x = xs[:, 16:144]
    w = (torch.randn((N, 128), generator=g) * 0.05).to(DEV)
    b = (torch.randn((N,), generator=g) * 0.1).to(DEV)
    F = frames
    # F frames as [real rows f0 | real rows f1 | ... | one padded row per frame], frame boundaries off the 32-row blocks
    real = {1: [0, rows - 1], 2: [0, 1501, rows - 2], 3: [0, 33333, 70001, rows - 3]}[F]
    desc = X.FramesDesc.make([0, 100, 230, 300][:F + 1], real, 35)
    row_w = torch.ones((rows,), device=DEV)
    row_w[-F:] = torch.tensor([2000.0, 3050.0, 77.0][:F])
    flags = _hip.split_flags(split, True) | _hip.FLAG_RELU
    y1, st1, mi1 = _run(True, x, w, b, rows, N, flags, row_w, desc, X.ROWS_FUSION, F)
    y0, st0, mi0 = _run(False, x, w, b, rows, N, flags, row_w, desc, X.ROWS_FUSION, F)
    assert torch.equal(y1, y0)
    assert rel(st1.sum(1), st0.sum(1)) < 1e-13
    assert rel(mi1, mi0) < 2e-7                                  # one f32 ulp at most (the sums differ in the last f64 bits)
    ref = torch.relu(x[:700].double() @ w.double().t() + b.double())
    assert rel(y1[:700], ref) < {3: 2e-6, 4: 4e-6, 2: 2e-4}[split]
    # the sums are the sums of what was written
    yw = y1.double() * row_w.double()[:, None]
    for f in range(F):
        rows_f = torch.zeros((rows,), dtype=torch.bool, device=DEV)
        rows_f[real[f]:real[f + 1]] = True
        rows_f[rows - F + f] = True
        assert rel(st1[f].sum(0)[0], yw[rows_f].sum(0)) < 1e-12
        assert rel(st1[f].sum(0)[1], (yw[rows_f] * y1[rows_f].double()).sum(0)) < 1e-12


def test_input_gradient_form_without_epilogue():
    """dx = dz w (no bias, no ReLU, no sums): 128 -> 768 in six column chunks, rows not a multiple of 32."""
    from modules import _hip
    from modules import Extension as X
    g = torch.Generator().manual_seed(12)
    rows, N = 7019, 768
    dz = (torch.randn((rows, 128), generator=g) * 1e-3).to(DEV)
    wt = (torch.randn((N, 128), generator=g) * 0.05).to(DEV)
    flags = _hip.split_flags(3, True)
    y1, _, _ = _run(True, dz, wt, None, rows, N, flags, None, None, X.ROWS_SINGLE, 1, stats=False)
    y0, _, _ = _run(False, dz, wt, None, rows, N, flags, None, None, X.ROWS_SINGLE, 1, stats=False)
    assert torch.equal(y1, y0)
    assert rel(y1, dz.double() @ wt.double().t()) < 2e-6
