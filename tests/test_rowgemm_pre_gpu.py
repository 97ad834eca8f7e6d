"""Row GEMMs on pre-cut operands (csrc/rowgemm_pre.hip; nn.Linear / 1x1 Conv2d of /root/reference modules/layers/Blocks.py:5-18,31-40
as used by modules/imhead/Pipe.py:84-104) through the C ABI: the planes ARE the operand (bf16x6: bit for bit), the forward is
bit-identical to the in-kernel-cut kernel, the weight gradient meets the bound of the other split kernels against float64 --
including row counts that end in a partial stage, partial tiles and the accumulate form -- and the BatchNorm backward's plane
output is the cut of its f32 output."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-300))


def test_planes_are_the_operand_exactly():
    from modules import _hip
    g = torch.Generator().manual_seed(0)
    x = (torch.randn((1000, 96), generator=g) * torch.logspace(-20, 20, 96)[None]).to(DEV)
    p = _hip.split_rows(x, 3)
    assert p.shape == (3, 1000, 96) and p.dtype == torch.int16
    hi, mid, lo = (p[q].view(torch.bfloat16).float() for q in range(3))
    assert torch.equal((hi + mid) + lo, x)                      # three bf16 pieces carry the whole f32 mantissa
    assert torch.equal(hi, x.bfloat16().float())


@pytest.mark.parametrize('rows,K,N,frames', [(5000, 768, 768, 1), (3001, 128, 768, 1), (4100, 768, 256, 2), (700, 64, 512, 1)])
def test_forward_is_bit_identical_to_the_in_kernel_cut_kernel(rows, K, N, frames):
    """Same products, same order of accumulation over k: y equals linear_fwd_split bit for bit; the per-frame BatchNorm sums
    (every term in f64, read back from an LDS copy of the tile) agree to f64 rounding."""
    from modules import _hip
    from modules import Extension as X
    g = torch.Generator().manual_seed(1)
    x = torch.randn((rows, K), generator=g).to(DEV)
    w = (torch.randn((N, K), generator=g) * 0.05).to(DEV)
    b = (torch.randn((N,), generator=g) * 0.1).to(DEV)
    xp, wp = _hip.split_rows(x, 3), _hip.split_rows(w, 3)
    y = torch.empty((rows, N), device=DEV)
    # F frames as [real rows f0 | real rows f1 | one padded row per frame] (MVX_ROWS_FUSION), the padded rows with large weights
    F = frames
    real = [0, rows - 1] if F == 1 else [0, 1500, rows - 2]
    desc = X.FramesDesc.make([0, 100, 230][:F + 1], real, 35)
    kind = X.ROWS_FUSION
    row_w = torch.ones((rows,), device=DEV)
    row_w[-F:] = torch.tensor([2000.0, 3050.0][:F])
    st = torch.zeros((F, _hip.STATS_REPLICAS, 2, N), dtype=torch.float64, device=DEV)
    st_old = torch.zeros_like(st)
    y_old = torch.empty((rows, N), device=DEV)
    flags = _hip.split_flags(3, True) | _hip.FLAG_RELU
    X.check(X.lib.mvx_linear_forward_pre_frames(X.ptr(xp), X.ptr(wp), X.ptr(b), X.ptr(y), N, X.ptr(st), X.ptr(row_w), rows, K, N, flags,
                                                1.0, None, 0.0, None, desc.ref(), kind, X.stream()), 'mvx_linear_forward_pre_frames')
    cnt = torch.zeros((1,), dtype=torch.float64, device=DEV)
    mi = torch.empty((F, 2, N), device=DEV)
    X.check(X.lib.mvx_linear_forward_bn_frames(X.ptr(x), K, X.ptr(w), K, 0, X.ptr(b), X.ptr(y_old), N, X.ptr(st_old), X.ptr(row_w), rows,
                                               K, N, flags, X.ptr(cnt), 1e-6, X.ptr(mi), desc.ref(), kind, X.stream()),
            'mvx_linear_forward_bn_frames')
    assert torch.equal(y, y_old)
    assert rel(st.sum(1), st_old.sum(1)) < 1e-13             # every term in f64, as in linear_fwd_split (another grouping of the rows)
    ref = torch.relu(x[:512].double() @ w.double().t() + b.double())
    assert rel(y[:512], ref) < 2e-6


@pytest.mark.parametrize('rows,K,N', [(9000, 768, 768), (4099, 256, 512), (2049, 768, 256), (37, 256, 256)])
def test_weight_gradient_matches_float64(rows, K, N):
    """dz^T x from planes: the bound of the other bf16x6 kernels (2e-6) at row counts that end in a partial 16-row stage (zeroed
    in LDS), with partial strips, and added into an existing gradient."""
    from modules import _hip
    g = torch.Generator().manual_seed(2)
    x = torch.randn((rows, K), generator=g).to(DEV)
    dz = torch.randn((rows, N), generator=g).to(DEV)
    ref = dz.double().t() @ x.double()
    xp, zp = _hip.split_rows(x, 3), _hip.split_rows(dz, 3)
    dw = _hip.linear_wgrad_pre(xp, zp)
    assert rel(dw, ref) < 2e-6
    assert torch.equal(dw, _hip.linear_wgrad_pre(xp, zp))       # slabs summed in a fixed order: reproducible bit for bit
    base = torch.randn((N, K), generator=g).to(DEV)
    acc = base.clone()
    _hip.linear_wgrad_pre(xp, zp, accumulate_into=acc)
    _hip.join_side_stream()
    assert rel(acc - base, ref) < 4e-6


def test_batchnorm_backward_writes_the_cut_of_its_f32_output():
    """mvx_bn_relu_backward_planes_frames == mvx_bn_relu_backward_frames followed by the cut: planes recombine to the f32 dz bit
    for bit, bias gradients are equal."""
    from modules import _hip
    from modules import Extension as X
    from modules import frames as fr
    g = torch.Generator().manual_seed(3)
    rows, C = 3000, 768
    y = torch.randn((rows, C), generator=g).to(DEV)
    gup = (torch.randn((rows, C), generator=g) * 1e-3).to(DEV)
    st = torch.stack([y.relu().double().sum(0), (y.relu().double() ** 2).sum(0)])[None].repeat(_hip.STATS_REPLICAS, 1, 1)
    st[1:] = 0
    mi = _hip.bn_finalize(st.contiguous(), rows, 1e-6)[None].contiguous()

    class FS:
        F = 1
        desc = X.FramesDesc.make([0, rows], [0, rows], 1)
    db1, db2 = torch.zeros((C,), device=DEV), torch.zeros((C,), device=DEV)
    yr = y.relu()
    dz = fr.bn_relu_backward(gup, yr, mi, FS, X.ROWS_REAL, None, db1)
    dzp = fr.bn_relu_backward(gup, yr, mi, FS, X.ROWS_REAL, None, db2, planes=True)
    hi, mid, lo = (dzp[q].view(torch.bfloat16).float() for q in range(3))
    assert torch.equal((hi + mid) + lo, dz)
    assert torch.equal(db1, db2)


@pytest.mark.parametrize('nparts', [2, 3, 7])
def test_backward_in_row_ranges_equals_the_one_pass_form(nparts):
    """mvx_bn_relu_backward_planes_part_frames + mvx_linear_wgrad_pre_rows (the step's tail enqueued range by range so that the
    product of a range runs beside the apply pass of the next): the ranges tile the rows, the planes are those of the one-pass
    form bit for bit, the bias gradient is equal, and the weight gradient summed over the ranges meets the same float64 bound."""
    from modules import _hip
    from modules import Extension as X
    from modules import frames as fr
    g = torch.Generator().manual_seed(5)
    rows, C, K = 5003, 768, 256
    y = torch.randn((rows, C), generator=g).to(DEV)
    x = torch.randn((rows, K), generator=g).to(DEV)
    gup = (torch.randn((rows, C), generator=g) * 1e-3).to(DEV)
    yr = y.relu()
    st = torch.stack([yr.double().sum(0), (yr.double() ** 2).sum(0)])[None].repeat(_hip.STATS_REPLICAS, 1, 1)
    st[1:] = 0
    mi = _hip.bn_finalize(st.contiguous(), rows, 1e-6)[None].contiguous()

    class FS:
        F = 1
        desc = X.FramesDesc.make([0, rows], [0, rows], 1)
    db1, db2 = torch.zeros((C,), device=DEV), torch.zeros((C,), device=DEV)
    whole = fr.bn_relu_backward(gup, yr, mi, FS, X.ROWS_REAL, None, db1, planes=True)
    xp = _hip.split_rows(x, 3)
    dw = torch.zeros((C, K), device=DEV)
    ranges, planes = [], None
    for dzp, lo, hi in fr.bn_relu_backward_planes_parts(gup, yr, mi, FS, X.ROWS_REAL, None, db2, nparts):
        ranges.append((lo, hi))
        planes = dzp
        _hip.linear_wgrad_pre(xp, dzp, accumulate_into=dw, rows=(lo, hi))
    _hip.join_side_stream()
    torch.cuda.synchronize()
    assert ranges[0][0] == 0 and ranges[-1][1] == rows and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    assert torch.equal(planes, whole)
    assert torch.equal(db1, db2)
    dz = sum(whole[q].view(torch.bfloat16).double() for q in range(3))
    assert rel(dw, dz.t() @ x.double()) < 4e-6


def test_sampler_writes_the_planes_of_its_rows():
    """mvx_feature_sample_rows_planes_frames (modules/imhead/Pipe.py:23-82 on compact rows): the f32 rows are those of
    mvx_feature_sample_rows_frames and the planes are their cut (mvx_split_rows), both bit for bit, including rows whose sample
    falls outside the map (zeros + status)."""
    import ctypes
    from modules import _hip
    from modules import Extension as X
    g = torch.Generator().manual_seed(9)
    n, vc, L, C = 3000, 9, 3, 256
    hw = [(47, 153), (24, 77), (12, 39)]
    feats = [torch.randn((h, w, C), generator=g).to(DEV) for h, w in hw]
    vox = torch.randn((n + 50, vc), generator=g)
    vox[:, vc - 2] = torch.rand(n + 50, generator=g) * 369.0
    vox[:, vc - 1] = torch.rand(n + 50, generator=g) * 1223.0
    vox[7, vc - 2] = -50.0                                        # outside the image: zeros, status bit
    vox = vox.to(DEV)
    rows_sel = torch.randperm(n + 50, generator=g)[:n].sort().values.to(torch.int32).to(DEV)
    rows_sel[3] = 7
    ptrs = (ctypes.c_void_p * L)(*[t.data_ptr() for t in feats])
    hwa = (ctypes.c_int32 * (2 * L))(*[v for p in hw for v in p])
    desc = X.FramesDesc.make([0, 100], [0, n], 35)
    out0, out1 = (torch.full((n + 1, L * C), float('nan'), device=DEV) for _ in range(2))
    st0, st1 = (torch.zeros((1,), dtype=torch.int32, device=DEV) for _ in range(2))
    planes = torch.full((3, n + 1, L * C), -1, dtype=torch.int16, device=DEV)
    X.check(X.lib.mvx_feature_sample_rows_frames(X.ptr(vox), vc, X.ptr(rows_sel), n, ptrs, hwa, L, C, 370.0, 1224.0, 1e-6, X.ptr(out0),
                                                 X.ptr(st0), desc.ref(), None, X.stream()), 'mvx_feature_sample_rows_frames')
    X.check(X.lib.mvx_feature_sample_rows_planes_frames(X.ptr(vox), vc, X.ptr(rows_sel), n, ptrs, hwa, L, C, 370.0, 1224.0, 1e-6,
                                                        X.ptr(out1), X.ptr(st1), desc.ref(), None, X.ptr(planes), n + 1, X.stream()),
            'mvx_feature_sample_rows_planes_frames')
    torch.cuda.synchronize()
    assert torch.equal(out0[:n], out1[:n]) and int(st0) == int(st1) == 1
    assert float(out1[3].abs().max()) == 0.0
    ref = _hip.split_rows(out1[:n].contiguous(), 3)
    assert torch.equal(planes[:, :n], ref)
    assert bool((planes[:, n] == -1).all())                      # rows past n_real are the caller's
