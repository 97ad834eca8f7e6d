"""CPU-only checks: the C-ABI library loads and exports every symbol that include/mvx_hip.h
declares (no compute calls without a GPU), the ctypes table mirrors the header, the host-side
configuration / sharding logic, and the product path never imports the oracle."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, 'include', 'mvx_hip.h')


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(mvx_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from modules import Extension as X
    lib = ctypes.CDLL(X.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), 'libmvx_hip.so lacks %s declared in include/mvx_hip.h' % n
    assert set(names) == set(X.PROTOTYPES), set(names) ^ set(X.PROTOTYPES)
    assert lib.mvx_abi_version() == X.ABI_VERSION


def test_argument_counts_match_header():
    from modules import Extension as X
    text = re.sub(r'/\*.*?\*/', '', open(HEADER).read(), flags=re.S)
    for name, (_, args) in X.PROTOTYPES.items():
        m = re.search(r'\b%s\s*\(([^;]*?)\)\s*;' % name, text, flags=re.S)
        assert m, name
        params = m.group(1).strip()
        n = 0 if params in ('', 'void') else params.count(',') + 1
        assert n == len(args), (name, n, len(args))


def test_no_cpu_fallback_and_loud_failure():
    from modules import Extension as X
    with pytest.raises(X.MvxHipError):
        X.ptr(torch.zeros(4))                      # CPU tensor -> refused, never silently computed
    if not torch.cuda.is_available():
        with pytest.raises(X.MvxHipError):
            X.device()
    # size queries are pure host code and must work without a GPU
    assert X.lib.mvx_voxelize_workspace_bytes(2, 20000) > 0
    assert X.lib.mvx_conv3d_packed_weight_bytes(64, 128) == 27 * 64 * 128 * 4
    assert X.lib.mvx_conv3d_wgrad_workspace_bytes(352, 400, 128, 64) > 0
    # argument validation happens before any launch
    assert X.lib.mvx_voxelize(None, None, None, None, 1, 10, 4, 0., 0., 0., 1., 1., 1., 35, 9, 10,
                              None, None, None, None, None, None, 0, None) == -1


def test_product_path_never_imports_the_oracle():
    pkg = os.path.join(REPO, 'mvxnet-makise_amd')
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(root, f)).read()
                assert 'mvx_oracle' not in src and 'liboracle' not in src and 'oracle/' not in src, f


def test_config_surface():
    import modules.config as cfg
    assert cfg.voxelshape == [352, 400, 10] and cfg.samplenum == 35
    assert cfg.voxelsize == [0.2, 0.2, 0.4]            # exactly the reference's doubles (Config.py:7)
    assert cfg.eps == 1e-6 and cfg.dtype == torch.float32
    with pytest.raises(AttributeError):
        cfg.no_such_key


def test_arithmetic_selection_logic():
    """Host side of `convmath` (modules/_hip.py): config value -> split code -> flag bits of a call; which row GEMMs take the
    split arithmetic; the fall-backs of fp16x3 for operands without a range tag.  No GPU involved."""
    import re as _re
    import modules.config as cfg
    from modules import _hip
    header = open(HEADER).read()
    for name, val in (('MVX_FLAG_SPLIT', _hip.FLAG_SPLIT), ('MVX_FLAG_SPLIT3', _hip.FLAG_SPLIT3), ('MVX_FLAG_SPLIT_F16', _hip.FLAG_SPLIT_F16),
                      ('MVX_FLAG_AMAX_COARSE', _hip.FLAG_AMAX_COARSE), ('MVX_FLAG_NO_BG_FILL', _hip.FLAG_NO_BG_FILL)):
        assert int(_re.search(r'#define %s (\d+)' % name, header).group(1)) == val, name
    assert cfg.config['convmath'] == os.environ.get('MVX_CONVMATH', 'bf16x6')   # the shipped default: operand-exact (config.yml:2 half: False)
    old = cfg.config['convmath']
    try:
        codes = {}
        for math, code in (('f32', 0), ('bf16x3', 2), ('bf16x6', 3), ('fp16x3', 4)):
            cfg.config['convmath'] = math
            assert _hip.split_pieces() == code
            codes[math] = {t: _hip.row_split(t) for t in ('fusion_768x768', 'vfe', 'conv1', 'rpn', 'dgrad', 'wgrad')}
        assert set(codes['f32'].values()) == {0}
        assert codes['bf16x3'] == {'fusion_768x768': 0, 'vfe': 0, 'conv1': 0, 'rpn': 2, 'dgrad': 2, 'wgrad': 2}    # forward rows stay exact f32
        assert set(codes['bf16x6'].values()) == {3} and set(codes['fp16x3'].values()) == {4}
        cfg.config['convmath'] = 'fp8'
        with pytest.raises(Exception):
            _hip.split_pieces()
    finally:
        cfg.config['convmath'] = old
    assert _hip.split_flags(0) == 0 and _hip.split_flags(2) == 0 and _hip.split_flags(2, True) == _hip.FLAG_SPLIT
    assert _hip.split_flags(3) == _hip.FLAG_SPLIT3 and _hip.split_flags(3, True) == _hip.FLAG_SPLIT | _hip.FLAG_SPLIT3
    assert _hip.split_flags(4) == _hip.FLAG_SPLIT_F16 and _hip.split_flags(4, True) == _hip.FLAG_SPLIT | _hip.FLAG_SPLIT_F16
    t = torch.zeros(4)
    assert _hip.amax_of(t) is None
    assert _hip.grad_split(4, t) == 3 and _hip.grad_split(3, t) == 3 and _hip.grad_split(0, t) == 0      # untagged gradient: bf16x6
    assert _hip.foreign_split(4, t) == (3, 0) and _hip.foreign_split(2, t) == (2, 0)                       # untagged foreign input: bf16x6
    _hip.tag_amax(t, torch.ones(1))
    assert _hip.grad_split(4, t) == 4 and _hip.foreign_split(4, t) == (4, _hip.FLAG_AMAX_COARSE)
    assert _hip.amax_of(t.view(2, 2)) is None                  # a view does not carry the tag (callers re-tag)


def test_weight_copies_follow_the_parameter():
    """The cached derived copies of a weight (zero-padded columns for rows written with a 4-float pitch, the row-major
    transpose for the input-gradient GEMMs) are remade when the parameter changes -- optimizer step, load_state_dict -- and die
    with it (they live in the parameter's __dict__)."""
    import torch
    from modules import _hip
    w = torch.nn.Parameter(torch.arange(16 * 23, dtype=torch.float32).reshape(16, 23))
    p1 = _hip.padded_weight(w, 24)
    assert p1.shape == (16, 24) and torch.equal(p1[:, :23], w.detach()) and float(p1[:, 23].abs().max()) == 0.0
    assert _hip.padded_weight(w, 24) is p1                       # cached
    assert _hip.padded_weight(w, 23) is w                        # nothing to pad
    t1 = _hip.transposed_weight(w)
    assert torch.equal(t1, w.detach().t())
    with torch.no_grad():
        w.add_(1.0)                                              # an in-place update bumps the version counter
    p2, t2 = _hip.padded_weight(w, 24), _hip.transposed_weight(w)
    assert p2 is not p1 and torch.equal(p2[:, :23], w.detach()) and torch.equal(t2, w.detach().t())


def test_2d_blocks_never_fall_to_torch_silently():
    """a stand-alone CRB2d / DeCRB2d on a CPU tensor raises (this package has no CPU path); the torch /
    MIOpen form only runs where the caller chose it (RPN.forward_torch, `crb2d_hip: false`) or with a RuntimeWarning."""
    import modules.config as cfg
    from modules.Extension import MvxHipError
    from modules.layers import CRB2d, DeCRB2d
    from modules.layers import Blocks
    x = torch.randn(1, 64, 8, 8)
    for m in (CRB2d(64, 64, 3, 1, 1), DeCRB2d(64, 64, 3, 1, 1)):
        with pytest.raises(MvxHipError):
            m(x)
        old = cfg.config.get('crb2d_hip', True)
        cfg.config['crb2d_hip'] = False                       # the explicit opt-out: torch modules, no complaint
        try:
            assert m(x).shape == (1, 64, 8, 8)
        finally:
            cfg.config['crb2d_hip'] = old
        Blocks._IN_FORWARD_TORCH[0] = True                    # the comparison path of the tests
        try:
            assert m(x).shape == (1, 64, 8, 8)
        finally:
            Blocks._IN_FORWARD_TORCH[0] = False


def test_state_dict_keys_and_param_counts():
    from MVXNet import MVXNet
    m = MVXNet()
    sd = m.state_dict()
    assert sum(p.numel() for k, p in sd.items() if k.startswith('backbone.')) == 6670608
    assert sum(p.numel() for k, p in sd.items() if k.startswith('head.fusion.')) == 707872
    assert 'head.fusion.conv1.conv.weight' in sd and sd['head.fusion.conv1.conv.weight'].shape == (128, 768, 1, 1)


def test_shard_frames_partition():
    from modules.parallel import shard_frames
    for world in (1, 2, 4, 8):
        got = sorted(sum((shard_frames(16, r, world) for r in range(world)), []))
        assert got == list(range(16))


def test_synthetic_frames_are_deterministic():
    from modules.data import Synthetic as S
    a, b = S.synth_uniform(5, 2000), S.synth_uniform(5, 2000)
    assert np.array_equal(a, b) and a.dtype == np.float32
    lo, hi = np.array(S.VELORANGE[:3]), np.array(S.VELORANGE[3:])
    assert np.all(a[:, :3] >= lo) and np.all(a[:, :3] < hi)
    assert sorted(S.synth_perm(3, 100).tolist()) == list(range(100))


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return str(sk.getsockname()[1])


@pytest.mark.parametrize('mode,world', [('toy', 2), ('chunks', 2), ('model', 2), ('model', 8)])
def test_data_parallel_gradient_exchange_gloo(mode, world):
    """CPU processes over gloo.  toy: the flat bucket all-reduce gives the mean over all frames.  model (world 2 and the
    north_star's world 8): the real GradBucket over MVXNet's 1,169,440 hot-path parameters in its two-part layout
    [early | late | count] with DIFFERENT per-frame gradients on every rank, exchanged by the two-part branch, equals the
    single-process sum over all frames, and the replicas stay identical after AdamW."""
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=_free_port(), OMP_NUM_THREADS='1')
    worker = os.path.join(REPO, 'tests', '_dp_worker.py')
    procs = [subprocess.Popen([sys.executable, worker, mode], env=dict(env, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert 'DP_OK' in o, o
