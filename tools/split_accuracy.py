"""Which bf16x3 row GEMM groups cost how much accuracy?  One full-size S2 frame through pipeline.train_step_full under
convmath bf16x3 with the wide row GEMMs of the groups named in each variant in split arithmetic (modules/_hip.py row_split),
against the float64 oracle run end to end (developer tool; tests/test_configs_gpu.py asserts the shipped setting).
Writes gpurun_out/split_accuracy.json."""
import json, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd')); sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'oracle'))
import bench
import mvx_oracle as O
import modules.config as cfg
import modules.pipeline as pl
from MVXNet import MVXNet
from modules import Calc, _hip, parallel
from modules.data import Preprocessing as pre
from modules.voxelnet import VoxelLoss

dev = torch.device('cuda')
P = 20000
batch = bench.make_batch([0], dev, P, 'S2')
torch.manual_seed(0)
model = MVXNet().to(dev)
bucket = parallel.GradBucket([p for p in model.parameters() if p.requires_grad])
anchors = pre.createAnchors(cfg.voxelshape[0] // 2, cfg.voxelshape[1] // 2, cfg.velorange, cfg.carsize)
bevs = Calc.bbox3d2bev(anchors.reshape(anchors.shape[:2] + (-1, 7))).to(dev).contiguous()
gg = np.random.default_rng(11)
n = 8
gt = torch.tensor(np.stack([gg.uniform(8, 60, n), gg.uniform(-30, 30, n), gg.uniform(-1.8, -0.6, n), gg.uniform(3.4, 4.4, n),
                            gg.uniform(1.5, 1.8, n), gg.uniform(1.4, 1.7, n),
                            gg.choice([0.0, np.pi / 2], n) + gg.normal(0, 0.05, n)], 1), dtype=torch.float32)
pi, ni, gi = Calc.classifyAnchors(Calc.bbox3d2bev(gt), gt[:, [0, 1]], bevs, cfg.velorange, 0.45, 0.6)
targets = [(pi, ni, gi, gt.to(dev))]

points6, _ = batch.prepared()
pts6 = points6.cpu().numpy()
P64 = {k: v.detach().cpu().double() for k, v in model.state_dict().items()}
bb = O.strip_prefix(P64, 'backbone.')
rv, ri, _ = O.group(pts6[0], O.synth_perm(0, P), O.VELORANGE, O.voxelsize(), 35)
V = rv.shape[0]
vox = torch.from_numpy(rv.astype(np.float32))
idx = torch.from_numpy(np.concatenate([np.zeros((V, 1), np.int64), ri.astype(np.int64)], 1))
fpn = [t[0].cpu() for t in batch.fpn_levels[0]]
with torch.no_grad():
    imf = O.feature_mapping(vox, fpn, torch.tensor([370.0, 1224.0]))
    imf64 = O.image_feature_fusion(imf.double(), P64, 'head.fusion.')
    mid = O.voxelnet_middle(torch.cat([vox[..., :7].double(), imf64], dim=-1), idx, bb)
    score, reg = O.rpn(mid, bb)
print('oracle done', flush=True)


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


res = {}
for math, groups in (('f32', ''), ('bf16x3', ''), ('bf16x3', 'fusion_768x768'), ('bf16x3', 'fusion_768x768,fusion_128x768'),
                     ('bf16x3', 'fusion_768x768,conv1,rpn'), ('bf16x3', 'fusion'), ('bf16x3', 'vfe'), ('bf16x3', 'conv1'), ('bf16x3', 'rpn'),
                     ('bf16x3', 'fusion,vfe,conv1,rpn')):
    cfg.config['convmath'] = math
    _hip.ROW_SPLIT = tuple(k for k in groups.split(',') if k) + ('dgrad',)
    bucket.zero()
    keep = {}
    out = pl.train_step_full(model, batch, targets, VoxelLoss(), anchors.to(dev), cfg.imsize, keep=keep)
    torch.cuda.synchronize()
    F, D3, H, W, C3, h1, w1 = keep['geom']
    x3 = keep['x3'].view(F, D3, H, W, C3).permute(0, 4, 1, 2, 3).reshape(F, C3 * D3, H, W).cpu()
    heads = keep['heads'].view(F, h1, w1, 16).cpu()
    res['%s rows[%s]' % (math, groups)] = {
        'bev_rel_maxnorm': rel(x3[0].double(), mid[0]),
        'cls_logit_rel_maxnorm': rel(heads[0, ..., :2].double(), torch.logit(score[0].permute(1, 2, 0))),
        'reg_rel_maxnorm': rel(heads[0, ..., 2:].double(), reg[0].permute(1, 2, 0))}
    print(list(res.items())[-1], flush=True)
os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
json.dump(res, open(os.path.join(REPO, 'gpurun_out', 'split_accuracy.json'), 'w'), indent=1)
