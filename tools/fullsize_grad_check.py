"""One full-size S2 frame: parameter gradients of the benchmarked HIP path, of the f32 CPU oracle and of a float64 run of
the oracle (the yardstick).  TEST TOOLING (imports oracle/); writes gpurun_out/fullsize_grads.json.
Answers: when HIP and the f32 oracle disagree on a gradient at full size, which one is off?"""
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = sys.argv[:1]
for p in (os.path.join(REPO, 'mvxnet-makise_amd'), os.path.join(REPO, 'oracle'), os.path.join(REPO, 'tests')):
    sys.path.insert(0, p)
import mvx_oracle as O                                       # noqa: E402
from test_fullsize_gpu import _host_projection             # noqa: E402


def main():
    import modules.pipeline as pl
    from MVXNet import MVXNet
    from modules import parallel
    from modules.pipeline import FrameBatch, train_step_frames
    dev = torch.device('cuda')
    fid, Pn = 0, 20000
    pc = O.synth_ring(fid, Pn)
    pts6 = np.zeros((1, Pn, 6), np.float32)
    pts6[0, :, :4] = pc
    pts6[0, :, 4:] = _host_projection(pc)
    perm = O.synth_perm(fid, Pn)[None]
    fpn_cpu = [torch.from_numpy(f) for f in O.synth_fpn(fid)]
    batch = FrameBatch(torch.from_numpy(pts6).to(dev), torch.from_numpy(perm).to(dev),
                       torch.full((1,), Pn, dtype=torch.int32, device=dev),
                       [[f[None].to(dev).contiguous(memory_format=torch.channels_last) for f in fpn_cpu]])
    torch.manual_seed(0)
    model = MVXNet().to(dev)
    hot = [(k, p) for k, p in model.named_parameters() if p.requires_grad and '.rpn.' not in k]
    bucket = parallel.GradBucket([p for _, p in hot])
    bucket.zero()
    g = torch.Generator(device='cpu').manual_seed(77)
    G = torch.randn((1, 128, 352, 400), generator=g) * 1e-3
    train_step_frames(model, batch, G.to(dev), [370.0, 1224.0])
    torch.cuda.synchronize()
    hip = {k: p.grad.detach().cpu().double() for k, p in hot}

    rv, ri, _ = O.group(pts6[0], perm[0], O.VELORANGE, O.voxelsize(), 35)
    V = rv.shape[0]
    idx = torch.from_numpy(np.concatenate([np.zeros((V, 1), np.int64), ri.astype(np.int64)], 1))
    out = {}
    res = {}
    for name, dt in (('f32', torch.float32), ('f64', torch.float64)):
        t0 = time.time()
        Pm = {k: v.detach().cpu().to(dt).clone().requires_grad_(True) for k, v in model.state_dict().items() if '.rpn.' not in k}
        vox32 = torch.from_numpy(rv.astype(np.float32))
        imf = O.feature_mapping(vox32, fpn_cpu, torch.tensor([370.0, 1224.0]))       # f32 sampling positions in both
        imf = O.image_feature_fusion(imf.to(dt), Pm, 'head.fusion.')
        v23 = torch.cat([vox32[..., :7].to(dt), imf], dim=-1)
        mid = O.voxelnet_middle(v23, idx, O.strip_prefix(Pm, 'backbone.'))
        mid.backward(G.to(dt))
        res[name] = {k: v.grad.double() for k, v in Pm.items()}
        print(name, 'oracle forward+backward %.1f s' % (time.time() - t0), flush=True)
    for k in hip:
        ref = res['f64'][k]
        den = float(ref.abs().max())
        out[k] = {'hip_vs_f64': float((hip[k] - ref).abs().max()) / den,
                  'oracle_f32_vs_f64': float((res['f32'][k] - ref).abs().max()) / den,
                  'hip_vs_oracle_f32': float((hip[k] - res['f32'][k]).abs().max()) / float(res['f32'][k].abs().max())}
        print('%-40s hip-f64 %.2e   f32-f64 %.2e   hip-f32 %.2e' % (k, out[k]['hip_vs_f64'], out[k]['oracle_f32_vs_f64'],
                                                                    out[k]['hip_vs_oracle_f32']), flush=True)
    os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(REPO, 'gpurun_out', 'fullsize_grads.json'), 'w') as fh:
        json.dump(out, fh, indent=1)


if __name__ == '__main__':
    main()
