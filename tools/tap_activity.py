"""How many executed (output tile, depth tap) stages of the conv2 / conv3 forward have a source halo WITHOUT an active site
(their contribution is a per-plane constant)?  Developer diagnostic for the benchmark frames."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = sys.argv[:1]
sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd'))
sys.path.insert(0, REPO)
import bench  # noqa: E402
import modules.config as cfg  # noqa: E402
from modules import frames as fr, pipeline as pl, _hip  # noqa: E402
from MVXNet import MVXNet  # noqa: E402
dev = torch.device('cuda')
torch.manual_seed(0)
model = MVXNet().to(dev)
for wl in ('S1', 'S2'):
    batch = bench.make_batch([0, 1, 2, 3], dev, 20000, wl)
    fs, live, counts, status = pl.prepare_frame_set(batch)
    model.prepack()
    _hip.arena_begin(dev, doubles=1 << 21)
    with torch.no_grad():
        feat, S = fr.rows_forward(model, fs, [batch.fpn_levels[f] for f in live], [370.0, 1224.0], [])
        fr.cml_forward(model, fs, feat, S, [])
    _hip.arena_end()
    torch.cuda.synchronize()
    F = fs.F
    H, W = cfg.voxelshape[0], cfg.voxelshape[1]
    ty, tx = (H + 7) // 8, (W + 15) // 16
    border = torch.zeros((ty, tx), dtype=torch.bool, device=dev)
    border[0, :] = border[-1, :] = True
    border[:, 0] = border[:, -1] = True
    border = border.reshape(-1)
    for li, rec in enumerate(S.convs):
        hf = rec['hflag_in'].view(F, rec['din'], -1) != 0
        din, dout, sd, pd = rec['din'], rec['dout'], rec['sd'], rec['pd']
        tot_exec = tot_inactive = tot_inactive_interior = idle_border = 0
        for d in range(dout):
            taps = [d * sd - pd + kd for kd in range(3) if 0 <= d * sd - pd + kd < din]
            act = torch.stack([hf[:, s] for s in taps], 0)            # [nk][F][tiles]
            computed = act.any(0) | border[None]
            tot_exec += int(computed.sum()) * len(taps)
            inactive = (~act) & computed[None]
            tot_inactive += int(inactive.sum())
            tot_inactive_interior += int((inactive & ~border[None, None]).sum())
            idle_border += int((border[None] & ~act.any(0)).sum()) * len(taps)      # border tiles computed without any active source
        print('%s conv%d forward: executed stages (x chunks) %d, with an inactive source halo %d (%.1f %%), of those in interior tiles %d (%.1f %%); '
              'stages of border tiles without any active source %d (%.1f %%)'
              % (wl, li + 2, tot_exec, tot_inactive, 100.0 * tot_inactive / tot_exec, tot_inactive_interior, 100.0 * tot_inactive_interior / tot_exec,
                 idle_border, 100.0 * idle_border / tot_exec))

# ---- weight-gradient step lists: entries per depth tap vs the share of workgroups each tap gets (by plane count)
print('weight-gradient lists (last workload): steps per depth tap, and steps per workgroup with the plane-count shares')
for li, rec in enumerate(S.convs):
    hf = rec['hflag_in'].view(F, rec['din'], -1) != 0
    din, dout, sd, pd = rec['din'], rec['dout'], rec['sd'], rec['pd']
    counts, nd = [], []
    for kd in range(3):
        c = 0
        n = 0
        for d in range(dout):
            s_ = d * sd - pd + kd
            if 0 <= s_ < din:
                c += int(hf[:, s_].sum())
                n += 1
        counts.append(c)
        nd.append(n)
    chunks = rec['w'].shape[1] // 64
    nstrips = max(1, 128 // chunks)
    shares = [min(nstrips, max(1, (2 * nstrips * n) // max(1, sum(nd)))) for n in nd]
    per = [c / s_ for c, s_ in zip(counts, shares)]
    print('  conv%d: steps %s, shares %s -> steps per workgroup %s (ideal %.1f)' % (li + 2, counts, shares, ['%.1f' % p for p in per], sum(counts) / sum(shares)))
