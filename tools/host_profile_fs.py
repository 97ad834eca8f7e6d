"""cProfile of the host side of the frame-set training step (developer tool)."""
import cProfile, os, pstats, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = sys.argv[:1]
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd'))
import bench
import modules.config as cfg
from modules import parallel, _hip
from modules.pipeline import train_step_frame_set
from MVXNet import MVXNet

dev = torch.device('cuda')
torch.manual_seed(0)
model = MVXNet().to(dev)
hot = [p for k, p in model.named_parameters() if p.requires_grad and '.rpn.' not in k]
bucket = parallel.GradBucket(hot)
opt = torch.optim.AdamW(hot, lr=1e-3, eps=cfg.eps)
batch = bench.make_batch([0, 1, 2, 3], dev, 20000)
grad_mid = torch.randn((1, 128, cfg.voxelshape[0], cfg.voxelshape[1]), device=dev) * 1e-3
imsize = [float(v) for v in cfg.imsize]
ready = [None]


def step():
    bucket.zero()
    _, _, ready[0] = train_step_frame_set(model, batch, grad_mid, imsize, ready=ready[0], prepare_next=batch)
    bucket.all_reduce_mean(4)
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
print('host enqueue %.2f ms/step, wall %.2f ms/step' % ((t1 - t0) / 5 * 1e3, (time.perf_counter() - t0) / 5 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(28)
