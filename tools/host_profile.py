"""cProfile of the host side of the training step (what the Python enqueue path costs) -- developer tool.
usage: host_profile.py [f32|bf16x3]"""
import cProfile, os, pstats, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'mvxnet-makise_amd'))
import bench
import modules.config as cfg
from modules import parallel, _hip
from modules.pipeline import train_step_frames
from MVXNet import MVXNet

if len(sys.argv) > 1:
    cfg.config['convmath'] = sys.argv[1]
dev = torch.device('cuda')
torch.manual_seed(0)
model = MVXNet().to(dev)
hot = [p for k, p in model.named_parameters() if p.requires_grad and '.rpn.' not in k]
bucket = parallel.GradBucket(hot)
opt = torch.optim.AdamW(hot, lr=1e-3, eps=cfg.eps)
batch = bench.make_batch([0, 1, 2, 3], dev, 20000)
grad_mid = torch.randn((1, 128, cfg.voxelshape[0], cfg.voxelshape[1]), device=dev) * 1e-3
imsize = [float(v) for v in cfg.imsize]


from modules import pipeline as pl_mod
pending = [None]
PIPE = os.environ.get('MVX_PIPELINE_INPUT', '0') != '0'


def step():
    ready = None
    if PIPE:
        if pending[0] is not None:
            ready = pl_mod.prepare_end(pending[0], model.head)
        pending[0] = pl_mod.prepare_begin(batch)
    bucket.zero()
    train_step_frames(model, batch, grad_mid, imsize, ready=ready)
    if PIPE:
        pl_mod.prepare_mid(pending[0], model.head)
    bucket.all_reduce_mean(4)
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
print('host enqueue %.2f ms/step, wall %.2f ms/step' % ((t1 - t0) / 3 * 1e3, (time.perf_counter() - t0) / 3 * 1e3))
if len(sys.argv) > 2 and sys.argv[2] == 'mock':
    # replace every kernel entry point by a no-op: what remains is the Python/torch side of the enqueue path
    from modules import Extension as X

    class _Mock:
        def __init__(self, real):
            self._real = real

        def __getattr__(self, name):
            fn = getattr(self._real, name)
            if name.endswith('_bytes') or name in ('mvx_abi_version', 'mvx_conv3d_tile_shape', 'mvx_voxelize', 'mvx_row_compact_map',
                                                   'mvx_voxel_row_offsets'):
                return fn
            return lambda *a: 0
    X.lib = _Mock(X.lib)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print('MOCK kernels: host %.2f ms/step' % ((t1 - t0) / 5 * 1e3))
    sys.exit(0)
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(35)
st.sort_stats('cumulative').print_stats(45)
