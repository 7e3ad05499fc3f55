"""cProfile of the host side of one LC step (where does the 54 ms of enqueue go?)."""
import cProfile, pstats, sys, os, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from msmdfusion_amd import synthetic as S
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = bench.FusionBackbone().to(dev).train()
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.AdamW(params, lr=1e-4)
clouds = [torch.from_numpy(S.lidar_sweep(i)).to(dev) for i in range(2)]
virt = [torch.from_numpy(S.virtual_points(i)).to(dev) for i in range(2)]
def step():
    out = model(clouds, virt)
    out.mean().backward()
    opt.step(); opt.zero_grad(set_to_none=True)
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(3): step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
