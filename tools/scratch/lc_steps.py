"""Per-step wall time of the LC workload (variance hunt)."""
import os, sys, time
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
ts = []
for i in range(40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    step()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(" ".join("%.1f" % t for t in ts))
