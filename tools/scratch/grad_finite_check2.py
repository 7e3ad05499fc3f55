"""As grad_finite_check.py but with no device sync between iterations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from msmdfusion_amd import synthetic as S
dev = torch.device("cuda:0")
clouds = [torch.from_numpy(S.lidar_sweep(i)).to(dev) for i in range(4)]
for trial in range(3):
    torch.manual_seed(0)
    model = bench.Backbone().to(dev).train()
    for it in range(3):
        model.zero_grad(set_to_none=True)
        out = model(clouds)
        (out * out).mean().backward()
    torch.cuda.synchronize()
    if os.environ.get("SHOW_MEM"):
        print(torch.cuda.memory_summary(abbreviated=True)[:1500])
    g = {n: p.grad for n, p in model.named_parameters()}
    bad = [n for n, v in g.items() if not torch.isfinite(v).all()]
    print("trial", trial, "non-finite:", len(bad))
    if trial == 0:
        for n, v in g.items():
            print("   ", "NaN" if not torch.isfinite(v).all() else "ok ", n, tuple(v.shape))
