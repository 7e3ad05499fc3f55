"""Full encoder fwd+bwd a few times: are all gradients finite and reproducible?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from msmdfusion_amd import synthetic as S
dev = torch.device("cuda:0")
clouds = [torch.from_numpy(S.lidar_sweep(i)).to(dev) for i in range(4)]
torch.manual_seed(0)
model = bench.Backbone().to(dev).train()
ref = None
for it in range(4):
    model.zero_grad(set_to_none=True)
    out = model(clouds)
    (out * out).mean().backward()
    torch.cuda.synchronize()
    g = {n: p.grad.clone() for n, p in model.named_parameters()}
    bad = [n for n, v in g.items() if not torch.isfinite(v).all()]
    same = None if ref is None else sum(torch.equal(g[n], ref[n]) for n in g)
    if ref is None: ref = g
    print("iter", it, "non-finite:", len(bad), bad[:3], "identical to iter0:", same, "of", len(g),
          "out max %.3e" % out.abs().max().item())
