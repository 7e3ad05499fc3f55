"""Which parameter / gradient goes non-finite first in a bench workload (debug)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B
from msmdfusion_amd import synthetic as S

wl = sys.argv[1] if len(sys.argv) > 1 else "lc_b4"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 150
dev = torch.device("cuda:0")
torch.manual_seed(0)
W = B.WORKLOADS[wl]
spg = W["spg"]
ids = list(range(spg))
model = (B.FusionDetector(ids) if wl == "lc_full" else B.FusionTailBackbone() if wl == "lc_tail"
         else B.FusionBackbone()).to(dev).train()
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.01, fused=True)
clouds = [torch.from_numpy(S.lidar_sweep(i)).to(dev) for i in ids]
batch = (clouds, [torch.from_numpy(S.virtual_points(i)).to(dev) for i in ids])
target = torch.randn(spg, W["bev_channels"], 180, 180, device=dev).contiguous(memory_format=torch.channels_last)
for it in range(steps):
    opt.zero_grad(set_to_none=True)
    bev = model(*batch)
    loss = (bev * target).mean()
    loss.backward()
    bad_g = [(n, int((~torch.isfinite(p.grad)).sum())) for n, p in model.named_parameters()
             if p.grad is not None and not torch.isfinite(p.grad).all()]
    gn = torch.nn.utils.clip_grad_norm_(params, 10.0)
    opt.step()
    bad_p = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
    if it % 10 == 0 or bad_g or bad_p:
        print("step %d loss %.4e gradnorm %.4e bev finite %s max|bev| %.3e" % (
            it, loss.item(), float(gn), bool(torch.isfinite(bev).all()), bev.abs().max().item()), flush=True)
    if bad_g or bad_p:
        print("bad grads:", bad_g[:12])
        print("bad params:", bad_p[:12])
        break
else:
    print("all finite after %d steps" % steps)
