#!/usr/bin/env python
"""cProfile of the TRAINING thread over pipelined LC steps (the worker thread is not
profiled): where the host time of the feature pass goes."""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from msmdfusion_amd import distributed as D  # noqa: E402
from msmdfusion_amd import synthetic as S  # noqa: E402
from msmdfusion_amd.prefetch import IndexPrefetcher  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = bench.FusionBackbone().to(dev).train()
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.01, fused=True)
clouds = [torch.from_numpy(S.lidar_sweep(i)).to(dev) for i in range(2)]
batch = (clouds, [torch.from_numpy(S.virtual_points(i)).to(dev) for i in range(2)])
target = torch.randn(2, 640, 180, 180, device=dev).contiguous(memory_format=torch.channels_last)
sys.setswitchinterval(0.0005)
pf = IndexPrefetcher(model.prepare, dev, threaded=True)
step = D.TrainStep(model, params, opt, lambda bev: (bev * target).mean(), pf, 10.0)
step.prime(batch)
for _ in range(25):
    step(batch)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step(batch)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
