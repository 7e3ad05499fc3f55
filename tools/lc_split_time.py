#!/usr/bin/env python
"""LC step decomposition: (a) prepare() alone (index pass: voxelization, rulebooks, tilings,
neighbour search), (b) the feature pass fwd+bwd+optimizer on an already prepared batch,
(c) the pipelined step of bench.py.  Tells which of the two streams bounds the step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from msmdfusion_amd import synthetic as S

dev = torch.device("cuda:0")
torch.manual_seed(0)
lc = len(sys.argv) < 2 or sys.argv[1] == "lc"
model = (bench.FusionBackbone() if lc else bench.Backbone()).to(dev).train()
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.01, fused=True)
spg = 2 if lc else 4
clouds = [torch.from_numpy(S.lidar_sweep(i)).to(dev) for i in range(spg)]
batch = (clouds, [torch.from_numpy(S.virtual_points(i)).to(dev) for i in range(spg)]) if lc else (clouds,)
target = torch.randn(spg, 640 if lc else 256, 180, 180, device=dev)
if lc:
    target = target.contiguous(memory_format=torch.channels_last)


def timed(fn, n=10, warm=6):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def prep():
    return model.prepare(*batch) if not lc else model.path.prepare(*batch[:1], [batch[1]] * 4, nn_side_stream=False)


def feature(prepared):
    bev = model(*batch, prepared=prepared)
    (bev * target).mean().backward()
    torch.nn.utils.clip_grad_norm_(params, 10.0)
    opt.step()
    opt.zero_grad(set_to_none=True)


print("prepare alone        %.2f ms" % timed(prep))
p = prep()
print("feature pass alone   %.2f ms (prepared batch reused)" % timed(lambda: feature(p)))
with torch.no_grad():
    print("forward only         %.2f ms" % timed(lambda: model(*batch, prepared=p)))
if lc:
    def prep_side():
        return model.path.prepare(*batch[:1], [batch[1]] * 4, nn_side_stream=True)
    print("prepare, NN on side streams: %.2f ms per call (device-complete)" % timed(prep_side))
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); prep_side(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print("   host returns after %.2f ms, device done after %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5):
        prep_side()
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(60)
    p = prep_side(); torch.cuda.synchronize()
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); feature(p); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print("feature pass: host returns after %.2f ms, device done after %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5):
        feature(p)
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(30)
