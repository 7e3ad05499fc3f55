"""Per-step GPU time of the headline workload (events on the main stream, no host sync
inside), with and without the index prefetcher."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from msmdfusion_amd import synthetic as S
from msmdfusion_amd.prefetch import IndexPrefetcher
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = bench.Backbone().to(dev).train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
clouds = [torch.from_numpy(S.lidar_sweep(i)).to(dev) for i in range(4)]
use = os.environ.get("MSMD_PREFETCH", "0") == "1"
pf = IndexPrefetcher(model.prepare, dev) if use else None
pending = [pf.submit(clouds)] if use else None
evs = []
import time
host = []
dalloc = []
def step():
    t0 = time.perf_counter()
    if use:
        pending.append(pf.submit(clouds)); t = pending.pop(0)
        out = model(clouds, prepared=pf.take(t))
    else:
        out = model(clouds)
    out.mean().backward()
    opt.step(); opt.zero_grad(set_to_none=True)
    if use: pf.retire(t)
    e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
    host.append((time.perf_counter() - t0) * 1e3)
    st = torch.cuda.memory_stats()
    dalloc.append((st.get("num_device_alloc", 0), len(pf._retired) if use else 0, st.get("reserved_bytes.all.current", 0) >> 20,
                   st.get("allocated_bytes.all.current", 0) >> 20, st.get("inactive_split_bytes.all.current", 0) >> 20))
for i in range(60):
    step()
torch.cuda.synchronize()
d = [evs[i].elapsed_time(evs[i + 1]) for i in range(len(evs) - 1)]
print("prefetch", use, " ".join("%.1f" % t for t in d[10:]))
print("mean of last 40: %.2f ms" % (sum(d[-40:]) / 40))
print("host ms/step:", " ".join("%.1f" % t for t in host[10:]))
print("device allocs / retired tickets / reservedMB / allocatedMB / inactive-splitMB at steps:", [dalloc[i] for i in (5, 10, 20, 25, 26, 27, 28, 30, 40, 59)])
