import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench, types
from msmdfusion_amd import distributed as D
args = types.SimpleNamespace(steps=20, warmup=5, diag=False)
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
D.init_distributed(device=dev)
for wl, planes in [("lc", None), ("lc_b4", None), ("lc", None), ("lc", "1"), ("lc", None)]:
    if planes: os.environ["MSMD_CONV_PLANES"] = planes
    else: os.environ.pop("MSMD_CONV_PLANES", None)
    r = bench.run_workload(wl, args, dev, 0, 1, False)
    print(wl, planes, r["value"], r["ms_per_step"], "reserved GB %.2f" % (torch.cuda.memory_reserved()/2**30), flush=True)
