import os, sys, argparse, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as B
from msmdfusion_amd import distributed as D
args = argparse.Namespace(steps=20, warmup=5, diag=False, gpus=1)
dev = torch.device("cuda:0")
D.init_distributed(device=dev)
seq = sys.argv[1:]
for item in seq:
    if item == "cpu":
        t = time.time(); r = B.cpu_baseline("lc"); print("cpu_baseline", r["value"], round(time.time() - t, 1), flush=True)
        continue
    pl, wl = item.split(":")
    os.environ["MSMD_CONV_PLANES"] = pl
    try:
        r = B.run_workload(wl, args, dev, 0, 1, False)
        print(item, r["value"], flush=True)
    except AssertionError as e:
        print(item, "FAILED", str(e)[:1500], flush=True)
