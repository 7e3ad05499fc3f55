#!/usr/bin/env python
"""On-device sweep of the implicit-GEMM conv kernel's tiling knobs on the
real voxel sets of the bench workload (4 synthetic clouds).  Each knob set
needs a fresh process (the library reads MSMD_FWD_* once), so this script
re-executes itself per configuration.

    python tools/conv_sweep.py            # run the sweep
"""
import itertools
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one():
    import torch
    import torch.nn.functional as F
    from msmdfusion_amd import kernels as K
    from msmdfusion_amd import synthetic as S
    from msmdfusion_amd.voxelize import Voxelization
    dev = torch.device("cuda:0")
    vox = Voxelization(S.VOXEL_SIZE, S.POINT_CLOUD_RANGE, 10, (120000, 160000))
    clouds = [torch.from_numpy(S.lidar_sweep(i)).to(dev) for i in range(4)]
    coors = [F.pad(c, (1, 0), value=b) for b, (_, c, _) in
             enumerate(vox.forward_batch(clouds, fused_mean=True))]
    idx = torch.cat(coors).contiguous()
    shape = list(S.SPARSE_SHAPE)
    stages = []
    for i, pad in enumerate([1, 1, [0, 1, 1]]):
        stages.append((idx, shape))
        idx, _, _, shape = K.rulebook_conv(idx, 4, shape, 3, 2, pad)
    stages.append((idx, shape))
    mode = os.environ.get("SWEEP_ORDER", "mask")
    res = []
    for si, c in [(3, 128), (2, 64), (1, 32)]:
        idx, shape = stages[si]
        n = idx.shape[0]
        nbr = K.rulebook_subm(idx, 4, shape, 3)
        pairs = int((nbr >= 0).sum())
        order = None if mode == "none" else K.row_mask_order(nbr)
        if mode == "mask_nocounter":
            os.environ["SWEEP_NOCOUNTER"] = "1"
        f = torch.randn(n, c, device=dev)
        w = K.pack_weight(torch.randn(27, c, c, device=dev) * 0.05)
        for _ in range(3):
            K.conv_forward(f, w, nbr, n, c, row_order=order)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            K.conv_forward(f, w, nbr, n, c, row_order=order)
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) * 100
        res.append("C%d n=%d %.0fus %.1fTF" % (c, n, us, 2.0 * pairs * c * c / us / 1e6))
    print(os.environ.get("MSMD_FWD_R", "-"), os.environ.get("MSMD_FWD_SLOTS", "-"), mode,
          "dbg=" + os.environ.get("MSMD_DBG", "0"), "pipe=" + os.environ.get("MSMD_FWD_PIPE", "1"), "kc=" + os.environ.get("MSMD_FWD_KC", "0"),
          " | ".join(res), flush=True)


if __name__ == "__main__":
    if os.environ.get("SWEEP_CHILD"):
        one()
    else:
        combos = [dict(MSMD_FWD_R=r, MSMD_FWD_SLOTS=sl, SWEEP_ORDER=m, MSMD_DBG=d, MSMD_FWD_PIPE=pp,
                       MSMD_FWD_KC=kc)
                  for kc, pp, r, sl, m, d in itertools.product(
                      os.environ.get("SW_KC", "0").split(","),
                      os.environ.get("SW_PIPE", "1").split(","),
                      os.environ.get("SW_R", "2").split(","),
                      os.environ.get("SW_SLOTS", "3").split(","),
                      os.environ.get("SW_ORDER", "mask,none").split(","),
                      os.environ.get("SW_DBG", "0,1,2,3").split(","))]
        for c in combos:
            subprocess.run([sys.executable, __file__], env=dict(os.environ, SWEEP_CHILD="1", **c))
