#!/usr/bin/env python
"""Per-layer wgrad timing on the bench workload's voxel sets."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from msmdfusion_amd import kernels as K
from msmdfusion_amd import synthetic as S
from msmdfusion_amd.voxelize import Voxelization
from tools.split_bench import timed

dev = torch.device("cuda:0")
vox = Voxelization(S.VOXEL_SIZE, S.POINT_CLOUD_RANGE, 10, (120000, 160000))
clouds = [torch.from_numpy(S.lidar_sweep(i)).to(dev) for i in range(4)]
coors = [F.pad(c, (1, 0), value=b) for b, (_, c, _) in
         enumerate(vox.forward_batch(clouds, fused_mean=True))]
idx = torch.cat(coors).contiguous()
shape = list(S.SPARSE_SHAPE)
stages, strided = [], []
for i, pad in enumerate([1, 1, [0, 1, 1]]):
    stages.append((idx, shape))
    oidx, nf, nb, oshape = K.rulebook_conv(idx, 4, shape, 3, 2, pad)
    strided.append((idx.shape[0], oidx.shape[0], nf))
    idx, shape = oidx, oshape
stages.append((idx, shape))
tot = 0.0
for si, cin, cout, cnt in [(0, 5, 16, 1), (0, 16, 16, 4), (1, 32, 32, 4), (2, 64, 64, 4), (3, 128, 128, 4)]:
    idx, shape = stages[si]
    n = idx.shape[0]
    nbr = K.rulebook_subm(idx, 4, shape, 3)
    pairs, num = K.rulebook_pairs(nbr)
    P = int(num.sum())
    f, g = torch.randn(n, cin, device=dev), torch.randn(n, cout, device=dev)
    t = timed(lambda: K.conv_wgrad(f, g, pairs, num))
    extra = ""
    if K.wgrad_split_supported(cin, cout):
        for np_ in (3, 2):
            ts = timed(lambda: K.conv_wgrad_split(f, g, pairs, num, np_))
            d32, dsp = K.conv_wgrad(f, g, pairs, num), K.conv_wgrad_split(f, g, pairs, num, np_)
            extra += " | split%d %.0f us %.1f TF maxdiff/max %.1e" % (
                np_, ts, 2.0 * P * cin * cout / ts / 1e6,
                (d32 - dsp).abs().max().item() / d32.abs().max().item())
            if np_ == 3:
                t = min(t, ts)
    tot += t * cnt
    print("subm %3d->%3d n=%6d pairs=%7d  %6.0f us  %5.1f TF  x%d%s" % (cin, cout, n, P, t, 2.0 * P * cin * cout / t / 1e6, cnt, extra), flush=True)
for li, (cin, cout) in enumerate([(16, 32), (32, 64), (64, 128)]):
    n_in, n_out, nf = strided[li]
    pairs, num = K.rulebook_pairs(nf, ld=max(n_in, n_out))
    P = int(num.sum())
    f, g = torch.randn(n_in, cin, device=dev), torch.randn(n_out, cout, device=dev)
    t = timed(lambda: K.conv_wgrad(f, g, pairs, num))
    tot += t
    print("down %3d->%3d n=%6d->%6d pairs=%7d  %6.0f us  %5.1f TF" % (cin, cout, n_in, n_out, P, t, 2.0 * P * cin * cout / t / 1e6), flush=True)
print("total per step ~ %.0f us" % tot)
