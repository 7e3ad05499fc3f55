#!/usr/bin/env python
"""fp32 forward kernels on the narrow (16-/32-channel) layers of the bench workload.
    MSMD_PIPE_MIN_NT=1 python tools/narrow_bench.py     (pipelined kernel on them too)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from msmdfusion_amd import kernels as K
from msmdfusion_amd import synthetic as S
from msmdfusion_amd.voxelize import Voxelization
from tools.split_bench import ref64, timed

dev = torch.device("cuda:0")
vox = Voxelization(S.VOXEL_SIZE, S.POINT_CLOUD_RANGE, 10, (120000, 160000))
clouds = [torch.from_numpy(S.lidar_sweep(i)).to(dev) for i in range(4)]
coors = [F.pad(c, (1, 0), value=b) for b, (_, c, _) in
         enumerate(vox.forward_batch(clouds, fused_mean=True))]
idx0 = torch.cat(coors).contiguous()
shape0 = list(S.SPARSE_SHAPE)
idx1, nf, _, shape1 = K.rulebook_conv(idx0, 4, shape0, 3, 2, 1)
torch.manual_seed(0)
for name, idx, shape, cin, cout, nbr in [
        ("subm 5->16", idx0, shape0, 5, 16, None), ("subm 16->16", idx0, shape0, 16, 16, None),
        ("strided 16->32", idx0, shape0, 16, 32, nf), ("subm 32->32", idx1, shape1, 32, 32, None)]:
    n_in = idx.shape[0]
    if nbr is None:
        nbr = K.rulebook_subm(idx, 4, shape, 3)
    n_out = nbr.shape[1]
    f = torch.randn(n_in, cin, device=dev)
    w = torch.randn(27, cin, cout, device=dev) * 0.05
    wp = K.pack_weight(w)
    order = K.rulebook_tiling(nbr, want_table=False)[0] if os.environ.get("NARROW_SORT") else None
    t = timed(lambda: K.conv_forward(f, wp, nbr, n_out, cout, row_order=order))
    o = K.conv_forward(f, wp, nbr, n_out, cout, row_order=order).double()
    r = ref64(f, w, nbr)
    print("%-15s n_out=%d pairs=%d | %.1f us | err %.2e | sum %.9e" % (
        name, n_out, int((nbr >= 0).sum()), t, (o - r).abs().max().item() / r.abs().max().item(),
        o.sum().item()), flush=True)
