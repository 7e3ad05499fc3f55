#!/usr/bin/env python
"""Times the forward conv kernel on bench-shaped layers (run under MSMD_FWD / MSMD_FWD_PW /
MSMD_FWD_DBG settings to ablate it; results are wrong by design with DBG != 0, which only
a library built with -DMSMD_FWD_BLOCK_DBG honours)."""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msmdfusion_amd import kernels as K  # noqa: E402
from msmdfusion_amd import synthetic as S  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, n=20):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


clouds = [torch.from_numpy(S.lidar_sweep(i)).to(dev) for i in range(4)]
res = K.hard_voxelize_batch(clouds, S.VOXEL_SIZE, S.POINT_CLOUD_RANGE, 10, 120000,
                            want_voxels=False, want_mean=True)
idx = torch.cat([F.pad(r[1], (1, 0), value=i) for i, r in enumerate(res)]).contiguous()
shape = list(S.SPARSE_SHAPE)
stages = []
for pad in [1, 1, [0, 1, 1]]:
    stages.append((idx, shape))
    idx, _, _, shape = K.rulebook_conv(idx, 4, shape, 3, 2, pad)
stages.append((idx, shape))
layers = [(3, 128, 128), (2, 64, 128), (1, 96, 96), (3, 192, 192)]
if len(sys.argv) > 1:
    layers = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
out = []
for si, cin, cout in layers:
    idx, shape = stages[si]
    n = idx.shape[0]
    nbr = K.rulebook_subm(idx, 4, shape, 3)
    P = int((nbr >= 0).sum())
    order = K.rulebook_tiling(nbr, want_table=False)[0]
    nbr_t = K.permute_cols(nbr, order)
    pre = K.tile_prefix(nbr_t, K.split_tile_rows(cout))
    gen = torch.Generator(device=dev).manual_seed(cin * 1000 + cout)
    f = torch.randn(n, cin, device=dev, generator=gen)
    w = torch.randn(27, cin, cout, device=dev, generator=gen) * 0.05
    ws = K.pack_weight_split(w, 3)
    t = timed(lambda: K.conv_forward_split(f, ws, nbr_t, n, cout, 3, row_order=order,
                                           tile_prefix=pre))
    out.append("%d->%d %.0f us (%.0f TF)" % (cin, cout, t, 2.0 * P * cin * cout / t / 1e6))
print("FWD=%s PW=%s DBG=%s: " % (os.environ.get("MSMD_FWD", "block"), os.environ.get("MSMD_FWD_PW", "4"),
                                os.environ.get("MSMD_FWD_DBG", "0")) + " | ".join(out))
