"""Phase timing of the whole-block wgrad kernel (library built with `make PROF=1`)."""
import ctypes
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from msmdfusion_amd import kernels as K  # noqa: E402
from msmdfusion_amd import synthetic as S  # noqa: E402

dev = torch.device("cuda:0")
h = ctypes.CDLL(os.path.join(ROOT, "msmdfusion_amd", "libmsmd_hip.so"))
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
names = ["P issue", "P convert", "P lds write", "P barrier", "C half0", "C half1", "C flush",
         "C barrier"]
for si, cin, cout in [(2, 128, 128), (1, 96, 96), (0, 80, 80), (2, 64, 64)]:
    idx, shape = stages[si]
    n = idx.shape[0]
    pairs, num = K.rulebook_pairs(K.rulebook_subm(idx, 4, shape, 3))
    steps = int(((num + 31) // 32).sum())
    f = torch.randn(n, cin, device=dev)
    g = torch.randn(n, cout, device=dev)
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.6:      # clocks ramp for several hundred ms from idle
        K.conv_wgrad_split(f, g, pairs, num, 3)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        K.conv_wgrad_split(f, g, pairs, num, 3)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 100
    buf = (ctypes.c_ulonglong * 16)()
    h.msmd_debug_wbprof(buf)
    K.conv_wgrad_split(f, g, pairs, num, 3)
    h.msmd_debug_wbprof(buf)
    v = list(buf)
    its = 16 * max(1, (steps + 255) // 256)     # iterations of the 16 timed workgroups
    print("%dx%d: %.0f us/call, %d rows, %d steps, ~%d per workgroup; cycles per iteration: %s | P %.0f C %.0f" % (
        cin, cout, us, n, steps, its // 16, ", ".join("%s %.0f" % (names[j], v[j] / its) for j in range(8)),
        sum(v[:4]) / its, sum(v[4:8]) / its))
