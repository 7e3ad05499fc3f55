#!/usr/bin/env python
"""Times nn_search at the LC stage shapes (2048 FPS representatives / raw queries against the
LiDAR voxels of one sample)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msmdfusion_amd import kernels as K  # noqa: E402
from msmdfusion_amd import synthetic as S  # noqa: E402

dev = torch.device("cuda:0")
for nq, nk, shape in [(2048, 19000, [41, 1440, 1440]), (2048, 37000, [21, 720, 720]),
                      (1500, 34000, [11, 360, 360]), (600, 22000, [5, 180, 180])]:
    q = torch.from_numpy(S.random_voxel_indices(nq, 1, shape, seed=1)[:, 1:]).to(dev).contiguous()
    k = torch.from_numpy(S.random_voxel_indices(nk, 1, shape, seed=2)[:, 1:]).to(dev).contiguous()
    for _ in range(3):
        K.nn_search(q, k, 13.3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        K.nn_search(q, k, 13.3)
    torch.cuda.synchronize()
    print("nn_search nq=%d nk=%d: %.1f us" % (q.shape[0], k.shape[0],
                                             (time.perf_counter() - t0) / 20 * 1e6))
