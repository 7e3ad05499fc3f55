#!/usr/bin/env python
"""Times furthest point sampling (2048 samples) on LC-shaped inputs: the voxel
coordinates of synthetic virtual points at stage 0 (0.075 m) and stage 1
(0.15 m), in first-touch order (what hard voxelization emits) and shuffled (the
pruned kernel's worst case), and checks every result against the oracle.

    python tools/fps_bench.py            # pruned kernel (default)
    MSMD_FPS_PRUNE=0 python tools/fps_bench.py
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msmdfusion_amd import kernels as K  # noqa: E402
from msmdfusion_amd import synthetic as S  # noqa: E402
from oracle import oracle as O  # noqa: E402


def voxel_coords(seed, scale, n_pts):
    p = S.virtual_points(seed, n=n_pts)[:, :3]
    vs = np.array(S.VOXEL_SIZE) * scale
    c = np.floor((p - np.array(S.POINT_CLOUD_RANGE[:3])) / vs).astype(np.int64)[:, ::-1]  # z,y,x
    _, first = np.unique(c, axis=0, return_index=True)
    return np.ascontiguousarray(c[np.sort(first)], dtype=np.float32)


def main():
    dev = torch.device("cuda:0")
    mode = "plain" if os.environ.get("MSMD_FPS_PRUNE") == "0" else "pruned"
    for name, scale, n_pts in [("stage0", 1, 50000), ("stage0-big", 1, 56000), ("stage1", 2, 50000)]:
        a, b = voxel_coords(0, scale, n_pts), voxel_coords(1, scale, n_pts)
        n = min(a.shape[0], b.shape[0])
        for order in ("first-touch", "shuffled"):
            xyz = np.stack([a[:n], b[:n]])
            if order == "shuffled":
                rng = np.random.RandomState(0)
                xyz = np.stack([x[rng.permutation(n)] for x in xyz])
            t = torch.from_numpy(xyz).to(dev)
            got = K.furthest_point_sample(t, 2048)
            torch.cuda.synchronize()
            ok = np.array_equal(got.cpu().numpy(), O.furthest_point_sample(xyz, 2048))
            ts = []
            for _ in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                K.furthest_point_sample(t, 2048)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            print("%-6s %-10s %-11s n=%5d x2  %.3f ms  (%.2f us/round)  oracle-exact=%s"
                  % (mode, name, order, n, min(ts), min(ts) / 2047 * 1e3, ok), flush=True)


if __name__ == "__main__":
    main()
