#!/usr/bin/env python
"""CPU simulation (oracle rulebooks, no GPU): how much structural-zero matrix work each
row ordering of the split conv kernel leaves, on the bench workload's stage-3 voxel set.

  issued(wave)  = |union of the 27-bit masks of the wave's 32 rows|
  issued(tile)  = |union over the tile's 128 rows|  (items the workgroup walks)
  useful        = sum of popcounts / 32
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from msmdfusion_amd import synthetic as S  # noqa: E402
from oracle import oracle as O  # noqa: E402


def stage_indices(n_samples, stage):
    idx_all = []
    for b in range(n_samples):
        v, c, n = O.hard_voxelize(S.lidar_sweep(b), S.VOXEL_SIZE, S.POINT_CLOUD_RANGE, 10, 120000)
        idx_all.append(np.concatenate([np.full((c.shape[0], 1), b, np.int32), c], 1))
    idx = np.concatenate(idx_all)
    shape = list(S.SPARSE_SHAPE)
    for pad in [1, 1, [0, 1, 1]][:stage]:
        idx, _, _, shape = O.get_indice_pairs(idx, n_samples, shape, 3, 2, pad, 1, False)
        shape = list(shape)
    return idx, shape


def masks_of(idx, shape, batch):
    oi, pairs, num, _ = O.get_indice_pairs(idx, batch, shape, 3, 1, 1, 1, True)
    m = np.zeros(idx.shape[0], np.uint32)
    for k in range(27):
        o = pairs[k, 1, :num[k]]
        m[o] |= np.uint32(1 << k)
    return m


def popc(x):
    x = x.astype(np.uint64)
    c = np.zeros(x.shape, np.int64)
    for k in range(27):
        c += ((x >> np.uint64(k)) & np.uint64(1)).astype(np.int64)
    return c


def report(name, m):
    n = m.shape[0]
    pad = (-n) % 128
    mm = np.concatenate([m, np.zeros(pad, np.uint32)])
    waves = np.bitwise_or.reduce(mm.reshape(-1, 32), axis=1)
    tiles = np.bitwise_or.reduce(mm.reshape(-1, 128), axis=1)
    useful = popc(m).sum() / 32.0
    w_iss = popc(waves).sum()
    t_iss = popc(tiles).sum() * 4
    print("%-34s wave-issued/useful %.3f   tile-coupled/useful %.3f" % (name, w_iss / useful,
                                                                        t_iss / useful))


def gray(x):
    return x ^ (x >> 1)


def main():
    batch = 4
    idx, shape = stage_indices(batch, 3)
    m = masks_of(idx, shape, batch)
    pc = popc(m)
    print("stage 3: %d rows, mean popcount %.2f" % (m.shape[0], pc.mean()))
    report("natural order", m)
    key = ((27 - pc).astype(np.uint64) << np.uint64(27)) | m.astype(np.uint64)
    report("current: (27-popcount, mask)", m[np.argsort(key, kind="stable")])
    report("mask only", m[np.argsort(m, kind="stable")])
    # offsets reordered so that the most frequent ones are the high bits of the key
    freq = np.array([((m >> k) & 1).sum() for k in range(27)])
    perm = np.argsort(freq)                     # rare offsets -> low bits
    m2 = np.zeros_like(m, dtype=np.uint64)
    for newbit, k in enumerate(perm):
        m2 |= ((m.astype(np.uint64) >> np.uint64(k)) & np.uint64(1)) << np.uint64(newbit)
    key2 = ((27 - pc).astype(np.uint64) << np.uint64(27)) | m2
    report("(27-popcount, freq-ranked mask)", m[np.argsort(key2, kind="stable")])
    report("freq-ranked mask only", m[np.argsort(m2, kind="stable")])
    # rare-first: rare offsets as the HIGH bits (rows sharing a rare offset cluster)
    m3 = np.zeros_like(m, dtype=np.uint64)
    for newbit, k in enumerate(perm[::-1]):
        m3 |= ((m.astype(np.uint64) >> np.uint64(k)) & np.uint64(1)) << np.uint64(newbit)
    key3 = ((27 - pc).astype(np.uint64) << np.uint64(27)) | m3
    report("(27-popcount, rare-high mask)", m[np.argsort(key3, kind="stable")])
    report("rare-high mask only", m[np.argsort(m3, kind="stable")])
    # static geometric rank: corners, then edges, then faces, centre (z-neighbours first)
    def cls(k):
        dz, dy, dx = k // 9 - 1, (k // 3) % 3 - 1, k % 3 - 1
        return (abs(dz) + abs(dy) + abs(dx), abs(dz), abs(dy))
    sperm = sorted(range(27), key=cls)            # common (centre) -> low bits
    m5 = np.zeros_like(m, dtype=np.uint64)
    for newbit, k in enumerate(sperm):
        m5 |= ((m.astype(np.uint64) >> np.uint64(k)) & np.uint64(1)) << np.uint64(newbit)
    order5 = np.argsort(m5, kind="stable")
    report("static corner-high mask only", m[order5])
    # ... and tiles of that order re-sequenced heaviest first (LPT at tile granularity)
    for nm, od in (("rare-high", np.argsort(m3, kind="stable")), ("static", order5)):
        ms = m[od]
        pad = (-ms.shape[0]) % 128
        mt = np.concatenate([ms, np.zeros(pad, np.uint32)]).reshape(-1, 128)
        cost = popc(np.bitwise_or.reduce(mt, axis=1))
        tperm = np.argsort(-cost, kind="stable")
        report(nm + " + tiles by cost", mt[tperm].reshape(-1)[: ms.shape[0] + pad])
        print("    tile cost: max %d mean %.1f min %d" % (cost.max(), cost.mean(), cost.min()))
    # coarse popcount bands (4 wide) then mask
    key4 = (((27 - pc) // 4).astype(np.uint64) << np.uint64(27)) | m3
    report("(popcount band of 4, rare-high)", m[np.argsort(key4, kind="stable")])


if __name__ == "__main__":
    main()
