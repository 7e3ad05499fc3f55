"""Per-tile timeline of the split forward kernel (library built with `make PROF=1`):
which CU each workgroup landed on, when each tile started / ended, slot occupancy
over the launch.   python tools/ktrace.py [channels]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.nn.functional as F
from msmdfusion_amd import kernels as K
from msmdfusion_amd import synthetic as S
from msmdfusion_amd.voxelize import Voxelization
dev = torch.device("cuda:0")
want = int(sys.argv[1]) if len(sys.argv) > 1 else 128
sk = len(sys.argv) > 2 and sys.argv[2] == "sk"     # stream-K scheduling
vox = Voxelization(S.VOXEL_SIZE, S.POINT_CLOUD_RANGE, 10, (120000, 160000))
clouds = [torch.from_numpy(S.lidar_sweep(i)).to(dev) for i in range(4)]
coors = [F.pad(c, (1, 0), value=b) for b, (_, c, _) in enumerate(vox.forward_batch(clouds, fused_mean=True))]
idx = torch.cat(coors).contiguous(); shape = list(S.SPARSE_SHAPE)
h = ctypes.CDLL(os.path.join(ROOT, "msmdfusion_amd", "libmsmd_hip.so"))
h.msmd_debug_ktrace.argtypes = [ctypes.c_void_p, ctypes.c_int]
for i, (pad, c) in enumerate([(1, 32), (1, 64), ([0, 1, 1], 128)]):
    idx, _, _, shape = K.rulebook_conv(idx, 4, shape, 3, 2, pad)
    if c != want:
        continue
    nbr = K.rulebook_subm(idx, 4, shape, 3); n = idx.shape[0]
    f = torch.randn(n, c, device=dev); ws = K.pack_weight_split(torch.randn(27, c, c, device=dev) * 0.05, 3)
    o, nt = K.rulebook_tiling(nbr)
    pre = K.tile_prefix(nt, K.split_tile_rows(c)) if sk else None
    for _ in range(3):
        K.conv_forward_split(f, ws, nt, n, c, 3, row_order=o, tile_prefix=pre)
    buf = np.zeros((16384, 8), dtype=np.uint64)
    h.msmd_debug_ktrace(buf.ctypes.data, 16384)
    K.conv_forward_split(f, ws, nt, n, c, 3, row_order=o, tile_prefix=pre)
    cnt = h.msmd_debug_ktrace(buf.ctypes.data, 16384)
    t = buf[:cnt].astype(np.int64)
    hw, xcc, blk, tile, t0, t1, items, mask = t.T
    base = t0.min()
    t0 = (t0 - base) / 100.0; t1 = (t1 - base) / 100.0       # us
    cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    cuid = ((xcc & 0xf) * 8 + se) * 32 + sh * 16 + cu
    print("n=%d tiles=%d traced=%d blocks=%d  kernel span %.1f us" % (n, (n + 127) // 128, cnt, len(set(blk)), t1.max()))
    per_cu = {}
    for b, cidx in zip(blk, cuid):
        per_cu.setdefault(cidx, set()).add(b)
    occ = np.bincount([len(v) for v in per_cu.values()])
    print("distinct CUs %d; workgroups per CU histogram %s; per XCC %s" % (
        len(per_cu), occ.tolist(), np.bincount(xcc[np.unique(blk, return_index=True)[1]] & 0xf).tolist()))
    dur = t1 - t0
    print("us per item: mean %.2f  p10 %.2f  p50 %.2f  p90 %.2f" % tuple(
        [float((dur / np.maximum(items, 1)).mean())] + [float(np.percentile(dur / np.maximum(items, 1), p)) for p in (10, 50, 90)]))
    # per-item time as a function of start time (is the loaded phase slower?)
    for lo in range(0, int(t1.max()) + 1, 40):
        m = (t0 >= lo) & (t0 < lo + 40)
        if m.any():
            act = ((t0 < lo + 20) & (t1 > lo + 20)).sum()
            print("  start in [%3d,%3d) us: %4d tiles, items/tile %.1f, us/item %.2f | tiles running at t=%d: %d" % (
                lo, lo + 40, m.sum(), items[m].mean(), (dur[m] / np.maximum(items[m], 1)).mean(), lo + 20, act))
    # per block: busy time and finish time
    fin = {}; busy = {}
    for b, a, e in zip(blk, t0, t1):
        fin[b] = max(fin.get(b, 0), e); busy[b] = busy.get(b, 0) + e - a
    f_ = np.array(list(fin.values())); b_ = np.array(list(busy.values()))
    print("block finish time: p10 %.0f p50 %.0f p90 %.0f max %.0f | busy per block mean %.0f (%.0f%% of span)" % (
        np.percentile(f_, 10), np.percentile(f_, 50), np.percentile(f_, 90), f_.max(), b_.mean(), 100 * b_.mean() / t1.max()))
    first = t0[np.unique(blk, return_index=True)[1]]
    print("first-tile start: p50 %.1f p90 %.1f max %.1f us" % (np.percentile(first, 50), np.percentile(first, 90), first.max()))
    k = np.argsort(-dur)[:5]
    print("longest tiles:", [(int(tile[i]), int(items[i]), round(float(t0[i]), 1), round(float(dur[i]), 1)) for i in k])
