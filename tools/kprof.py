"""Phase timing of the split forward kernel (an instrumented build of the library:
`make -C msmdfusion_amd/csrc PROF=1 OUT=../libmsmd_hip_prof.so`, then
`MSMD_LIB=msmdfusion_amd/libmsmd_hip_prof.so python tools/kprof.py [--lc]`).
Cycles per item of wave 1 of the first 8 workgroups, summed per phase.  4-wave kernel:
top wait / barrier / weight DMA issue / gather issue + index fetch / wait rows / convert +
multiply / glue.  Ping-pong kernel (> 64 output channels): LOAD segment = glue, weight DMA
issue, gather issue + idx, wait rows, convert, barrier; MULTIPLY segment = MFMAs, wait for
the next weights, barrier."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from msmdfusion_amd import kernels as K
from msmdfusion_amd import synthetic as S
from msmdfusion_amd._lib import LIB_PATH
from msmdfusion_amd.voxelize import Voxelization
dev = torch.device("cuda:0")
vox = Voxelization(S.VOXEL_SIZE, S.POINT_CLOUD_RANGE, 10, (120000, 160000))
clouds = [torch.from_numpy(S.lidar_sweep(i)).to(dev) for i in range(4)]
coors = [F.pad(c, (1, 0), value=b) for b, (_, c, _) in enumerate(vox.forward_batch(clouds, fused_mean=True))]
idx = torch.cat(coors).contiguous(); shape = list(S.SPARSE_SHAPE)
NAMES4 = {0: "top wait (weights)", 1: "barrier", 2: "weight DMA issue", 3: "gather issue + idx",
          4: "wait rows", 5: "convert + multiply", 6: "loop glue"}
NAMESPP = {6: "glue", 2: "weight DMA issue", 3: "gather issue + idx", 4: "wait rows", 8: "convert",
           1: "barrier (end of load)", 5: "MFMAs", 0: "wait next weights", 9: "barrier (end of multiply)"}
h = ctypes.CDLL(LIB_PATH)
stages = []
for pad in (1, 1, [0, 1, 1]):
    stages.append((idx, shape))
    idx, _, _, shape = K.rulebook_conv(idx, 4, shape, 3, 2, pad)
stages.append((idx, shape))
layers = [(1, 32), (2, 64), (3, 128)]
if "--lc" in sys.argv:
    layers = [(0, 80), (1, 96), (3, 128), (3, 192)]
for si, c in layers:
    idx, shape = stages[si]
    nbr = K.rulebook_subm(idx, 4, shape, 3); n = idx.shape[0]
    f = torch.randn(n, c, device=dev); ws = K.pack_weight_split(torch.randn(27, c, c, device=dev) * 0.05, 3)
    o, nt = K.rulebook_tiling(nbr); pre = K.tile_prefix(nt, K.split_tile_rows(c))
    for _ in range(3):
        K.conv_forward_split(f, ws, nt, n, c, 3, row_order=o, tile_prefix=pre)
    buf = (ctypes.c_ulonglong * 16)()
    h.msmd_debug_kprof(buf)
    K.conv_forward_split(f, ws, nt, n, c, 3, row_order=o, tile_prefix=pre)
    h.msmd_debug_kprof(buf)
    v = list(buf)
    items = max(v[7], 1)
    pp = K.split_tile_rows(c) == 256
    names = NAMESPP if pp else NAMES4
    tot = sum(v[j] for j in names)
    print("subm %d -> %d, %d rows (%s): %d items over 8 waves; cycles per item: %s | total %.0f" % (
        c, c, n, "ping-pong" if pp else "4 waves", items,
        ", ".join("%s %.0f" % (names[j], v[j] / items) for j in names), tot / items))
