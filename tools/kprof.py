"""Phase timing of the split forward kernel (library built with `make PROF=1`)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from msmdfusion_amd import kernels as K
from msmdfusion_amd import synthetic as S
from msmdfusion_amd._lib import lib
from msmdfusion_amd.voxelize import Voxelization
dev = torch.device("cuda:0")
vox = Voxelization(S.VOXEL_SIZE, S.POINT_CLOUD_RANGE, 10, (120000, 160000))
clouds = [torch.from_numpy(S.lidar_sweep(i)).to(dev) for i in range(4)]
coors = [F.pad(c, (1, 0), value=b) for b, (_, c, _) in enumerate(vox.forward_batch(clouds, fused_mean=True))]
idx = torch.cat(coors).contiguous(); shape = list(S.SPARSE_SHAPE)
names = ["top wait (weights)", "barrier", "weight DMA issue", "gather issue + idx", "wait rows",
         "convert + multiply", "loop glue", "items"]
for i, (pad, c) in enumerate([(1, 32), (1, 64), ([0, 1, 1], 128)]):
    idx, _, _, shape = K.rulebook_conv(idx, 4, shape, 3, 2, pad)
    nbr = K.rulebook_subm(idx, 4, shape, 3); n = idx.shape[0]
    f = torch.randn(n, c, device=dev); ws = K.pack_weight_split(torch.randn(27, c, c, device=dev) * 0.05, 3)
    o, nt = K.rulebook_tiling(nbr); pre = K.tile_prefix(nt, K.split_tile_rows(c))
    for _ in range(3):
        K.conv_forward_split(f, ws, nt, n, c, 3, row_order=o, tile_prefix=pre)
    buf = (ctypes.c_ulonglong * 16)()
    ctypes.CDLL(os.path.join(ROOT, "msmdfusion_amd", "libmsmd_hip.so")).msmd_debug_kprof(buf)
    K.conv_forward_split(f, ws, nt, n, c, 3, row_order=o, tile_prefix=pre)
    h = ctypes.CDLL(os.path.join(ROOT, "msmdfusion_amd", "libmsmd_hip.so"))
    h.msmd_debug_kprof(buf)
    v = list(buf)
    items = max(v[7], 1)
    tot = sum(v[:7])
    print("subm %d: %d items over 8 waves; cycles per item: %s | total %.0f" % (
        c, items, ", ".join("%s %.0f" % (names[j], v[j] / items) for j in range(7)), tot / items))
