"""Phase timing of the wgrad plane kernel (library built with `make PROF=1`)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from msmdfusion_amd import kernels as K
from msmdfusion_amd import synthetic as S
from msmdfusion_amd.voxelize import Voxelization
dev = torch.device("cuda:0")
vox = Voxelization(S.VOXEL_SIZE, S.POINT_CLOUD_RANGE, 10, (120000, 160000))
clouds = [torch.from_numpy(S.lidar_sweep(i)).to(dev) for i in range(4)]
coors = [F.pad(c, (1, 0), value=b) for b, (_, c, _) in enumerate(vox.forward_batch(clouds, fused_mean=True))]
idx = torch.cat(coors).contiguous(); shape = list(S.SPARSE_SHAPE)
for pad in (1, 1, [0, 1, 1]):
    idx, _, _, shape = K.rulebook_conv(idx, 4, shape, 3, 2, pad)
n = idx.shape[0]; c = 128
nbr = K.rulebook_subm(idx, 4, shape, 3)
pairs, num = K.rulebook_pairs(nbr)
fp = K.split_planes(torch.randn(n, c, device=dev), 3); gp = K.split_planes(torch.randn(n, c, device=dev), 3)
h = ctypes.CDLL(os.path.join(ROOT, "msmdfusion_amd", "libmsmd_hip.so"))
buf = (ctypes.c_ulonglong * 16)()
for _ in range(3):
    K.conv_wgrad_planes(fp, gp, pairs, num)
h.msmd_debug_wprof(buf)
K.conv_wgrad_planes(fp, gp, pairs, num)
h.msmd_debug_wprof(buf)
v = list(buf); st = max(v[7], 1)
names = ["prologue", "DMA wait", "barrier", "DMA issue", "row indices", "LDS reads + MFMAs"]
print("stages sampled %d; cycles per stage, consumer wave: %s" % (
    st, ", ".join("%s %.0f" % (names[j], v[j] / st) for j in (0, 2, 5))))
print("                                   producer wave: %s" % (
    ", ".join("%s %.0f" % (names[j], v[8 + j] / st) for j in range(5))))
