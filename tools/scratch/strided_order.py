"""Does the mask-sorted tiling order pay for the strided convs (each order is used by one launch)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from msmdfusion_amd import kernels as K
from msmdfusion_amd import synthetic as S
from msmdfusion_amd.voxelize import Voxelization
from tools.split_bench import timed
dev = torch.device("cuda:0")
vox = Voxelization(S.VOXEL_SIZE, S.POINT_CLOUD_RANGE, 10, (120000, 160000))
clouds = [torch.from_numpy(S.lidar_sweep(i)).to(dev) for i in range(4)]
coors = [F.pad(c, (1, 0), value=b) for b, (_, c, _) in enumerate(vox.forward_batch(clouds, fused_mean=True))]
idx = torch.cat(coors).contiguous()
shape = list(S.SPARSE_SHAPE)
for i, (pad, cin, cout) in enumerate([(1, 16, 32), (1, 32, 64), ([0, 1, 1], 64, 128)]):
    oidx, nf, nb, oshape = K.rulebook_conv(idx, 4, shape, 3, 2, pad)
    n_in, n_out = idx.shape[0], oidx.shape[0]
    if K.split_supported(cin, cout):
        f = torch.randn(n_in, cin, device=dev); g = torch.randn(n_out, cout, device=dev)
        w = torch.randn(27, cin, cout, device=dev) * 0.05
        ws, wt = K.pack_weight_split(w, 3), K.pack_weight_split(w, 3, transpose=True)
        t_order = timed(lambda: K.permute_cols(nf, K.row_mask_order(nf)))
        of = K.row_mask_order(nf); nft = K.permute_cols(nf, of)
        ob = K.row_mask_order(nb); nbt = K.permute_cols(nb, ob)
        print("down %d->%d n %d->%d: order+permute %.0f us | fwd sorted %.0f us natural %.0f us | dgrad sorted %.0f us natural %.0f us" % (
            cin, cout, n_in, n_out, t_order,
            timed(lambda: K.conv_forward_split(f, ws, nft, n_out, cout, 3, row_order=of)),
            timed(lambda: K.conv_forward_split(f, ws, nf, n_out, cout, 3)),
            timed(lambda: K.conv_forward_split(g, wt, nbt, n_in, cin, 3, row_order=ob)) if K.split_supported(cout, cin) else -1,
            timed(lambda: K.conv_forward_split(g, wt, nb, n_in, cin, 3)) if K.split_supported(cout, cin) else -1), flush=True)
    idx, shape = oidx, oshape
# SubM at stage 3 for comparison
nbr = K.rulebook_subm(idx, 4, shape, 3); n = idx.shape[0]
f = torch.randn(n, 128, device=dev); ws = K.pack_weight_split(torch.randn(27, 128, 128, device=dev) * 0.05, 3)
o = K.row_mask_order(nbr); nt = K.permute_cols(nbr, o)
print("subm 128: sorted %.0f us natural %.0f us" % (timed(lambda: K.conv_forward_split(f, ws, nt, n, 128, 3, row_order=o)),
                                                   timed(lambda: K.conv_forward_split(f, ws, nbr, n, 128, 3))))
# SubM at stages 1..3, sorted vs natural, forward and dgrad (flip)
idx = torch.cat(coors).contiguous(); shape = list(S.SPARSE_SHAPE)
for i, (pad, c) in enumerate([(1, 32), (1, 64), ([0, 1, 1], 128)]):
    idx, _, _, shape = K.rulebook_conv(idx, 4, shape, 3, 2, pad)
    nbr = K.rulebook_subm(idx, 4, shape, 3); n = idx.shape[0]
    f = torch.randn(n, c, device=dev); ws = K.pack_weight_split(torch.randn(27, c, c, device=dev) * 0.05, 3)
    t_order = timed(lambda: K.permute_cols(nbr, K.row_mask_order(nbr)))
    o = K.row_mask_order(nbr); nt = K.permute_cols(nbr, o)
    print("subm %d n=%d: order+permute %.0f us | sorted %.0f us natural %.0f us (x8 launches per step)" % (
        c, n, t_order, timed(lambda: K.conv_forward_split(f, ws, nt, n, c, 3, row_order=o)),
        timed(lambda: K.conv_forward_split(f, ws, nbr, n, c, 3))), flush=True)
