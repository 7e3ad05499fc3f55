"""Run-to-run determinism of the split conv kernels at a given plane count on bench-shaped layers
(a race shows as a result that differs from the first call's)."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from msmdfusion_amd import kernels as K
from msmdfusion_amd import synthetic as S
dev = torch.device("cuda:0")
planes = int(sys.argv[1]) if len(sys.argv) > 1 else 2
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 4
clouds = [torch.from_numpy(S.lidar_sweep(i)).to(dev) for i in range(nb)]
res = K.hard_voxelize_batch(clouds, S.VOXEL_SIZE, S.POINT_CLOUD_RANGE, 10, 120000, want_voxels=False, want_mean=True)
idx = torch.cat([F.pad(r[1], (1, 0), value=i) for i, r in enumerate(res)]).contiguous()
shape = list(S.SPARSE_SHAPE)
stages = []
for pad in [1, 1, [0, 1, 1]]:
    stages.append((idx, shape))
    idx, _, _, shape = K.rulebook_conv(idx, nb, shape, 3, 2, pad)
stages.append((idx, shape))
side = torch.cuda.Stream()
for si, cin, cout in [(2, 64, 64), (2, 128, 128), (3, 192, 192), (1, 96, 96), (0, 80, 80), (1, 64, 128), (3, 128, 128), (0, 32, 32)]:
    idx, shape = stages[si]
    n = idx.shape[0]
    nbr = K.rulebook_subm(idx, nb, shape, 3)
    plan = K.rulebook_plan(nbr, tile_rows=(128,), want_pairs=True)
    pairs, num = plan["pairs"]
    order, tiled, pre = plan["order"], plan["tiled"], plan["prefix"][128]
    gen = torch.Generator(device=dev).manual_seed(cin * 1000 + cout)
    f = torch.randn(n, cin, device=dev, generator=gen)
    g = torch.randn(n, cout, device=dev, generator=gen)
    w = torch.randn(27, cin, cout, device=dev, generator=gen) * 0.05
    bad_w = bad_f = 0
    dw0 = K.conv_wgrad_split(f, g, pairs, num, planes) if K.wgrad_split_supported(cin, cout) else None
    ws = K.pack_weight_split(w, planes)
    o0 = K.conv_forward_split(f, ws, tiled, n, cout, planes, row_order=order, tile_prefix=pre)
    for r in range(reps):
        # a second stream keeps the chip busy with other conv work, as the bench's side streams do
        with torch.cuda.stream(side):
            K.conv_forward_split(f, ws, tiled, n, cout, planes, row_order=order, tile_prefix=pre)
        if dw0 is not None:
            dw = K.conv_wgrad_split(f, g, pairs, num, planes)
            bad_w += int(not torch.equal(dw, dw0))
        o = K.conv_forward_split(f, ws, tiled, n, cout, planes, row_order=order, tile_prefix=pre)
        bad_f += int(not torch.equal(o, o0))
    torch.cuda.synchronize()
    print("planes %d  %d->%d n=%d: wgrad differs %d/%d, fwd differs %d/%d, finite %s" % (
        planes, cin, cout, n, bad_w, reps, bad_f, reps,
        bool(torch.isfinite(o0).all()) and (dw0 is None or bool(torch.isfinite(dw0).all()))), flush=True)
