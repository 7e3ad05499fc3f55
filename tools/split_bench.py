#!/usr/bin/env python
"""fp32-MFMA vs split-bf16 implicit GEMM on the bench workload's real voxel
sets: time, and error of both against an fp64 evaluation of the same sum.

    python tools/split_bench.py [--planes 3] [--wgrad]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, n=10):
    import torch
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1000.0 / n


def ref64(f, w, nbr):
    import torch
    out = torch.zeros((nbr.shape[1], w.shape[2]), dtype=torch.float64, device=f.device)
    f64, w64 = f.double(), w.double()
    for k in range(nbr.shape[0]):
        idx = nbr[k].long()
        m = idx >= 0
        out[m] += f64[idx[m]] @ w64[k]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--planes", type=int, default=3)
    ap.add_argument("--wgrad", action="store_true")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--lc", action="store_true", help="the fusion stack's channel widths")
    args = ap.parse_args()
    import torch
    import torch.nn.functional as F
    from msmdfusion_amd import kernels as K
    from msmdfusion_amd import synthetic as S
    from msmdfusion_amd.voxelize import Voxelization
    dev = torch.device("cuda:0")
    vox = Voxelization(S.VOXEL_SIZE, S.POINT_CLOUD_RANGE, 10, (120000, 160000))
    clouds = [torch.from_numpy(S.lidar_sweep(i)).to(dev) for i in range(4)]
    coors = [F.pad(c, (1, 0), value=b) for b, (_, c, _) in
             enumerate(vox.forward_batch(clouds, fused_mean=True))]
    idx = torch.cat(coors).contiguous()
    shape = list(S.SPARSE_SHAPE)
    stages = []
    for i, pad in enumerate([1, 1, [0, 1, 1]]):
        stages.append((idx, shape))
        idx, _, _, shape = K.rulebook_conv(idx, 4, shape, 3, 2, pad)
    stages.append((idx, shape))
    torch.manual_seed(0)
    layers = [(3, 128, 128), (2, 64, 64), (2, 64, 128), (1, 32, 32), (1, 32, 64)]
    if args.lc:
        layers = [(0, 80, 80), (0, 80, 96), (1, 96, 96), (1, 96, 128), (2, 128, 192),
                  (3, 192, 192)]
    for si, cin, cout in layers:
        idx, shape = stages[si]
        n = idx.shape[0]
        nbr = K.rulebook_subm(idx, 4, shape, 3)
        pairs_n = int((nbr >= 0).sum())
        order = K.rulebook_tiling(nbr, want_table=False)[0]
        f = torch.randn(n, cin, device=dev)
        w = torch.randn(27, cin, cout, device=dev) * 0.05
        flops = 2.0 * pairs_n * cin * cout
        wp = K.pack_weight(w)
        t32 = timed(lambda: K.conv_forward(f, wp, nbr, n, cout, row_order=order))
        line = "fwd %3d->%3d n=%d pairs=%d | fp32 %.0f us %.1f TF" % (
            cin, cout, n, pairs_n, t32, flops / t32 / 1e6)
        if K.split_supported(cin, cout):
            ws = K.pack_weight_split(w, args.planes)
            nbr_t = K.permute_cols(nbr, order)
            ts = timed(lambda: K.conv_forward_split(f, ws, nbr_t, n, cout, args.planes,
                                                    row_order=order))
            line += " | split%d %.0f us %.1f TF" % (args.planes, ts, flops / ts / 1e6)
            pre = K.tile_prefix(nbr_t, K.split_tile_rows(cout))
            tk = timed(lambda: K.conv_forward_split(f, ws, nbr_t, n, cout, args.planes,
                                                    row_order=order, tile_prefix=pre))
            line += " | stream-K %.0f us %.1f TF (frac %.3f)" % (tk, flops / tk / 1e6,
                                                                flops / tk / 1e6 / 419.4)
            if args.check:
                ok = K.conv_forward_split(f, ws, nbr_t, n, cout, args.planes, row_order=order,
                                          tile_prefix=pre).double()
                r = ref64(f, w, nbr)
                line += " | sk err %.2e" % ((ok - r).abs().max().item() / r.abs().max().item())
                o32 = K.conv_forward(f, wp, nbr, n, cout, row_order=order).double()
                osp = K.conv_forward_split(f, ws, nbr_t, n, cout, args.planes,
                                           row_order=order).double()
                onat = K.conv_forward_split(f, ws, nbr, n, cout, args.planes,
                                            split_tiles=False).double()
                otil = K.conv_forward_split(f, ws, nbr_t, n, cout, args.planes, row_order=order,
                                            split_tiles=False).double()
                assert torch.equal(onat, otil), "tile order changed the result"
                sc = r.abs().max().item()
                line += " | max|err|/max|out|: fp32 %.2e split %.2e" % (
                    (o32 - r).abs().max().item() / sc, (osp - r).abs().max().item() / sc)
        print(line, flush=True)
        if args.wgrad and hasattr(K, "conv_wgrad_split"):
            pairs, num = K.rulebook_pairs(nbr)
            g = torch.randn(n, cout, device=dev)
            t32 = timed(lambda: K.conv_wgrad(f, g, pairs, num))
            line = "wgrad %3d x %3d | fp32 %.0f us %.1f TF" % (cin, cout, t32, flops / t32 / 1e6)
            if K.wgrad_split_supported(cin, cout):
                ts = timed(lambda: K.conv_wgrad_split(f, g, pairs, num, args.planes))
                line += " | split%d %.0f us %.1f TF" % (args.planes, ts, flops / ts / 1e6)
                if args.check:
                    r = torch.zeros(27, cin, cout, dtype=torch.float64, device=dev)
                    for k in range(27):
                        m = nbr[k] >= 0
                        r[k] = f.double()[nbr[k][m].long()].t() @ g.double()[m]
                    d32 = K.conv_wgrad(f, g, pairs, num).double()
                    dsp = K.conv_wgrad_split(f, g, pairs, num, args.planes).double()
                    sc = r.abs().max().item()
                    line += " | err fp32 %.2e split %.2e" % (
                        (d32 - r).abs().max().item() / sc, (dsp - r).abs().max().item() / sc)
            print(line, flush=True)


if __name__ == "__main__":
    main()
