#!/usr/bin/env python
"""fp32-MFMA vs split-bf16 implicit GEMM on the bench workload's real voxel
sets: time, and error of both against an fp64 evaluation of the same sum.

    python tools/split_bench.py [--planes 3] [--wgrad]
    python tools/split_bench.py --lc-b2     every conv launch of the LC step (configs[2],
                                            2 samples per GPU) alone on the chip vs inside the
                                            pipelined step: under-fill vs co-tenancy
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


def lc_b2(reps=10, sampled_steps=5, tail=False):
    """Every conv launch of one LC training step at its B = 2 shape: (a) replayed ALONE on
    an otherwise idle chip (kernels.CAPTURE keeps each launch's closure), (b) inside the
    pipelined step (index prefetch + neighbour search on their own queues, as bench.py runs
    it), timed by HIP events on the launch stream.  One line per launch, then the sums per
    template instantiation."""
    import statistics
    import torch
    import bench
    from msmdfusion_amd import distributed as D
    from msmdfusion_amd import kernels as K
    from msmdfusion_amd import synthetic as S
    from msmdfusion_amd.prefetch import IndexPrefetcher
    from msmdfusion_amd.spconv.functional import conv_planes
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    spg = bench.WORKLOADS["lc"]["spg"]
    ids = list(range(spg))
    model = (bench.FusionTailBackbone() if tail else bench.FusionBackbone()).to(dev).train()
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.01, fused=True)
    batch = ([torch.from_numpy(S.lidar_sweep(i)).to(dev) for i in ids],
             [torch.from_numpy(S.virtual_points(i)).to(dev) for i in ids])
    target = torch.randn(spg, 512 if tail else 640, 180, 180, device=dev).contiguous(
        memory_format=torch.channels_last)
    loss_fn = lambda bev: bench.mean_of_product(bev, target)    # noqa: E731
    plain = D.TrainStep(model, params, opt, loss_fn, None, 10.0)
    for _ in range(30):             # clocks + allocator
        plain(batch)
    torch.cuda.synchronize()
    K.CAPTURE = []
    plain(batch)
    cap, K.CAPTURE = K.CAPTURE, None
    torch.cuda.synchronize()
    alone = []
    for kind, meta, call in cap:
        alone.append(timed(call, reps))
    # (b) the same launches inside the pipelined step
    pf = IndexPrefetcher(model.prepare, dev, threaded=True, priority=-1, depth=2, workers=1)
    sys.setswitchinterval(0.0005)
    step = D.TrainStep(model, params, opt, loss_fn, pf, 10.0)
    step.prime(batch)
    for _ in range(60):
        step(batch)
    torch.cuda.synchronize()
    insitu = [[] for _ in cap]
    for s_ in range(sampled_steps):
        for _ in range(4):
            step(batch)
        K.PROFILE = []
        step(batch)
        prof, K.PROFILE = K.PROFILE, None
        torch.cuda.synchronize()
        assert len(prof) == len(cap), (len(prof), len(cap))
        for i, (kind, s, e, meta) in enumerate(prof):
            assert kind == cap[i][0] and meta["c_out"] == cap[i][1]["c_out"]
            insitu[i].append(s.elapsed_time(e) * 1e3)
    step.drain()
    groups = {}
    pair_cache = {}
    print("%-48s %9s %8s %9s | %8s %8s | %6s %6s" % (
        "kernel", "cin->cout", "rows", "pairs", "alone us", "step us", "TF al", "TF st"))
    for i, (kind, meta, _) in enumerate(cap):
        if "nbr" in meta:
            nbr = meta["nbr"]
            key = (nbr.data_ptr(), nbr.shape[1])
            if key not in pair_cache:
                pair_cache[key] = int((nbr >= 0).sum().item())
            pairs = pair_cache[key]
        else:
            pairs = int(meta["num"].sum().item())
        if kind == "spconv_fwd_split":
            name, _n = bench.split_instantiation(meta["c_out"], conv_planes())
        elif kind == "spconv_wgrad_split":
            name = "spconv_wgrad_block_kernel" if K.wgrad_split_supported(
                meta["c_in"], meta["c_out"]) else "spconv_wgrad_split_var_kernel"
        else:
            name = kind
        flops = 2.0 * pairs * meta["c_in"] * meta["c_out"]
        st = statistics.median(insitu[i])
        print("%-48s %4d->%-4d %8d %9d | %8.1f %8.1f | %6.1f %6.1f" % (
            name, meta["c_in"], meta["c_out"], meta["n_out"], pairs, alone[i], st,
            flops / alone[i] / 1e6, flops / st / 1e6), flush=True)
        g = groups.setdefault(name, [0, 0.0, 0.0, 0.0])
        g[0] += 1
        g[1] += alone[i]
        g[2] += st
        g[3] += flops
    print()
    print("%-48s %5s %10s %10s %8s %8s %7s" % ("per instantiation", "n", "alone us", "step us",
                                                 "TF alone", "TF step", "step/al"))
    tot = [0.0, 0.0, 0.0]
    for name, (n, a, st, fl) in sorted(groups.items(), key=lambda kv: -kv[1][2]):
        print("%-48s %5d %10.1f %10.1f %8.1f %8.1f %7.2f" % (name, n, a, st, fl / a / 1e6,
                                                            fl / st / 1e6, st / a))
        tot[0] += a
        tot[1] += st
        tot[2] += fl
    print("%-48s %5d %10.1f %10.1f %8.1f %8.1f %7.2f" % (
        "all conv launches of a step", len(cap), tot[0], tot[1], tot[2] / tot[0] / 1e6,
        tot[2] / tot[1] / 1e6, tot[1] / tot[0]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--planes", type=int, default=3)
    ap.add_argument("--wgrad", action="store_true")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--lc", action="store_true", help="the fusion stack's channel widths")
    ap.add_argument("--lc-b2", action="store_true",
                    help="every conv launch of the LC step alone vs inside the pipelined step")
    ap.add_argument("--tail", action="store_true",
                    help="with --lc-b2: the step with the dense BEV tail (row f1) behind it")
    args = ap.parse_args()
    if args.lc_b2:
        return lc_b2(tail=args.tail)
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
