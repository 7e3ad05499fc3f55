#!/usr/bin/env python
"""Times the dense BEV tail (SPP + SECOND + SECONDFPN, LC widths) forward+backward
on a [B,640,180,180] map in the four layout/dtype combinations, and the two ways
of producing its input (dense()+view+cat vs the joint channels-last scatter).

    python tools/bev_tail_bench.py [--batch 2] [--iters 10]
"""
import argparse
import sys
import os
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msmdfusion_amd import configs as C  # noqa: E402
from msmdfusion_amd import spconv, synthetic as S  # noqa: E402
from msmdfusion_amd.spconv import functional as Fsp  # noqa: E402


def timed(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B = a.batch
    x = torch.randn(B, 640, 180, 180, device=dev)
    for cl in (False, True):
        for dt in (None, torch.bfloat16):
            tail = C.build_bev_tail(C.MSMDFUSION_LC, compute_dtype=dt, rows=False)
            tail.channels_last = cl
            if not cl:
                tail = tail.to(memory_format=torch.contiguous_format)
            tail = tail.to(dev).train()
            xin = x.contiguous(memory_format=torch.channels_last) if cl else x

            def step():
                xi = xin.detach().requires_grad_(True)
                out = tail(xi)[0]
                out.float().mean().backward()
            t0 = time.perf_counter()
            step()
            torch.cuda.synchronize()
            first = time.perf_counter() - t0
            ms = timed(step, a.iters)

            def fwd():
                with torch.no_grad():
                    tail(xin)
            print("bev tail  layout=%s dtype=%s  fwd+bwd %.2f ms  fwd %.2f ms  (first call %.1f s)"
                  % ("nhwc" if cl else "nchw", "bf16" if dt else "fp32", ms, timed(fwd, a.iters),
                     first), flush=True)
            del tail
    # the whole tail on pixel rows (sparse-conv kernels, fp32-equivalent)
    tail = C.build_bev_tail(C.MSMDFUSION_LC).to(dev).train()
    xcl0 = x.contiguous(memory_format=torch.channels_last)

    def step_rows():
        xi = xcl0.detach().requires_grad_(True)
        tail(xi)[0].mean().backward()

    def fwd_rows():
        with torch.no_grad():
            tail(xcl0)
    print("bev tail  on rows (sparse-conv kernels, 3 bf16 planes)  fwd+bwd %.2f ms  fwd %.2f ms"
          % (timed(step_rows, a.iters), timed(fwd_rows, a.iters)), flush=True)
    del tail
    # SPP alone: MIOpen fp32 / bf16 against the sparse-conv row kernels (fp32-equivalent)
    from msmdfusion_amd.bev import SPPModule
    from msmdfusion_amd.grid_conv import SPPModuleRows
    xcl = x.contiguous(memory_format=torch.channels_last)
    for name, mod, xin, dt in [("MIOpen fp32 nchw", SPPModule(), x, None),
                               ("MIOpen bf16 nchw", SPPModule(), x, torch.bfloat16),
                               ("row kernels (3 bf16 planes)", SPPModuleRows(), xcl, None)]:
        mod = mod.to(dev).train()

        def step():
            xi = xin.detach().requires_grad_(True)
            if dt is None:
                out = mod(xi)
            else:
                with torch.autocast("cuda", dtype=dt):
                    out = mod(xi)
            out.float().mean().backward()

        def fwd():
            with torch.no_grad():
                if dt is None:
                    mod(xin)
                else:
                    with torch.autocast("cuda", dtype=dt):
                        mod(xin)
        print("SPP alone  %-28s fwd+bwd %.2f ms  fwd %.2f ms" % (name, timed(step, a.iters),
                                                                  timed(fwd, a.iters)), flush=True)
        del mod
    # hand-over
    shape = [2, 180, 180]
    sp = []
    for c, n in [(128, 21000 * B // 2), (192, 38000 * B // 2)]:
        idx = torch.from_numpy(S.random_voxel_indices(n, B, shape, seed=c)).to(dev)
        sp.append(spconv.SparseConvTensor(torch.randn(idx.shape[0], c, device=dev), idx, shape, B))

    def old():
        return torch.cat([t.dense().view(B, -1, 180, 180) for t in sp], 1).contiguous(
            memory_format=torch.channels_last)

    def old_nchw():
        return torch.cat([t.dense().view(B, -1, 180, 180) for t in sp], 1)

    def new():
        return Fsp.bev_concat(sp)
    print("hand-over: dense+view+cat (nchw) %.3f ms | + to channels_last %.3f ms | joint nhwc "
          "scatter %.3f ms" % (timed(old_nchw, 20), timed(old, 20), timed(new, 20)))


if __name__ == "__main__":
    main()
