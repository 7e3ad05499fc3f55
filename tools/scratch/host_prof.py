#!/usr/bin/env python
"""Where the HOST time of a step goes (cProfile over N steps, no prefetch thread).

    python tools/host_prof.py [--lc] [--prefetch]
"""
import argparse
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lc", action="store_true")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--top", type=int, default=45)
    args = ap.parse_args()
    import torch
    import bench
    from msmdfusion_amd import synthetic as S
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    lc = args.lc
    spg = 2 if lc else 4
    model = (bench.FusionBackbone() if lc else bench.Backbone()).to(dev).train()
    params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.01)
    clouds = [torch.from_numpy(S.lidar_sweep(i)).to(dev) for i in range(spg)]
    virtual = [torch.from_numpy(S.virtual_points(i)).to(dev) for i in range(spg)] if lc else None
    target = torch.randn(spg, 640 if lc else 256, 180, 180, device=dev)
    batch = (clouds, virtual) if lc else (clouds,)

    def prep():
        return model.prepare(*batch)

    def feat(p):
        bev = model(*batch, prepared=p)
        loss = (bev * target).mean()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 10.0)
        opt.step()
        opt.zero_grad(set_to_none=True)

    for _ in range(8):
        feat(prep())
    torch.cuda.synchronize()
    for name, fn in (("prepare", None), ("feature pass", None)):
        pr = cProfile.Profile()
        ps = [prep() for _ in range(args.steps)] if name != "prepare" else None
        torch.cuda.synchronize()
        pr.enable()
        if name == "prepare":
            for _ in range(args.steps):
                prep()
        else:
            for p in ps:
                feat(p)
        pr.disable()
        torch.cuda.synchronize()
        st = pstats.Stats(pr)
        print("=" * 30, name, "(%d steps; divide by that)" % args.steps)
        st.sort_stats("tottime").print_stats(args.top)


if __name__ == "__main__":
    main()
