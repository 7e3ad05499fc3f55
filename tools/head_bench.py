#!/usr/bin/env python
"""TransFusionHead at the LC size on a fixed [B, 512, 180, 180] feature map: forward, loss,
backward, per section (HIP events), for the torch / MIOpen convolutions and for the row
kernels.    python tools/head_bench.py [batch]      MSMD_HEAD_HEATMAP_ROWS=0: heat-map convs
on MIOpen while shared_conv stays on rows."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msmdfusion_amd import configs as C  # noqa: E402
from tools.head_loss_bench import make_case  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    if os.environ.get("MSMD_HEAD_BENCH_NOMIOPEN") == "1":
        torch.backends.cudnn.enabled = False
    if os.environ.get("MSMD_HEAD_BENCH_NOGC") == "1":
        import gc
        gc.disable()
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    rs = np.random.RandomState(0)
    x = torch.randn(B, 512, 180, 180, device=dev).contiguous(memory_format=torch.channels_last)
    which = os.environ.get("MSMD_HEAD_BENCH_ROWS")      # "0" / "1": one variant only
    for rows in ((False, True) if which is None else (which == "1",)):
        torch.manual_seed(0)
        head = C.build_head(C.MSMDFUSION_LC, rows=rows).to(dev).train()
        _, gts, labs = make_case(rs, B, head.num_proposals, dev, boxes_per_sample=40)
        gts = [g.to(dev) for g in gts]
        labs = [l.to(dev) for l in labs]
        mode = os.environ.get("MSMD_HEAD_BENCH_MODE")       # "fwd" / "fwdbwd": wall time of parts
        if mode:
            import time

            def run():
                head.zero_grad(set_to_none=True)
                (p,) = head(x)[0]
                if mode == "fwdbwd":
                    sum(v.float().mean() for k, v in p.items() if v.requires_grad).backward()
            for _ in range(24):        # the caching allocator needs ~16 iterations to settle
                run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                run()
            torch.cuda.synchronize()
            print("mode %s rows=%s: %.2f ms per iteration" % (mode, rows,
                                                              (time.perf_counter() - t0) * 100))
            continue
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        acc = np.zeros(3)
        n, settle = 12, 24               # (hipMalloc in every iteration before that)
        for it in range(n + settle):
            head.zero_grad(set_to_none=True)
            marks[0].record()
            preds = head(x if rows else x.contiguous())
            marks[1].record()
            losses = head.loss(gts, labs, preds)
            total = sum(v for k, v in losses.items() if "loss" in k)
            marks[2].record()
            total.backward()
            marks[3].record()
            torch.cuda.synchronize()
            if it >= settle:
                acc += [marks[i].elapsed_time(marks[i + 1]) for i in range(3)]
        if os.environ.get("MSMD_HEAD_BENCH_CPROFILE") == "1":      # where the host time goes
            import cProfile
            import pstats
            profs = [cProfile.Profile() for _ in range(3)]
            for _ in range(5):
                head.zero_grad(set_to_none=True)
                torch.cuda.synchronize()
                profs[0].enable()
                preds = head(x if rows else x.contiguous())
                profs[0].disable()
                torch.cuda.synchronize()
                profs[1].enable()
                losses = head.loss(gts, labs, preds)
                total = sum(v for k, v in losses.items() if "loss" in k)
                profs[1].disable()
                torch.cuda.synchronize()
                profs[2].enable()
                total.backward()
                profs[2].disable()
            for name, pr in zip(("forward", "loss", "backward"), profs):
                print("----", name, "(5 iterations)")
                pstats.Stats(pr).sort_stats("tottime").print_stats(14)
        if os.environ.get("MSMD_HEAD_BENCH_TORCHPROF") == "1":     # host time per operator
            from torch.profiler import ProfilerActivity, profile
            with profile(activities=[ProfilerActivity.CPU]) as tp:
                for _ in range(3):
                    head.zero_grad(set_to_none=True)
                    losses = head.loss(gts, labs, head(x if rows else x.contiguous()))
                    sum(v for k, v in losses.items() if "loss" in k).backward()
                torch.cuda.synchronize()
            print(tp.key_averages().table(sort_by="self_cpu_time_total", row_limit=25,
                                          max_name_column_width=60))
        if os.environ.get("MSMD_HEAD_BENCH_PHASES", "0") == "1":   # host enqueue vs device drain
            import time
            rec = np.zeros(6)
            for it in range(8):
                head.zero_grad(set_to_none=True)
                torch.cuda.synchronize()
                t = [time.perf_counter()]
                preds = head(x if rows else x.contiguous())
                t.append(time.perf_counter())
                torch.cuda.synchronize()
                t.append(time.perf_counter())
                losses = head.loss(gts, labs, preds)
                total = sum(v for k, v in losses.items() if "loss" in k)
                t.append(time.perf_counter())
                torch.cuda.synchronize()
                t.append(time.perf_counter())
                total.backward()
                t.append(time.perf_counter())
                torch.cuda.synchronize()
                t.append(time.perf_counter())
                if it >= 2:
                    rec += np.diff(t) * 1e3
            rec /= 6
            print("   phases (host enqueue + drain): forward %.2f + %.2f, loss %.2f + %.2f, "
                  "backward %.2f + %.2f ms" % tuple(rec), flush=True)
        f, l, b = acc / n
        print("B=%d rows=%s heatmap_rows=%s: forward %.2f ms, loss %.2f ms, backward %.2f ms, "
              "sum %.2f ms" % (B, rows, os.environ.get("MSMD_HEAD_HEATMAP_ROWS", "1"), f, l, b,
                               f + l + b), flush=True)


if __name__ == "__main__":
    main()
