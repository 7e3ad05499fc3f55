#!/usr/bin/env python
"""bench.py -- the MSMDFusion sparse-voxel hot path on MI355X.

Workload (BASELINE.json configs[1]): TransFusion-L voxel backbone
(configs/transfusion_nusc_voxel_L.py: voxelize 0.075 m -> HardSimpleVFE ->
SparseEncoder 5->16->32->64->128 -> BEV [B,256,180,180]), forward + backward +
AdamW step, samples_per_gpu = 4, synthetic nuScenes-shaped clouds (seeded,
resident in HBM before the timed region), fp32 (the reference's precision).

One "step" = voxelize 4 clouds + forward + backward + optimizer step on every
rank; value = samples/s over all ranks.  Prints ONE JSON line (rank 0).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--no-cpu-baseline]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_MFMA_TFLOPS = 2516.6   # 256 CUs x 4096 flop/clk x 2.4 GHz, dense
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak

from msmdfusion_amd.configs import MSMDFUSION_LC, TRANSFUSION_L  # noqa: E402

ENCODER_CFG = TRANSFUSION_L["model"]["pts_middle_encoder"]   # transfusion_nusc_voxel_L.py:161-169
SAMPLES_PER_GPU = TRANSFUSION_L["samples_per_gpu"]           # :116


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--diag", action="store_true", help="print host enqueue vs step time")
    ap.add_argument("--workload", choices=["transfusion_l", "lc"], default="transfusion_l",
                    help="transfusion_l = BASELINE configs[1] (the headline); lc = configs[2], "
                         "the full MSMDFusion-LC sparse path (virtual points, GMA-Conv, sparse_add)")
    ap.add_argument("--no-profile", action="store_true",
                    help="skip per-launch event timing of the conv kernels")
    return ap.parse_args()


class Backbone(torch.nn.Module):
    """pts_voxel_layer + pts_voxel_encoder + pts_middle_encoder of
    TransFusionDetector.extract_pts_feat (mmdet3d/models/detectors/transfusion.py:61-74)."""

    def __init__(self):
        super().__init__()
        from msmdfusion_amd import synthetic as S
        from msmdfusion_amd.registry import build_middle_encoder
        from msmdfusion_amd.voxelize import Voxelization
        import msmdfusion_amd.sparse_encoder  # noqa: F401  (registers SparseEncoder)
        self.voxel_layer = Voxelization(S.VOXEL_SIZE, S.POINT_CLOUD_RANGE, S.MAX_NUM_POINTS,
                                        S.MAX_VOXELS)
        self.middle_encoder = build_middle_encoder(ENCODER_CFG)

    @torch.no_grad()
    def voxelize(self, points):
        """transfusion.py:76-101 with the VFE fused into the gather."""
        feats, coors = [], []
        for b, (mean, c, _) in enumerate(self.voxel_layer.forward_batch(points, fused_mean=True)):
            feats.append(mean)
            coors.append(F.pad(c, (1, 0), mode="constant", value=b))
        return torch.cat(feats, 0), torch.cat(coors, 0)

    def prepare(self, points):
        """The index-only part of a step (needs no weights, no previous step):
        voxelization + every rulebook / tiling order / pair list of the encoder."""
        feats, coors = self.voxelize(points)
        planned, _ = self.middle_encoder.plan(coors, len(points))
        return feats, coors, planned

    def forward(self, points, prepared=None):
        feats, coors, planned = prepared if prepared is not None else self.prepare(points)
        bev, _ = self.middle_encoder(feats, coors, len(points), planned=planned)
        return bev


class FusionBackbone(torch.nn.Module):
    """MSMDFusionDetector.extract_pts_feat's sparse section
    (mmdet3d/models/detectors/MSMDFusion.py:421-443) with the LC config
    (configs/MSMDFusion_nusc_voxel_LC.py:141-190): LiDAR encoder frozen
    (freeze_lidar_components, tools/train.py:185-219), fusion stack trained."""

    def __init__(self):
        super().__init__()
        from msmdfusion_amd import synthetic as S
        from msmdfusion_amd.distributed import freeze_unused_fusion_blocks
        from msmdfusion_amd.fusion import SparseFusionPath
        from msmdfusion_amd.registry import build_middle_encoder
        from msmdfusion_amd.voxelize import Voxelization
        vox = Voxelization(S.VOXEL_SIZE, S.POINT_CLOUD_RANGE, S.MAX_NUM_POINTS, S.MAX_VOXELS)
        enc = build_middle_encoder(ENCODER_CFG)
        mm = build_middle_encoder(MSMDFUSION_LC["model"]["multimodal_middle_encoder"])
        for p in enc.parameters():
            p.requires_grad = False
        for m in enc.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.track_running_stats = False
        freeze_unused_fusion_blocks(mm)     # built, never called: no find_unused_parameters
        self.path = SparseFusionPath(vox, enc, mm)

    def prepare(self, points, virtual):
        """Index-only part of a step, for the prefetcher (its own stream, a step ahead)."""
        # the FPS / nearest-voxel chain (9 + 2 ms, two workgroups wide) goes to the path's own
        # side stream: on the prefetcher's stream the NEXT batch's host reads would queue
        # behind it and the prepare chain (25 ms) would set the step time
        return self.path.prepare(points, [virtual] * 4, nn_side_stream=True)

    def forward(self, points, virtual, prepared=None):
        x, x_mm = self.path(points, [virtual] * 4, prepared=prepared)
        return torch.cat([x, x_mm], 1)


def _cpu_baseline_one(seed):
    """Reference algorithm on the host (oracle port, OpenMP): one cloud through
    voxelization, every rulebook and all 21 sparse convs forward + backward
    (dgrad + wgrad); BN/ReLU (elementwise, <1 % of the work) are skipped."""
    from msmdfusion_amd import synthetic as S
    from oracle import oracle as O
    # memory-bound gather/scatter loops stop scaling well before the socket is
    # full (256 hardware threads were slower than 8 here): cap at 32
    cores = O.set_threads(min(os.cpu_count() or 1, 32))
    pts = S.lidar_sweep(seed)
    t0 = time.perf_counter()
    v, c, n = O.hard_voxelize(pts, S.VOXEL_SIZE, S.POINT_CLOUD_RANGE, 10, 120000)
    feat = O.voxel_mean(v, n)
    idx = np.concatenate([np.zeros((c.shape[0], 1), np.int32), c], 1)
    rng = np.random.RandomState(0)
    shape = list(S.SPARSE_SHAPE)
    layers = [("subm", 5, 16)]
    for i, blocks in enumerate(ENCODER_CFG["encoder_channels"]):
        cin = layers[-1][2]
        for j, cout in enumerate(blocks):
            last = j == len(blocks) - 1 and i != 3
            if last:
                layers.append(("down%d" % i, cin, cout))
            else:
                layers += [("subm", cout, cout), ("subm", cout, cout)]
            cin = cout
    layers.append(("out", 128, 128))
    pads = {0: 1, 1: 1, 2: [0, 1, 1]}
    macs = 0
    cache = {}
    for kind, cin, cout in layers:
        if kind == "subm":
            key = (idx.shape[0], tuple(shape))
            if key not in cache:
                cache[key] = O.get_indice_pairs(idx, 1, shape, 3, 1, 1, 1, True)
            oi, pr, nm, osz = cache[key]
            w = rng.randn(27, cin, cout).astype(np.float32) * 0.05
            out = O.indice_conv_fwd(feat, w, pr, nm, oi.shape[0], subm=True)
            O.indice_conv_bwd(feat, w, out, pr, nm, subm=True)
        else:
            ks, st, pd = (3, 2, pads[int(kind[4])]) if kind != "out" else ([3, 1, 1], [2, 1, 1], 0)
            oi, pr, nm, osz = O.get_indice_pairs(idx, 1, shape, ks, st, pd, 1, False)
            w = rng.randn(pr.shape[0], cin, cout).astype(np.float32) * 0.05
            out = O.indice_conv_fwd(feat, w, pr, nm, oi.shape[0])
            O.indice_conv_bwd(feat, w, out, pr, nm)
            idx, shape = oi, osz
        macs += int(nm.sum()) * cin * cout
        feat = np.maximum(out, 0)
    dt = time.perf_counter() - t0
    return dict(value=round(1.0 / dt, 4), unit="samples/s", cores=cores, kind="port",
                sample="1 synthetic cloud (seed %d, %d pts, %d voxels): voxelize + all rulebooks + "
                       "21 sparse convs fwd+dgrad+wgrad with oracle/msmd_oracle.c (OpenMP, %d "
                       "threads), %.1f GMAC fwd, %.1f s" % (seed, pts.shape[0], c.shape[0], cores,
                                                           macs / 1e9, dt))


def cpu_baseline(seed, budget_s=12.0, max_clouds=24):
    """The host baseline on a bounded sample: whole synthetic clouds, one after the
    other, until ~budget_s of CPU work is done (the first one also warms the
    OpenMP pool and the page cache and is not counted when more follow)."""
    runs = []
    t_all = time.perf_counter()
    while len(runs) < max_clouds and (time.perf_counter() - t_all < budget_s or len(runs) < 2):
        runs.append(_cpu_baseline_one(seed + len(runs)))
    timed = runs[1:] if len(runs) > 1 else runs
    secs = [1.0 / r["value"] for r in timed]
    out = dict(timed[-1])
    out["value"] = round(len(secs) / sum(secs), 4)
    out["sample"] = ("%d synthetic clouds (seeds %d..%d, ~28.7k pts / ~18.9k voxels each, first one "
                     "untimed warm-up), each: voxelize + all rulebooks + 21 sparse convs "
                     "fwd+dgrad+wgrad with oracle/msmd_oracle.c (OpenMP, %d threads), 29.2 GMAC fwd; "
                     "%.1f s of CPU work, %.2f s per cloud"
                     % (len(secs), seed + 1, seed + len(runs) - 1, out["cores"], sum(secs),
                        sum(secs) / len(secs)))
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)   # RCCL on ROCm
    assert world == args.gpus or world == 1, "launch with torch.distributed.run for --gpus > 1"

    from msmdfusion_amd import kernels as K
    from msmdfusion_amd import synthetic as S

    torch.manual_seed(0)
    lc = args.workload == "lc"
    spg = 2 if lc else SAMPLES_PER_GPU      # configs/MSMDFusion_nusc_voxel_LC.py:104
    model = (FusionBackbone() if lc else Backbone()).to(dev).train()
    params = [p for p in model.parameters() if p.requires_grad]
    net = model
    if world > 1:
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank],
                                                        gradient_as_bucket_view=True)
    # AdamW lr=1e-4, wd=0.01: configs/transfusion_nusc_voxel_L.py optimizer
    # (fused=True: torch's single multi-tensor kernel per step instead of ~10 foreach launches)
    opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.01,
                            fused=os.environ.get("MSMD_FUSED_ADAMW", "1") == "1")

    clouds = [torch.from_numpy(S.lidar_sweep(rank * spg + i)).to(dev) for i in range(spg)]
    virtual = [torch.from_numpy(S.virtual_points(rank * spg + i)).to(dev) for i in range(spg)] \
        if lc else None
    target = torch.randn(spg, 640 if lc else 256, 180, 180, device=dev)

    prefetch = None
    if os.environ.get("MSMD_PREFETCH", "1") == "1":
        from msmdfusion_amd.prefetch import IndexPrefetcher
        # worker thread: pays on the LC path (its prepare() waits ~10 ms on host reads,
        # 70 -> 77 samples/s); configs[1] is GPU-bound either way (366 vs 368)
        threaded = os.environ.get("MSMD_PREFETCH_THREAD", "1" if lc else "0") == "1"
        if threaded:    # two threads share the GIL: hand it over promptly (default 5 ms)
            sys.setswitchinterval(float(os.environ.get("MSMD_SWITCH_INTERVAL", "0.0005")))
        prefetch = IndexPrefetcher(model.prepare, dev, threaded=threaded)
        batch = (clouds, virtual) if lc else (clouds,)
        pending = [prefetch.submit(*batch)]

    def step():
        if prefetch is not None:
            pending.append(prefetch.submit(*batch))      # next step's batch
            ticket = pending.pop(0)
            bev = net(*batch, prepared=prefetch.take(ticket))
        else:
            bev = net(clouds, virtual) if lc else net(clouds)
        loss = (bev * target).mean()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 10.0)     # grad_clip max_norm=10 (config)
        opt.step()
        opt.zero_grad(set_to_none=True)
        if prefetch is not None:
            prefetch.retire(ticket)
        return loss

    # Setup, untimed: let torch's caching allocator reach its steady state before
    # the W warm-up steps.  The LC path allocates on two streams (record_stream
    # defers block reuse), and needs ~8 steps before no step calls hipMalloc any
    # more (37 ms -> 28 ms per step, tools/lc_steps.py).
    # (LC with the prefetcher allocates on four streams: 16.)
    for _ in range((16 if lc else 10) if (lc or prefetch is not None) else 2):
        step()
    for _ in range(args.warmup):
        step()
    if args.diag and rank == 0:     # host enqueue time vs device time, outside the timed region
        for _ in range(3):
            torch.cuda.synchronize()
            a = time.perf_counter()
            step()
            b = time.perf_counter()
            torch.cuda.synchronize()
            c = time.perf_counter()
            print("diag: enqueue %.2f ms, step %.2f ms" % ((b - a) * 1e3, (c - a) * 1e3),
                  file=sys.stderr)
    # Per-launch HIP events (recorded on the launch stream) bracket every conv
    # launch of a few timed steps only: on ROCm a timing event is a barrier
    # packet that drains the queue, so bracketing all ~85 launches of every
    # step would slow the measured throughput by ~25 %.
    prof = None if args.no_profile else []
    sampled = set() if prof is None else {0, args.steps // 2}
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        K.PROFILE = prof if i in sampled else None
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    K.PROFILE = None
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert torch.isfinite(loss).item()
    # untimed sanity step: every parameter and every gradient of the trained modules is
    # finite (a skipped tile or a stale buffer shows up as NaN/garbage here, not in the rate)
    loss = None
    bev = net(clouds, virtual) if lc else net(clouds)
    (bev * target).mean().backward()
    bad = [n for n, p in model.named_parameters()
           if p.requires_grad and (p.grad is None and "blocks_2D" not in n and "blocks_mix" not in n
                                   or p.grad is not None and not torch.isfinite(p.grad).all())]
    bad += [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
    assert not bad, "non-finite parameters / gradients after the timed steps: %s" % bad[:5]
    opt.zero_grad(set_to_none=True)

    if rank == 0:
        n_samples = args.steps * spg * world
        out = {
            "metric": ("samples/sec MSMDFusion-LC sparse fusion path fwd+bwd (nuScenes 0.075m voxel)"
                       if lc else
                       "samples/sec TransFusion-L voxel backbone fwd+bwd (nuScenes 0.075m voxel)"),
            "value": round(n_samples / elapsed, 3), "unit": "samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("configs[2]: MSMDFusion-LC sparse path (LiDAR SparseEncoder "
                                    "frozen + 4-scale virtual-point voxels + modality split + "
                                    "GMA-Conv + sparse_add + downscale -> BEV 640ch), fwd+bwd+"
                                    "AdamW, 2 x (28.7k LiDAR + 50k virtual pts)/GPU, fp32"
                                    if lc else
                                    "configs[1]: TransFusion-L voxel backbone (voxelize+VFE+"
                                    "SparseEncoder->BEV), fwd+bwd+AdamW, 4 synthetic ~28.7k-pt "
                                    "clouds/GPU, 0.075 m voxels, fp32"),
                       "global_batch": spg * world, "parallelism": "dp%d" % world,
                       "index_prefetch": prefetch is not None},
        }
        out["roofline"] = roofline(prof) if prof else None
        if world == 1 and not args.no_cpu_baseline and not lc:
            out["cpu_baseline"] = cpu_baseline(0)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def pmc_traffic(kernel_name):
    """HBM bytes per launch of the dominant kernel from the committed PMC summary."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
    if not os.path.exists(path):
        return None, None
    kernels = json.load(open(path))["kernels"]
    if "NT=" in kernel_name:    # "spconv_fwd_split_kernel<NT=8>" -> "spconv_fwd_split_kernel<8, ..."
        nt = kernel_name.split("NT=")[1].rstrip(">")
        prefix = kernel_name.split("<")[0] + "<" + nt + ","
    else:                       # wgrad: the widest instantiation present
        prefix = kernel_name + "<"
    cands = sorted(k for k in kernels if k.startswith(prefix))
    key = cands[-1] if cands else None
    e = kernels.get(key, {})
    return e.get("hbm_bytes_per_launch"), e.get("mfma_pipe_busy_frac")


def roofline(prof):
    """Dominant kernel = the conv kernel class with the most accumulated time.
    achieved = algorithmic flops (2 * pairs * Cin * Cout, the reference's MAC
    count mmdet3d/apis/flops_counter.py:9-12) / measured launch duration."""
    pair_cache = {}
    groups = {}
    for kind, s, e, meta in prof:
        ms = s.elapsed_time(e)
        if kind in ("spconv_fwd", "spconv_fwd_split"):
            nbr = meta["nbr"]
            key = (nbr.data_ptr(), nbr.shape[1])
            if key not in pair_cache:
                pair_cache[key] = int((nbr >= 0).sum().item())
            pairs = pair_cache[key]
            nt = (meta["c_out"] + 15) // 16
            if kind == "spconv_fwd_split":
                name = "spconv_fwd_split_kernel<NT=%d>" % nt
            else:
                name = ("spconv_fwd_pipe_kernel<NT=%d>" if nt >= 4 and meta["c_in"] % 16 == 0
                        else "spconv_fwd_kernel<NT=%d>") % nt
        else:
            pairs = int(meta["num"].sum().item())
            name = "spconv_wgrad_split_kernel" if kind == "spconv_wgrad_split" \
                else "spconv_wgrad_kernel"
        g = groups.setdefault(name, dict(ms=0.0, flops=0.0, launches=0))
        g["ms"] += ms
        g["flops"] += 2.0 * pairs * meta["c_in"] * meta["c_out"]
        g["launches"] += 1
    total_ms = sum(g["ms"] for g in groups.values())
    name, g = max(groups.items(), key=lambda kv: kv[1]["ms"])
    achieved = g["flops"] / (g["ms"] * 1e-3) / 1e12
    traffic, mfma_busy = pmc_traffic(name)
    peak, peak_note = PEAK_F32_MFMA_TFLOPS, "dense fp32 MFMA (v_mfma_f32_16x16x4_f32)"
    if name.startswith("spconv_fwd_split") or name.startswith("spconv_wgrad_split"):
        from msmdfusion_amd.spconv.functional import conv_planes
        products = {3: 6, 2: 3, 1: 1}[conv_planes()]
        peak = round(PEAK_BF16_MFMA_TFLOPS / products, 1)
        peak_note = ("dense bf16 MFMA peak %.0f TF / %d bf16 products per fp32-equivalent product "
                     "(operands split into %d bf16 planes, fp32 accumulate)"
                     % (PEAK_BF16_MFMA_TFLOPS, products, conv_planes()))
    return {"bound": "mfma", "kernel": name, "achieved": round(achieved, 3),
            "peak": peak, "peak_note": peak_note, "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4),
            "frac_of_fp32_mfma_peak": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
            "traffic": traffic,
            "traffic_note": "HBM bytes per launch from the committed rocprofv3 --pmc passes "
                            "(profiles/r01_pmc_summary.json: (2*FETCH_SIZE + WRITE_SIZE) KiB, "
                            "gfx950 correction), not re-measured in this run",
            "mfma_pipe_busy_frac_pmc": mfma_busy,
            "avg_launch_us": round(g["ms"] / g["launches"] * 1e3, 2), "launches": g["launches"],
            "share_of_conv_time": round(g["ms"] / total_ms, 3),
            "all_conv_kernels": {k: {"ms": round(v["ms"], 3),
                                     "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 3),
                                     "launches": v["launches"]} for k, v in groups.items()}}


if __name__ == "__main__":
    main()
