#!/usr/bin/env python
"""bench.py -- the MSMDFusion sparse-voxel fusion hot path on MI355X.

Headline workload = BASELINE.json configs[2], the configuration the metric is
quoted on: MSMDFusion-LC (configs/MSMDFusion_nusc_voxel_LC.py) sparse section of
MSMDFusionDetector.extract_pts_feat (MSMDFusion.py:421-443): LiDAR voxelization
+ SparseEncoder (frozen, tools/train.py:185-219), virtual-point voxels at four
scales, voxel_modality_split, GMA-Conv with FPS / ball-query neighbour search,
sparse_add, downscale convs -> BEV [B,640,180,180]; forward + backward + AdamW,
samples_per_gpu = 2 (config :104), synthetic nuScenes-shaped inputs (seeded,
resident in HBM before the timed region), fp32-equivalent arithmetic.

One "step" = the whole path over one batch on every rank; value = samples/s over
all ranks.  Rank 0 prints ONE JSON line.  At N = 1 the line also carries
  roofline      the dominant conv kernel, timed live with HIP events
  cpu_baseline  the oracle port of the same path on the host cores (bounded sample)
  also          the same measurement for configs[1] (TransFusion-L voxel backbone,
                4 clouds/GPU, everything trained) -- a secondary line, not the headline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload lc|transfusion_l]
                    [--no-cpu-baseline] [--no-also] [--no-profile]

`--gpus N` started as a plain process re-launches itself as N ranks under
torch.distributed.run (one process per GPU, RCCL over xGMI); started by
torch.distributed.run it reads RANK/LOCAL_RANK/WORLD_SIZE from the environment.
"""
import argparse
import json
import os
import sys
import time

# One rank drives one GPU: its few host-side tensor ops (index lists, box targets) must not
# fan out over every core of the node.  torch's intra-op pool defaults to all 256 hardware
# threads; on a box whose cgroup grants 16 CPUs their spinning exhausts the quota and the
# kernel throttles the whole process for the rest of each 100 ms period (measured with
# /sys/fs/cgroup/cpu.stat: LC line 117 -> 128 samples/s, the head's step 33 -> 13 ms).
# torch.distributed.run sets OMP_NUM_THREADS=1 for multi-rank launches already.
os.environ.setdefault("OMP_NUM_THREADS", "8")
# The LC step keeps six streams busy (feature pass, index prefetch, four neighbour-search
# streams); the HIP runtime multiplexes a process's streams onto 4 hardware queues by default.
# One queue each: 129.4 -> 131.9-133.5 samples/s (read before the runtime initialises).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# ... and the rank's threads stay on a few CPUs (before anything creates threads): unpinned,
# the same binary on the same box reads 100-133 samples/s from run to run, pinned 128-130
from msmdfusion_amd.hostcpu import pin_host_threads  # noqa: E402
# (only as a program: a test session that imports this module for its model classes keeps
# its own CPUs -- pinned to four, the oracle's OpenMP loops made the GPU suite five times
# slower; tools that want the bench's placement set MSMD_PIN_ON_IMPORT=1 first)
PINNED_CPUS = pin_host_threads() if (__name__ == "__main__" or
                                     os.environ.get("MSMD_PIN_ON_IMPORT") == "1") else None

import torch  # noqa: E402

# more busy host threads than the cgroup pays CPUs for (8 ranks on a 16-CPU box): waits
# block instead of spinning (device flag: before the first HIP call of the process)
from msmdfusion_amd.hostcpu import set_blocking_sync_if_oversubscribed  # noqa: E402
BLOCKING_SYNC = set_blocking_sync_if_oversubscribed() if __name__ == "__main__" else False

PEAK_BF16_MFMA_TFLOPS = 2516.6   # 256 CUs x 4096 flop/clk x 2.4 GHz, dense
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak

from msmdfusion_amd.configs import MSMDFUSION_LC, TRANSFUSION_L  # noqa: E402

ENCODER_CFG = TRANSFUSION_L["model"]["pts_middle_encoder"]   # transfusion_nusc_voxel_L.py:161-169
SAMPLES_PER_GPU = TRANSFUSION_L["samples_per_gpu"]           # :116

WORKLOADS = {
    "lc": dict(
        metric="samples/sec MSMDFusion fwd+bwd nuScenes 0.075m voxel (MSMDFusion-LC sparse "
               "fusion path)",
        name="configs[2]: MSMDFusion-LC sparse path (LiDAR SparseEncoder frozen + 4-scale "
             "virtual-point voxels + modality split + GMA-Conv + sparse_add + downscale -> BEV "
             "640ch), fwd+bwd+AdamW, 2 x (28.7k LiDAR + 50k virtual pts)/GPU, 0.075 m voxels, fp32",
        spg=MSMDFUSION_LC["samples_per_gpu"], bev_channels=640, settle=16),
    "lc_b4": dict(
        metric="samples/sec MSMDFusion fwd+bwd nuScenes 0.075m voxel (MSMDFusion-LC sparse "
               "fusion path, 4 samples per GPU)",
        name="configs[2] at configs[3]'s per-GPU batch (batch_size=4/GPU): MSMDFusion-LC sparse "
             "path, fwd+bwd+AdamW, 4 x (28.7k LiDAR + 50k virtual pts)/GPU, 0.075 m voxels, fp32",
        spg=4, bev_channels=640, settle=16),
    "lc_quirks": dict(
        metric="samples/sec MSMDFusion fwd+bwd nuScenes 0.075m voxel (MSMDFusion-LC sparse "
               "fusion path, reference_quirks=True)",
        name="configs[2] with reference_quirks=True: the same step with the reference's float32 "
             "voxel keys in voxel_modality_split (MSMDFusion.py:251-325,27-45: false 'mixed' "
             "voxels where keys alias) and its non-cumulative batch offsets -- the mode that "
             "matches a checkpoint trained with the reference; 2 x (28.7k LiDAR + 50k virtual "
             "pts)/GPU, fwd+bwd+AdamW, fp32",
        spg=MSMDFUSION_LC["samples_per_gpu"], bev_channels=640, settle=16),
    "lc_tail": dict(
        metric="samples/sec MSMDFusion fwd+bwd nuScenes 0.075m voxel (MSMDFusion-LC sparse "
               "fusion path + dense BEV tail)",
        name="configs[2] + row f1: the MSMDFusion-LC sparse path followed by bev_fusion "
             "(SPPModule) + SECOND + SECONDFPN -> [B,512,180,180], the dense tail computed on "
             "channels-last pixel rows by the sparse-conv kernels (fp32-equivalent), "
             "fwd+bwd+AdamW, 2 x (28.7k LiDAR + 50k virtual pts)/GPU, fp32",
        spg=MSMDFUSION_LC["samples_per_gpu"], bev_channels=512, settle=16),
    "lc_full": dict(
        metric="samples/sec MSMDFusion fwd+bwd nuScenes 0.075m voxel (MSMDFusion-LC LiDAR + "
               "virtual-point path from points to losses)",
        name="configs[2] + rows f1 + f3: the MSMDFusion-LC sparse path, bev_fusion (SPPModule) "
             "+ SECOND + SECONDFPN, TransFusionHead (200 heat-map queries, 1 decoder layer) and "
             "its loss (Hungarian targets for 40 synthetic boxes per sample, focal / L1 / Gaussian "
             "focal), fwd+bwd+AdamW, 2 x (28.7k LiDAR + 50k virtual pts)/GPU, fp32 (image backbone "
             "out of scope: virtual points arrive with their 49 image channels)",
        spg=MSMDFUSION_LC["samples_per_gpu"], bev_channels=0, settle=16),
    "lc_img": dict(
        metric="samples/sec MSMDFusion fwd+bwd nuScenes 0.075m voxel (MSMDFusion-LC sparse "
               "fusion path fed from image feature maps)",
        name="configs[2] + rows a13 (image half) / f2: the virtual points are MADE inside the "
             "step -- synthetic FPN maps [B*6,256,112x200 / 56x100 / 28x50] + per-camera pixel "
             "lists (~50k foreground points per sample, host arrays as the loader hands them) -> "
             "pack_foreground (H2D) -> depth canvas -> DepthAwareChannelCompression (3 x Conv2d "
             "257->49 + BN + ReLU) -> msmd_fg_gather_f32 x 4 scales -> ScoreNet -> 4-scale "
             "voxelization (MSMDFusion.py:335-393,169-238; no gradient flows back through it: "
             "voxelize is @no_grad, :462) -> the MSMDFusion-LC sparse path, fwd+bwd+AdamW, fp32",
        spg=MSMDFUSION_LC["samples_per_gpu"], bev_channels=640, settle=16),
    "transfusion_l": dict(
        metric="samples/sec TransFusion-L voxel backbone fwd+bwd (nuScenes 0.075m voxel)",
        name="configs[1]: TransFusion-L voxel backbone (voxelize+VFE+SparseEncoder->BEV), "
             "fwd+bwd+AdamW, 4 synthetic ~28.7k-pt clouds/GPU, 0.075 m voxels, fp32",
        spg=SAMPLES_PER_GPU, bev_channels=256, settle=10),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true",
                    help="skip the secondary configs[1] measurement (N = 1 only)")
    ap.add_argument("--diag", action="store_true", help="print host enqueue vs step time")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="lc",
                    help="lc = BASELINE configs[2], the metric's configuration (the headline); "
                         "transfusion_l = configs[1], the LiDAR-only voxel backbone")
    ap.add_argument("--no-profile", action="store_true",
                    help="skip per-launch event timing of the conv kernels")
    return ap.parse_args()


def _sparse_only(model_cfg):
    """The config's `model` dict without the dense tail and head: the sparse hot path."""
    return {k: v for k, v in model_cfg.items()
            if k not in ("pts_backbone", "pts_neck", "pts_bbox_head")}


class Backbone(torch.nn.Module):
    """configs[1]: TransFusionDetector (msmdfusion_amd/detector.py) built from
    configs/transfusion_nusc_voxel_L.py's model dict up to the BEV map --
    pts_voxel_layer + pts_voxel_encoder + pts_middle_encoder of extract_pts_feat
    (mmdet3d/models/detectors/transfusion.py:61-74)."""

    def __init__(self):
        super().__init__()
        from msmdfusion_amd.detector import build_detector
        self.det = build_detector(_sparse_only(TRANSFUSION_L["model"]))
        self.voxel_layer, self.middle_encoder = self.det.pts_voxel_layer, self.det.pts_middle_encoder

    def prepare(self, points):
        return self.det.prepare(points)

    def forward(self, points, prepared=None):
        return self.det.extract_sparse_feat(points, prepared=prepared)


class FusionBackbone(torch.nn.Module):
    """configs[2]: MSMDFusionDetector built from configs/MSMDFusion_nusc_voxel_LC.py's model
    dict, its sparse section (MSMDFusion.py:421-443): LiDAR encoder frozen
    (freeze_lidar_components, tools/train.py:185-219), fusion stack trained."""

    def __init__(self, tail=False, head=False, reference_quirks=False):
        super().__init__()
        from msmdfusion_amd.detector import build_detector, freeze_lidar_components
        cfg = dict(MSMDFUSION_LC["model"]) if tail else _sparse_only(MSMDFUSION_LC["model"])
        if reference_quirks:
            cfg["reference_quirks"] = True
        if head:
            from msmdfusion_amd.configs import _PTS_BBOX_HEAD, _TEST_CFG_PTS, _TRAIN_CFG_PTS
            cfg.update(pts_bbox_head=dict(_PTS_BBOX_HEAD), train_cfg=dict(pts=dict(_TRAIN_CFG_PTS)),
                       test_cfg=dict(pts=dict(_TEST_CFG_PTS)))
        self.det = build_detector(cfg)
        if MSMDFUSION_LC["freeze_lidar_components"]:
            freeze_lidar_components(self.det)
        if not tail:     # the image-side glue and SPP block take no part in the sparse section
            for m in (self.det.conv1x1_blocks, self.det.score_net, self.det.bev_fusion):
                for p in m.parameters():
                    p.requires_grad = False
        else:            # reached only through @no_grad voxelization (the LC config's
            for m in (self.det.conv1x1_blocks, self.det.score_net):   # unused parameters)
                for p in m.parameters():
                    p.requires_grad = False
        self.path = self.det._path      # (tests and tools reach the sparse section here)

    def prepare(self, points, virtual):
        """Index-only part of a step, for the prefetcher (its own stream, a step ahead)."""
        # the FPS / nearest-voxel chain (9 + 2 ms, two workgroups wide) goes to the path's own
        # side stream: on the prefetcher's stream the NEXT batch's host reads would queue
        # behind it and the prepare chain (25 ms) would set the step time
        return self.det.prepare(points, virtual, nn_side_stream=True)

    def forward(self, points, virtual, prepared=None):
        # cat([x, x_mm], 1) -- bev_fusion's input (MSMDFusion.py:440) -- as ONE
        # channels-last map both sparse tensors scatter into (no dense()+view+cat)
        return self.det.extract_sparse_feat(points, virtual, prepared=prepared)


class FusionImageBackbone(FusionBackbone):
    """FusionBackbone fed from IMAGE FEATURES: MSMDFusionDetector.extract_multiscale_voxel_feat's
    image half (MSMDFusion.py:400-407 -> depth_aware_channel_compression :335-368,
    get_foreground2D :169-238) makes the per-scale virtual points inside the step.  It is
    input + frozen-weight work (the reference's voxelize is @no_grad: conv1x1_blocks and
    score_net receive no gradient, configs/...LC.py:309 find_unused_parameters), so it runs
    with the index pass, a step ahead on the prefetcher's stream."""

    def prepare(self, points, img_feats, metas):
        with torch.no_grad():
            virt = self.det.virtual_points_from_images(img_feats, metas)
        return dict(virt=virt, index=self.det.prepare(points, virt, nn_side_stream=True))

    def forward(self, points, img_feats, metas, prepared=None):
        p = prepared if prepared is not None else self.prepare(points, img_feats, metas)
        return self.det.extract_sparse_feat(points, p["virt"], prepared=p["index"])


def synthetic_image_batch(sample_ids, dev, cams=6, input_hw=(448, 800)):
    """Image-side inputs of a batch: FPN maps (strides 4 / 8 / 16 of the 800 x 448 input,
    configs/...LC.py:17,159-163) and img_metas whose foreground2D_info holds, per camera, the
    foreground pixel list (x, y, depth), the virtual points' own 15 attributes (the first 15
    columns of synthetic.virtual_points: the image channels come from the maps now) and the
    real-point pixels of the sparse depth map -- numpy arrays, as MyLoadForeground2D leaves
    them (my_loading_multi_proj.py:38-97)."""
    import numpy as np
    from msmdfusion_amd import synthetic as S
    H, W = input_hw
    B = len(sample_ids)
    g = torch.Generator(device="cpu").manual_seed(1234)
    feats = [torch.randn(B * cams, 256, H // s, W // s, generator=g).to(dev) for s in (4, 8, 16)]
    metas = []
    for i in sample_ids:
        rs = np.random.RandomState(500 + i)
        pts = S.virtual_points(i)[:, :15]
        parts = np.array_split(np.arange(pts.shape[0]), cams)
        pix, fpts, real, l2i = [], [], [], []
        for j in range(cams):
            n = parts[j].shape[0]
            p = np.stack([rs.uniform(0, W - 1, n), rs.uniform(0, H - 1, n),
                          rs.uniform(1, 60, n)], 1).astype(np.float32)
            pix.append(p)
            fpts.append(np.ascontiguousarray(pts[parts[j]]))
            m = n // 4                   # LiDAR returns that hit the camera: integer pixels
            real.append(np.stack([rs.randint(0, W, m), rs.randint(0, H, m),
                                  rs.uniform(1, 60, m)], 1).astype(np.float32))
            l2i.append(rs.randn(4, 4).astype(np.float32))
        metas.append(dict(foreground2D_info=dict(fg_pixels=pix, fg_points=fpts, fg_real_pixels=real),
                          lidar2img=l2i, input_shape=(H, W), pad_shape=(H, W, 3)))
    return feats, metas


def image_glue_kernel_times(model, batch, reps=10):
    """The image half alone, after the timed steps: ms per call of virtual_points_from_images
    (the whole glue) and us / GB/s of one msmd_fg_gather_f32 launch at each scale's map
    (algorithmic bytes: per point the C gathered floats read + the two rows written)."""
    from msmdfusion_amd import kernels as K
    from msmdfusion_amd.image_glue import pack_foreground
    _, feats, metas = batch
    det = model.det

    def timed(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / reps
    with torch.no_grad():
        whole = timed(lambda: det.virtual_points_from_images(feats, metas))
        pack = pack_foreground(metas, feats[0].device)
        comp = det._compress(feats, metas, pack=pack)
        out = {"virtual_points_from_images_ms": round(whole, 3), "fg_gather": []}
        n = pack.pixels.shape[0]
        for f in comp:
            c = f.shape[1]
            ds = f.shape[-1] / metas[0]["input_shape"][-1]
            ms = timed(lambda: K.fg_gather(f, pack.pixels, pack.plane, ds, pack.points,
                                           pack.lidar2img))
            nbytes = n * 4 * (c + 3 + 15 + (15 + c) + (c + 17) + 1)
            out["fg_gather"].append({"map": list(f.shape), "points": int(n), "us": round(ms * 1e3, 1),
                                     "GBps": round(nbytes / (ms * 1e-3) / 1e9, 1),
                                     "frac_hbm": round(nbytes / (ms * 1e-3) / 1e12 / HBM_PEAK_TBPS, 4)})
            # the launch the step actually makes (no gradient: gather + score_net + scaling
            # fused, msmd_fg_gather_scored_f32): per point C floats read + one row written
            lin = det.score_net[0]
            ms = timed(lambda: K.fg_gather_scored(f, pack.pixels, pack.plane, ds, pack.points,
                                                  pack.lidar2img, lin.weight, lin.bias))
            nbytes = n * 4 * (c + 3 + 15 + (15 + c) + 1)
            out.setdefault("fg_gather_scored", []).append(
                {"map": list(f.shape), "points": int(n), "us": round(ms * 1e3, 1),
                 "GBps": round(nbytes / (ms * 1e-3) / 1e9, 1),
                 "frac_hbm": round(nbytes / (ms * 1e-3) / 1e12 / HBM_PEAK_TBPS, 4)})
    return out


def modality_split_times(model, batch, reps=10):
    """voxel_modality_split of the four scales of this batch (one modality_split_many call,
    as prepare() makes it), alone on the chip: the reference's float32 keys + two-pointer
    merge + reference batch offsets (csrc/modality_float.hip) beside the exact-key bitmap
    split of the default mode; and how many voxels the float keys match falsely."""
    from msmdfusion_amd import kernels as K
    path = model.det._path
    with torch.no_grad():
        p = path.prepare(batch[0], [batch[1]] * 4, nn_side_stream=False)
    torch.cuda.synchronize()
    B = len(batch[0])
    jobs = []
    for i in range(4):
        idx3 = p["stages"][i][0]
        i2 = p["v2"][i].indices
        idx2 = torch.cat([i2[:, :1], i2[:, 2:]], 1).contiguous()
        shape = [max(a, b) for a, b in zip(p["stages"][i][1], p["v2"][i].spatial_shape)]
        jobs.append((idx3, idx2, shape))

    def timed(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / reps * 1e3
    out = {"call": "kernels.modality_split_many over the 4 scales (incl. its one host read)",
           "float_keys_us": round(timed(lambda: K.modality_split_many(
               jobs, B, float_keys=True, reference_offsets=True)), 1),
           "exact_keys_us": round(timed(lambda: K.modality_split_many(jobs, B)), 1)}
    fk = K.modality_split_many(jobs, B, float_keys=True, reference_offsets=True)
    ex = K.modality_split_many(jobs, B)
    out["mixed_voxels_float_keys"] = [int((a[0] > 0).sum().item()) for a in fk]
    out["mixed_voxels_exact_keys"] = [int((a[0] > 0).sum().item()) for a in ex]
    return out


class FusionTailBackbone(FusionBackbone):
    """FusionBackbone + the dense BEV tail (extract_pts_feat to its end,
    MSMDFusion.py:440-447): bev_fusion, pts_backbone, pts_neck, trained."""

    def __init__(self, head=False):
        super().__init__(tail=True, head=head)

    def forward(self, points, virtual, prepared=None):
        return self.det.extract_pts_feat(points, virtual_points=virtual, prepared=prepared)[0]


class FusionDetector(FusionTailBackbone):
    """MSMDFusionDetector.forward_train's point branch (MSMDFusion.py:494-560 ->
    forward_pts_train :562-590): extract_pts_feat, pts_bbox_head, its loss.  forward()
    returns the summed losses of the batch's (synthetic, fixed) ground truth."""

    def __init__(self, sample_ids=(), boxes_per_sample=40):
        super().__init__(head=True)
        import numpy as np
        from msmdfusion_amd.head_loss import LiDARBoxes
        self.head = self.det.pts_bbox_head
        self.gt_boxes, self.gt_labels = [], []
        for i in sample_ids:                       # nuScenes-like sizes inside the range
            rs = np.random.RandomState(1000 + i)
            g = boxes_per_sample
            b = np.zeros((g, 9), np.float32)
            b[:, 0:2] = rs.uniform(-50, 50, (g, 2))
            b[:, 2] = rs.uniform(-2.5, -0.5, g)
            b[:, 3:6] = rs.uniform((0.5, 0.5, 1.0), (2.5, 6.0, 3.0), (g, 3))
            b[:, 6] = rs.uniform(-3.14, 3.14, g)
            b[:, 7:9] = rs.uniform(-5, 5, (g, 2))
            self.gt_boxes.append(LiDARBoxes(torch.from_numpy(b)))
            self.gt_labels.append(torch.from_numpy(rs.randint(0, 10, g).astype(np.int64)))

    def _apply(self, fn, *a, **kw):                # ground truth travels with the module
        out = super()._apply(fn, *a, **kw)
        self.gt_boxes = [type(b)(fn(b.tensor)) for b in self.gt_boxes]
        self.gt_labels = [fn(l) for l in self.gt_labels]
        return out

    def forward(self, points, virtual, prepared=None):
        losses = self.det.forward_train(points=points, virtual_points=virtual, prepared=prepared,
                                        gt_bboxes_3d=self.gt_boxes, gt_labels_3d=self.gt_labels)
        return sum(v for k, v in losses.items() if "loss" in k)


def mean_of_product(bev, target):
    """(bev * target).mean() -- the stand-in for everything downstream of the BEV map -- as one
    dot product over the map as it lies in memory (channels-last for the LC path, NCHW for
    configs[1]): the same value and the same gradient (target / numel), without the
    strided-elementwise multiply, the [B,C,H,W] temporary and its reduction that torch
    runs for the product-then-mean form (0.75 ms of a 13.3 ms LC step went to this
    scaffolding, profiles/r03_lc_stream_summary.txt: elementwise 383 + 205 + 83 us,
    reduce 52 us)."""
    if bev.dim() == 4 and bev.permute(0, 2, 3, 1).is_contiguous() \
            and target.permute(0, 2, 3, 1).is_contiguous():
        a, b = bev.permute(0, 2, 3, 1).reshape(-1), target.permute(0, 2, 3, 1).reshape(-1)
    elif bev.is_contiguous() and target.is_contiguous():
        a, b = bev.reshape(-1), target.reshape(-1)
    else:
        return (bev * target).mean()
    return torch.dot(a, b) / a.numel()


def lc_worker_init(sample_ids, seed):
    """Runs in the index worker process (msmdfusion_amd/prefetch_proc.py): the synthetic
    data source and the index half of the LC step.  Weights play no part in prepare();
    the model is built for its structure (conv lists, channel counts) only."""
    from msmdfusion_amd import synthetic as S
    dev = torch.device("cuda", torch.cuda.current_device())
    torch.manual_seed(seed)
    model = FusionBackbone().to(dev).train()
    clouds = [torch.from_numpy(S.lidar_sweep(i)).to(dev) for i in sample_ids]
    virt = [torch.from_numpy(S.virtual_points(i)).to(dev) for i in sample_ids]

    def produce():
        return model.prepare(clouds, virt)
    return produce


def effective_cpus():
    """CPUs this process may actually use: affinity mask, capped by the cgroup's CPU quota
    (threads beyond it only get the whole group throttled)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(workload, seed=0, budget_s=14.0, max_samples=24):
    """The host baseline on a bounded sample: whole synthetic samples through the
    oracle port (oracle/baseline.py), one after the other, until ~budget_s of CPU
    work is done (the first one also warms the OpenMP pool and the page cache and
    is not counted when more follow)."""
    from msmdfusion_amd.hostcpu import unpinned
    with unpinned():         # the rank's own threads are pinned to four CPUs: not this leg
        return _cpu_baseline(workload, seed, budget_s, max_samples)


def _cpu_baseline(workload, seed, budget_s, max_samples):
    from oracle import baseline as B
    from oracle import oracle as O
    # memory-bound gather/scatter loops stop scaling well before the socket is
    # full: cap at 32, and at what the cgroup grants
    cores = O.set_threads(min(effective_cpus(), 32))
    one = B.lc_sample if workload.startswith("lc") else B.transfusion_l_sample
    runs = []
    t_all = time.perf_counter()
    while len(runs) < max_samples and (time.perf_counter() - t_all < budget_s or len(runs) < 2):
        runs.append(one(seed + len(runs)))
    timed = runs[1:] if len(runs) > 1 else runs
    secs = sum(r["seconds"] for r in timed)
    r = timed[-1]
    if workload.startswith("lc"):
        what = ("LiDAR voxelize + rulebooks + 21 encoder convs fwd (frozen), 4-scale virtual-point "
                "voxelize + mean VFE, modality split, FPS/ball-query neighbour search, gates, 16 "
                "fusion-stack convs fwd+dgrad+wgrad + sparse_add; %.1f GMAC fwd (%.1f in the "
                "fusion stack); ~%dk + %dk pts -> ~%dk + %dk voxels"
                % (r["gmac_fwd"], r["gmac_fwd_fusion"], r["points"] // 1000,
                   r["virtual_points"] // 1000, r["voxels"] // 1000, r["virtual_voxels"] // 1000))
    else:
        what = ("voxelize + all rulebooks + 21 sparse convs fwd+dgrad+wgrad; %.1f GMAC fwd; ~%.1fk "
                "pts -> ~%.1fk voxels" % (r["gmac_fwd"], r["points"] / 1e3, r["voxels"] / 1e3))
    # the conv part through the reference's own compiled CPU code, beside the port
    ref = None
    try:
        torch.set_num_threads(cores)
        ref = B.reference_conv_leg(seed)
    except Exception as e:     # the leg is a report, never a reason to lose the line
        ref = dict(error=repr(e)[:200])
    # ... and the index part (voxelization, rulebooks) likewise
    try:
        ref_index = B.reference_index_leg(seed)
    except Exception as e:
        ref_index = dict(error=repr(e)[:200])
    return dict(value=round(len(timed) / secs, 4), unit="samples/s", cores=cores, kind="port",
                reference_index=ref_index if ref_index is not None else
                "oracle/_ref/libmsmd_ref.so is not on this box",
                reference_conv=ref if ref is not None else
                "oracle/_ref/libmsmd_ref.so (the reference's gather / torch::mm / scatter-add "
                "loop, built from /root/reference by oracle/Makefile) is not on this box",
                sample="%d synthetic samples (seeds %d..%d, first one untimed warm-up), each: %s; "
                       "oracle/msmd_oracle.c restatement of the reference CPU path, OpenMP %d "
                       "threads; BN/ReLU/optimizer skipped (elementwise); reference_conv = the conv "
                       "part alone through the reference's compiled gather / torch::mm / scatter "
                       "loop beside the port; %.1f s of CPU work, %.2f s per sample"
                       % (len(timed), seed + (1 if len(runs) > 1 else 0), seed + len(runs) - 1,
                          what, cores, secs, secs / len(timed)))


def run_workload(workload, args, dev, rank, world, profile):
    """Settle + warm-up + K timed steps of one workload on this rank.
    -> (result dict for rank 0's JSON, profile list | None)."""
    from msmdfusion_amd import distributed as D
    from msmdfusion_amd import kernels as K
    from msmdfusion_amd import synthetic as S
    from msmdfusion_amd.prefetch import IndexPrefetcher

    wl = WORKLOADS[workload]
    lc = workload in ("lc", "lc_tail", "lc_b4", "lc_full", "lc_img", "lc_quirks")
    spg = wl["spg"]
    torch.manual_seed(0)
    ids = D.shard_sample_ids(rank, world, spg)      # disjoint samples per rank (weak scaling)
    model = (FusionDetector(ids) if workload == "lc_full" else
             FusionTailBackbone() if workload == "lc_tail" else
             FusionImageBackbone() if workload == "lc_img" else
             FusionBackbone(reference_quirks=True) if workload == "lc_quirks" else
             FusionBackbone() if lc
             else Backbone()).to(dev).train()
    params = [p for p in model.parameters() if p.requires_grad]
    net = D.wrap_data_parallel(model, device_ids=[dev.index])
    # AdamW lr=1e-4, wd=0.01: the configs' optimizer
    # (fused=True: torch's single multi-tensor kernel per step instead of ~10 foreach launches)
    opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.01,
                            fused=os.environ.get("MSMD_FUSED_ADAMW", "1") == "1")

    clouds = [torch.from_numpy(S.lidar_sweep(i)).to(dev) for i in ids]
    batch = (clouds,)
    if workload == "lc_img":
        batch = (clouds,) + synthetic_image_batch(ids, dev)
    elif lc:
        batch = (clouds, [torch.from_numpy(S.virtual_points(i)).to(dev) for i in ids])
    if workload == "lc_full":       # the module returns its loss: (loss * 1).mean()
        target = torch.ones((), device=dev)
    else:
        target = torch.randn(spg, wl["bev_channels"], 180, 180, device=dev)
    if lc and workload != "lc_full":      # the LC path hands its BEV map over channels-last (see FusionBackbone.forward)
        target = target.contiguous(memory_format=torch.channels_last)

    prefetch = None
    if os.environ.get("MSMD_PREFETCH", "1") == "1":
        # worker thread: pays on the LC path (its prepare() waits ~10 ms on host reads,
        # 70 -> 77 samples/s); configs[1] is GPU-bound either way (366 vs 368)
        threaded = os.environ.get("MSMD_PREFETCH_THREAD", "1" if lc else "0") == "1"
        if threaded:    # two threads share the GIL: hand it over promptly (default 5 ms)
            sys.setswitchinterval(float(os.environ.get("MSMD_SWITCH_INTERVAL", "0.0005")))
        if lc and os.environ.get("MSMD_PREFETCH_PROC", "0") == "1":
            # index work in a worker process: no interpreter lock shared with this loop
            from msmdfusion_amd.prefetch_proc import ProcessPrefetcher
            prefetch = ProcessPrefetcher(lc_worker_init, (list(ids), 0), dev)
        else:
            prefetch = IndexPrefetcher(model.prepare, dev, threaded=threaded,
                                       priority=int(os.environ.get("MSMD_INDEX_PRIORITY", "-1")),
                                       # two batches queued on ONE worker: prepare() calls run
                                       # back to back instead of each waiting for the step
                                       # thread to submit it
                                       depth=int(os.environ.get("MSMD_PREFETCH_DEPTH", "2")),
                                       workers=int(os.environ.get("MSMD_PREFETCH_WORKERS", "1")) or None)
    # grad_clip max_norm=10 (config); the step structure is msmdfusion_amd.distributed.TrainStep
    train_step = D.TrainStep(net, params, opt, lambda bev: mean_of_product(bev, target), prefetch,
                             10.0)
    calls = [0]          # every step of this workload, timed or not (profile normalisation)

    def step(b):
        calls[0] += 1
        return train_step(b)
    step.prime = train_step.prime
    # experiment (MSMD_CU_PARTITION="i,s"): the feature pass on the CUs the index / search
    # streams do not own -- this thread's current stream becomes a CU-masked one
    from msmdfusion_amd.prefetch import cu_partition, masked_stream
    part = cu_partition()
    if part is not None and sum(part) > 0 and not getattr(run_workload, "_masked_main", None):
        run_workload._masked_main = masked_stream(dev, range(sum(part), 256))
    main_ctx = torch.cuda.stream(run_workload._masked_main) \
        if getattr(run_workload, "_masked_main", None) is not None else None
    if main_ctx is not None:
        torch.cuda.synchronize()
        main_ctx.__enter__()
    step.prime(batch)

    # Setup, untimed: let torch's caching allocator reach its steady state before
    # the W warm-up steps.  The LC path allocates on four streams (record_stream
    # defers block reuse) and needs ~16 steps before no step calls hipMalloc any more.
    # ... and the GPU its clocks: from idle they ramp over several hundred ms of load (the
    # configs[1] line reads 326 samples/s after 15 untimed steps = 0.14 s, 454 after 70), so
    # the untimed setup also lasts at least MSMD_BENCH_SETTLE_S seconds of stepping
    settle_s = float(os.environ.get("MSMD_BENCH_SETTLE_S", "1.5"))
    D.settle_steps(lambda: step(batch), wl["settle"] if (lc or prefetch is not None) else 2,
                   settle_s, device=dev)      # same count on every rank (each step all-reduces)
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step(batch)
    if args.diag and rank == 0:     # host enqueue time vs device time, outside the timed region
        for _ in range(3):
            torch.cuda.synchronize()
            a = time.perf_counter()
            step(batch)
            b = time.perf_counter()
            torch.cuda.synchronize()
            c = time.perf_counter()
            print("diag: enqueue %.2f ms, step %.2f ms" % ((b - a) * 1e3, (c - a) * 1e3),
                  file=sys.stderr)
    # Per-launch HIP events (recorded on the launch stream) bracket every conv
    # launch of a few timed steps only: on ROCm a timing event is a barrier
    # packet that drains the queue, so bracketing all launches of every
    # step would slow the measured throughput by ~25 %.
    prof = [] if profile else None
    # three sampled steps, spread over the timed region; the roofline figure is the MEDIAN
    # step's (one sampled step that happens to share the chip with a burst of the index pass
    # read 88 instead of 122 TF for the same kernel: DESIGN.md 9.5)
    sampled = set() if prof is None else {0, args.steps // 3, (2 * args.steps) // 3}
    marks = []          # where each sampled step's records start
    torch.cuda.synchronize()
    D.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i in sampled:
            marks.append(len(prof))
        K.PROFILE = prof if i in sampled else None
        loss = step(batch)
    torch.cuda.synchronize()
    D.barrier()
    elapsed = time.perf_counter() - t0
    K.PROFILE = None
    if main_ctx is not None:
        main_ctx.__exit__(None, None, None)
    elapsed = D.global_max(elapsed, device=dev)
    assert torch.isfinite(loss).item()
    # untimed sanity step: every parameter and every gradient of the trained modules is
    # finite (a skipped tile or a stale buffer shows up as NaN/garbage here, not in the rate)
    loss = None
    train_step.drain()      # (the index work queued ahead of the last step: see TrainStep.drain)
    torch.cuda.synchronize()
    bev = net(*batch)
    mean_of_product(bev, target).backward()
    bad = []
    for n, p in model.named_parameters():
        if p.requires_grad and p.grad is None and "blocks_2D" not in n and "blocks_mix" not in n:
            bad.append((n, "no gradient"))
        elif p.grad is not None and not torch.isfinite(p.grad).all():
            bad.append((n, "%d non-finite gradient entries" % int((~torch.isfinite(p.grad)).sum())))
        if not torch.isfinite(p).all():
            bad.append((n, "%d non-finite entries" % int((~torch.isfinite(p)).sum())))
    assert not bad, "workload %s: %d bad parameters after the timed steps: %s" % (
        workload, len(bad), bad[:8])
    opt.zero_grad(set_to_none=True)
    n_samples = args.steps * spg * world
    res = {"value": round(n_samples / elapsed, 3), "unit": "samples/s",
           "ms_per_step": round(elapsed / args.steps * 1e3, 3),
           "config": {"workload": wl["name"], "global_batch": spg * world,
                      "parallelism": "dp%d" % world, "index_prefetch": prefetch is not None,
                      "trainable_params": sum(p.numel() for p in params)}}
    if workload == "lc_img":
        res["image_glue"] = image_glue_kernel_times(model, batch)
    if workload == "lc_quirks":
        res["modality_split"] = modality_split_times(model, batch)
    res["roofline"] = roofline(prof, workload, marks) if prof else None
    if res["roofline"]:
        r = res["roofline"]
        r["step_frac"] = round(r["step_gflop"] / res["ms_per_step"] / r["peak_tflops"], 4)
    if rank == 0:   # tools/stream_prof.py divides a rocprofv3 trace of this run by this count
        print("[bench] workload=%s steps_total=%d (+1 untimed check step)" % (workload, calls[0]),
              file=sys.stderr)
    if hasattr(prefetch, "close"):
        prefetch.close()
    del step, net, model, opt, prefetch
    torch.cuda.empty_cache()
    return res


def main():
    args = parse()
    from msmdfusion_amd import distributed as D
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as a plain process: become N ranks (one per GPU) and relay the line
        D.require_gpus(args.gpus)
        raise SystemExit(D.launch_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:]))
    rank, local_rank, world = D.env_world()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with `python -m torch.distributed.run "
                         "--nproc-per-node %d bench.py --gpus %d ...` (or start bench.py plainly "
                         "and let it spawn its ranks)" % (args.gpus, world, args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    D.require_gpus(local_rank + 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    torch.zeros(1, device=dev)          # the context exists: its flags can be read back
    from msmdfusion_amd.hostcpu import device_schedule_flags
    sched_flags = device_schedule_flags()     # of THIS rank's device (0x4 = blocking waits)
    D.init_distributed(device=dev)      # RCCL (torch backend "nccl") when world > 1

    head = run_workload(args.workload, args, dev, rank, world, not args.no_profile)
    out = None
    if rank == 0:
        out = {"metric": WORKLOADS[args.workload]["metric"], "value": head["value"],
               "unit": head["unit"], "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None,
               "dtype": "f32 via 3xbf16 split MFMA (each fp32 operand = exact sum of 3 bf16 planes, 6 "
                        "products, fp32 accumulate; fp32 storage)",
               "data": "synthetic",
               "rccl_ranks": D.rccl_ranks(),
               "host": {"pinned_cpus": PINNED_CPUS, "blocking_sync": sched_flags == 0x4
                        if sched_flags is not None else bool(BLOCKING_SYNC),
                        "blocking_sync_requested": bool(BLOCKING_SYNC),
                        "device_schedule_flags": sched_flags},
               "config": head["config"],
               "roofline": head["roofline"]}
        for extra in ("image_glue", "modality_split"):
            if extra in head:
                out[extra] = head[extra]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(
                args.workload, budget_s=float(os.environ.get("MSMD_BENCH_CPU_BUDGET", "14.0")))
    if world == 1 and not args.no_also and args.workload == "lc":
        # secondary lines: a leg that fails reports its error instead of costing the headline
        out["also"] = {}

        def leg(key, wl, profile=False, dtype=None, cpu=False):
            """One secondary workload in a process of its own (this script again, `--workload
            wl --no-also`): what an earlier leg leaves behind in the process -- allocator
            pools, cached dense-grid tables, worker threads, packed-weight images -- costs the
            later B = 2 legs up to 20 % (round 6: the same leg read 214, 194 or 174 samples/s
            depending on how many legs had run before it; 205-213 in a fresh process), so
            every line is measured the way the headline is: first in its process."""
            import subprocess
            cmd = [sys.executable, os.path.abspath(__file__), "--workload", wl, "--no-also",
                   "--steps", str(args.steps), "--warmup", str(args.warmup)]
            if not profile:
                cmd.append("--no-profile")
            if not cpu:
                cmd.append("--no-cpu-baseline")
            env = dict(os.environ, MSMD_BENCH_CPU_BUDGET="8.0")
            try:
                p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                   text=True, timeout=900)
                sys.stderr.write(p.stderr[-4000:])
                lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
                if p.returncode != 0 or not lines:
                    raise RuntimeError("exit code %d: %s" % (p.returncode, p.stderr[-300:]))
                d = json.loads(lines[-1])
                r = {k: d[k] for k in ("value", "unit", "ms_per_step", "config", "roofline",
                                       "metric", "image_glue", "modality_split", "cpu_baseline")
                     if k in d}
                if dtype:
                    r["dtype"] = dtype
            except Exception as e:      # noqa: BLE001 -- reported in the JSON line
                r = {"error": "%s: %s" % (type(e).__name__, str(e)[:400])}
                print("[bench] also[%s] failed: %s" % (key, r["error"]), file=sys.stderr)
            out["also"][key] = r

        leg("configs[1]", "transfusion_l", profile=not args.no_profile, cpu=not args.no_cpu_baseline)
        if os.environ.get("MSMD_BENCH_TAIL", "1") == "1":
            leg("configs[2]+image glue", "lc_img")
            leg("configs[2]+f1", "lc_tail")
            leg("configs[2]+f1+f3", "lc_full")
            leg("configs[2] @ 4/GPU", "lc_b4")
            try:
                out["also"]["configs[4]"] = configs4_leg()
            except Exception as e:      # noqa: BLE001 -- reported in the JSON line
                out["also"]["configs[4]"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:400])}
            # BASELINE.json's configs[2] names bf16: the same path with plain bf16 conv
            # operands (one plane, one product; fp32 accumulation, features / BN / BEV stay
            # fp32 in HBM) -- a second line, the headline keeps the reference's fp32 arithmetic
            try:
                os.environ["MSMD_CONV_PLANES"] = "1"
                one = "bf16 conv operands, f32 accumulate and storage"
                leg("configs[2] bf16 operands", "lc", dtype=one)
                leg("configs[2] @ 4/GPU bf16 operands", "lc_b4", dtype=one)
                # two bf16 planes per operand (three products): max error 4.3e-6 of the
                # layer's LARGEST output (element by element within 2e-4), no longer the
                # fp32-equivalent of the headline
                os.environ["MSMD_CONV_PLANES"] = "2"
                two = "two bf16 planes per conv operand (3 products), f32 accumulate"
                leg("configs[1] two bf16 planes", "transfusion_l", dtype=two)
                leg("configs[2] @ 4/GPU two bf16 planes", "lc_b4", dtype=two)
            finally:
                os.environ.pop("MSMD_CONV_PLANES", None)
            # the reference-exact mode (float32-key modality split + the reference's batch
            # offsets): the headline step with reference_quirks=True, and the split alone
            leg("configs[2] reference_quirks", "lc_quirks")
    if rank == 0:
        print(json.dumps(out))
    D.shutdown()


def pmc_summary_path(workload):
    """The newest committed PMC summary of this workload (profiles/rNN_pmc_summary_<wl>.json;
    round 1 wrote one file, for configs[1])."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary_%s.json" % workload)))
    if not found and workload == "transfusion_l":
        found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
    return found[-1] if found else None


def pmc_entry(kernel_name, workload):
    """(matched key, entry, file) of the committed PMC summary for a kernel named as rocprofv3
    names it (a template instantiation: `spconv_fwd_split_kernel<6, 1, 3, 4>`); falls back to
    the instantiation of the same kernel with the most sampled launches."""
    path = pmc_summary_path(workload)
    if path is None:
        return None, {}, None
    kernels = json.load(open(path))["kernels"]
    key = kernel_name if kernel_name in kernels else None
    if key is None and "<" in kernel_name:
        # the same instantiation under a summary written before / after a template parameter
        # was appended (round 4 added the weight-buffer count as a fifth argument)
        args = kernel_name.split("<", 1)[1].rstrip(">").split(", ")
        for k in kernels:
            if k.split("<")[0] == kernel_name.split("<")[0] and "<" in k:
                ka = k.split("<", 1)[1].rstrip(">").split(", ")
                n = min(len(ka), len(args))
                if n >= 4 and ka[:n] == args[:n]:
                    key = k
                    break
    if key is None:
        stem = kernel_name.split("<")[0]
        cands = [k for k in kernels if k.split("<")[0] == stem]
        key = max(cands, key=lambda k: kernels[k].get("launches_sampled", 0)) if cands else None
    return key, kernels.get(key, {}), os.path.relpath(path, ROOT)


def rocprof_entry(kernel_name, workload):
    """(average launch us, calls, file, head) of `kernel_name` in the newest committed rocprofv3
    kernel-stats summary of this workload (profiles/rNN_<wl>_kernel_stats.csv, written by
    tools/prof_bench.sh from `rocprofv3 --kernel-trace --stats` of this command); head = the
    commit the summary was taken at (profiles/rNN_<wl>_kernel_stats.head, when recorded)."""
    import csv
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_kernel_stats.csv" % workload)))
    if not found:
        return None
    path = found[-1]
    want = kernel_name.replace(" ", "")
    for row in csv.DictReader(open(path)):
        # "void msmd::(anonymous namespace)::spconv_fwd_split_kernel<8, 1, 3, 4, 2, false, 2>(float const*, ..."
        name = row["Name"].split("(anonymous namespace)::")[-1].replace(" ", "")
        if name.startswith(want + "("):
            head_file = path[:-4] + ".head"
            head = open(head_file).read().strip() if os.path.exists(head_file) else None
            return float(row["AverageNs"]) / 1e3, int(row["Calls"]), os.path.relpath(path, ROOT), head
    return None


def split_instantiation(c_out, planes):
    """The template instantiation msmd_spconv_fwd_split runs for a layer with c_out output
    channels -> (name as rocprofv3 prints it, kernel launches per call).  Asked of the library
    (msmd_spconv_fwd_split_instantiation: csrc/spconv_split.hip's own dispatch under the
    current environment), not re-derived here: 4 waves / 128-row tiles below
    MSMD_FWD_PP_MIN = 161 output channels (80 and 96 channels run `<6, 1, ..>`, 128 runs
    `<8, 1, ..>`), the 8-wave ping-pong form from there up, 161..192 channels as one 12-tile
    pass."""
    from msmdfusion_amd import kernels as K
    i = K.split_instantiation(c_out)
    return "spconv_fwd_split_kernel<%d, %d, %d, %d, %d, %s, %d>" % (
        i["nt"], i["ub"], planes, i["waves"], i["buffers"], "true" if i["pingpong"] else "false",
        i["tables"]), i["passes"]


HBM_PEAK_TBPS = 8.0     # MI355X_MICROARCH.md: HBM3E spec peak (6.29 TB/s measured copy rate)


def roofline(prof, workload, marks=None):
    """Dominant kernel = the conv kernel (by TEMPLATE INSTANTIATION, the unit rocprofv3
    reports) with the most accumulated time.  SURVEY 8(d): two fractions of ALGORITHMIC work,
      mfma  = flops (2 * pairs * Cin * Cout, the reference's MAC count,
              mmdet3d/apis/flops_counter.py:9-12) / launch time / the matrix peak of its
              arithmetic;
      bytes = algorithmic bytes (features in + out once, the packed weights, the table:
              DESIGN.md section 3) / launch time / 8 TB/s;
    `bound` names the larger one, `frac` is its value.  `traffic` = HBM bytes per launch from
    the PMC counters, beside it: traffic_over_algorithmic is the re-fetch ratio."""
    from msmdfusion_amd.spconv.functional import conv_planes
    pair_cache = {}
    groups = {}
    per_step = {}       # kernel -> [per sampled step: ms, flops, launches]
    bounds = sorted(set(marks or [0])) + [len(prof)]
    step_of = [max(i for i, b in enumerate(bounds[:-1]) if b <= r) for r in range(len(prof))]
    for rec, (kind, s, e, meta) in enumerate(prof):
        ms = s.elapsed_time(e)
        launches = 1
        if kind in ("spconv_fwd", "spconv_fwd_split"):
            nbr = meta["nbr"]
            key = (nbr.data_ptr(), nbr.shape[1])
            if key not in pair_cache:
                pair_cache[key] = int((nbr >= 0).sum().item())
            pairs = pair_cache[key]
            nt = (meta["c_out"] + 15) // 16
            kvol = nbr.shape[0]
            if kind == "spconv_fwd_split":
                name, launches = split_instantiation(meta["c_out"], conv_planes())
                wbytes = 2 * conv_planes()
            else:
                name = ("spconv_fwd_pipe_kernel<NT=%d>" if nt >= 4 and meta["c_in"] % 16 == 0
                        else "spconv_fwd_kernel<NT=%d>") % nt
                wbytes = 4
            algo = 4 * (meta["n_in"] * meta["c_in"] + meta["n_out"] * meta["c_out"]) + \
                wbytes * kvol * meta["c_in"] * meta["c_out"] + 4 * kvol * meta["n_out"]
        else:
            pairs = int(meta["num"].sum().item())
            # (msmd_spconv_wgrad_split runs the whole-block kernel where both widths are
            # multiples of 16 and >= 64 -- csrc/spconv_wgrad_block.hip -- else the slab kernel)
            block = meta["c_in"] % 16 == 0 and meta["c_out"] % 16 == 0 and \
                min(meta["c_in"], meta["c_out"]) >= 64 and os.environ.get("MSMD_WGRAD") != "var"
            name = ("spconv_wgrad_block_kernel<%d>" % conv_planes() if block
                    else "spconv_wgrad_split_var_kernel<%d>" % conv_planes()) \
                if kind == "spconv_wgrad_split" else "spconv_wgrad_kernel"
            kvol = int(meta["num"].numel())
            # both operands once + the pair lists + dW
            algo = 4 * (meta.get("n_in", 0) * meta["c_in"] + meta.get("n_out", 0) * meta["c_out"]) \
                + 8 * pairs + 4 * kvol * meta["c_in"] * meta["c_out"]
        flops = 2.0 * pairs * meta["c_in"] * meta["c_out"]
        if os.environ.get("MSMD_BENCH_LAYERS") == "1":     # per-launch lines (stderr)
            print("[layer] %-40s %4d->%-4d pairs %8d  %7.1f us  %6.1f TF" % (
                name, meta["c_in"], meta["c_out"], pairs, ms * 1e3, flops / (ms * 1e-3) / 1e12),
                file=sys.stderr)
        for g in (groups.setdefault(name, dict(ms=0.0, flops=0.0, launches=0, algo=0.0)),
                  per_step.setdefault(name, [dict(ms=0.0, flops=0.0, launches=0, algo=0.0)
                                             for _ in range(len(bounds) - 1)])[step_of[rec]]):
            g["ms"] += ms
            g["flops"] += flops
            g["launches"] += launches
            g["algo"] += algo
    total_ms = sum(g["ms"] for g in groups.values())
    name, g_all = max(groups.items(), key=lambda kv: kv[1]["ms"])
    # the dominant kernel's MEDIAN sampled step (by its total time)
    steps_of_class = sorted((p for p in per_step[name] if p["launches"]), key=lambda p: p["ms"])
    g = steps_of_class[len(steps_of_class) // 2]
    achieved = g["flops"] / (g["ms"] * 1e-3) / 1e12
    avg_us = g["ms"] / g["launches"] * 1e3
    pmc_key, pmc, pmc_file = pmc_entry(name, workload)
    traffic = pmc.get("hbm_bytes_per_launch")
    peak, peak_note = PEAK_F32_MFMA_TFLOPS, "dense fp32 MFMA (v_mfma_f32_16x16x4_f32)"
    if name.startswith(("spconv_fwd_split", "spconv_wgrad_split", "spconv_wgrad_block")):
        products = {3: 6, 2: 3, 1: 1}[conv_planes()]
        peak = round(PEAK_BF16_MFMA_TFLOPS / products, 1)
        peak_note = ("dense bf16 MFMA peak %.0f TF / %d bf16 products per fp32-equivalent product "
                     "(operands split into %d bf16 planes, fp32 accumulate)"
                     % (PEAK_BF16_MFMA_TFLOPS, products, conv_planes()))
    frac_mfma = achieved / peak
    algo_per_launch = g["algo"] / g["launches"]
    # the fraction is ALGORITHMIC work / launch time / peak on either roof --
    # flops against the matrix peak of the arithmetic, algorithmic bytes (each feature row in
    # and out once, the packed weights, the table) against 8 TB/s -- and `bound` names the
    # larger.  Counter bytes (`traffic`) are reported beside it, never priced as achievement:
    # re-fetched lines are waste, not work.
    algo_tbps = algo_per_launch / (avg_us * 1e-6) / 1e12
    frac_algo_bytes = algo_tbps / HBM_PEAK_TBPS
    hbm_bound = frac_algo_bytes > frac_mfma
    tbps = None if traffic is None else traffic / (avg_us * 1e-6) / 1e12
    # every conv kernel of the median sampled step together, against the same matrix peak
    # (the 5- / 16-channel layers run fp32 MFMAs: their flops are priced against it too)
    med = sorted(range(len(bounds) - 1),
                 key=lambda i: sum(p[i]["ms"] for p in per_step.values()))[(len(bounds) - 1) // 2]
    step_flops = sum(p[med]["flops"] for p in per_step.values())
    step_conv_ms = sum(p[med]["ms"] for p in per_step.values())
    out = {"bound": "hbm" if hbm_bound else "mfma", "kernel": name,
           "achieved": round(algo_tbps * 1e3, 1) if hbm_bound else round(achieved, 3),
           "peak": HBM_PEAK_TBPS * 1e3 if hbm_bound else peak,
           "unit": "GB/s" if hbm_bound else "TFLOP/s",
           "frac": round(frac_algo_bytes if hbm_bound else frac_mfma, 4),
           "frac_mfma": round(frac_mfma, 4), "achieved_tflops": round(achieved, 3),
           "peak_tflops": peak, "peak_note": peak_note,
           "frac_algorithmic_bytes": round(frac_algo_bytes, 4),
           "algorithmic_gbps": round(algo_tbps * 1e3, 1),
           "frac_of_fp32_mfma_peak": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
           "traffic": traffic,
           "traffic_gbps": None if tbps is None else round(tbps * 1e3, 1),
           "algorithmic_bytes": int(algo_per_launch),
           "traffic_over_algorithmic": None if traffic is None else
           round(traffic / algo_per_launch, 2),
           "traffic_note": "HBM bytes per launch of `%s` from the rocprofv3 --pmc passes of this "
                           "command under its default (pipelined) schedule (%s: (2*FETCH_SIZE + "
                           "WRITE_SIZE) KiB, gfx950 correction; Infinity-Cache hits are counted, "
                           "an upper bound); algorithmic_bytes = features in + out once, packed "
                           "weights, neighbour table, mean over the same launches"
                           % (pmc_key, pmc_file),
           "step_gflop": round(step_flops / 1e9, 1),
           "step_conv_ms": round(step_conv_ms, 3),
           "step_frac_note": "step_frac (set by the caller) = step_gflop / ms_per_step / peak_tflops: "
                             "all conv flops of a step over the WHOLE step time",
           "l2_hit_rate_pmc": pmc.get("l2_hit_rate"),
           "mfma_pipe_busy_frac_pmc": pmc.get("mfma_pipe_busy_frac"),
           "avg_launch_us": round(avg_us, 2), "launches": g["launches"],
           "avg_launch_us_clock": "HIP events on the launch stream around each launch, inside the "
                                  "pipelined step (queue gaps and co-running index kernels "
                                  "included); frac / frac_mfma are priced with it.  "
                                  "frac_rocprof prices the same algorithmic work with the "
                                  "kernel's own duration from the committed rocprofv3 summary",
           "sampling": "HIP events around every conv call of %d sampled steps of the timed "
                       "region (a call on > 128 output channels is two launches of the kernel: "
                       "time and flops are per launch); figures of the median step (TF of "
                       "each: %s)" % (
                           len(steps_of_class),
                           ", ".join("%.1f" % (p["flops"] / (p["ms"] * 1e-3) / 1e12)
                                     for p in per_step[name] if p["launches"])),
           "share_of_conv_time": round(g_all["ms"] / total_ms, 3),
           "all_conv_kernels": {k: {"ms": round(v["ms"], 3),
                                    "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 3),
                                    "launches": v["launches"]} for k, v in groups.items()}}
    rp = rocprof_entry(name, workload)
    if rp is not None:      # the same algorithmic flops / bytes per launch over rocprof's duration
        us, calls, rfile, rhead = rp
        fl_per_launch = g["flops"] / g["launches"]
        f_mfma = fl_per_launch / (us * 1e-6) / 1e12 / peak
        f_bytes = algo_per_launch / (us * 1e-6) / 1e12 / HBM_PEAK_TBPS
        out.update(frac_rocprof=round(max(f_mfma, f_bytes), 4),
                   rocprof_avg_launch_us=round(us, 2), rocprof_calls=calls, rocprof_file=rfile,
                   rocprof_head=rhead)
    return out


def configs4_leg():
    """BASELINE configs[4]: the integer kernels at the dense-scene stress size (4 x 10-sweep
    ~290k-pt clouds, 0.05 m voxels, ~0.7 M active voxels, grid 41x2160x2160): hard
    voxelization, SubM and strided rulebook builders, pair lists -- time per call (HIP
    events), algorithmic bytes and GB/s against the 8 TB/s HBM peak.
    The same measurement tools/rulebook_bench.py prints (profiles/rNN_rulebook_*.jsonl)."""
    import importlib.util
    from msmdfusion_amd import synthetic as S
    spec = importlib.util.spec_from_file_location(
        "rulebook_bench", os.path.join(ROOT, "tools", "rulebook_bench.py"))
    rb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rb)
    dev = torch.device("cuda", torch.cuda.current_device())
    stress = [torch.from_numpy(S.lidar_sweep(10 + i, sweeps=10)).to(dev) for i in range(4)]
    rows = rb.case("stress: 4 x 10-sweep ~290k pts, 0.05 m, grid 41x2160x2160", stress,
                   [0.05, 0.05, 0.2], [41, 2160, 2160], 1200000)
    keep = [r for r in rows if "frac_hbm" in r]
    head = max((r for r in keep if r["kernel"].startswith("rulebook_subm3d")),
               key=lambda r: r["GBps"])
    return {"metric": "rulebook HBM GB/s vs roofline, dense-scene stress (configs[4])",
            "value": head["GBps"], "unit": "GB/s", "frac_hbm": head["frac_hbm"],
            "kernel": head["kernel"], "peak": 8000.0, "kernels": keep}


if __name__ == "__main__":
    main()
