"""Detector-side glue of the hot path, as free functions / a small module:
MSMDFusionDetector.voxelize (MSMDFusion.py:462-491), fetch_2D_voxels' voxel half
(:371-393), voxel_modality_split (:251-325), extract_pts_feat's sparse part
(:421-443).  The image branch, score_net and the dense BEV tail are outside the
hot path proper; virtual points arrive here as ready [N,64] tensors
(image_glue.get_foreground2D builds them; bev.BevTail consumes the result).
"""
import os
import threading

import torch
from torch import nn
from torch.nn import functional as F

from . import kernels as K
from . import spconv
from .spconv.core import plan_batch


@torch.no_grad()
def voxelize_batch(voxel_layer, points, downscale_factor=1.0, base_voxel_size=None,
                   fused_mean=False):
    """MSMDFusion.py:462-491: per-sample hard voxelization at voxel size
    base * downscale_factor, concatenated, batch id prepended to coors.
    Returns (voxels | mean, num_points, coors[M,4]) -- the reference's order."""
    base = list(base_voxel_size) if base_voxel_size is not None else [0.075, 0.075, 0.2]
    # the reference rescales the layer's attribute (:475-478); passed per call here, so
    # that batches prepared concurrently (IndexPrefetcher depth > 1) cannot see each
    # other's scale
    size = [v * downscale_factor for v in base]
    feats, coors, nums = [], [], []
    for f, c, n in voxel_layer.forward_batch(points, fused_mean=fused_mean, voxel_size=size):
        feats.append(f)
        coors.append(c)
        nums.append(n)
    coors_batch = [F.pad(c, (1, 0), mode="constant", value=i) for i, c in enumerate(coors)]
    return torch.cat(feats, 0), torch.cat(nums, 0), torch.cat(coors_batch, 0)


def virtual_points_to_voxels(voxel_layer, fg_points, spatial_shape, downscale_factor, batch_size,
                             base_voxel_size=None):
    """fetch_2D_voxels after get_foreground2D (MSMDFusion.py:374-393): zero-pad
    empty samples to 100 points (:376-380), voxelize at the stage's scale, mean
    VFE over all 64 channels, xyz /= (13.5, 13.5, 2.0) (:388-389)."""
    pts = []
    for p in fg_points:
        if p.shape[0] == 0:
            p = p.new_zeros((100, p.shape[1]))
        pts.append(p)
    mean, _, coors = voxelize_batch(voxel_layer, pts, downscale_factor, base_voxel_size,
                                    fused_mean=True)
    norm = _xyz_norm(mean.device) if mean.is_cuda else mean.new_tensor([13.5, 13.5, 2.0])
    mean = torch.cat([mean[:, :3] / norm[None, :], mean[:, 3:]], 1)
    return spconv.SparseConvTensor(mean, coors, spatial_shape, batch_size)


def modality_split_indices(idx3, idx2, batch_size, spatial_shape, float_keys=False,
                           reference_offsets=False):
    """Index-level voxel_modality_split: 4-column (b,z,y,x) indices of the two
    voxel sets -> (5-column indices of each with the mix flag inserted, matched
    row lists).  Needs no features."""
    mix3, mix2, pair3, pair2 = K.modality_split(idx3, idx2, batch_size, spatial_shape,
                                                float_keys=float_keys,
                                                reference_offsets=reference_offsets)
    idx3_5 = torch.cat([idx3[:, :1], mix3[:, None], idx3[:, 1:]], 1).contiguous()
    idx2_5 = torch.cat([idx2[:, :1], mix2[:, None], idx2[:, 1:]], 1).contiguous()
    return idx3_5, idx2_5, pair3.long(), pair2.long()


def voxel_modality_split(voxel_3D, voxel_2D, batch_size, float_keys=False,
                         reference_offsets=None):
    """MSMDFusion.py:251-325: mark voxels present in both modalities.
    indices become 5 columns (batch, mix_flag, z, y, x); syn_mix_3D / syn_mix_2D
    list the matched rows of each tensor, aligned, in ascending key order per sample.
    Default: exact integer keys (the reference's float32 keys alias for z >= 17 or
    x >= 1000 and mark voxels that merely share a rounded key: SURVEY Appendix B.3).
    float_keys=True: the reference's keys and two-pointer merge, bit for bit
    (csrc/modality_float.hip) -- what a checkpoint trained with the reference has seen;
    reference_offsets (default: follows float_keys): pair rows numbered with the reference's
    non-cumulative batch offsets (:288-289,313-314; the same rows for batch <= 2)."""
    idx3, idx2 = voxel_3D.indices, voxel_2D.indices
    assert idx3.shape[1] == 4 and idx2.shape[1] == 4
    shape = [max(a, b) for a, b in zip(voxel_3D.spatial_shape, voxel_2D.spatial_shape)]
    voxel_3D.indices, voxel_2D.indices, pair3, pair2 = modality_split_indices(
        idx3, idx2, batch_size, shape, float_keys=float_keys,
        reference_offsets=float_keys if reference_offsets is None else reference_offsets)
    return voxel_3D, voxel_2D, pair3, pair2


_NN_STREAMS_LOCK = threading.Lock()     # see prepare(): the shared neighbour-search streams
_XYZ_NORM = {}


def _xyz_norm(device):
    """(13.5, 13.5, 2.0) on the device (MSMDFusion.py:388-389), made once: `new_tensor` of a
    Python list is a blocking pageable copy -- one more wait for the stream per prepare()."""
    t = _XYZ_NORM.get(device)
    if t is None:
        t = _XYZ_NORM[device] = torch.tensor([13.5, 13.5, 2.0], dtype=torch.float32,
                                             device=device)
    return t


class SparseFusionPath(nn.Module):
    """extract_pts_feat's sparse section (MSMDFusion.py:421-443) behind one
    module: LiDAR clouds + per-stage virtual points -> (x[B,256,180,180],
    x_mm[B,384,180,180]) ready for bev_fusion."""

    def __init__(self, voxel_layer, middle_encoder, multimodal_encoder,
                 spatial_shapes=([41, 1440, 1440], [21, 720, 720], [11, 360, 360], [5, 180, 180]),
                 downscale_factors=(1, 2, 4, 8), fps_num_list=(2048,) * 4,
                 radius_list=(6, 3, 2, 1), max_cluster_samples_list=(200, 100, 50, 25),
                 dist_thresh_list=(13.3, 6.6, 3.3, 1.6), base_voxel_size=(0.075, 0.075, 0.2),
                 reference_quirks=False):
        """reference_quirks=True: the behaviours of the reference a published checkpoint was
        trained with, bit for bit, instead of their fixes -- float32 voxel keys in
        voxel_modality_split (MSMDFusion.py:271-272: false "mixed" voxels wherever keys
        alias, on every real frame at the 0.075 m scale) and non-cumulative batch offsets
        (:288-289,313-314 and sparse_multimodal_encoder_painting.py:355-369; same rows for
        batch <= 2).  INTEGRATION.md, "Which mode reproduces a published checkpoint"."""
        super().__init__()
        self.reference_quirks = bool(reference_quirks)
        multimodal_encoder.reference_quirks = self.reference_quirks
        self.pts_voxel_layer = voxel_layer
        self.pts_middle_encoder = middle_encoder
        self.multimodal_middle_encoder = multimodal_encoder
        self.spatial_shapes = [list(s) for s in spatial_shapes]
        self.downscale_factors = list(downscale_factors)
        self.fps_num_list = list(fps_num_list)
        self.radius_list = list(radius_list)
        self.max_cluster_samples_list = list(max_cluster_samples_list)
        self.dist_thresh_list = list(dist_thresh_list)
        self.base_voxel_size = list(base_voxel_size)

    @torch.no_grad()
    def _voxelize_all(self, points, virtual_points_per_stage, B):
        """voxelize(pts) (MSMDFusion.py:425) and the four fetch_2D_voxels
        voxelizations (:382, one per image scale) as ONE batch of 5*B clouds with
        per-cloud voxel sizes.  -> (LiDAR mean features, LiDAR coors, [voxel_2D x4])."""
        base = self.base_voxel_size
        clouds, sizes = list(points), [list(base)] * B
        for i in range(4):
            for p in virtual_points_per_stage[i]:
                clouds.append(p if p.shape[0] else p.new_zeros((100, p.shape[1])))   # :376-380
                sizes.append([v * self.downscale_factors[i] for v in base])
        res = self.pts_voxel_layer.forward_batch(clouds, fused_mean=True, voxel_size=sizes)

        def joined(group):
            coors = torch.cat([F.pad(c, (1, 0), mode="constant", value=b)
                               for b, (_, c, _) in enumerate(group)], 0)
            return torch.cat([f for f, _, _ in group], 0), coors
        feats, coors = joined(res[:B])
        norm = _xyz_norm(feats.device)
        v2 = []
        for i in range(4):
            mean, c2 = joined(res[B * (i + 1):B * (i + 2)])
            mean = torch.cat([mean[:, :3] / norm[None, :], mean[:, 3:]], 1)     # :388-389
            v2.append(spconv.SparseConvTensor(mean, c2, self.spatial_shapes[i], B))
        return feats, coors, v2

    def prepare(self, points, virtual_points_per_stage, nn_side_stream=True):
        """Everything of a step that depends on the INPUTS alone (no weights, no
        previous step): LiDAR voxelization, the encoder's rulebooks, the
        virtual-point voxels, the modality split and the FPS / nearest-voxel
        search of all four stages (9 of the step's 38 ms, two workgroups wide).

        nn_side_stream=True enqueues the neighbour search on a high-priority side
        stream so that it runs underneath the LiDAR encoder's feature pass of the
        SAME step; a caller that runs prepare() a step ahead on its own stream
        (msmdfusion_amd/prefetch.py) passes False."""
        B = len(points)
        enc, mm = self.pts_middle_encoder, self.multimodal_middle_encoder
        # Host reads are what this function's latency is made of (each one waits for a
        # short chain of small kernels that queue behind the feature pass's chip-filling
        # ones): the five voxelizations share one read, the four modality splits share
        # one -- which also brings the per-sample row counts every later selection needs.
        # (plan_batch: the tilings / pair lists of all ~21 tables built below are computed
        # together when the context closes -- one launch set, see spconv/core.py)
        with plan_batch("all"):
            feats, coors, v2 = self._voxelize_all(points, virtual_points_per_stage, B)
            planned, stages = enc.plan(coors, B)
            jobs = []
            for i in range(4):
                shape = [max(a, b) for a, b in zip(stages[i][1], v2[i].spatial_shape)]
                jobs.append((stages[i][0], v2[i].indices, shape))
            idx3_5, s3, s2, plans = [], [], [], []
            q = self.reference_quirks
            splits = K.modality_split_many(jobs, B, float_keys=q, reference_offsets=q)
            # the row lists of the unmatched voxels of both sets at all four scales: one scan
            lists = K.rows_where_eq_many(
                [(m, 0, sum(st[key])) for (m3, m2, _, _, st) in splits
                 for m, key in ((m3, "c3_plain"), (m2, "c2_plain"))])
            for i, (mix3, mix2, pa, pb, stats) in enumerate(splits):
                idx3, idx2 = jobs[i][0], jobs[i][1]
                i3 = torch.cat([idx3[:, :1], mix3[:, None], idx3[:, 1:]], 1).contiguous()
                v2[i].indices = torch.cat([idx2[:, :1], mix2[:, None], idx2[:, 1:]], 1).contiguous()
                idx3_5.append(i3); s3.append(pa.long()); s2.append(pb.long())
                plans.append(mm.plan_stage_rows(i3, v2[i].indices, B, stats, bzyx3=idx3, bzyx2=idx2,
                                                mix3=mix3, mix2=mix2,
                                                plain_rows=(lists[2 * i], lists[2 * i + 1])))
            # the fusion stack's own voxel sets and rulebooks, stage by stage (each needs
            # the previous stage's output set).  Before the neighbour search is enqueued:
            # these calls read counts back, and must not wait behind 9 ms of FPS
            need_grad = torch.is_grad_enabled()
            # the sets every stage builds from its own inputs first (no host read), then the
            # chain through the stages -- union with the previous stage's output, down-scaling
            # conv -- counted on the device in one go: one read instead of two per stage
            for i in range(4):
                mm.plan_stage_sets(plans[i], s2[i], stages[i][1], self.spatial_shapes[i], B, i,
                                   need_grad)
            mm.plan_stage_chain(plans, B, need_grad)
        counts = [p["counts_host"] for p in plans]
        main = torch.cuda.current_stream()
        # one side stream PER STAGE: a stage's chain is FPS (2047 serial rounds, one
        # workgroup per sample: 2 of 256 CUs busy for 6 ms at stage 0, 3 ms at stage 1) ->
        # nearest voxel -> ball query -> assignment, and the four chains are independent.
        # On one stream they took 11.7 ms back to back -- longer than the feature pass
        # they hide under, i.e. the LC step time; side by side the longest one (7.5 ms) counts.
        # The search streams are process-wide (slots 8..11) and their scratch is per stream:
        # ONE thread at a time may drive them.  Two prepare() calls can overlap -- a prefetcher
        # of depth 2, or a forward pass run inline while the worker prepares the next batch
        # (round 3: bench.py's sanity step did, and once in ~10 runs read another batch's
        # neighbour indices) -- so the enqueue of the four chains is a critical section.
        with _NN_STREAMS_LOCK:
            for i in range(4):
                side = self._side_stream(feats.device, i) if nn_side_stream else main
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    mm.plan_stage_nn(plans[i], counts[i], B, self.fps_num_list[i],
                                     self.radius_list[i], self.max_cluster_samples_list[i],
                                     self.dist_thresh_list[i])
                    if nn_side_stream:
                        plans[i]["nn3"].record_stream(main)
                        # (read by the assembly's backward on the main stream: a caller that
                        # drops the plan early must not hand their blocks back to the search
                        # stream's allocator while that kernel is still queued)
                        for seg in plans[i].get("nn_segments") or ():
                            seg.record_stream(main)
                        plans[i]["ready"] = torch.cuda.Event()
                        plans[i]["ready"].record(side)
        return dict(feats=feats, coors=coors, planned=planned, stages=stages, v2=v2,
                    idx3_5=idx3_5, s3=s3, s2=s2, plans=plans)

    def forward(self, points, virtual_points_per_stage, prepared=None, joint_bev=False):
        """points: list of B [N,5] clouds; virtual_points_per_stage: 4 lists of
        B [Nv,64] tensors (what get_foreground2D yields per image scale).

        Order of work (results are those of MSMDFusion.py:421-443; only the
        schedule differs): the index-only part (prepare) first, then the two
        feature passes.  `prepared` = a prepare() result computed ahead of time.
        joint_bev=True returns cat([x, x_mm], 1) -- bev_fusion's input -- as one
        channels-last [B,640,H,W] map both sparse tensors scatter into."""
        B = len(points)
        enc, mm = self.pts_middle_encoder, self.multimodal_middle_encoder
        p = prepared if prepared is not None else self.prepare(points, virtual_points_per_stage)
        x, encode_features = enc(p["feats"], p["coors"], B, planned=p["planned"],
                                 dense_out=not joint_bev)
        v3 = [spconv.SparseConvTensor(encode_features[i].features, p["idx3_5"][i],
                                      p["stages"][i][1], B) for i in range(4)]
        stage_outs = mm(v3, p["v2"], p["s3"], p["s2"], self.fps_num_list, self.radius_list,
                        self.max_cluster_samples_list, self.dist_thresh_list,
                        stage_plans=p["plans"])
        if joint_bev:
            return spconv.functional.bev_concat([x, stage_outs[-1]])
        mm_dense = stage_outs[-1].dense()
        n, c, d, h, w = mm_dense.shape
        return x, mm_dense.view(n, c * d, h, w)

    def _side_stream(self, device, stage=0):
        # high priority: the FPS workgroups are 1024 threads x 128 registers -- a whole
        # CU each -- and must win the CU when one drains between the main stream's
        # chip-filling persistent kernels, or the search starts late.  Process-wide
        # streams (slots 8..11; the prefetcher owns the low slots): see prefetch.side_stream
        from .prefetch import side_stream
        # MSMD_NN_STREAMS (1..4): stages share streams round robin, paired long + short
        # (stage 0 with 3, 1 with 2) when there are two
        n = max(1, min(4, int(os.environ.get("MSMD_NN_STREAMS", "2"))))
        slot = stage if n == 4 else (min(stage, 3 - stage) if n == 2 else stage % n)
        return side_stream(device, int(os.environ.get("MSMD_NN_PRIORITY", "-1")), slot=8 + slot)
