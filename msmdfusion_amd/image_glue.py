"""Image-feature side of the virtual points (SURVEY 8 rows a13 image half, f2):
MSMDFusionDetector.depth_aware_channel_compression (MSMDFusion.py:335-368) and
get_foreground2D + score_net (:169-238), same inputs (`img_metas` dicts with
'foreground2D_info', 'lidar2img', 'input_shape', 'pad_shape'), same outputs.

The reference walks B x 6 cameras on the host, per image scale: ~40 numpy ->
device copies and ~60 small launches per scale, four scales per step.  Here the
per-camera arrays are concatenated ONCE per batch (`pack_foreground`: one pinned
staging buffer per dtype, one copy each) and every scale is one gather launch
plus score_net; the depth canvas is one launch.

Gradients: both functions are differentiable w.r.t. the image features and the
module weights.  In the reference's training step nothing flows back through
them, because their output goes straight into `voxelize`, which is
@torch.no_grad() (MSMDFusion.py:462) -- conv1x1_blocks and score_net are
"unused parameters" (configs/MSMDFusion_nusc_voxel_LC.py:309 sets
find_unused_parameters=True for that reason).
"""
import os

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

from . import kernels as K


def _as_numpy(a):
    if hasattr(a, "tensor"):          # mmdet3d LiDARPoints / BasePoints
        a = a.tensor
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.asarray(a)


class ForegroundPack:
    """All cameras of all samples, concatenated in (sample, camera) order and
    resident on the device.  plane = sample * cameras + camera."""

    def __init__(self, pixels, plane, points, real_pixels, real_plane, lidar2img,
                 sample_counts, batch_size, cameras):
        self.pixels, self.plane, self.points = pixels, plane, points
        self.real_pixels, self.real_plane = real_pixels, real_plane
        self.lidar2img = lidar2img
        self.sample_counts = sample_counts      # python ints: points per sample
        self.batch_size, self.cameras = batch_size, cameras


def _stage(arrays, dtype, cols, device):
    """Concatenate host arrays into one pinned buffer and copy once."""
    n = sum(a.shape[0] for a in arrays)
    tdtype = {np.dtype("float32"): torch.float32, np.dtype("float64"): torch.float64,
              np.dtype("int32"): torch.int32}[np.dtype(dtype)]
    shape = (n, cols) if cols else (n,)
    pin = device.type == "cuda"
    host = torch.empty(shape, dtype=tdtype, pin_memory=pin)
    view = host.numpy()
    o = 0
    for a in arrays:
        m = a.shape[0]
        if m:
            view[o:o + m] = a if cols == 0 else a[:, :cols]
        o += m
    return host.to(device, non_blocking=True)


def _pixel_dtype(arrays):
    kinds = {np.asarray(a).dtype for a in arrays if np.asarray(a).size}
    # the product fg_pxl * downscale_factor keeps the array's dtype in numpy: float32
    # pixels are scaled in float32, anything else (float64, integers) in float64
    return np.float32 if kinds and kinds <= {np.dtype("float32")} else np.float64


def pack_foreground(img_metas, device):
    """img_metas[i]['foreground2D_info'] = dict(fg_pixels=[cam arrays [n,3]
    (x, y, depth)], fg_points=[cam [n,15]], fg_real_pixels=[cam [m,3]]) and
    img_metas[i]['lidar2img'] = [cam 4x4] (what MyLoadForeground2D /
    the nuScenes loader produce: my_loading_multi_proj.py:38-97)."""
    device = torch.device(device)
    B = len(img_metas)
    cams = len(img_metas[0]["foreground2D_info"]["fg_pixels"])
    pix, pts, rpix, plane, rplane, l2i, counts = [], [], [], [], [], [], []
    for b, meta in enumerate(img_metas):
        info = meta["foreground2D_info"]
        if len(info["fg_pixels"]) != cams:
            raise ValueError("every sample must carry the same number of cameras")
        total = 0
        for j in range(cams):
            p = _as_numpy(info["fg_pixels"][j]).reshape(-1, 3)
            q = _as_numpy(info["fg_points"][j])
            if q.shape[0] != p.shape[0]:
                raise ValueError("fg_pixels and fg_points of sample %d camera %d disagree: "
                                 "%d vs %d rows" % (b, j, p.shape[0], q.shape[0]))
            pix.append(p)
            pts.append(q.astype(np.float32, copy=False))
            plane.append(np.full((p.shape[0],), b * cams + j, np.int32))
            total += p.shape[0]
            if "fg_real_pixels" in info:
                r = _as_numpy(info["fg_real_pixels"][j]).reshape(-1, 3)
                rpix.append(r)
                rplane.append(np.full((r.shape[0],), b * cams + j, np.int32))
            l2i.append(np.asarray(meta["lidar2img"][j], dtype=np.float32).reshape(1, 16))
        counts.append(total)
    pts_dim = next((q.shape[1] for q in pts if q.ndim == 2), 15)
    pts = [q.reshape(-1, pts_dim) for q in pts]
    return ForegroundPack(
        pixels=_stage(pix, _pixel_dtype(pix), 3, device),
        plane=_stage(plane, np.int32, 0, device),
        points=_stage(pts, np.float32, pts_dim, device),
        real_pixels=_stage(rpix, _pixel_dtype(rpix), 3, device) if rpix else None,
        real_plane=_stage(rplane, np.int32, 0, device) if rpix else None,
        lidar2img=_stage(l2i, np.float32, 16, device),
        sample_counts=counts, batch_size=B, cameras=cams)


def raise_if_any_bad(collected):
    """One host read for the out-of-map counters a whole batch collected (check=[] passed
    to get_foreground2D / DepthAwareChannelCompression): the reference's IndexError, raised
    once per batch instead of after every gather (five waits for the stream per step)."""
    if not collected:
        return
    counts = torch.stack([b.reshape(()) for b, _ in collected]).tolist()
    for n, (_, what) in zip(counts, collected):
        if n:
            raise IndexError("%s: %d pixel(s) fall outside the map (the reference's advanced "
                             "indexing raises IndexError here too)" % (what, n))


def _raise_if_bad(bad, what):
    n = int(bad.item())
    if n:
        raise IndexError("%s: %d pixel(s) fall outside the map (the reference's advanced "
                         "indexing raises IndexError here too)" % (what, n))


class _ForegroundGather(torch.autograd.Function):
    """img_feat -> ([pts | feat], [feat | depth | lidar2img]); the backward
    scatter-adds both feature blocks into the map."""

    @staticmethod
    def forward(ctx, img_feat, pack, downscale, check):
        fg, sc, cells, bad = K.fg_gather(img_feat, pack.pixels, pack.plane, downscale,
                                         pack.points, pack.lidar2img, want_cells=True)
        if isinstance(check, list):      # deferred: the caller reads all counters at once
            check.append((bad, "get_foreground2D"))
        elif check:
            _raise_if_bad(bad, "get_foreground2D")
        ctx.save_for_backward(cells)
        ctx.like = img_feat
        ctx.pts_dim = pack.points.shape[1]
        ctx.bad = bad
        return fg, sc

    @staticmethod
    def backward(ctx, g_fg, g_sc):
        (cells,) = ctx.saved_tensors
        c = ctx.like.shape[1]
        g = g_fg[:, ctx.pts_dim:] + g_sc[:, :c]
        return K.fg_scatter_add(g, 0, c, cells, ctx.like), None, None, None


def get_foreground2D(img_feats, img_metas, score_net, pack=None, check=True,
                     reference_quirks=False):
    """MSMDFusion.py:169-238.  img_feats: [B*cams, C, h, w] (any memory format).
    Returns batch_fg_pcd_cams: B tensors [n_b, pts_dim + C], rows in (camera,
    point) order, the C image channels scaled by score_net.

    Default: every sample's channels are scaled.  The reference scales a torch.cat COPY
    (:226-228) and copies the scaled channels back for sample 0 always and for sample 1
    `if B == 2` only ("only suit for bs = 2", :229-234): identical for B <= 2, the only
    sizes it trains with; reference_quirks=True reproduces it for larger batches too
    (B >= 3: samples 1.. keep their unscaled channels)."""
    if pack is None:
        pack = pack_foreground(img_metas, img_feats.device)
    downscale = img_feats.shape[-1] / img_metas[0]["input_shape"][-1]
    C = img_feats.shape[1]
    if img_feats.shape[0] != pack.batch_size * pack.cameras:
        raise ValueError("img_feats has %d maps for %d samples x %d cameras"
                         % (img_feats.shape[0], pack.batch_size, pack.cameras))
    lin = score_net[0] if isinstance(score_net, nn.Sequential) and len(score_net) == 2 else None
    if (_FG_FUSED and not torch.is_grad_enabled() and isinstance(lin, nn.Linear)
            and isinstance(score_net[1], nn.ReLU) and lin.out_features == 1
            and lin.bias is not None and lin.in_features == C + 17 and C <= 64):
        # no gradient wanted (the training step itself: this feeds voxelize(), which is
        # @torch.no_grad(), MSMDFusion.py:462): gather, score_net and the scaling in one launch
        n_scaled = None
        if reference_quirks and pack.batch_size > 2:        # :229-234: sample 0 only
            n_scaled = pack.sample_counts[0]
        fg, bad = K.fg_gather_scored(img_feats.float(), pack.pixels, pack.plane, downscale,
                                     pack.points, pack.lidar2img, lin.weight, lin.bias, n_scaled)
        if isinstance(check, list):
            check.append((bad, "get_foreground2D"))
        elif check:
            _raise_if_bad(bad, "get_foreground2D")
        return list(torch.split(fg, pack.sample_counts, 0))
    fg, score_in = _ForegroundGather.apply(img_feats.float(), pack, downscale, check)
    scores = score_net(score_in)                             # [n,1], Linear(66,1)+ReLU
    scaled = torch.cat([fg[:, :-C], fg[:, -C:] * scores], 1)
    out = list(torch.split(scaled, pack.sample_counts, 0))
    if reference_quirks and len(out) > 2:
        out[1:] = list(torch.split(fg, pack.sample_counts, 0))[1:]
    return out


def sparse_depth_canvas(img_metas, H, W, device, pack=None, check=True):
    """MSMDFusion.py:336-356 -> canvas [B*cams, 1, H, W]."""
    if pack is None:
        pack = pack_foreground(img_metas, device)
    planes = pack.batch_size * pack.cameras
    if pack.real_pixels is None:
        return torch.zeros((planes, 1, H, W), dtype=torch.float32, device=device)
    canvas, bad = K.depth_canvas(pack.real_pixels, pack.real_plane, planes, H, W)
    if isinstance(check, list):
        check.append((bad, "depth_aware_channel_compression"))
    elif check:
        _raise_if_bad(bad, "depth_aware_channel_compression")
    return canvas.view(planes, 1, H, W)


_FG_FUSED = os.environ.get("MSMD_FG_FUSED", "1") == "1"
_GLUE_NHWC = int(os.environ.get("MSMD_GLUE_NHWC", "1"))


class DepthAwareChannelCompression(nn.Module):
    """conv1x1_blocks of MSMDFusionDetector (MSMDFusion.py:106-123): three
    Conv2d(256+1 -> 49, k 5/5/3, bias=False) + BN2d(eps 1e-3, momentum 0.01) +
    ReLU, applied to [feature map | bilinearly resized sparse depth map].
    Child names match the reference, so `detector.conv1x1_blocks.*` checkpoint
    keys load with prefix 'conv1x1_blocks.'."""

    def __init__(self, in_channels=256, out_channels=49, kernel_sizes=(5, 5, 3)):
        super().__init__()
        self.conv1x1_blocks = nn.ModuleList([
            nn.Sequential(
                nn.Conv2d(in_channels + 1, out_channels, kernel_size=k, stride=1, padding=k // 2,
                          bias=False),
                nn.BatchNorm2d(out_channels, eps=0.001, momentum=0.01),
                nn.ReLU(),
            ) for k in kernel_sizes])

    def forward(self, feat_list, img_metas, pack=None, check=True):
        H, W = img_metas[0]["pad_shape"][:2]
        canvas = sparse_depth_canvas(img_metas, H, W, feat_list[0].device, pack, check)
        out = []
        # Memory format of the compressed maps.  get_foreground2D's gather reads, per virtual
        # point, the C = 49 channels of ONE pixel: 49 floats H*W*4 bytes apart in an NCHW map
        # (49 separate sectors per point), 196 contiguous bytes in a pixel-major
        # (channels-last) one.  MSMD_GLUE_NHWC: 1 (default) = NCHW convs + ONE conversion pass
        # per map, 0 = NCHW maps as in rounds 2-5, 2 = the block itself channels-last.
        # Measured (MI355X, lc_img, profiles/r06_image_glue_formats.txt): gather on the 112 x
        # 200 / 56 x 100 / 28 x 50 maps 75.5 / 39.3 / 31.7 us (0.13 / 0.25 / 0.31 of HBM) ->
        # 24.9 / 24.0 / 22.4 us (0.40 / 0.41 / 0.44) with pixel-major maps; whole glue 5.21 ->
        # 5.02 ms, step 106.1 -> 110.0 samples/s; MIOpen's fp32 NHWC kernels (mode 2) take
        # 7.10 ms for the three blocks (101.6 samples/s).  Same values in any format
        # (K.fg_gather takes the map's strides).
        mode = _GLUE_NHWC
        for i, block in enumerate(self.conv1x1_blocks):
            feat = feat_list[i]
            depth = F.interpolate(canvas, feat.shape[-2:], mode="bilinear")
            x = torch.cat([feat, depth], 1)
            if mode == 2:
                x = x.contiguous(memory_format=torch.channels_last)
            y = block(x)
            if mode >= 1:
                y = y.contiguous(memory_format=torch.channels_last)
            out.append(y)
        return out


class ScoreNet(nn.Sequential):
    """score_net of MSMDFusionDetector (MSMDFusion.py:125-128)."""

    def __init__(self, in_features=50 + 16):
        super().__init__(nn.Linear(in_features, 1), nn.ReLU())
