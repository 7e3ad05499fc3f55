"""CPU restatement of the image-feature glue (SURVEY 8 rows a13 image half, f2)
and of the channels-last BEV hand-over.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  numpy loops over samples
and cameras exactly as the reference walks them; pinned by
tests/golden/image_glue_vectors.npz, which holds outputs of the reference's own
functions (tests/golden/make_image_glue_golden.py runs them from
/root/reference in the build container).
"""
import numpy as np


def foreground_cells(fg_pixels, downscale):
    """MSMDFusion.py:209-212: fg_feat_pxl = (fg_pxl * downscale_factor).long();
    the product keeps the numpy dtype of fg_pxl, .long() truncates toward zero.
    -> (coord_h, coord_w)."""
    scaled = np.asarray(fg_pixels) * downscale
    cells = np.trunc(scaled).astype(np.int64)
    return cells[:, 1], cells[:, 0]


def get_foreground2d(img_feats, metas, w_score, b_score, reference_write_back=False):
    """MSMDFusionDetector.get_foreground2D (MSMDFusion.py:169-238).
    img_feats [B*N, C, H, W]; metas as the reference's img_metas (fg_points entries
    are arrays here); score_net = ReLU(Linear(C+17 -> 1)) with weight w_score
    [1, C+17], bias b_score [1].  Returns B arrays [n_b, 15 + C].
    reference_write_back: the scaled channels are computed on a torch.cat COPY
    (:226-228) and copied back for sample 0 always and for sample 1 `if B == 2` only
    (:230-234, "only suit for bs = 2"): every other sample keeps its unscaled channels."""
    B = len(metas)
    BN, C, H, W = img_feats.shape
    N = BN // B
    downscale = W / metas[0]["input_shape"][-1]                       # :182
    feats = img_feats.reshape(B, N, C, H, W)
    out = []
    for b in range(B):
        info = metas[b]["foreground2D_info"]
        rows = []
        for v in range(N):
            pxl = np.asarray(info["fg_pixels"][v]).reshape(-1, 3)
            pts = np.asarray(info["fg_points"][v], dtype=np.float32)
            pts = pts.reshape(pxl.shape[0], pts.shape[-1] if pts.ndim == 2 else 15)
            trans = np.asarray(metas[b]["lidar2img"][v], dtype=np.float32).reshape(1, 16)  # :206
            ch, cw = foreground_cells(pxl, downscale)
            fg_feat = feats[b, v].transpose(1, 2, 0)[ch, cw]          # :213 (python wrap of <0)
            depth = pxl[:, 2:3].astype(np.float32)                    # :208
            score_in = np.concatenate([fg_feat, depth, np.repeat(trans, pxl.shape[0], 0)], 1)
            score = np.maximum(score_in.astype(np.float32) @ w_score.T.astype(np.float32)
                               + b_score.astype(np.float32), 0)       # :227
            scaled = b == 0 or (b == 1 and B == 2) or not reference_write_back   # :230-234
            rows.append(np.concatenate([pts, fg_feat * score if scaled else fg_feat], 1))  # :221, :228
        out.append(np.concatenate(rows, 0).astype(np.float32))
    return out


def depth_canvas(metas, H, W, cam_num=6):
    """depth_aware_channel_compression's canvas (MSMDFusion.py:336-356):
    canvas[i, j][y, x] = depth, in row order -- a later row overwrites an earlier
    one on the same pixel (numpy's documented rule for repeated indices, and what
    torch's CPU index_put_ does)."""
    B = len(metas)
    canvas = np.zeros((B, cam_num, H, W), np.float32)
    for i in range(B):
        real = metas[i]["foreground2D_info"]["fg_real_pixels"]
        for j in range(cam_num):
            r = np.asarray(real[j]).reshape(-1, 3)
            xy = np.trunc(r[:, :2]).astype(np.int64)                  # .long()
            for k in range(r.shape[0]):
                canvas[i, j, xy[k, 1], xy[k, 0]] = r[k, 2]
    return canvas.reshape(-1, 1, H, W)


def bilinear_resize(x, h, w):
    """F.interpolate(x, (h, w), mode='bilinear') with align_corners=False
    (MSMDFusion.py:363): source coordinate (dst + 0.5) * in/out - 0.5, clamped at
    0; the upper neighbour clamps at the border."""
    def axis(n_in, n_out):
        src = (np.arange(n_out, dtype=np.float32) + np.float32(0.5)) * np.float32(n_in / n_out) \
            - np.float32(0.5)
        src = np.maximum(src, np.float32(0))
        i0 = np.minimum(np.floor(src).astype(np.int64), n_in - 1)
        i1 = np.minimum(i0 + 1, n_in - 1)
        lam = (src - i0.astype(np.float32)).astype(np.float32)
        return i0, i1, lam
    y0, y1, ly = axis(x.shape[-2], h)
    x0, x1, lx = axis(x.shape[-1], w)
    top = x[..., y0, :][..., x0] * (1 - lx) + x[..., y0, :][..., x1] * lx
    bot = x[..., y1, :][..., x0] * (1 - lx) + x[..., y1, :][..., x1] * lx
    return (top * (1 - ly)[:, None] + bot * ly[:, None]).astype(np.float32)


def bev_concat(tensors, batch_size):
    """torch.cat([dense(t).view(N, C*D, H, W) for t in tensors], 1)
    (structure.py:55-64, MSMDFusion.py:436-440); tensors = [(feat [n,c], indices
    [n,4] (b,z,y,x), (D,H,W)), ...]."""
    maps = []
    for feat, idx, (D, H, W) in tensors:
        c = feat.shape[1]
        dense = np.zeros((batch_size, c, D, H, W), np.float32)
        dense[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]] = feat
        maps.append(dense.reshape(batch_size, c * D, H, W))
    return np.concatenate(maps, 1)
