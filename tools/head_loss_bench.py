#!/usr/bin/env python
"""Times TransFusionHead.loss (row f3, training half) at the LC config's size -- 200
proposals, 10 classes, 180 x 180 heat map, ~50 boxes per sample -- and the pieces under it:

  * the batched path (one IoU launch, one host read, one painter launch, fused heat-map loss);
  * the reference's structure rebuilt from the same parts: per-sample assign() (its own IoU
    launch, cost matrix and device->host copy per sample), a python loop over the boxes that
    paints one Gaussian at a time with torch ops, the heat-map loss as separate torch ops
    with .item() for the normaliser;
  * kernels alone: IoU, painter, fused heat-map loss against its torch composition.

    python tools/head_loss_bench.py [batch ...]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from msmdfusion_amd import configs as C  # noqa: E402
from msmdfusion_amd import head_loss as HL  # noqa: E402
from msmdfusion_amd import kernels as K  # noqa: E402


def make_case(rs, B, P, dev, boxes_per_sample=50):
    gts, labs = [], []
    for _ in range(B):
        g = boxes_per_sample + rs.randint(-10, 11)
        b = np.zeros((g, 9), np.float32)
        b[:, 0:2] = rs.uniform(-50, 50, (g, 2))
        b[:, 2] = rs.uniform(-2, 0, g)
        b[:, 3:6] = rs.uniform(0.5, 5, (g, 3))
        b[:, 6] = rs.uniform(-3.1, 3.1, g)
        gts.append(HL.LiDARBoxes(torch.from_numpy(b)))
        labs.append(torch.from_numpy(rs.randint(0, 10, g).astype(np.int64)))
    pred = dict(heatmap=rs.standard_normal((B, 10, P)) * 2, center=rs.uniform(0, 180, (B, 2, P)),
                height=rs.uniform(-2, 1, (B, 1, P)), dim=rs.uniform(-0.5, 1.6, (B, 3, P)),
                rot=rs.uniform(-1, 1, (B, 2, P)), vel=rs.standard_normal((B, 2, P)),
                dense_heatmap=rs.standard_normal((B, 10, 180, 180)) * 3)
    pred = {k: torch.from_numpy(v.astype(np.float32)).to(dev).requires_grad_(True)
            for k, v in pred.items()}
    return pred, gts, labs


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))


def loop_targets(head, gts, labs, pred):
    """get_targets_single's structure (transfusion_head.py:1092-1219), one sample at a time."""
    cfg, coder = head.train_cfg, head.bbox_coder
    dev = pred["heatmap"].device
    P = head.num_proposals
    out = []
    for b in range(len(gts)):
        one = {k: v[b:b + 1].detach() for k, v in pred.items()}
        dec = coder.decode(one["heatmap"], one["rot"], one["dim"], one["center"], one["height"],
                           one["vel"])[0]["bboxes"]
        gt = gts[b].tensor.to(dev)
        lab = labs[b].to(dev)
        ar = head.bbox_assigner.assign(dec[:P], gt, lab, one["heatmap"][..., :P], cfg)
        pos = torch.nonzero(ar.gt_inds > 0).squeeze(-1)
        tgt = dec.new_zeros((P, coder.code_size))
        tgt[pos] = HL.encode_boxes(gt[ar.gt_inds[pos] - 1], coder.pc_range, coder.out_size_factor,
                                   coder.voxel_size, coder.code_size)
        heat = dec.new_zeros((head.num_classes, 180, 180))
        for i in range(gt.shape[0]):                           # the per-box painter
            w = gt[i, 3] / cfg["voxel_size"][0] / cfg["out_size_factor"]
            l = gt[i, 4] / cfg["voxel_size"][1] / cfg["out_size_factor"]
            if w > 0 and l > 0:
                r = max(cfg["min_radius"], int(HL.gaussian_radius(l, w, cfg["gaussian_overlap"])))
                x = int((gt[i, 0] - cfg["point_cloud_range"][0]) / cfg["voxel_size"][0] / 8)
                y = int((gt[i, 1] - cfg["point_cloud_range"][1]) / cfg["voxel_size"][1] / 8)
                ys, xs = np.ogrid[-r:r + 1, -r:r + 1]
                g = torch.from_numpy(np.exp(-(xs * xs + ys * ys) / (2 * ((2 * r + 1) / 6) ** 2))
                                     ).to(dev, torch.float32)
                l0, r0, t0, b0 = min(x, r), min(180 - x, r + 1), min(y, r), min(180 - y, r + 1)
                view = heat[int(lab[i]), y - t0:y + b0, x - l0:x + r0]
                if min(view.shape) > 0:
                    torch.max(view, g[r - t0:r + b0, r - l0:r + r0], out=view)
        out.append((ar, tgt, heat))
    return out


def main():
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(0)
    batches = [int(a) for a in sys.argv[1:]] or [2, 4]
    for B in batches:
        head = C.build_head(C.MSMDFUSION_LC).to(dev)
        pred, gts, labs = make_case(rs, B, head.num_proposals, dev)
        n_gt = sum(len(g) for g in gts)

        def step():
            for v in pred.values():
                v.grad = None
            losses = head.loss(gts, labs, ([pred],))
            sum(v for k, v in losses.items() if "loss" in k).backward()
        t_loss = timed(step)
        t_targets = timed(lambda: head.get_targets(gts, labs, [pred]))
        t_loop = timed(lambda: loop_targets(head, gts, labs, pred), reps=5)

        gt, lab, counts = HL._pad_ground_truth(gts, labs, dev)
        dec = torch.stack([d["bboxes"] for d in head.bbox_coder.decode(
            pred["heatmap"].detach(), pred["rot"].detach(), pred["dim"].detach(),
            pred["center"].detach(), pred["height"].detach(), pred["vel"].detach())])
        nbv = torch.tensor(counts, dtype=torch.int32, device=dev)
        t_iou = timed(lambda: K.boxes_iou3d(dec, gt, nb_valid=nbv))
        flat = gt.reshape(-1, gt.shape[-1])
        cx, cy, rad = HL.heatmap_boxes(flat[:, 0:2], flat[:, 3:5], head.train_cfg)
        plane = (torch.arange(B, device=dev)[:, None] * 10 + lab).reshape(-1).int()
        heat = torch.zeros((B, 10, 180, 180), device=dev)
        t_paint = timed(lambda: K.heatmap_gaussian(heat, plane, cx, cy, rad))
        x = pred["dense_heatmap"].detach().clone().requires_grad_(True)

        def fused():
            x.grad = None
            head.loss_heatmap.from_logits(x, heat).backward()

        def composed():
            x.grad = None
            avg = max(heat.eq(1).float().sum().item(), 1)
            head.loss_heatmap(HL.clip_sigmoid(x), heat, avg_factor=avg).backward()
        t_fused, t_comp = timed(fused), timed(composed)
        print("B=%d  %d boxes: loss fwd+bwd %.2f ms | get_targets batched %.2f ms, per-sample/"
              "per-box structure %.1f ms | iou3d %.3f ms (%d pairs), painter %.3f ms, heat-map "
              "loss fused %.3f ms vs torch ops %.3f ms"
              % (B, n_gt, t_loss, t_targets, t_loop, t_iou, dec.shape[1] * n_gt, t_paint, t_fused,
                 t_comp), flush=True)


if __name__ == "__main__":
    main()
