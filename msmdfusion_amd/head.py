"""TransFusionHead, LiDAR branch: the consumer of the dense BEV tail.

Reference: mmdet3d/models/dense_heads/transfusion_head.py -- PositionEmbeddingLearned
(:25-41), TransformerDecoderLayer (:44-122), MultiheadAttention (:125-504, a copy of
torch.nn.MultiheadAttention's parameters and arithmetic), FFN (:507-591), TransFusionHead
(:594-1379); box decoding: mmdet3d/core/bbox/coders/transfusion_bbox_coder.py:8-130.
Built here: the path the LC config takes (`fuse_img` unset, `initialize_by_heatmap=True`,
one decoder layer, configs/MSMDFusion_nusc_voxel_LC.py:207-241) -- forward_single's
heatmap-initialised queries, the decoder layers over the flattened BEV map, the prediction
heads, and get_bboxes' score composition + box decoding without NMS (`nms_type=None`).
The training half -- get_targets (HungarianAssigner3D, heat-map targets) and loss
(:1051-1286) -- lives in msmdfusion_amd/head_loss.py and is reached through this class's
`get_targets` / `loss`.  Not built: the image-fusion decoder stages (`fuse_img=True`), the
HeuristicAssigner and NMS variants the configs do not use.

Parameter names and shapes are the reference's (`shared_conv.weight`,
`heatmap_head.0.conv.weight`, `decoder.0.self_attn.in_proj_weight`,
`prediction_heads.0.center.0.conv.weight`, ...), so `pts_bbox_head.*` checkpoint keys
load unchanged.  Everything is torch / rocBLAS work except the 512 -> 128 3x3
`shared_conv`, which can run on the sparse-conv row kernels (`rows=True`) when its input
is the channels-last map the row tail produces.
"""
import copy
import os

import torch
from torch import nn
from torch.nn import functional as F


class Conv1d(nn.Conv1d):
    """nn.Conv1d (same parameters, same state-dict keys); a kernel-size-1 convolution runs as
    the matrix product it is.  MIOpen serves these through its 3x3 Winograd kernels --
    0.6 ms per launch on the 32 400 key positions of the LC map, 32 launches (20 ms) per
    training step for the head's position embeddings and prediction heads; rocBLAS needs
    ~30 us for the same product."""

    def forward(self, x):
        if self.kernel_size != (1,) or self.stride != (1,) or self.groups != 1 or x.dim() != 3:
            return super().forward(x)
        y = torch.matmul(self.weight[:, :, 0], x)
        return y if self.bias is None else y + self.bias[:, None]


class ConvModule(nn.Module):
    """The slice of mmcv.cnn.ConvModule the head uses: conv -> norm -> ReLU with the
    attribute names `conv` / `bn` (checkpoint keys), bias='auto' = no conv bias under a
    norm."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias="auto",
                 conv="Conv1d", norm="BN1d"):
        super().__init__()
        conv_cls = {"Conv1d": Conv1d, "Conv2d": nn.Conv2d}[conv]
        norm_cls = {"BN1d": nn.BatchNorm1d, "BN2d": nn.BatchNorm2d, None: None}[norm]
        if bias == "auto":
            bias = norm_cls is None
        self.conv = conv_cls(in_channels, out_channels, kernel_size, stride=stride,
                             padding=padding, bias=bool(bias))
        self.bn = norm_cls(out_channels) if norm_cls is not None else None
        self.activate = nn.ReLU(inplace=True)

    def forward(self, x):
        x = self.conv(x)
        if self.bn is not None:
            x = self.bn(x)
        return self.activate(x)


class PositionEmbeddingLearned(nn.Module):
    """:25-41 -- Conv1d(k=1) + BN1d + ReLU + Conv1d(k=1) over [B, P, 2] positions."""

    def __init__(self, input_channel, num_pos_feats=288):
        super().__init__()
        self.position_embedding_head = nn.Sequential(
            Conv1d(input_channel, num_pos_feats, kernel_size=1),
            nn.BatchNorm1d(num_pos_feats), nn.ReLU(inplace=True),
            Conv1d(num_pos_feats, num_pos_feats, kernel_size=1))

    def forward(self, xyz):
        return self.position_embedding_head(xyz.transpose(1, 2).contiguous())


class TransformerDecoderLayer(nn.Module):
    """:44-122 -- post-norm decoder layer: self-attention over the queries, cross-attention
    into the keys (positions added to queries, keys AND values), FFN.  Tensors come and go
    as [B, C, P]; nn.MultiheadAttention carries the same parameters as the reference's
    private copy of it."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu",
                 self_posembed=None, cross_posembed=None, cross_only=False):
        super().__init__()
        self.cross_only = cross_only
        if not cross_only:
            self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.multihead_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1, self.norm2, self.norm3 = (nn.LayerNorm(d_model) for _ in range(3))
        self.dropout1, self.dropout2, self.dropout3 = (nn.Dropout(dropout) for _ in range(3))
        self.activation = {"relu": F.relu, "gelu": F.gelu, "glu": F.glu}[activation]
        self.self_posembed, self.cross_posembed = self_posembed, cross_posembed

    def forward(self, query, key, query_pos, key_pos, attn_mask=None):
        q_pos = self.self_posembed(query_pos).permute(2, 0, 1) \
            if self.self_posembed is not None else None
        k_pos = self.cross_posembed(key_pos).permute(2, 0, 1) \
            if self.cross_posembed is not None else None
        query, key = query.permute(2, 0, 1), key.permute(2, 0, 1)       # [P, B, C]

        def plus(t, p):
            return t if p is None else t + p
        if not self.cross_only:
            q = plus(query, q_pos)
            query = self.norm1(query + self.dropout1(self.self_attn(q, q, value=q)[0]))
        kv = plus(key, k_pos)
        attended = self.multihead_attn(query=plus(query, q_pos), key=kv, value=kv,
                                       attn_mask=attn_mask)[0]
        query = self.norm2(query + self.dropout2(attended))
        ffn = self.linear2(self.dropout(self.activation(self.linear1(query))))
        query = self.norm3(query + self.dropout3(ffn))
        return query.permute(1, 2, 0)


class FFN(nn.Module):
    """:507-591 -- one small Conv1d stack per predicted quantity:
    heads = {name: (channels, num_conv)}; the heatmap head's last bias starts at -2.19."""

    def __init__(self, in_channels, heads, head_conv=64, final_kernel=1, init_bias=-2.19,
                 bias="auto"):
        super().__init__()
        self.heads = heads
        self.init_bias = init_bias
        for name, (classes, num_conv) in heads.items():
            layers, c_in = [], in_channels
            for _ in range(num_conv - 1):
                layers.append(ConvModule(c_in, head_conv, final_kernel, 1, final_kernel // 2,
                                         bias=bias))
                c_in = head_conv
            layers.append(Conv1d(head_conv, classes, final_kernel, 1, final_kernel // 2,
                                 bias=True))
            setattr(self, name, nn.Sequential(*layers))

    def init_weights(self):
        for name in self.heads:
            if name == "heatmap":
                getattr(self, name)[-1].bias.data.fill_(self.init_bias)

    def forward(self, x):
        return {name: getattr(self, name)(x) for name in self.heads}


class TransFusionBBoxCoder:
    """transfusion_bbox_coder.py:8-130, decode half.  Inputs are NOT modified (the
    reference scales `center` and exponentiates `dim` in place)."""

    def __init__(self, pc_range, out_size_factor, voxel_size, post_center_range=None,
                 score_threshold=None, code_size=8):
        self.pc_range, self.out_size_factor, self.voxel_size = pc_range, out_size_factor, voxel_size
        self.post_center_range, self.score_threshold = post_center_range, score_threshold
        self.code_size = code_size

    def decode(self, heatmap, rot, dim, center, height, vel, filter=False):
        labels = heatmap.max(1).indices
        scores = heatmap.max(1).values
        scale = self.out_size_factor
        cx = center[:, 0:1] * scale * self.voxel_size[0] + self.pc_range[0]
        cy = center[:, 1:2] * scale * self.voxel_size[1] + self.pc_range[1]
        dim = dim.exp()
        height = height - dim[:, 2:3] * 0.5            # gravity centre -> bottom centre
        yaw = torch.atan2(rot[:, 0:1], rot[:, 1:2])
        parts = [cx, cy, height, dim, yaw] + ([vel] if vel is not None else [])
        boxes = torch.cat(parts, dim=1).permute(0, 2, 1)
        if not filter:
            return [dict(bboxes=boxes[i], scores=scores[i], labels=labels[i])
                    for i in range(heatmap.shape[0])]
        if self.post_center_range is None:
            raise NotImplementedError("Need to reorganize output as a batch, only support "
                                      "post_center_range is not None for now!")
        rng = torch.as_tensor(self.post_center_range, device=heatmap.device, dtype=boxes.dtype)
        mask = (boxes[..., :3] >= rng[:3]).all(2) & (boxes[..., :3] <= rng[3:]).all(2)
        if self.score_threshold:                                    # (0.0 filters nothing: :117)
            mask &= scores > self.score_threshold
        return [dict(bboxes=boxes[i, mask[i]], scores=scores[i, mask[i]], labels=labels[i, mask[i]])
                for i in range(heatmap.shape[0])]


class TransFusionHead(nn.Module):
    """The LiDAR branch of transfusion_head.py:594-1379 (see the module docstring)."""

    def __init__(self, num_proposals=128, auxiliary=True, in_channels=128 * 3, hidden_channel=128,
                 num_classes=4, num_decoder_layers=3, num_heads=8, learnable_query_pos=False,
                 initialize_by_heatmap=False, nms_kernel_size=1, ffn_channel=256, dropout=0.1,
                 bn_momentum=0.1, activation="relu", common_heads=None, num_heatmap_convs=2,
                 bias="auto", bbox_coder=None, test_cfg=None, train_cfg=None, fuse_img=False,
                 loss_cls=None, loss_bbox=None, loss_heatmap=None, rows=False, **unused):
        super().__init__()
        if fuse_img:
            raise NotImplementedError("TransFusionHead: the image-fusion stages are not built "
                                      "(the LC config leaves fuse_img unset)")
        if initialize_by_heatmap and learnable_query_pos:
            raise ValueError("initialized by heatmap is conflicting with learnable query position")
        self.num_classes, self.num_proposals, self.auxiliary = num_classes, num_proposals, auxiliary
        self.num_decoder_layers, self.bn_momentum = num_decoder_layers, bn_momentum
        self.initialize_by_heatmap, self.nms_kernel_size = initialize_by_heatmap, nms_kernel_size
        self.test_cfg, self.train_cfg, self.rows = test_cfg, train_cfg, rows
        self.bbox_coder = TransFusionBBoxCoder(**{k: v for k, v in (bbox_coder or {}).items()
                                                  if k != "type"}) if bbox_coder else None
        self.shared_conv = nn.Conv2d(in_channels, hidden_channel, 3, padding=1, bias=bool(bias))
        if initialize_by_heatmap:
            self.heatmap_head = nn.Sequential(
                ConvModule(hidden_channel, hidden_channel, 3, padding=1, bias=bias, conv="Conv2d",
                           norm="BN2d"),
                nn.Conv2d(hidden_channel, num_classes, 3, padding=1, bias=bool(bias)))
            self.class_encoding = Conv1d(num_classes, hidden_channel, 1)
        else:
            self.query_feat = nn.Parameter(torch.randn(1, hidden_channel, num_proposals))
            self.query_pos = nn.Parameter(torch.rand([1, num_proposals, 2]),
                                          requires_grad=learnable_query_pos)
        self.decoder = nn.ModuleList([
            TransformerDecoderLayer(hidden_channel, num_heads, ffn_channel, dropout, activation,
                                    self_posembed=PositionEmbeddingLearned(2, hidden_channel),
                                    cross_posembed=PositionEmbeddingLearned(2, hidden_channel))
            for _ in range(num_decoder_layers)])
        self.prediction_heads = nn.ModuleList()
        for _ in range(num_decoder_layers):
            heads = copy.deepcopy(common_heads or {})
            heads.update(dict(heatmap=(num_classes, num_heatmap_convs)))
            self.prediction_heads.append(FFN(hidden_channel, heads, bias=bias))
        self.init_weights()
        x_size = test_cfg["grid_size"][0] // test_cfg["out_size_factor"]
        y_size = test_cfg["grid_size"][1] // test_cfg["out_size_factor"]
        self.bev_pos = self.create_2D_grid(x_size, y_size)
        self._bev_pos_cache = {}
        self.query_labels = None
        from . import head_loss as HL
        self.loss_cls = HL.build_loss(loss_cls or dict(type="FocalLoss"))        # :609-611
        self.loss_bbox = HL.build_loss(loss_bbox or dict(type="L1Loss"))
        self.loss_heatmap = HL.build_loss(loss_heatmap or dict(type="GaussianFocalLoss"))
        self.bbox_assigner = None                        # _init_assigner_sampler (:778-792);
        if train_cfg is not None:                        # the sampler is PseudoSampler: none
            self.bbox_assigner = HL.build_assigner(train_cfg["assigner"])
        self.heatmap_painter = HL.HeatmapPainter()

    @staticmethod
    def create_2D_grid(x_size, y_size):
        """:755-762 -- cell centres (x + 0.5, y + 0.5), flattened y-major like the map."""
        by, bx = torch.meshgrid(torch.linspace(0, x_size - 1, x_size),
                                torch.linspace(0, y_size - 1, y_size), indexing="ij")
        coord = torch.cat([(bx + 0.5)[None], (by + 0.5)[None]], dim=0)[None]
        return coord.view(1, 2, -1).permute(0, 2, 1)

    def init_weights(self):
        for p in self.decoder.parameters():                      # :764-770
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
                m.momentum = self.bn_momentum

    def _bev_pos_on(self, device, batch):
        """bev_pos.repeat(B, 1, 1).to(device) (:800) without its per-call host work: the
        reference repeats 64 800 floats on the CPU and copies them from pageable memory --
        a blocking copy that drains the stream -- in every forward."""
        key = (device, batch)
        if self._bev_pos_cache.get("key") != key:
            self._bev_pos_cache = {"key": key,
                                   "value": self.bev_pos.to(device).repeat(batch, 1, 1)}
        return self._bev_pos_cache["value"]

    def _shared_conv(self, inputs):
        """-> (feature map [B, C, H, W] contiguous, its pixel rows | None, grid | None)"""
        if self.rows and inputs.is_cuda:
            from .grid_conv import grid_conv2d, map_of, rows_of
            rows, grid = rows_of(inputs)
            y, grid = grid_conv2d(rows.contiguous(), grid, self.shared_conv.weight, 1, 1, 1)
            if self.shared_conv.bias is not None:
                y = y + self.shared_conv.bias
            return map_of(y, grid).contiguous(), y, grid
        return self.shared_conv(inputs), None, None

    def _heatmap(self, lidar_feat, rows, grid):
        """heatmap_head (:686-692): ConvModule 3x3 + Conv2d 3x3 -> class heat-map logits.  On
        the row path both convolutions run on the sparse-conv kernels (the class conv with
        its output channels zero-padded to the kernels' narrowest width, 32): MIOpen's fp32
        Winograd kernels need 11 ms per training step for this pair at 180 x 180."""
        if rows is None or os.environ.get("MSMD_HEAD_HEATMAP_ROWS", "1") != "1":
            return self.heatmap_head(lidar_feat)
        from .grid_conv import grid_conv2d, map_of
        from .spconv import functional as Fsp
        block, last = self.heatmap_head[0], self.heatmap_head[1]
        y, grid = grid_conv2d(rows, grid, block.conv.weight, 1, 1, 1)
        if block.conv.bias is not None:
            y = y + block.conv.bias
        y = Fsp.bn_act(y, block.bn, relu=True)
        c = last.weight.shape[0]
        width = max(32, (c + 3) // 4 * 4)
        z, grid = grid_conv2d(y, grid, F.pad(last.weight, (0, 0, 0, 0, 0, 0, 0, width - c)), 1, 1, 1)
        z = z[:, :c]
        if last.bias is not None:
            z = z + last.bias
        return map_of(z.contiguous(), grid).contiguous()

    def forward_single(self, inputs):
        """:795-1027, fuse_img False.  inputs [B, C, H, W] -> [dict] with center, height,
        dim, rot, vel, heatmap ([B, *, num_proposals * layers] when auxiliary) and, for
        heatmap-initialised queries, query_heatmap_score and dense_heatmap."""
        B = inputs.shape[0]
        lidar_feat, rows, grid = self._shared_conv(inputs)
        flat = lidar_feat.reshape(B, lidar_feat.shape[1], -1)
        bev_pos = self._bev_pos_on(lidar_feat.device, B)
        if self.initialize_by_heatmap:
            dense_heatmap = self._heatmap(lidar_feat, rows, grid)
            heatmap = dense_heatmap.detach().sigmoid()
            pad = self.nms_kernel_size // 2
            local_max = torch.zeros_like(heatmap)
            inner = F.max_pool2d(heatmap, kernel_size=self.nms_kernel_size, stride=1, padding=0)
            local_max[:, :, pad:heatmap.shape[2] - pad, pad:heatmap.shape[3] - pad] = inner
            dataset = self.test_cfg["dataset"]
            keep = {"nuScenes": (8, 9), "Waymo": (1, 2)}.get(dataset, ())
            for c in keep:                     # small classes: every cell is its own maximum
                local_max[:, c] = heatmap[:, c]
            heatmap = (heatmap * (heatmap == local_max)).reshape(B, heatmap.shape[1], -1)
            top = heatmap.reshape(B, -1).argsort(dim=-1, descending=True)[..., :self.num_proposals]
            top_class = top // heatmap.shape[-1]
            top_index = top % heatmap.shape[-1]
            query_feat = flat.gather(index=top_index[:, None, :].expand(-1, flat.shape[1], -1),
                                     dim=-1)
            self.query_labels = top_class
            one_hot = F.one_hot(top_class, num_classes=self.num_classes).permute(0, 2, 1)
            query_feat = query_feat + self.class_encoding(one_hot.float())
            query_pos = bev_pos.gather(
                index=top_index[:, None, :].permute(0, 2, 1).expand(-1, -1, bev_pos.shape[-1]), dim=1)
        else:
            query_feat = self.query_feat.repeat(B, 1, 1)
            query_pos = self.query_pos.repeat(B, 1, 1).to(lidar_feat.device)
        ret = []
        for i in range(self.num_decoder_layers):
            query_feat = self.decoder[i](query_feat, flat, query_pos, bev_pos)
            res = self.prediction_heads[i](query_feat)
            res["center"] = res["center"] + query_pos.permute(0, 2, 1)
            ret.append(res)
            query_pos = res["center"].detach().clone().permute(0, 2, 1)
        if self.initialize_by_heatmap:
            ret[0]["query_heatmap_score"] = heatmap.gather(
                index=top_index[:, None, :].expand(-1, self.num_classes, -1), dim=-1)
            ret[0]["dense_heatmap"] = dense_heatmap
        if not self.auxiliary:
            return [ret[-1]]
        keep_first = ("dense_heatmap", "dense_heatmap_old", "query_heatmap_score")
        return [{k: (ret[0][k] if k in keep_first else torch.cat([r[k] for r in ret], dim=-1))
                 for k in ret[0]}]

    def forward(self, feats, img_feats=None, img_metas=None):
        """:1029-1047 -- multi_apply over feature levels: a tuple with, per result slot of
        forward_single, the list over levels (one level, one slot here: `([dict],)`)."""
        if isinstance(feats, torch.Tensor):
            feats = [feats]
        res = tuple(map(list, zip(*[self.forward_single(x) for x in feats])))
        assert len(res) == 1, "only support one level features."
        return res

    def get_targets(self, gt_bboxes_3d, gt_labels_3d, preds_dict):
        """:1051-1090 (whole batch at once, head_loss.get_targets)."""
        from . import head_loss as HL
        return HL.get_targets(self, gt_bboxes_3d, gt_labels_3d, preds_dict)

    def loss(self, gt_bboxes_3d, gt_labels_3d, preds_dicts, **kwargs):
        """:1221-1286 -> dict(loss_heatmap, layer_-1_loss_cls, layer_-1_loss_bbox, ...,
        matched_ious)."""
        if self.train_cfg is None or self.bbox_assigner is None:
            raise RuntimeError("TransFusionHead.loss needs train_cfg (assigner, code_weights, ...)")
        from . import head_loss as HL
        return HL.loss(self, gt_bboxes_3d, gt_labels_3d, preds_dicts)

    def get_bboxes(self, preds_dicts):
        """:1285-1379 with nms_type None: score = sigmoid(heatmap) * query heatmap score *
        one-hot(query class) over the LAST layer's proposals, boxes decoded and filtered by
        post_center_range / score_threshold.  -> per sample dict(bboxes [n, code], scores,
        labels); the reference additionally wraps bboxes in its box class."""
        if self.test_cfg.get("nms_type") is not None:
            raise NotImplementedError("circle / rotated NMS after decoding is not built "
                                      "(the LC config runs with nms_type=None)")
        (pred,) = preds_dicts[0] if isinstance(preds_dicts[0], (list, tuple)) else (preds_dicts[0],)
        n = self.num_proposals
        score = pred["heatmap"][..., -n:].sigmoid()
        one_hot = F.one_hot(self.query_labels, num_classes=self.num_classes).permute(0, 2, 1)
        score = score * pred["query_heatmap_score"] * one_hot
        vel = pred["vel"][..., -n:] if "vel" in pred else None
        return self.bbox_coder.decode(score, pred["rot"][..., -n:], pred["dim"][..., -n:],
                                      pred["center"][..., -n:], pred["height"][..., -n:], vel,
                                      filter=True)
