#!/usr/bin/env python
"""Generates tests/golden/head_vectors.npz by RUNNING the reference's own code for row f3:

    PositionEmbeddingLearned, TransformerDecoderLayer, MultiheadAttention (+ its
    multi_head_attention_forward), FFN, TransFusionHead.forward_single / create_2D_grid
        mmdet3d/models/dense_heads/transfusion_head.py:25-591, 755-1027
    TransFusionBBoxCoder.decode     mmdet3d/core/bbox/coders/transfusion_bbox_coder.py:41-130

mmcv / mmdet are absent, so the definitions are taken from the reference FILES at run time
(ast) and executed as they stand.  The names they import from those packages get inert or
minimal stand-ins -- registry decorators that return the class, `build_conv_layer` = the
torch layer of that name, `ConvModule` = conv -> norm -> ReLU with mmcv's attribute names --
and the head object is assembled by hand (its __init__ needs the loss / assigner builders):
the layers are built from the reference's OWN classes, then forward_single and decode run on
seeded inputs.  Parameters are set from their names (synthetic.seeded_parameters), which the
reimplementation shares, so no weights are stored -- only the outputs.
"""
import ast
import copy
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from msmdfusion_amd import synthetic as S  # noqa: E402

HEAD = "/root/reference/mmdet3d/models/dense_heads/transfusion_head.py"
CODER = "/root/reference/mmdet3d/core/bbox/coders/transfusion_bbox_coder.py"
OUT = os.path.join(ROOT, "tests", "golden", "head_vectors.npz")

CFG = dict(num_proposals=24, in_channels=32, hidden_channel=32, num_classes=10,
           num_decoder_layers=2, num_heads=4, nms_kernel_size=3, ffn_channel=48,
           common_heads=dict(center=(2, 2), height=(1, 2), dim=(3, 2), rot=(2, 2), vel=(2, 2)),
           grid=(20, 20))
CODER_CFG = dict(pc_range=[-54.0, -54.0], out_size_factor=8, voxel_size=[0.075, 0.075],
                 post_center_range=[-61.2, -61.2, -10.0, 61.2, 61.2, 10.0], score_threshold=0.0,
                 code_size=10)


class _Registry:
    def register_module(self):
        return lambda cls: cls


class _ConvModule(nn.Module):      # mmcv.cnn.ConvModule: conv / bn / activate, bias='auto'
    def __init__(self, cin, cout, kernel_size, stride=1, padding=0, bias="auto", conv_cfg=None,
                 norm_cfg=None):
        super().__init__()
        conv = {"Conv1d": nn.Conv1d, "Conv2d": nn.Conv2d}[conv_cfg["type"]]
        norm = {"BN1d": nn.BatchNorm1d, "BN2d": nn.BatchNorm2d}[norm_cfg["type"]]
        self.conv = conv(cin, cout, kernel_size, stride=stride, padding=padding,
                         bias=False if bias == "auto" else bool(bias))
        self.bn = norm(cout)
        self.activate = nn.ReLU(inplace=True)

    def forward(self, x):
        return self.activate(self.bn(self.conv(x)))


def _build_conv_layer(cfg, *args, **kw):
    return {"Conv1d": nn.Conv1d, "Conv2d": nn.Conv2d}[cfg["type"]](*args, **kw)


def reference_namespace():
    ns = {"torch": torch, "nn": nn, "F": F, "np": np, "copy": copy,
          "Parameter": nn.Parameter, "Linear": nn.Linear,
          "xavier_uniform_": nn.init.xavier_uniform_, "constant_": nn.init.constant_,
          "ConvModule": _ConvModule, "build_conv_layer": _build_conv_layer,
          "kaiming_init": lambda m: None, "HEADS": _Registry(), "BBOX_CODERS": _Registry(),
          "BaseBBoxCoder": object, "force_fp32": lambda **kw: (lambda f: f)}
    tree = ast.parse(open(HEAD).read())
    want = {"PositionEmbeddingLearned", "TransformerDecoderLayer", "MultiheadAttention",
            "multi_head_attention_forward", "FFN"}
    body = [n for n in tree.body if isinstance(n, (ast.ClassDef, ast.FunctionDef))
            and n.name in want]
    head = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "TransFusionHead")
    body += [n for n in head.body if isinstance(n, ast.FunctionDef)
             and n.name in ("forward_single", "create_2D_grid")]
    exec(compile(ast.Module(body=body, type_ignores=[]), HEAD, "exec"), ns)
    coder = ast.parse(open(CODER).read())
    exec(compile(ast.Module(body=[n for n in coder.body if isinstance(n, ast.ClassDef)],
                            type_ignores=[]), CODER, "exec"), ns)
    return ns


def build_reference_head(ns):
    """What TransFusionHead.__init__ (:664-706) assembles, from the reference's classes."""
    c = CFG
    head = nn.Module()
    head.num_classes, head.num_proposals, head.auxiliary = c["num_classes"], c["num_proposals"], True
    head.num_decoder_layers, head.fuse_img = c["num_decoder_layers"], False
    head.initialize_by_heatmap, head.nms_kernel_size = True, c["nms_kernel_size"]
    head.test_cfg = dict(dataset="nuScenes", grid_size=[c["grid"][0] * 8, c["grid"][1] * 8, 40],
                         out_size_factor=8)
    hid = c["hidden_channel"]
    head.shared_conv = nn.Conv2d(c["in_channels"], hid, 3, padding=1, bias=True)
    head.heatmap_head = nn.Sequential(
        _ConvModule(hid, hid, 3, padding=1, bias="auto", conv_cfg=dict(type="Conv2d"),
                    norm_cfg=dict(type="BN2d")),
        nn.Conv2d(hid, c["num_classes"], 3, padding=1, bias=True))
    head.class_encoding = nn.Conv1d(c["num_classes"], hid, 1)
    head.decoder = nn.ModuleList([
        ns["TransformerDecoderLayer"](hid, c["num_heads"], c["ffn_channel"], 0.1, "relu",
                                      self_posembed=ns["PositionEmbeddingLearned"](2, hid),
                                      cross_posembed=ns["PositionEmbeddingLearned"](2, hid))
        for _ in range(c["num_decoder_layers"])])
    head.prediction_heads = nn.ModuleList()
    for _ in range(c["num_decoder_layers"]):
        heads = copy.deepcopy(c["common_heads"])
        heads.update(dict(heatmap=(c["num_classes"], 2)))
        head.prediction_heads.append(ns["FFN"](hid, heads, conv_cfg=dict(type="Conv1d"),
                                               norm_cfg=dict(type="BN1d"), bias="auto"))
    head.create_2D_grid = lambda x, y: ns["create_2D_grid"](head, x, y)
    head.bev_pos = head.create_2D_grid(*c["grid"])
    return head


def main():
    ns = reference_namespace()
    head = S.seeded_parameters(build_reference_head(ns), seed=21).eval()
    x = torch.from_numpy(np.random.RandomState(22).standard_normal(
        (2, CFG["in_channels"], *CFG["grid"])).astype(np.float32))
    with torch.no_grad():
        (res,) = ns["forward_single"](head, x, None, None)
    out = {"fs_" + k: v.numpy() for k, v in res.items()}
    out["query_labels"] = head.query_labels.numpy()
    out["state_dict_keys"] = np.array(sorted(head.state_dict().keys()))
    out["bev_pos"] = head.bev_pos.numpy()
    # get_bboxes' score composition (:1299-1304) + the reference coder's decode
    n = CFG["num_proposals"]
    coder = ns["TransFusionBBoxCoder"](**CODER_CFG)
    score = res["heatmap"][..., -n:].sigmoid()
    one_hot = F.one_hot(head.query_labels, num_classes=CFG["num_classes"]).permute(0, 2, 1)
    score = score * res["query_heatmap_score"] * one_hot
    dec = coder.decode(score.clone(), res["rot"][..., -n:].clone(), res["dim"][..., -n:].clone(),
                       res["center"][..., -n:].clone(), res["height"][..., -n:].clone(),
                       res["vel"][..., -n:].clone(), filter=True)
    for i, d in enumerate(dec):
        for k, v in d.items():
            out["dec_%d_%s" % (i, k)] = v.numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
