"""Dense BEV tail of extract_pts_feat: bev_fusion (SPPModule,
MSMDFusion.py:47-90) -> pts_backbone (SECOND, backbones/second.py:9-88) ->
pts_neck (SECONDFPN, necks/second_fpn.py:10-93), with the reference's
attribute names, so its checkpoints load key for key
(`bev_fusion.conv3x3.0.weight`, `pts_backbone.blocks.1.3.weight`,
`pts_neck.deblocks.1.0.weight`, ...).

The classes here are the torch / MIOpen form of the three modules.  Measured on MI355X,
B=2, [2,640,180,180] input, forward+backward (tools/bev_tail_bench.py,
profiles/r02_bev_tail.txt; 0.87 TFLOP forward):

    MIOpen fp32  NCHW 30.0 ms  NHWC 34.7 ms     MIOpen bf16 autocast  13.3 / 13.5 ms
    msmdfusion_amd.grid_conv (same modules on pixel rows, sparse-conv kernels,
    fp32-equivalent)                            19.7 ms   (SPP block alone: 12.2 against 21.5)

Channels-last does not pay with this MIOpen build; what does is running these dense
convolutions on the sparse-conv kernels (grid_conv.py: a dense channels-last map is a sparse
tensor with every cell active, its neighbour table is arithmetic on the pixel index), which
`configs.build_bev_tail` therefore builds by default -- `SPPModuleRows`, `SECONDRows`,
`SECONDFPNRows` subclass the modules below and share their parameters and checkpoint keys.
The sparse side hands them ONE [B,H,W,640] buffer that both sparse tensors scatter into
(spconv.functional.bev_concat: 0.05 ms against 0.24 ms for dense() + view + cat).
"""
import torch
from torch import nn

from .registry import BACKBONES, NECKS, build_conv_layer, build_norm_layer

_BN = dict(type="BN", eps=1e-3, momentum=0.01)


def _conv_bn_relu(cin, cout, k, stride=1, padding=0, dilation=1, norm_cfg=_BN,
                  conv_cfg=None, inplace=False):
    conv_cfg = dict(type="Conv2d", bias=False) if conv_cfg is None else conv_cfg
    return [build_conv_layer(conv_cfg, cin, cout, k, stride=stride, padding=padding,
                             dilation=dilation),
            build_norm_layer(norm_cfg, cout)[1],
            nn.ReLU(inplace=inplace)]


class SPPModule(nn.Module):
    """Four parallel 640->256 branches (1x1, 3x3, 3x3 dilation 6, 3x3 dilation 12)
    and a 1024->256 1x1 fuse; every conv bias-free + BN(eps 1e-3, momentum 0.01)
    + ReLU (MSMDFusion.py:47-90)."""
    BRANCHES = (("conv1x1", 1, 0, 1), ("conv3x3", 3, 1, 1),
                ("dilated_conv3x3_rate6", 3, 6, 6), ("dilated_conv3x3_rate12", 3, 12, 12))

    def __init__(self, in_channels=384 + 256, channels=256):
        super().__init__()
        for name, k, pad, dil in self.BRANCHES:
            setattr(self, name, nn.Sequential(*_conv_bn_relu(in_channels, channels, k, 1, pad, dil)))
        self.fuse = nn.Sequential(*_conv_bn_relu(channels * len(self.BRANCHES), channels, 1))

    def forward(self, x):
        return self.fuse(torch.cat([getattr(self, b[0])(x) for b in self.BRANCHES], dim=1))


@BACKBONES.register_module()
class SECOND(nn.Module):
    """backbones/second.py:9-88: per stage one strided 3x3 conv + layer_num 3x3
    convs, each + BN + ReLU; returns the output of every stage."""

    def __init__(self, in_channels=128, out_channels=(128, 128, 256), layer_nums=(3, 5, 5),
                 layer_strides=(2, 2, 2), norm_cfg=_BN, conv_cfg=dict(type="Conv2d", bias=False)):
        super().__init__()
        if not len(layer_strides) == len(layer_nums) == len(out_channels):
            raise ValueError("out_channels, layer_nums and layer_strides must have one entry "
                             "per stage")
        widths = [in_channels, *out_channels]
        self.blocks = nn.ModuleList()
        for i, depth in enumerate(layer_nums):
            layers = _conv_bn_relu(widths[i], widths[i + 1], 3, layer_strides[i], 1,
                                   norm_cfg=norm_cfg, conv_cfg=conv_cfg, inplace=True)
            for _ in range(depth):
                layers += _conv_bn_relu(widths[i + 1], widths[i + 1], 3, 1, 1, norm_cfg=norm_cfg,
                                        conv_cfg=conv_cfg, inplace=True)
            self.blocks.append(nn.Sequential(*layers))

    def init_weights(self, pretrained=None):
        """The reference leaves the conv layers at their default init (:61-68)."""

    def forward(self, x):
        outs = []
        for block in self.blocks:
            x = block(x)
            outs.append(x)
        return tuple(outs)


@NECKS.register_module()
class SECONDFPN(nn.Module):
    """necks/second_fpn.py:10-93: one up-sampling block per input level
    (ConvTranspose2d with kernel = stride; a plain conv when the stride is 1 and
    use_conv_for_no_stride, or the stride is a fraction) + BN + ReLU, outputs
    concatenated on channels and returned as a one-element list."""

    def __init__(self, in_channels=(128, 128, 256), out_channels=(256, 256, 256),
                 upsample_strides=(1, 2, 4), norm_cfg=_BN,
                 upsample_cfg=dict(type="deconv", bias=False),
                 conv_cfg=dict(type="Conv2d", bias=False), use_conv_for_no_stride=False):
        super().__init__()
        if not len(out_channels) == len(upsample_strides) == len(in_channels):
            raise ValueError("in_channels, out_channels and upsample_strides must have one "
                             "entry per level")
        self.in_channels, self.out_channels = list(in_channels), list(out_channels)
        up = dict(upsample_cfg)
        if up.pop("type") != "deconv":
            raise NotImplementedError("SECONDFPN: only upsample_cfg type 'deconv' is built")
        self.deblocks = nn.ModuleList()
        for cin, cout, stride in zip(in_channels, out_channels, upsample_strides):
            if stride > 1 or (stride == 1 and not use_conv_for_no_stride):
                first = nn.ConvTranspose2d(cin, cout, kernel_size=stride, stride=stride, **up)
            else:
                k = int(round(1 / stride))
                first = build_conv_layer(conv_cfg, cin, cout, kernel_size=k, stride=k)
            self.deblocks.append(nn.Sequential(first, build_norm_layer(norm_cfg, cout)[1],
                                               nn.ReLU(inplace=True)))

    def init_weights(self):
        """second_fpn.py:69-75: kaiming (fan_out, relu) on Conv2d, 1 on norms."""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        if len(x) != len(self.in_channels):
            raise ValueError("SECONDFPN got %d maps for %d levels" % (len(x), len(self.in_channels)))
        ups = [deblock(x[i]) for i, deblock in enumerate(self.deblocks)]
        return [torch.cat(ups, dim=1) if len(ups) > 1 else ups[0]]


def to_channels_last(module):
    """Weights of every 2-D conv in NHWC (KRSC) order: with a channels-last input
    MIOpen then runs its NHWC kernels and no transpose is inserted per layer."""
    return module.to(memory_format=torch.channels_last)


class BevTail(nn.Module):
    """x = bev_fusion(cat([x, x_mm], 1)); x = pts_backbone(x); x = pts_neck(x)
    (MSMDFusion.py:440-447).  Attribute names are the detector's."""

    def __init__(self, bev_fusion=None, pts_backbone=None, pts_neck=None, compute_dtype=None,
                 channels_last=False):
        super().__init__()
        self.bev_fusion = bev_fusion if bev_fusion is not None else SPPModule()
        self.pts_backbone = pts_backbone if pts_backbone is not None else SECOND(
            in_channels=256, out_channels=[128, 256], layer_nums=[5, 5], layer_strides=[1, 2])
        self.pts_neck = pts_neck if pts_neck is not None else SECONDFPN(
            in_channels=[128, 256], out_channels=[256, 256], upsample_strides=[1, 2],
            use_conv_for_no_stride=True)
        self.compute_dtype = compute_dtype
        self.channels_last = channels_last
        if channels_last:
            to_channels_last(self)

    def forward(self, x, x_mm=None):
        """x: the joint [B,640,H,W] map (channels-last view from
        spconv.functional.bev_concat), or x and x_mm separately as the reference
        holds them."""
        if x_mm is not None:
            x = torch.cat([x, x_mm], dim=1)
        if self.channels_last:
            x = x.contiguous(memory_format=torch.channels_last)
        if self.compute_dtype is None:
            x = self.bev_fusion(x)
            x = self.pts_backbone(x)
            return self.pts_neck(x)
        with torch.autocast(x.device.type, dtype=self.compute_dtype):
            x = self.bev_fusion(x)
            x = self.pts_backbone(x)
            return self.pts_neck(x)
