"""Dense 2-D convolutions of the BEV tail on the sparse-conv kernels (row f1, round 2).

A dense H x W map in channels-last layout IS a sparse tensor with every cell active: its
rows are the pixels, `[B*H*W, C]`, and a k x k convolution (any stride / dilation) is the
output-stationary neighbour-table convolution of csrc/spconv_split.hip with a table that
needs no hash and no search -- it is arithmetic on the pixel index, built once per
(grid, geometry) and cached for the life of the process together with its tiling order,
stream-K prefix and pair lists (a dense grid never changes, so the index pass of these
layers costs nothing per step).  What that buys on MI355X: fp32-EQUIVALENT results on the
bf16 matrix cores (three bf16 planes per operand) at the rate of the sparse
kernels (~150 TF useful) where MIOpen's fp32 convolutions run the 640 -> 256 SPP branches at
~90 TF, with no layout change anywhere: the joint channels-last BEV buffer that both sparse
tensors scatter into (spconv.functional.bev_concat) is the first layer's input as it lies.

Modules keep torch's parameter shapes and names (`weight [Cout,Cin,kh,kw]`), so the
reference's checkpoints load unchanged: `SPPModuleRows` IS a `bev.SPPModule` with a
different forward.
"""
import torch
from torch import nn

from .bev import SECOND, SECONDFPN, SPPModule
from .spconv import functional as Fsp
from .spconv.core import IndiceData

_TABLES = {}


def _pair(v):
    return (int(v), int(v)) if isinstance(v, int) else (int(v[0]), int(v[1]))


def grid_rulebook(batch, height, width, ksize, stride=1, padding=0, dilation=1, device="cuda"):
    """IndiceData of a dense conv over a [batch, height, width] grid (rows in b, y, x
    order).  -> (rulebook, out_height, out_width).  Cached per geometry."""
    (kh, kw), (sh, sw), (ph, pw), (dh, dw) = map(_pair, (ksize, stride, padding, dilation))
    device = torch.device(device)
    key = (device, batch, height, width, kh, kw, sh, sw, ph, pw, dh, dw)
    hit = _TABLES.get(key)
    if hit is not None:
        return hit
    ho = (height + 2 * ph - dh * (kh - 1) - 1) // sh + 1
    wo = (width + 2 * pw - dw * (kw - 1) - 1) // sw + 1
    b = torch.arange(batch, device=device).view(1, batch, 1, 1)
    ky = torch.arange(kh, device=device).repeat_interleave(kw).view(-1, 1, 1, 1)
    kx = torch.arange(kw, device=device).repeat(kh).view(-1, 1, 1, 1)
    # forward: output pixel (yo, xo) reads input (yo*s - p + ky*d, xo*s - p + kx*d)
    yo = torch.arange(ho, device=device).view(1, 1, ho, 1)
    xo = torch.arange(wo, device=device).view(1, 1, 1, wo)
    yi, xi = yo * sh - ph + ky * dh, xo * sw - pw + kx * dw
    ok = (yi >= 0) & (yi < height) & (xi >= 0) & (xi < width)
    fwd = torch.where(ok, (b * height + yi) * width + xi, -1)
    nbr_fwd = fwd.reshape(kh * kw, batch * ho * wo).int().contiguous()
    same = sh == sw == 1 and ho == height and wo == width and kh % 2 == 1 and kw % 2 == 1 \
        and 2 * ph == dh * (kh - 1) and 2 * pw == dw * (kw - 1)
    nbr_bwd = None
    if not same:
        # backward: input pixel (yi, xi) feeds output ((yi + p - ky*d) / s, ...) if divisible
        yi_ = torch.arange(height, device=device).view(1, 1, height, 1)
        xi_ = torch.arange(width, device=device).view(1, 1, 1, width)
        ty, tx = yi_ + ph - ky * dh, xi_ + pw - kx * dw
        yo_, xo_ = torch.div(ty, sh, rounding_mode="floor"), torch.div(tx, sw, rounding_mode="floor")
        ok = (ty >= 0) & (tx >= 0) & (yo_ * sh == ty) & (xo_ * sw == tx) & (yo_ < ho) & (xo_ < wo)
        bwd = torch.where(ok, (b * ho + yo_) * wo + xo_, -1)
        nbr_bwd = bwd.reshape(kh * kw, batch * height * width).int().contiguous()
    n_in, n_out = batch * height * width, batch * ho * wo
    rb = IndiceData(torch.empty((n_out, 0), dtype=torch.int32, device=device),
                    torch.empty((n_in, 0), dtype=torch.int32, device=device), nbr_fwd, nbr_bwd,
                    same, [1, height, width], [1, ho, wo], [1, kh, kw], [1, sh, sw], [0, ph, pw],
                    [1, dh, dw])
    _TABLES[key] = (rb, ho, wo)
    return _TABLES[key]


def grid_conv2d(rows, grid, weight, stride=1, padding=0, dilation=1):
    """rows [B*H*W, Cin] (channels-last pixels), grid = (B, H, W), weight the nn.Conv2d
    parameter [Cout, Cin, kh, kw] -> (rows [B*Ho*Wo, Cout], (B, Ho, Wo)).  Differentiable
    in rows and weight (forward / dgrad: spconv_fwd_split; wgrad: spconv_wgrad_split)."""
    b, h, w = grid
    if rows.shape[0] != b * h * w:
        raise ValueError("grid_conv2d: %d rows for a %dx%dx%d grid" % (rows.shape[0], b, h, w))
    rb, ho, wo = grid_rulebook(b, h, w, weight.shape[2:], stride, padding, dilation, rows.device)
    # KRSC view of the torch parameter: [Cout, 1, kh, kw, Cin]; gradients flow back through it
    krsc = weight.permute(0, 2, 3, 1).unsqueeze(1)
    c_in, c_out = weight.shape[1], weight.shape[0]
    rb.prepare(torch.is_grad_enabled() and (weight.requires_grad or rows.requires_grad),
               c_in, c_out)
    return Fsp.sparse_conv(rows.contiguous(), krsc.contiguous(), rb, krsc=True), (b, ho, wo)


def rows_of(x):
    """NCHW tensor (any memory format) -> (rows [B*H*W, C], (B, H, W)); free for a
    channels-last tensor."""
    b, c, h, w = x.shape
    return x.permute(0, 2, 3, 1).reshape(b * h * w, c), (b, h, w)


def map_of(rows, grid):
    """rows [B*H*W, C] -> the [B, C, H, W] map as a channels-last view (no copy)."""
    b, h, w = grid
    return rows.view(b, h, w, rows.shape[1]).permute(0, 3, 1, 2)


def conv_bn_relu_rows(block, rows, grid):
    """One nn.Sequential(Conv2d(bias=False), BatchNorm2d, ReLU) of the reference on pixel
    rows: the conv through grid_conv2d, BN + ReLU through the fused row kernels
    (BatchNorm2d over NHWC pixels is BatchNorm1d over rows: same statistics)."""
    conv, bn = block[0], block[1]
    if conv.bias is not None or conv.groups != 1:
        raise NotImplementedError("grid conv: bias-free, ungrouped convolutions only")
    y, grid = grid_conv2d(rows, grid, conv.weight, conv.stride, conv.padding, conv.dilation)
    return Fsp.bn_act(y, bn, relu=True), grid


class SPPModuleRows(SPPModule):
    """bev.SPPModule (MSMDFusion.py:47-90) with the same parameters and state-dict keys,
    computed on pixel rows by the sparse-conv kernels.  forward takes and returns
    [B, C, H, W] maps (channels-last views: feed it spconv.functional.bev_concat's result
    directly)."""

    def forward(self, x):
        rows, grid = rows_of(x)
        rows = rows.contiguous()
        outs = [conv_bn_relu_rows(getattr(self, name), rows, grid)[0]
                for name, _, _, _ in self.BRANCHES]
        y, grid = conv_bn_relu_rows(self.fuse, torch.cat(outs, 1), grid)
        return map_of(y, grid)


def deconv_rulebook(batch, height, width, stride, device="cuda"):
    """ConvTranspose2d with kernel == stride (no overlap: second_fpn.py:46-52): output
    pixel (Y, X) has exactly one contribution, input (Y // s, X // s) through kernel tap
    (Y % s, X % s).  -> (rulebook, out_height, out_width), cached."""
    s = int(stride)
    device = torch.device(device)
    key = (device, "deconv", batch, height, width, s)
    hit = _TABLES.get(key)
    if hit is not None:
        return hit
    ho, wo = height * s, width * s
    b = torch.arange(batch, device=device).view(1, batch, 1, 1)
    ky = torch.arange(s, device=device).repeat_interleave(s).view(-1, 1, 1, 1)
    kx = torch.arange(s, device=device).repeat(s).view(-1, 1, 1, 1)
    yo = torch.arange(ho, device=device).view(1, 1, ho, 1)
    xo = torch.arange(wo, device=device).view(1, 1, 1, wo)
    ok = (yo % s == ky) & (xo % s == kx)
    fwd = torch.where(ok, (b * height + yo // s) * width + xo // s, -1)
    yi = torch.arange(height, device=device).view(1, 1, height, 1)
    xi = torch.arange(width, device=device).view(1, 1, 1, width)
    bwd = (b * ho + yi * s + ky) * wo + xi * s + kx
    n_in, n_out = batch * height * width, batch * ho * wo
    rb = IndiceData(torch.empty((n_out, 0), dtype=torch.int32, device=device),
                    torch.empty((n_in, 0), dtype=torch.int32, device=device),
                    fwd.reshape(s * s, n_out).int().contiguous(),
                    bwd.reshape(s * s, n_in).int().contiguous(), False,
                    [1, height, width], [1, ho, wo], [1, s, s], [1, s, s], [0, 0, 0], [1, 1, 1])
    _TABLES[key] = (rb, ho, wo)
    return _TABLES[key]


def grid_deconv2d(rows, grid, weight, stride):
    """nn.ConvTranspose2d(kernel_size=stride, stride=stride, bias=False) on pixel rows;
    weight is the torch parameter [Cin, Cout, s, s]."""
    b, h, w = grid
    s = int(stride[0] if isinstance(stride, (tuple, list)) else stride)
    if tuple(weight.shape[2:]) != (s, s):
        raise NotImplementedError("grid deconv: kernel_size == stride only")
    rb, ho, wo = deconv_rulebook(b, h, w, s, rows.device)
    c_in, c_out = weight.shape[0], weight.shape[1]
    rb.prepare(torch.is_grad_enabled() and (weight.requires_grad or rows.requires_grad),
               c_in, c_out)
    kio = weight.permute(2, 3, 0, 1).reshape(s * s, c_in, c_out)      # [K, Cin, Cout]
    return Fsp.sparse_conv(rows.contiguous(), kio.contiguous(), rb, krsc=False), (b, ho, wo)


class SECONDRows(SECOND):
    """bev.SECOND (backbones/second.py:9-88), same parameters and keys, on pixel rows."""

    def forward(self, x):
        rows, grid = rows_of(x)
        rows = rows.contiguous()
        outs = []
        for block in self.blocks:
            for j in range(0, len(block), 3):          # (conv, BN, ReLU) triples
                rows, grid = conv_bn_relu_rows(block[j:j + 3], rows, grid)
            outs.append(map_of(rows, grid))
        return tuple(outs)


class SECONDFPNRows(SECONDFPN):
    """bev.SECONDFPN (necks/second_fpn.py:10-93) on pixel rows: 1x1 / strided convs through
    grid_conv2d, kernel == stride transposed convs through grid_deconv2d."""

    def forward(self, x):
        if len(x) != len(self.in_channels):
            raise ValueError("SECONDFPN got %d maps for %d levels" % (len(x), len(self.in_channels)))
        ups = []
        for xi, deblock in zip(x, self.deblocks):
            rows, grid = rows_of(xi)
            up, bn = deblock[0], deblock[1]
            if isinstance(up, nn.ConvTranspose2d):
                y, grid = grid_deconv2d(rows, grid, up.weight, up.stride)
            else:
                y, grid = grid_conv2d(rows, grid, up.weight, up.stride, up.padding, up.dilation)
            ups.append((Fsp.bn_act(y, bn, relu=True), grid))
        if any(g != ups[0][1] for _, g in ups):
            raise ValueError("SECONDFPN levels end on different grids: %s" % [g for _, g in ups])
        rows = torch.cat([u for u, _ in ups], 1) if len(ups) > 1 else ups[0][0]
        return [map_of(rows, ups[0][1])]
