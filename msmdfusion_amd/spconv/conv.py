"""SubMConv3d / SparseConv3d modules: the module contract of spconv-2.1.21's
SparseConvolution as the reference ships it (bug_fix/conv.py:41-462, classes
registered in mmcv's CONV_LAYERS at :487-899), on the HIP implicit-GEMM path.

Kept: constructor signature and defaults (bug_fix/conv.py:46-64,873-885), KRSC
weight layout [Cout, kd, kh, kw, Cin] of the GPU build (:114-117) so published
checkpoints load, kaiming-uniform init with a=sqrt(5) over fan_in = Cin*K
(:163-183), output shape rule (:196-204), indice_key reuse for SubM only
(:364-375), bias add (:448-449).  Transposed / inverse convs and groups are
outside the hot path and raise.
"""
import math
from typing import List, Optional, Tuple, Union

import numpy as np
import torch
from torch.nn import init
from torch.nn.parameter import Parameter

from .. import kernels as K
from ..registry import CONV_LAYERS
from . import functional as Fsp
from .core import IndiceData, SparseConvTensor
from .modules import SparseModule


class ConvAlgo:
    """Names of spconv.core.ConvAlgo; every value runs the same HIP kernel."""
    Native = 0
    MaskImplicitGemm = 1
    MaskSplitImplicitGemm = 2


def expand_nd(ndim, val):
    if isinstance(val, (list, tuple)):
        assert len(val) == ndim
        return [int(v) for v in val]
    return [int(val)] * ndim


class SparseConvolution(SparseModule):
    __constants__ = ["stride", "padding", "dilation", "groups", "bias", "subm", "inverse",
                     "transposed", "output_padding"]

    def __init__(self, ndim: int, in_channels: int, out_channels: int,
                 kernel_size: Union[int, List[int], Tuple[int, ...]] = 3,
                 stride: Union[int, List[int], Tuple[int, ...]] = 1,
                 padding: Union[int, List[int], Tuple[int, ...]] = 0,
                 dilation: Union[int, List[int], Tuple[int, ...]] = 1,
                 groups: int = 1, bias: bool = True, subm: bool = False,
                 output_padding: Union[int, List[int], Tuple[int, ...]] = 0,
                 transposed: bool = False, inverse: bool = False,
                 indice_key: Optional[str] = None, algo=None, fp32_accum=None, name=None):
        super().__init__(name=name)
        assert groups == 1, "don't support groups for now"
        if ndim != 3:
            raise NotImplementedError("the MSMDFusion hot path is 3-D only")
        if transposed or inverse:
            raise NotImplementedError("transposed / inverse sparse convs are outside the hot path")
        self.ndim = ndim
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = expand_nd(ndim, kernel_size)
        self.stride = expand_nd(ndim, stride)
        self.dilation = expand_nd(ndim, dilation)
        self.padding = expand_nd(ndim, padding)
        kv = int(np.prod(self.kernel_size))
        self.conv1x1 = kv == 1
        if not subm:
            self.conv1x1 &= int(np.prod(self.stride)) == 1
            if self.conv1x1:
                assert self.padding == [0] * ndim, "padding must be zero for 1x1 conv (k=1,s=1)"
        self.transposed = transposed
        self.inverse = inverse
        self.output_padding = expand_nd(ndim, output_padding)
        self.groups = groups
        self.subm = subm
        self.indice_key = indice_key
        self.algo = ConvAlgo.MaskImplicitGemm if algo is None else algo
        self.fp32_accum = fp32_accum
        # KRSC
        self.weight = Parameter(torch.empty(out_channels, *self.kernel_size, in_channels))
        if bias:
            self.bias = Parameter(torch.empty(out_channels))
        else:
            self.register_parameter("bias", None)
        self.reset_parameters()

    def extra_repr(self):
        s = "{in_channels}, {out_channels}, kernel_size={kernel_size}, stride={stride}"
        if self.padding != [0] * len(self.padding):
            s += ", padding={padding}"
        if self.dilation != [1] * len(self.dilation):
            s += ", dilation={dilation}"
        if self.bias is None:
            s += ", bias=False"
        return s.format(**self.__dict__)

    def _calculate_fan_in_and_fan_out(self):
        rf = 1
        for s in self.kernel_size:
            rf *= s
        return self.in_channels * rf, self.out_channels * rf

    def reset_parameters(self):
        fan_in, _ = self._calculate_fan_in_and_fan_out()
        gain = init.calculate_gain("leaky_relu", math.sqrt(5))
        bound = math.sqrt(3.0) * gain / math.sqrt(fan_in)
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)
            if self.bias is not None:
                self.bias.uniform_(-1 / math.sqrt(fan_in), 1 / math.sqrt(fan_in))

    def weight_kio(self):
        """[K, Cin, Cout] view-copy of the KRSC parameter (autograd-tracked)."""
        kv = int(np.prod(self.kernel_size))
        return self.weight.reshape(self.out_channels, kv, self.in_channels).permute(1, 2, 0)

    def _load_from_state_dict(self, *args, **kwargs):
        # copy_ into a Parameter bumps its version, but loaders that write through `.data`
        # do not: the packed images of frozen weights are rebuilt after any load
        from . import functional as Fsp
        Fsp.invalidate_packed_weights()
        return super()._load_from_state_dict(*args, **kwargs)

    def train(self, mode=True):
        # hooks that swap or rewrite parameters through `.data` (EMA, fp16 master copies) do
        # it at train()/eval() boundaries and leave `_version` where it was: repack
        from . import functional as Fsp
        Fsp.invalidate_packed_weights(self.weight)
        return super().train(mode)

    def forward(self, input: SparseConvTensor):
        assert isinstance(input, SparseConvTensor)
        assert input.features.shape[1] == self.in_channels, "channel size mismatch"
        features = input.features
        spatial_shape = input.spatial_shape
        if not self.subm:
            out_spatial_shape = K.conv_output_size(spatial_shape, self.kernel_size, self.stride,
                                                   self.padding, self.dilation)
        else:
            out_spatial_shape = spatial_shape
        out_tensor = input.shadow_copy()
        if self.conv1x1:
            kio = self.weight_kio()[0]
            feats = torch.mm(features, kio)
            if self.bias is not None:
                feats = feats + self.bias
            out_tensor = out_tensor.replace_feature(feats)
            out_tensor.spatial_shape = out_spatial_shape
            return out_tensor
        indice_dict = input.indice_dict.copy()
        datas = input.find_indice_pair(self.indice_key)
        if datas is not None:
            assert isinstance(datas, IndiceData)
            assert self.subm, "only support reuse subm indices"
            self._check_subm_reuse_valid(input, spatial_shape, datas)
        else:
            datas = input.cached_rulebook(self.kernel_size, self.stride, self.padding,
                                          self.dilation, self.subm)
            if self.indice_key is not None:
                msg = f"your indice key {self.indice_key} already exists in this sparse tensor."
                assert self.indice_key not in indice_dict, msg
                indice_dict[self.indice_key] = datas
        if input.indices.shape[0] == 0 and datas.n_out == 0:
            out_features = features.new_zeros((0, self.out_channels))
            stats = None
        else:
            # a BatchNorm1d follows (SparseSequential / SparseBasicBlock set the flag): let the
            # conv kernel leave the per-tile sums the BN's statistics pass would recompute
            stats = [None] if (getattr(self, "emit_bn_stats", False) and self.bias is None
                               and features.is_cuda) else None
            out_features = Fsp.sparse_conv(features, self.weight, datas, krsc=True,
                                           bn_stats=stats)
        if self.bias is not None:
            out_features = out_features + self.bias
        out_tensor.indices = datas.out_indices
        out_tensor = out_tensor.replace_feature(out_features)
        # (an attribute of THIS tensor object only: whatever replaces its features drops it)
        out_tensor.bn_stats = stats[0] if stats else None
        out_tensor.indice_dict = indice_dict
        out_tensor.spatial_shape = out_spatial_shape
        return out_tensor

    def _check_subm_reuse_valid(self, inp, spatial_shape, datas):
        assert datas.is_subm, "only support reuse subm indices"
        if self.kernel_size != datas.ksize:
            raise ValueError(f"subm with same indice_key must have same kernel size, expect "
                             f"{datas.ksize}, this layer {self.kernel_size}")
        if self.dilation != datas.dilation:
            raise ValueError(f"subm with same indice_key must have same dilation, expect "
                             f"{datas.dilation}, this layer {self.dilation}")
        if inp.spatial_shape != datas.spatial_shape:
            raise ValueError(f"subm with same indice_key must have same spatial structure, "
                             f"expect {datas.spatial_shape}, input {spatial_shape}")
        if inp.indices.shape[0] != datas.indices.shape[0]:
            raise ValueError("subm with same indice_key must have the same number of voxels")


@CONV_LAYERS.register_module()
class SparseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, indice_key=None, algo=None, fp32_accum=None, name=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation,
                         groups, bias, indice_key=indice_key, algo=algo, fp32_accum=fp32_accum,
                         name=name)


@CONV_LAYERS.register_module()
class SubMConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias=True, indice_key=None, algo=None, fp32_accum=None, name=None):
        super().__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation,
                         groups, bias, True, indice_key=indice_key, algo=algo,
                         fp32_accum=fp32_accum, name=name)
