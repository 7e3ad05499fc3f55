"""SparseModule / SparseSequential: container semantics of
mmdet3d/ops/spconv/modules.py:44-137 (spconv-2.x keeps the same contract):
sparse modules receive the SparseConvTensor, every other nn.Module is applied
to .features only, and only when the tensor has at least one active voxel."""
import sys
from collections import OrderedDict

from torch import nn

from .core import SparseConvTensor


class SparseModule(nn.Module):
    """Marker base: subclasses take a SparseConvTensor inside SparseSequential."""

    def __init__(self, name=None):
        super().__init__()
        self.name = name
        self._sparse_unique_name = ""


def is_spconv_module(module):
    return isinstance(module, SparseModule)


class SparseSequential(SparseModule):

    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)
        for name, module in kwargs.items():
            if sys.version_info < (3, 6):
                raise ValueError("kwargs only supported in py36+")
            if name in self._modules:
                raise ValueError("name exists.")
            self.add_module(name, module)
        self._sparity_dict = {}

    def __getitem__(self, idx):
        if not (-len(self) <= idx < len(self)):
            raise IndexError("index {} is out of range".format(idx))
        if idx < 0:
            idx += len(self)
        return list(self._modules.values())[idx]

    def __len__(self):
        return len(self._modules)

    @property
    def sparity_dict(self):
        return self._sparity_dict

    def add(self, module, name=None):
        if name is None:
            name = str(len(self._modules))
            if name in self._modules:
                raise KeyError("name exists")
        self.add_module(name, module)

    def forward(self, input):
        from .functional import bn_act
        mods = list(self._modules.items())
        i = 0
        while i < len(mods):
            k, module = mods[i]
            i += 1
            if is_spconv_module(module):
                assert isinstance(input, SparseConvTensor)
                self._sparity_dict[k] = input.sparity
                input = module(input)
            elif isinstance(input, SparseConvTensor):
                if input.indices.shape[0] != 0:
                    if isinstance(module, nn.BatchNorm1d):
                        # [BN1d, ReLU] (make_sparse_convmodule) runs as one fused op
                        relu = i < len(mods) and isinstance(mods[i][1], nn.ReLU)
                        input = input.replace_feature(bn_act(input.features, module, relu=relu))
                        i += int(relu)
                    else:
                        input = input.replace_feature(module(input.features))
            else:
                input = module(input)
        return input


class ToDense(SparseModule):
    def forward(self, x: SparseConvTensor):
        return x.dense()
