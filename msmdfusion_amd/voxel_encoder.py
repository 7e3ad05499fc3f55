"""HardSimpleVFE: mmdet3d/models/voxel_encoders/voxel_encoder.py:14-46."""
from torch import nn

from . import kernels as K
from .registry import VOXEL_ENCODERS


@VOXEL_ENCODERS.register_module()
class HardSimpleVFE(nn.Module):
    """Mean of the points of each voxel (sum over the slots / num_points)."""

    def __init__(self, num_features=4):
        super().__init__()
        self.num_features = num_features
        self.fp16_enabled = False

    def forward(self, features, num_points, coors):
        return K.voxel_mean(features, num_points, self.num_features)
