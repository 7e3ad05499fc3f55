"""Voxelization module: mmdet3d/ops/voxel/voxelize.py:10-114 on the HIP path.

Same constructor, same mutable `.voxel_size` (MSMDFusionDetector.voxelize
rescales it per call, MSMDFusion.py:475-478), same (train, test) max_voxels
pair, same return triple (voxels[M,max_points,C], coors[M,3] zyx,
num_points[M]).  `voxelization_mean` is the fused Voxelization+HardSimpleVFE
form (the [M,max_points,C] tensor is never materialised)."""
import torch
from torch import nn
from torch.nn.modules.utils import _pair

from . import kernels as K


def voxelization(points, voxel_size, coors_range, max_points=35, max_voxels=20000):
    """_Voxelization.forward (voxelize.py:12-59).  Dynamic voxelization
    (max_points == -1) is not on the hot path and raises."""
    if max_points == -1 or max_voxels == -1:
        raise NotImplementedError("dynamic voxelization is outside the MSMDFusion hot path")
    with torch.no_grad():
        voxels, coors, num_points, _ = K.hard_voxelize(points, voxel_size, coors_range,
                                                       max_points, max_voxels)
    return voxels, coors, num_points


def voxelization_mean(points, voxel_size, coors_range, max_points, max_voxels):
    """-> (mean_features[M,C], coors[M,3], num_points[M])."""
    with torch.no_grad():
        _, coors, num_points, mean = K.hard_voxelize(points, voxel_size, coors_range, max_points,
                                                     max_voxels, want_voxels=False,
                                                     want_mean=True)
    return mean, coors, num_points


class Voxelization(nn.Module):

    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000):
        super().__init__()
        self.voxel_size = voxel_size
        self.point_cloud_range = point_cloud_range
        self.max_num_points = max_num_points
        self.max_voxels = max_voxels if isinstance(max_voxels, tuple) else _pair(max_voxels)
        pcr = torch.tensor(point_cloud_range, dtype=torch.float32)
        vs = torch.tensor(voxel_size, dtype=torch.float32)
        grid_size = torch.round((pcr[3:] - pcr[:3]) / vs).long()
        self.grid_size = grid_size
        self.pcd_shape = [*grid_size[:2], 1][::-1]

    def _max_voxels(self):
        return self.max_voxels[0] if self.training else self.max_voxels[1]

    def forward(self, input):
        return voxelization(input, self.voxel_size, self.point_cloud_range, self.max_num_points,
                            self._max_voxels())

    def forward_mean(self, input):
        return voxelization_mean(input, self.voxel_size, self.point_cloud_range,
                                 self.max_num_points, self._max_voxels())

    @torch.no_grad()
    def forward_batch(self, points_list, fused_mean=False, voxel_size=None):
        """The per-sample loop of the detectors' voxelize()
        (MSMDFusion.py:479-483) with all launches enqueued before the single
        host read of the voxel counts.  -> list of (voxels|mean, coors, num).
        voxel_size overrides the layer's for this call (the detector rescales it
        per image scale; two batches prepared concurrently must not do that
        through the shared attribute)."""
        res = K.hard_voxelize_batch(points_list,
                                    self.voxel_size if voxel_size is None else voxel_size,
                                    self.point_cloud_range,
                                    self.max_num_points, self._max_voxels(),
                                    want_voxels=not fused_mean, want_mean=fused_mean)
        return [((mu if fused_mean else v), c, n) for v, c, n, mu in res]

    def __repr__(self):
        return (f"{self.__class__.__name__}(voxel_size={self.voxel_size}, point_cloud_range="
                f"{self.point_cloud_range}, max_num_points={self.max_num_points}, "
                f"max_voxels={self.max_voxels})")
