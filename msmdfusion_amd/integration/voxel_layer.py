"""`voxel_layer` -- the pybind module of mmdet3d/ops/voxel (voxelization.cpp:6-11),
on the C ABI.  Only `hard_voxelize` is on the hot path; dynamic voxelization
raises (not used by either target config)."""
import ctypes as C

import torch

from .._lib import check, float_arr, lib


def hard_voxelize(points, voxels, coors, num_points_per_voxel, voxel_size, coors_range,
                  max_points, max_voxels, NDim=3):
    """voxelization.h:51-69: fills the caller's `voxels[max_voxels,max_points,C]`,
    `coors[max_voxels,3]` (z,y,x) and `num_points_per_voxel[max_voxels]` in place and
    returns the number of voxels (mmdet3d/ops/voxel/voxelize.py:41-59 slices by it).
    Rows [0, voxel_num) are written completely, zero padding included; rows beyond
    keep the caller's contents (voxelize.py:46-50 zero-fills them)."""
    if NDim != 3:
        raise RuntimeError("hard_voxelize: only NDim == 3 is built")
    if not points.is_cuda:
        raise RuntimeError("hard_voxelize: points must live on the GPU (no CPU path)")
    if points.dtype != torch.float32 or voxels.dtype != torch.float32 or \
            coors.dtype != torch.int32 or num_points_per_voxel.dtype != torch.int32:
        raise RuntimeError("hard_voxelize: float32 points/voxels and int32 coors/num_points")
    if not (points.is_contiguous() and voxels.is_contiguous() and coors.is_contiguous()
            and num_points_per_voxel.is_contiguous()):
        raise RuntimeError("hard_voxelize: tensors must be contiguous")
    n, c = points.shape
    if tuple(voxels.shape) != (max_voxels, max_points, c) or tuple(coors.shape) != (max_voxels, 3) \
            or num_points_per_voxel.shape[0] != max_voxels:
        raise RuntimeError("hard_voxelize: output tensors do not match max_voxels / max_points")
    with torch.cuda.device(points.device):
        count = torch.empty((1,), dtype=torch.int32, device=points.device)
        nbytes = lib.msmd_voxelize_workspace_bytes(n, int(max_voxels), int(max_points))
        ws = torch.empty((max(int(nbytes), 256),), dtype=torch.uint8, device=points.device)
        stream = torch.cuda.current_stream(points.device).cuda_stream
        check(lib.msmd_hard_voxelize(
            C.c_void_p(points.data_ptr()), n, c, float_arr(voxel_size), float_arr(coors_range),
            int(max_points), int(max_voxels), C.c_void_p(voxels.data_ptr()),
            C.c_void_p(coors.data_ptr()), C.c_void_p(num_points_per_voxel.data_ptr()), None,
            C.c_void_p(count.data_ptr()), C.c_void_p(ws.data_ptr()), nbytes, C.c_void_p(stream)),
            "msmd_hard_voxelize")
        return int(count.item())      # the one host read the reference has too


def dynamic_voxelize(points, coors, voxel_size, coors_range, NDim=3):
    raise RuntimeError("dynamic_voxelize is outside the MSMDFusion hot path (not built)")
