"""Extension-level drop-ins: Python modules with the SIGNATURES OF THE REFERENCE'S
PYBIND11 EXTENSIONS, bound to the C ABI of libmsmd_hip.so.

A maintainer who keeps mmdet3d/ops/voxel/voxelize.py and mmdet3d/ops/spconv/ops.py
as they are replaces the two compiled extension modules they import
(`from . import voxel_layer`, `from . import sparse_conv_ext`) with these files:

    voxel_layer.hard_voxelize              mmdet3d/ops/voxel/src/voxelization.h:51-69
    sparse_conv_ext.get_indice_pairs_3d    mmdet3d/ops/spconv/src/all.cc:24-25,
                                           include/spconv/spconv_ops.h:33-140
    sparse_conv_ext.indice_conv_fp32       all.cc:30, spconv_ops.h:260-361
    sparse_conv_ext.indice_conv_backward_fp32   all.cc:31-32, spconv_ops.h:363-456

Same arguments, same return values and formats (indicePairs[K,2,N] / indiceNum[K],
spconv_ops.h:55-59), RuntimeError where the reference raises through
TV_ASSERT_RT_ERR / TORCH_CHECK.  tests/test_gpu_integration.py calls each of them
the way the reference's Python does and checks the results against the oracle.
"""
from . import sparse_conv_ext, voxel_layer  # noqa: F401
