"""msmdfusion_amd -- MI355X (gfx950) implementation of MSMDFusion's sparse-voxel
fusion hot path behind the reference's mmdet3d / spconv module API.

Only the hot path lives here: hard voxelization,
SubMConv3d / SparseConv3d (rulebooks + implicit-GEMM arithmetic), sparse_add,
the GMA-Conv fusion block and its neighbour-search helpers, and BEV scatter.
Everything computes in libmsmd_hip.so (hand-written HIP, C ABI in
include/msmd_hip.h); importing the compute modules without that library built
raises -- there is no CPU fallback.
"""
__version__ = "0.1.0"
