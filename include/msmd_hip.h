/*
 * msmd_hip.h -- C ABI of libmsmd_hip.so: the MI355X (gfx950) implementation of
 * MSMDFusion's sparse-voxel fusion hot path.
 *
 * The reference has no C ABI for this path; its boundary is pybind11 torch
 * extensions plus the spconv-2.x Python package.  Every entry point below
 * names the reference interface it replaces (paths relative to the reference
 * checkout).  INTEGRATION.md shows the binding a reference maintainer adds.
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes only; every pointer is DEVICE memory unless the
 *     parameter is documented "host";
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it,
 *     nothing synchronises the device, nothing allocates;
 *   - outputs and scratch are caller-allocated; scratch sizes come from the
 *     matching *_workspace_bytes() query; workspace base must be 256-B aligned;
 *   - the return value is an msmd_status (0 = ok, <0 = error, nothing was
 *     enqueued on error); msmd_status_string() names it;
 *   - re-entrant: no global mutable state; two streams may call concurrently
 *     with distinct workspaces.
 *   - tensors are dense row-major; `indices` rows are (batch, z, y, x) int32.
 */
#ifndef MSMD_HIP_H_
#define MSMD_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* msmd_stream_t; /* hipStream_t */

enum msmd_status {
  MSMD_OK = 0,
  MSMD_ERR_INVALID_ARG = -1,
  MSMD_ERR_WORKSPACE = -2,   /* workspace too small or misaligned            */
  MSMD_ERR_UNSUPPORTED = -3, /* shape / parameter outside the built kernels  */
  MSMD_ERR_LAUNCH = -4,      /* hipGetLastError() != hipSuccess after launch */
  MSMD_ERR_RANGE = -5        /* linear voxel id would not fit 32 bits        */
};

const char* msmd_status_string(int status);
/* hipGetErrorString() of the launch that made the calling thread's most recent
 * MSMD_ERR_LAUNCH. */
const char* msmd_last_launch_error(void);
/* ABI version: bumped whenever a signature below changes. */
int msmd_abi_version(void);
/* 1 when a gfx950 device is visible to this process, else 0 (host call). */
int msmd_device_ok(void);

/* ------------------------------------------------------------------------ *
 * a1/a3  Hard voxelization (+ fused mean VFE)
 * replaces: voxel_layer.hard_voxelize  mmdet3d/ops/voxel/src/voxelization.h:51-69
 *           (CPU mmdet3d/ops/voxel/src/voxelization_cpu.cpp:44-142,
 *            CUDA mmdet3d/ops/voxel/src/voxelization_cuda.cu:184-326)
 *           HardSimpleVFE.forward  mmdet3d/models/voxel_encoders/voxel_encoder.py:29-46
 * Semantics kept bit-for-bit: float32 floor((p-min)/size) coordinates, grid =
 * round((max-min)/size), voxel id = rank of the voxel's first point in input
 * order, slot = number of earlier same-voxel points (< max_points kept), the
 * `break` at the point that would open voxel number max_voxels.
 * Only rows [0, *voxel_num) of voxels/coors/num_points_per_voxel are written
 * (padding slots inside a written voxel row are zero-filled); the caller
 * reads *voxel_num (device int32) after the stream reaches this point.
 * ------------------------------------------------------------------------ */
size_t msmd_voxelize_workspace_bytes(int num_points, int max_voxels,
                                     int max_points);

int msmd_hard_voxelize(const float* points, int num_points, int num_features,
                       const float* voxel_size /* host[3] x,y,z */,
                       const float* coors_range /* host[6] xyzxyz min,max */,
                       int max_points, int max_voxels,
                       float* voxels /* [max_voxels,max_points,C] or NULL */,
                       int32_t* coors /* [max_voxels,3] (z,y,x) */,
                       int32_t* num_points_per_voxel /* [max_voxels] */,
                       float* voxel_mean /* [max_voxels,C] or NULL: fused VFE */,
                       int32_t* voxel_num /* [1] */, void* workspace,
                       size_t workspace_bytes, msmd_stream_t stream);

/* The same for SEVERAL clouds in one launch set (the B LiDAR sweeps of a batch and the
 * virtual points of its four image scales: MSMDFusion.py:462-491 calls the voxel layer once
 * per sample and scale, each call its own voxel size): blocks find their cloud, the scan
 * restarts per cloud, one fill initialises every cloud's tables.  Results per cloud as
 * msmd_hard_voxelize's, bit for bit.  The workspace is shared by the descriptors of a call. */
typedef struct msmd_voxelize_desc {
  const float* points;             /* [num_points, num_features] */
  int32_t num_points, num_features;
  float voxel_size[3];
  float coors_range[6];
  int32_t max_points, max_voxels;
  float* voxels;                   /* [max_voxels, max_points, num_features] or NULL */
  int32_t* coors;                  /* [max_voxels, 3] (z, y, x) */
  int32_t* num_points_per_voxel;   /* [max_voxels] */
  float* voxel_mean;               /* [max_voxels, num_features] or NULL */
  int32_t* voxel_num;              /* [1], device */
} msmd_voxelize_desc;
size_t msmd_hard_voxelize_many_workspace_bytes(const msmd_voxelize_desc* descs, int n_desc);
int msmd_hard_voxelize_many(const msmd_voxelize_desc* descs /* host */, int n_desc,
                            void* workspace, size_t workspace_bytes, msmd_stream_t stream);

/* HardSimpleVFE on an already materialised voxel tensor (unfused form). */
int msmd_voxel_mean(const float* voxels /* [M,max_points,C] */,
                    const int32_t* num_points_per_voxel, int num_voxels,
                    int max_points, int num_features, int out_features,
                    float* out /* [M,out_features] */, msmd_stream_t stream);

/* ------------------------------------------------------------------------ *
 * a5  Submanifold rulebook (hash-based voxel index)
 * replaces: sparse_conv_ext.get_indice_pairs_3d(..., subM=1)
 *           mmdet3d/ops/spconv/src/all.cc:21-27,
 *           mmdet3d/ops/spconv/include/spconv/spconv_ops.h:28-107,
 *           algorithm geometry.h:247-297 (getIndicePairsSubM), CUDA
 *           indice.cu.h:148-203; spconv-2.x ops.get_indice_pairs_implicit_gemm
 *           (bug_fix/conv.py:382-396).
 * Output is output-stationary: nbr[k*n + o] = input row feeding output row o
 * through kernel offset k ((kz*KH+ky)*KW+kx, geometry.h:62-73), or -1.
 * msmd_rulebook_pairs() converts it to the reference's indicePairs/indiceNum.
 * ------------------------------------------------------------------------ */
size_t msmd_rulebook_subm_workspace_bytes(int n);

int msmd_rulebook_subm3d(const int32_t* indices /* [n,4] */, int n,
                         int batch_size, const int* spatial_shape /* host[3] */,
                         const int* ksize /* host[3] */,
                         int32_t* nbr /* [K,n] */, void* workspace,
                         size_t workspace_bytes, msmd_stream_t stream);

/* The same table through an occupancy bitmap of the grid instead of the hash (1 bit per
 * cell, a popcount prefix per 256-cell block, a rank -> row table): cost = a clear and a
 * counting pass over batch_size * D*H*W / 8 bytes plus ~9 word reads per voxel, against 27
 * random slot reads per voxel for the hash -- the better choice for large voxel sets and
 * for the small grids of the deeper stages.  Identical output (duplicates keep the last
 * row). */
size_t msmd_rulebook_subm_bitmap_workspace_bytes(int n, int batch_size,
                                                 const int* spatial_shape);
int msmd_rulebook_subm3d_bitmap(const int32_t* indices, int n, int batch_size,
                                const int* spatial_shape, const int* ksize,
                                int32_t* nbr, void* workspace, size_t workspace_bytes,
                                msmd_stream_t stream);

/* msmd_rulebook_subm3d / _bitmap for MANY voxel sets in one launch set (2 fills + at most 6
 * kernels whatever the number of tables): an index pass builds the SubM tables of all its
 * voxel sets together at its end -- nothing in the index chain reads one.  descs: HOST
 * array; method 0 = hash index, 1 = occupancy bitmap (as the single calls); tables with
 * n = 0 are skipped.  Results identical to the single calls. */
typedef struct msmd_subm_desc {
  const int32_t* indices;      /* [n,4] (b,z,y,x) */
  int32_t n, batch_size;
  int32_t spatial_shape[3], ksize[3];
  int32_t method, reserved;
  int32_t* nbr;                /* [K, n] */
} msmd_subm_desc;
size_t msmd_rulebook_subm3d_many_workspace_bytes(const msmd_subm_desc* descs, int n_desc);
int msmd_rulebook_subm3d_many(const msmd_subm_desc* descs, int n_desc, void* workspace,
                              size_t workspace_bytes, msmd_stream_t stream);

/* ------------------------------------------------------------------------ *
 * a6  Strided (regular) sparse conv rulebook, two phases with one host read
 * replaces: sparse_conv_ext.get_indice_pairs_3d(..., subM=0)
 *           spconv_ops.h:108-137, geometry.h:144-194 (getIndicePairsConv),
 *           CUDA indice.cu.h:22-65,112-145 + torch::_unique.
 * Output rows are in ascending linear (b,z,y,x) id, the order of the
 * reference's CUDA path (spconv_ops.h:130).  Phase 1 marks the occupied
 * output cells and counts them (*n_out, device); the caller reads it,
 * allocates, then phase 2 fills out_indices / nbr_fwd / nbr_bwd.  The same
 * workspace must be passed, untouched, to both phases.
 * An input set that repeats a coordinate (only the reference_quirks unified sets can): the
 * table is output-stationary, one input row per (offset, output row) -- the LAST row of the
 * coordinate, deterministically, as every SubM look-up; nbr_bwd still lists the output row
 * for every input row (the earlier rows receive the gradient of a contribution the forward
 * pass did not take from them: INTEGRATION.md).
 * ------------------------------------------------------------------------ */
size_t msmd_rulebook_conv_workspace_bytes(int batch_size,
                                          const int* out_shape /* host[3] */);

int msmd_rulebook_conv3d_count(const int32_t* indices /* [n,4] */, int n,
                               int batch_size, const int* out_shape,
                               const int* ksize, const int* stride,
                               const int* padding, int32_t* n_out /* [1] */,
                               void* workspace, size_t workspace_bytes,
                               msmd_stream_t stream);

int msmd_rulebook_conv3d_fill(const int32_t* indices, int n, int batch_size,
                              const int* out_shape, const int* ksize,
                              const int* stride, const int* padding, int n_out,
                              int32_t* out_indices /* [n_out,4] */,
                              int32_t* nbr_fwd /* [K,n_out] in-row or -1 */,
                              int32_t* nbr_bwd /* [K,n]    out-row or -1 */,
                              void* workspace, size_t workspace_bytes,
                              msmd_stream_t stream);

/* A chain of strided convs whose input set is the previous one's output set
 * (SparseEncoder: sparse_encoder.py:175-187, only SubM convs in between): all levels
 * counted back to back -- level l+1 is marked from level l's occupancy bitmap -- so the
 * host reads n_out[0..levels) ONCE instead of once per level (the reference's
 * getIndicePairs syncs per conv).  out_shapes / ksizes / strides / paddings: [levels][3].
 * The workspace is the concatenation of the per-level msmd_rulebook_conv_workspace_bytes
 * blocks (each rounded up to 256 bytes): afterwards msmd_rulebook_conv3d_fill runs per
 * level on its own block, level l's input rows being level l-1's out_indices. */
size_t msmd_rulebook_conv_chain_workspace_bytes(int batch_size, int levels,
                                                const int* out_shapes);
int msmd_rulebook_conv3d_count_chain(const int32_t* indices, int n, int batch_size,
                                     int levels, const int* out_shapes, const int* ksizes,
                                     const int* strides, const int* paddings,
                                     int32_t* n_out /* [levels] */, void* workspace,
                                     size_t workspace_bytes, msmd_stream_t stream);

/* The same idea for the fusion stack's stage chain (sparse_multimodal_encoder_painting.py:
 * 413-428): level l's strided conv takes the UNION (sparse_add) of an extra voxel set
 * extra[l] (host array of device pointers, [n_extra[l], 4] each) and the previous level's
 * output set; level 0 takes extra[0] as it is.  counts[2l] = |union_l| (l >= 1),
 * counts[2l + 1] = |out_l|, all counted back to back: ONE host read for the whole chain.
 * After the read the caller fills level by level with msmd_sparse_add_fill (workspace =
 * level l's union region) and msmd_rulebook_conv3d_fill (its conv region): the regions lie
 * in `workspace` in the order conv_0, union_1, conv_1, union_2, ..., each
 * align256(msmd_rulebook_conv_workspace_bytes(batch, its grid)) bytes
 * (msmd_sparse_add_workspace_bytes is the same layout).  in_shapes[l] = out_shapes[l - 1]. */
size_t msmd_rulebook_add_conv_chain_workspace_bytes(int batch_size, int levels,
                                                    const int* in_shapes, const int* out_shapes);
int msmd_rulebook_add_conv_count_chain(const int32_t* const* extra, const int* n_extra,
                                       int batch_size, int levels, const int* in_shapes,
                                       const int* out_shapes, const int* ksizes,
                                       const int* strides, const int* paddings, int32_t* counts,
                                       void* workspace, size_t workspace_bytes,
                                       msmd_stream_t stream);

/* nbr table -> reference rulebook format: indice_pairs[K,2,ld] (-1 padded,
 * pairs of one offset sorted by output row) and indice_num[K]
 * (spconv_ops.h:55-59).  n_rows = number of output rows of `nbr`. */
size_t msmd_rulebook_pairs_workspace_bytes(int kernel_volume, int n_rows);

int msmd_rulebook_pairs(const int32_t* nbr /* [K,n_rows] */, int kernel_volume,
                        int n_rows, int32_t* indice_pairs /* [K,2,ld] */,
                        int ld, int32_t* indice_num /* [K] */, void* workspace,
                        size_t workspace_bytes, msmd_stream_t stream);

/* ------------------------------------------------------------------------ *
 * a7/a8  Sparse convolution arithmetic (implicit GEMM on MFMA)
 * replaces: sparse_conv_ext.indice_conv_fp32 / indice_conv_backward_fp32
 *           all.cc:28-51, spconv_ops.h:260-456 (+ reordering.cc:20-50);
 *           spconv-2.x Fsp.implicit_gemm (bug_fix/conv.py:441-447).
 *   out[o,:] = sum_k in[nbr[k,o],:] @ W[k]          (W[k] is [c_in,c_out])
 * `weight` is plain [K,c_in,c_out] fp32; msmd_spconv_pack_weight() converts
 * it (optionally transposing each W[k]) to the MFMA fragment order the
 * kernels read; packed size is K*round16(c_in)*round16(c_out) floats.
 * dgrad = the same kernel on the backward table with transposed weights.
 * `weight_flip` != 0 pairs table row k with weight K-1-k: a SubM (odd kernel)
 * forward table read that way IS its backward table, so SubM dgrad needs no
 * second rulebook.
 * ------------------------------------------------------------------------ */
size_t msmd_spconv_packed_weight_elems(int kernel_volume, int c_in, int c_out);

int msmd_spconv_pack_weight(const float* weight, int kernel_volume, int c_in,
                            int c_out,
                            int flags /* bit0: pack W[k]^T (dgrad);
                                         bit1: weight is KRSC [c_out,K,c_in]
                                               (bug_fix/conv.py:114-117) instead
                                               of [K,c_in,c_out] */,
                            float* packed, msmd_stream_t stream);

/* `row_order` (NULL or a permutation of [0,n_out)): the order in which output
 * rows are tiled.  Any permutation gives bit-identical results; sorting rows
 * by msmd_rulebook_row_masks() lets whole waves / workgroups skip the kernel
 * offsets none of their rows is connected through. */
int msmd_spconv_fwd_f32(const float* in_feat /* [n_in,c_in] */, int n_in,
                        int c_in, const float* packed_weight,
                        const int32_t* nbr /* [K,ld] */, int ld, int n_out,
                        int kernel_volume, int weight_flip,
                        const int32_t* row_order /* [n_out] or NULL */,
                        int32_t* tile_counter /* [1] scratch or NULL */,
                        float* out_feat /* [n_out,c_out] */, int c_out,
                        msmd_stream_t stream);
/* tile_counter != NULL selects the persistent form: a fixed number of resident
 * workgroups draw row tiles from *tile_counter -- with a heaviest-first
 * row_order this balances the very uneven per-tile cost.  *tile_counter must
 * be 0 on entry; the kernel's last draw puts it back to 0, so one zeroed word
 * per stream serves every launch (no memset between launches). */

/* ---- the same convolution at bf16 MFMA rate, fp32-equivalent results -------
 * Every fp32 operand is the exact sum of three bf16 values (h + m + l); six
 * bf16 products accumulated in fp32 reproduce the fp32 product to ~2^-23 (the
 * dropped cross terms are below fp32's own rounding), at 2.7x fewer matrix-core
 * cycles than the fp32 MFMA.  `planes` = 3 is that mode; 2 keeps three products
 * (relative error ~2^-17); 1 is plain bf16 operands.  Features are read as fp32
 * and split in registers; weights are split by the pack call.
 * Covers c_in % 8 == 0 and c_out % 4 == 0, both >= 32, K <= 32 (ask
 * msmd_spconv_fwd_split_supported; c_out > 128 runs in column passes); other
 * layers use msmd_spconv_fwd_f32.
 * With `row_order`, `nbr` must be in TILE order: nbr[k][p] refers to output row
 * row_order[p] (msmd_rulebook_permute_cols).  `tile_counter` is required.      */
int msmd_spconv_fwd_split_supported(int c_in, int c_out, int kernel_volume);

size_t msmd_spconv_packed_split_bytes(int kernel_volume, int c_in /* contraction */,
                                      int c_out /* outputs */, int planes);

int msmd_spconv_pack_weight_split(const float* weight, int kernel_volume, int c_in,
                                  int c_out, int flags /* as msmd_spconv_pack_weight */,
                                  int planes, void* packed, msmd_stream_t stream);
/* The same plus the image of the opposite transposition (flags ^ 1), one launch:
 * forward and dgrad of a conv read one each. */
int msmd_spconv_pack_weight_split_pair(const float* weight, int kernel_volume, int c_in,
                                       int c_out, int flags, int planes, void* packed,
                                       void* packed_transposed, msmd_stream_t stream);

/* Several weights in one launch (a training step repacks every trained conv's weight after
 * the optimizer has moved it: one launch instead of one per conv).  descs: n_desc descriptors
 * in DEVICE memory, 48 bytes each
 *   { const float* weight; void* packed; void* packed_transposed (or NULL); int64_t start;
 *     int32_t kernel_volume, c_in, c_out, flags; }
 * with start = the running count of work units in front of this weight (descriptor order; a
 * weight's units = msmd_spconv_packed_split_bytes / (16 * planes) of each image it writes) and
 * total_units their sum.  Images as msmd_spconv_pack_weight_split[_pair]. */
int msmd_spconv_pack_weight_split_many(const void* descs, int n_desc, long total_units,
                                       int planes, msmd_stream_t stream);

/* `tile_counter`: `sync_ints` zeroed int32 owned by the stream ([0] = the tile
 * counter, [1 + t] = exchange flag of row tile t); every launch leaves all of them
 * at 0 again.  `workspace` (msmd_spconv_fwd_split_workspace_bytes; may be NULL):
 * exchange buffer that lets the heaviest 128-row tiles run as two scheduling units
 * -- used when sync_ints >= 1 + ceil(n_out / 128) and the row tiles alone would
 * leave the workgroup slots unbalanced; without it every tile is one unit.  The
 * sum order per output element is fixed either way (deterministic), but differs
 * between the two modes in the last bit.                                        */
size_t msmd_spconv_fwd_split_workspace_bytes(int n_out, int c_out);

int msmd_spconv_fwd_split(const float* in_feat /* [n_in,c_in] */, int n_in, int c_in,
                          const void* packed_weight, const int32_t* nbr /* [K,ld] */,
                          int ld, int n_out, int kernel_volume, int weight_flip,
                          const int32_t* row_order /* [n_out] or NULL */,
                          int32_t* tile_counter /* [sync_ints] */, int sync_ints,
                          float* out_feat /* [n_out,c_out] */, int c_out, int planes,
                          void* workspace, size_t workspace_bytes,
                          const int32_t* tile_prefix /* msmd_rulebook_tile_prefix of `nbr`,
                                                        or NULL: dynamic tile scheduler */,
                          msmd_stream_t stream);

/* The same call that also leaves, per row tile of the output (128 or 256 rows:
 * msmd_spconv_fwd_split_tile_rows(c_out)), the column sums and sums of squares of the rows
 * it wrote: bn_partials[msmd_spconv_fwd_split_stats_blocks(n_out, c_out)][2][c_out] -- the statistics
 * pass of the BatchNorm1d that follows a conv in make_sparse_convmodule / SparseBasicBlock
 * (mmdet3d/ops/sparse_block.py:87-117,161-190) without reading the output again
 * (msmd_bn_act_fwd_from_partials_f32 takes them).  bn_partials = NULL: msmd_spconv_fwd_split. */
int msmd_spconv_fwd_split_stats(const float* in_feat, int n_in, int c_in,
                                const void* packed_weight, const int32_t* nbr, int ld,
                                int n_out, int kernel_volume, int weight_flip,
                                const int32_t* row_order, int32_t* tile_counter, int sync_ints,
                                float* out_feat, int c_out, int planes, void* workspace,
                                size_t workspace_bytes, const int32_t* tile_prefix,
                                float* bn_partials, msmd_stream_t stream);
int msmd_spconv_fwd_split_stats_blocks(int n_out, int c_out);

/* Rows per tile the split kernel uses for a layer of c_out output channels: 256 (the
 * ping-pong form: 8 waves, one workgroup and one weight stream per CU; 161..192 channels as
 * one 12-tile pass) from 161 channels up, 128 below -- the rows_per_tile to compute its
 * tile_prefix with.  Environment: MSMD_FWD_PP_MIN moves the threshold, MSMD_FWD_PP=0 = 128
 * everywhere. */
int msmd_spconv_fwd_split_tile_rows(int c_out);

/* Which instantiation of the split kernel msmd_spconv_fwd_split launches for a layer of c_out
 * output channels (under the current environment): params[7] = {NT column tiles per pass,
 * UB units per item, waves per workgroup, weight buffers, ping-pong 0|1, table buffers,
 * column passes = kernel launches per call}.  rocprofv3 names the kernel
 * spconv_fwd_split_kernel<NT, UB, planes, waves, buffers, pingpong, tables>.  For tools
 * (bench.py's roofline grouping, tools/split_bench.py); replaces nothing in the reference. */
int msmd_spconv_fwd_split_instantiation(int c_out, int* params);

/* Stream-K work table of a neighbour table (in the order the conv kernel tiles it):
 * prefix[t] = number of (row tile, active offset) work items before tile t, prefix[n_tiles]
 * = all of them (n_tiles = ceil(n_rows / rows_per_tile); an empty tile counts 1).  With it
 * msmd_spconv_fwd_split gives every workgroup the same share of the launch (see
 * csrc/spconv_split.hip); no reference counterpart. */
int msmd_rulebook_tile_prefix(const int32_t* nbr /* [K,ld] */, int kernel_volume, int ld,
                              int n_rows, int rows_per_tile,
                              int32_t* prefix /* [n_tiles + 1] */, msmd_stream_t stream);

/* wgrad with the same operand splitting (both operands are read as fp32 and split
 * into bf16 planes on the way to the matrix cores); c_in, c_out >= 64 and multiples of 4.
 * Widths that are multiples of 16 take the whole-block kernel (csrc/spconv_wgrad_block.hip:
 * one workgroup owns up to 128 x 128 channels of an offset, producer waves fetch and split
 * each pair's rows once, consumer waves only multiply); the others 64 x 64 slabs.
 * replaces: sparse_conv_ext.indice_conv_backward_fp32's filter-gradient half
 *           (mmdet3d/ops/spconv/include/spconv/spconv_ops.h:363-456).
 * Workspace as msmd_spconv_wgrad_workspace_bytes. */
int msmd_spconv_wgrad_split_supported(int c_in, int c_out);

int msmd_spconv_wgrad_split(const float* in_feat, int c_in, const float* d_out, int c_out,
                            const int32_t* indice_pairs /* [K,2,ld] */,
                            const int32_t* indice_num /* [K] device */, int ld,
                            int kernel_volume, int planes,
                            float* d_weight /* [K,c_in,c_out] */,
                            int krsc_out /* != 0: d_weight is [c_out,K,c_in] */,
                            void* workspace, size_t workspace_bytes, msmd_stream_t stream);

/* The same with the whole-block kernel's step sequence in ROW-CHUNK-major order: the pairs of
 * every offset are cut where their OUTPUT row crosses a multiple of chunk_rows, and the 27
 * pieces of a chunk are processed next to each other (same XCD, same time), so that a chunk's
 * d_out rows are fetched from HBM once for all offsets instead of once per offset
 * (csrc/spconv_wgrad_block.hip).  seg_table = msmd_rulebook_pair_segments' output for these
 * pair lists (index data: once per rulebook), n_chunks its chunk count; NULL = one chunk (the
 * plain offset-major sequence, what msmd_spconv_wgrad_split runs).  Same sums per (offset,
 * channel pair) in another fixed order: results agree to fp32 rounding, run to run identical.
 * Workspace: msmd_spconv_wgrad_segments_workspace_bytes.
 * msmd_rulebook_pair_segments_ints returns 0 for kernel volumes the whole-block kernel does
 * not take (> 64, e.g. 5x5x5): build no table, pass seg_table = NULL, the slab kernel runs. */
size_t msmd_rulebook_pair_segments_ints(int kernel_volume, int n_chunks);
int msmd_rulebook_pair_segments(const int32_t* indice_pairs /* [K,2,ld] */,
                                const int32_t* indice_num /* [K] device */, int ld,
                                int kernel_volume, int chunk_rows, int n_chunks /* <= 256,
                                chunk_rows * n_chunks >= ld */,
                                int32_t* table /* [msmd_rulebook_pair_segments_ints] */,
                                msmd_stream_t stream);
size_t msmd_spconv_wgrad_segments_workspace_bytes(int kernel_volume, int ld, int c_in, int c_out,
                                                  int n_chunks);
int msmd_spconv_wgrad_split_segments(const float* in_feat, int c_in, const float* d_out,
                                     int c_out, const int32_t* indice_pairs,
                                     const int32_t* indice_num, int ld, int kernel_volume,
                                     int planes, float* d_weight, int krsc_out,
                                     const int32_t* seg_table, int n_chunks, void* workspace,
                                     size_t workspace_bytes, msmd_stream_t stream);

/* out[k][p] = nbr[k][order[p]] for p < n: the neighbour table in tile order. */
int msmd_rulebook_permute_cols(const int32_t* nbr, int kernel_volume, int ld, int n,
                               const int32_t* order, int32_t* out /* [K,n] */,
                               msmd_stream_t stream);

/* masks[o] = bitset over k of (nbr[k][o] >= 0); sort_keys[o]: ascending order makes
 * rows with similar masks adjacent (3x3x3: the mask with its bits ranked centre <
 * faces < edges < corners; other volumes: heaviest mask first).  Either output may be
 * NULL.  kernel_volume <= 64.
 * msmd_rulebook_tile_costs: cost[t] = K - |union of the masks of rows
 * order[t*rows_per_tile ..)| -- ascending = the tile sequence (heaviest first) the
 * persistent conv kernels balance best on; `order` may be NULL (natural order).
 * Both only choose a tiling: conv results do not depend on it. */
/* The whole tiling of a table in one call: order[p] = output row at tile position p
 * (rows sorted by msmd_rulebook_row_masks' key, full tiles re-sequenced by
 * msmd_rulebook_tile_costs, a partial last tile stays last) and, if `tiled` is not
 * NULL, tiled[k][p] = nbr[k][order[p]] (= msmd_rulebook_permute_cols).  In-library
 * radix sorts over the key's significant bits; kernel_volume <= 31, nbr is [K,n_rows]. */
size_t msmd_rulebook_tiling_workspace_bytes(int n_rows, int rows_per_tile);
int msmd_rulebook_tiling(const int32_t* nbr, int kernel_volume, int n_rows,
                         int rows_per_tile, int32_t* order /* [n_rows] */,
                         int32_t* tiled /* [K,n_rows] or NULL */, void* workspace,
                         size_t workspace_bytes, msmd_stream_t stream);

/* msmd_rulebook_tiling + msmd_rulebook_tile_prefix (128- and/or 256-row tiles; NULL to
 * skip) + msmd_rulebook_pairs (indice_pairs NULL to skip) of one table in one call: same
 * results, one host call instead of four (the index pass is host-bound). */
size_t msmd_rulebook_plan_workspace_bytes(int kernel_volume, int n_rows, int rows_per_tile);
int msmd_rulebook_plan(const int32_t* nbr, int kernel_volume, int n_rows,
                       int rows_per_tile, int32_t* order, int32_t* tiled,
                       int32_t* prefix128, int32_t* prefix256, int32_t* indice_pairs,
                       int ld, int32_t* indice_num, void* workspace,
                       size_t workspace_bytes, msmd_stream_t stream);

/* msmd_rulebook_plan (+ the one-chunk msmd_rulebook_pair_segments table) for MANY tables in
 * one launch set: an index pass plans every table of the step at its end -- nothing in the
 * index chain reads a plan -- with 7 kernels and one radix sort whatever the number of
 * tables (table id = the key's top bits; stable, so each table's order is its own sort's).
 * descs: HOST array.  Per table: order is required, tiled / prefix128 / prefix256 /
 * indice_pairs (+ indice_num) / segtab optional (NULL); prefixes need tiled, segtab needs
 * indice_pairs; ld >= n_rows.  Results identical to the single calls.  Tables the launch
 * set cannot take (kernel volumes whose key needs more than 27 bits, empty tables,
 * MSMD_TILE_LPT=1) run through the single calls inside. */
typedef struct msmd_plan_desc {
  const int32_t* nbr;        /* [K, n_rows] */
  int32_t kvol, n_rows;
  int32_t* order;            /* [n_rows] */
  int32_t* tiled;            /* [K, n_rows] */
  int32_t* prefix128;        /* [ceil(n_rows / 128) + 1] */
  int32_t* prefix256;        /* [ceil(n_rows / 256) + 1] */
  int32_t* indice_pairs;     /* [K, 2, ld] */
  int32_t* indice_num;       /* [K] */
  int32_t* segtab;           /* [msmd_rulebook_pair_segments_ints(K, 1)] */
  int32_t ld, reserved;
} msmd_plan_desc;
size_t msmd_rulebook_plan_many_workspace_bytes(const msmd_plan_desc* descs, int n_desc);
int msmd_rulebook_plan_many(const msmd_plan_desc* descs, int n_desc, void* workspace,
                            size_t workspace_bytes, msmd_stream_t stream);

int msmd_rulebook_tile_costs(const int32_t* nbr /* [K,n_rows] */, int kernel_volume,
                             int n_rows, const int32_t* order, int rows_per_tile,
                             int32_t* cost /* [ceil(n_rows / rows_per_tile)] */,
                             msmd_stream_t stream);

int msmd_rulebook_row_masks(const int32_t* nbr /* [K,n_rows] */,
                            int kernel_volume, int n_rows, uint64_t* masks,
                            int64_t* sort_keys, msmd_stream_t stream);

/* dW[k] = sum over pairs p of offset k: in[pairs[k,0,p],:]^T (x) dout[pairs[k,1,p],:]
 * (spconv_ops.h:399,438).  Deterministic two-pass reduction. */
size_t msmd_spconv_wgrad_workspace_bytes(int kernel_volume, int ld, int c_in,
                                         int c_out);

int msmd_spconv_wgrad_f32(const float* in_feat, int c_in, const float* d_out,
                          int c_out, const int32_t* indice_pairs /* [K,2,ld] */,
                          const int32_t* indice_num /* [K] device */, int ld,
                          int kernel_volume, float* d_weight /* [K,c_in,c_out] */,
                          int krsc_out /* != 0: d_weight is [c_out,K,c_in] */,
                          void* workspace, size_t workspace_bytes,
                          msmd_stream_t stream);

/* ------------------------------------------------------------------------ *
 * a9/a10  BatchNorm1d (+ residual) (+ ReLU) on sparse-tensor features [n,c]
 * replaces: the nn.BatchNorm1d + nn.ReLU(inplace) pair make_sparse_convmodule
 *           appends to every sparse conv (mmdet3d/ops/sparse_block.py:161-190)
 *           and SparseBasicBlock's relu(bn2(conv2(x)) + identity) (:103-126).
 * training != 0: batch statistics (biased variance for normalisation,
 * running_var updated with the unbiased one, torch semantics; running_* may be
 * NULL = track_running_stats False, tools/train.py:205-211); else running stats.
 * save_mean / save_invstd [c] are outputs the backward needs.  c % 4 == 0.
 * ------------------------------------------------------------------------ */
size_t msmd_bn_workspace_bytes(int n, int c);

int msmd_bn_act_fwd_f32(const float* x /* [n,c] */, const float* residual /* or NULL */,
                        int n, int c, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, int training,
                        float momentum, float eps, int relu, float* y,
                        float* save_mean, float* save_invstd, void* workspace,
                        size_t workspace_bytes, msmd_stream_t stream);

/* Training-mode forward whose statistics pass was done by the producer of x:
 * partials[n_partials][2][c] = column sums / sums of squares of disjoint row blocks covering x
 * (msmd_spconv_fwd_split_stats).  Same outputs and running-stat update as msmd_bn_act_fwd_f32. */
int msmd_bn_act_fwd_from_partials_f32(const float* x, const float* residual, int n, int c,
                                      const float* gamma, const float* beta,
                                      float* running_mean, float* running_var, float momentum,
                                      float eps, int relu, float* y, float* save_mean,
                                      float* save_invstd, const float* partials, int n_partials,
                                      msmd_stream_t stream);

int msmd_bn_act_bwd_f32(const float* x, const float* y /* fwd output, for the ReLU mask */,
                        const float* dy, int n, int c, const float* gamma,
                        const float* save_mean, const float* save_invstd,
                        int training, int relu, float* dx,
                        float* dresidual /* or NULL */, float* dgamma, float* dbeta,
                        void* workspace, size_t workspace_bytes, msmd_stream_t stream);
/* BatchNorm + ReLU WITHOUT a residual, backward without y: the ReLU mask is recomputed from x
 * with the forward pass's own arithmetic (bit-identical decision), so the two streaming passes
 * read one array less each.  Results = msmd_bn_act_bwd_f32(relu = 1) given the forward's y. */
int msmd_bn_relu_bwd_f32(const float* x, const float* dy, int n, int c, const float* gamma,
                         const float* beta, const float* save_mean, const float* save_invstd,
                         int training, float* dx, float* dgamma, float* dbeta,
                         void* workspace, size_t workspace_bytes, msmd_stream_t stream);

/* ------------------------------------------------------------------------ *
 * a12  SparseConvTensor.dense(): BEV scatter
 * replaces: scatter_nd + permute + contiguous
 *           mmdet3d/ops/spconv/structure.py:5-18,55-64
 * out is [B,C,D,H,W] contiguous (channels-first) and is fully written
 * (zero-filled then scattered) by this call.  The backward gathers.
 * ------------------------------------------------------------------------ */
int msmd_dense_scatter_f32(const float* feat /* [n,c] */,
                           const int32_t* indices /* [n,4] */, int n, int c,
                           int batch_size, const int* spatial_shape,
                           float* out, msmd_stream_t stream);
int msmd_dense_gather_f32(const float* dense /* [B,C,D,H,W] */,
                          const int32_t* indices, int n, int c, int batch_size,
                          const int* spatial_shape, float* feat /* [n,c] */,
                          msmd_stream_t stream);

/* ------------------------------------------------------------------------ *
 * a12 -> f1  channels-last BEV hand-over
 * replaces: stage_outs[-1].dense(); .view(N, C*D, H, W); torch.cat([x, x_mm], 1)
 *           mmdet3d/models/detectors/MSMDFusion.py:436-440 (and
 *           mmdet3d/models/middle_encoders/sparse_encoder.py:187-190 for x)
 * bev is ONE [B,H,W,total_channels] buffer (the NHWC image of the reference's
 * [B, total_channels, H, W] tensor) the caller has zero-filled; a tensor with c
 * channels and depth D fills channels [channel_offset, channel_offset + c*D),
 * channel index c_i*D + z as .view(N, C*D, H, W) orders them.  gather = backward.
 * ------------------------------------------------------------------------ */
int msmd_bev_scatter_nhwc_f32(const float* feat /* [n,c] */,
                              const int32_t* indices /* [n,4] b,z,y,x */,
                              int n, int c, int batch_size,
                              const int* spatial_shape /* D,H,W */, float* bev,
                              int total_channels, int channel_offset,
                              msmd_stream_t stream);
int msmd_bev_gather_nhwc_f32(const float* bev, const int32_t* indices, int n,
                             int c, int batch_size, const int* spatial_shape,
                             float* feat /* [n,c] */, int total_channels,
                             int channel_offset, msmd_stream_t stream);

/* ------------------------------------------------------------------------ *
 * a13 (image half)  MSMDFusionDetector.get_foreground2D, gather part
 * replaces: the per-(sample, camera) loop -- (fg_pxl * downscale).long(),
 *           img_feat.permute(1,2,0)[coord_h, coord_w], the torch.cat calls
 *           mmdet3d/models/detectors/MSMDFusion.py:195-224
 * All cameras of all samples in one call.  img_feat is [planes, c, h, w] with
 * arbitrary element strides (strides[4] = plane, channel, row, column), planes =
 * B * cameras.  pixels is [n,3] (x, y, depth) in float32 or float64, the numpy
 * dtype the loader produced: the product with `downscale` is taken in that
 * dtype and truncated toward zero like .long().  Negative cells wrap like
 * torch indexing; cells outside [-size, size) are COUNTED in *n_bad (the
 * reference raises IndexError there; the Python mirror raises on n_bad != 0)
 * and read as zeros.
 *   fg_pcd   [n, pts_dim + c]  = [pts | feat]
 *   score_in [n, c + 17]       = [feat | depth | lidar2img[plane] (16)]
 *   cells    [n] (optional)    = linear (plane, h, w) cell per point, -1 = bad;
 *                                input of msmd_fg_scatter_add_f32 (the backward)
 * ------------------------------------------------------------------------ */
int msmd_fg_gather_f32(const float* img_feat, const int64_t* strides,
                       int planes, int c, int h, int w, const void* pixels,
                       int pixel_is_f64, const int32_t* plane /* [n] */,
                       double downscale, const float* pts /* [n,pts_dim] */,
                       int pts_dim, const float* lidar2img /* [planes,16] */,
                       int n, float* fg_pcd, float* score_in, int32_t* cells,
                       int32_t* n_bad, msmd_stream_t stream);
/* The same gather with get_foreground2D's tail folded in, for the no-gradient case (the
 * reference's own training step: the result feeds voxelize(), which is @torch.no_grad(),
 * MSMDFusion.py:462): score = ReLU([feat | depth | lidar2img[plane]] . score_weight +
 * score_bias) (score_net = Linear(c + 17, 1) + ReLU, :125-128, :225-227) and
 *   fg_pcd [n, pts_dim + c] = [pts | feat * score]   for points [0, n_scaled),
 *                             [pts | feat]           for the rest (:229-234 copies the
 * scaled channels back for sample 0, and sample 1 when B == 2, only: reference_quirks).
 * score_in is never materialised.  c <= 64.  The dot product is summed in a fixed order. */
int msmd_fg_gather_scored_f32(const float* img_feat, const int64_t* strides,
                              int planes, int c, int h, int w, const void* pixels,
                              int pixel_is_f64, const int32_t* plane /* [n] */,
                              double downscale, const float* pts /* [n,pts_dim] */,
                              int pts_dim, const float* lidar2img /* [planes,16] */,
                              const float* score_weight /* [c + 17] */,
                              const float* score_bias /* [1] */, int n, int n_scaled,
                              float* fg_pcd, int32_t* n_bad, msmd_stream_t stream);
/* grad_img[plane, :, h, w] += grad[i, col0 : col0 + c]   (atomic fp32 adds: the
 * order, hence the last bit, is not fixed -- as index_put(accumulate) on a GPU) */
int msmd_fg_scatter_add_f32(const float* grad, int grad_stride, int col0,
                            const int32_t* cells, int n, int planes, int c,
                            int h, int w, const int64_t* strides,
                            float* grad_img, msmd_stream_t stream);

/* ------------------------------------------------------------------------ *
 * f2  depth_aware_channel_compression, sparse depth canvas
 * replaces: canvas[i,j].index_put_((y, x), depth) for B x 6 cameras
 *           mmdet3d/models/detectors/MSMDFusion.py:336-356
 * canvas is [planes, h, w], fully written.  Where several pixels land on one
 * cell the row with the highest index wins (what a sequential index_put_
 * leaves; the reference's CUDA index_put_ is unordered there).
 * ------------------------------------------------------------------------ */
size_t msmd_depth_canvas_workspace_bytes(int planes, int h, int w);
int msmd_depth_canvas_f32(const void* pixels /* [n,3] x,y,depth */,
                          int pixel_is_f64, const int32_t* plane, int n,
                          int planes, int h, int w, float* canvas,
                          int32_t* n_bad, void* workspace,
                          size_t workspace_bytes, msmd_stream_t stream);

/* ------------------------------------------------------------------------ *
 * a17  spconv.pytorch.functional.sparse_add(a, b)
 * call site: mmdet3d/models/middle_encoders/sparse_multimodal_encoder_painting.py:455
 * Union of two coordinate sets on one grid, rows in ascending linear id,
 * features summed where both are present.  Two phases like the strided
 * rulebook; map_a/map_b give each input row's output row (for autograd).
 * ------------------------------------------------------------------------ */
size_t msmd_sparse_add_workspace_bytes(int batch_size, const int* spatial_shape);

int msmd_sparse_add_count(const int32_t* idx_a, int n_a, const int32_t* idx_b,
                          int n_b, int batch_size, const int* spatial_shape,
                          int32_t* n_out /* [1] */, void* workspace,
                          size_t workspace_bytes, msmd_stream_t stream);

int msmd_sparse_add_fill(const float* feat_a, const int32_t* idx_a, int n_a,
                         const float* feat_b, const int32_t* idx_b, int n_b,
                         int c, int batch_size, const int* spatial_shape,
                         int n_out, int32_t* out_indices /* [n_out,4] */,
                         float* out_feat /* [n_out,c] */,
                         int32_t* map_a /* [n_a] */, int32_t* map_b /* [n_b] */,
                         void* workspace, size_t workspace_bytes,
                         msmd_stream_t stream);
/* c == 0 makes msmd_sparse_add_fill an index-only pass (out_indices + maps; the
 * feature pointers may be NULL).  msmd_sparse_add_rows is then the feature half
 * on its own: out_feat[map_a[i]] += feat_a[i], out_feat[map_b[j]] += feat_b[j]
 * over a zeroed out_feat -- the same sums as the fused call, bit for bit (at
 * most two addends per element).  Together they let a caller that knows the
 * coordinate sets ahead of the features (step pipelining) keep the host read of
 * n_out off the feature pass. */
int msmd_sparse_add_rows(const float* feat_a, const int32_t* map_a, int n_a,
                         const float* feat_b, const int32_t* map_b, int n_b,
                         int c, int n_out, float* out_feat /* [n_out,c] */,
                         msmd_stream_t stream);

/* The same feature half as a gather, without a zero fill and without float atomics on the
 * common path: inv_x[j] = the last row of x that lands on output row j (msmd_rows_inverse of
 * map_x, once per batch); out[j] = a[inv_a[j]] + b[inv_b[j]].  Rows of one tensor that share
 * their coordinates (sparse_add sums those too) are added by a fix-up pass.  c % 4 == 0. */
int msmd_rows_inverse(const int32_t* map, int n, int n_out, int32_t* inv /* [n_out] */,
                      msmd_stream_t stream);
int msmd_sparse_add_rows_gather(const float* feat_a, const int32_t* map_a, const int32_t* inv_a,
                                int n_a, const float* feat_b, const int32_t* map_b,
                                const int32_t* inv_b, int n_b, int c, int n_out, float* out_feat,
                                msmd_stream_t stream);

/* ------------------------------------------------------------------------ *
 * a16  GMA-Conv stage assembly (one launch each way)
 * replaces: the index / cat / pad / mul chain of
 *           SparseMultiModalEncoderPaint.grouped_sparse_conv
 *           mmdet3d/models/middle_encoders/sparse_encoder_multimodal_encoderpaint_double_aware.py:349-421
 * out rows, in this order ([c3 | c2] columns):
 *   n_o3      only-3D rows   [ conv3[i]                       | 0 ]
 *   n_o2      only-2D rows   [ 0 | cross_gate[nn3[i] < 0 ? n3 : nn3[i]] * feat2[rows_o2[i]] ]
 *   n_o2_pad  zero rows (samples without an only-2D voxel, :208-225)
 *   n_mix     mixed rows     [ feat3[rows_m3[i]] | gate[i] * feat2[rows_m2[i]] ]
 *   n_mix_pad zero rows
 * backward: d_conv3 = left block of the only-3D rows, d_gate = right block of the
 * mixed rows * feat2, d_cross_gate[t] = sum over the only-2D rows whose nearest voxel is t
 * of right block * feat2.  With `order` (the only-2D rows sorted by nearest voxel, -1 as
 * n3) and `starts` (starts[t] .. starts[t+1] = target t's slice of `order`) the sum runs
 * in a fixed order, one wave per target (c2 <= 64); without them: a zero fill + float
 * atomics, like the index_add_ it replaces.  feat3 / feat2 receive no gradient.
 * c3, c2 multiples of 4. */
int msmd_gma_assemble_fwd_f32(const float* conv3, int n_o3, int c3, const float* cross_gate,
                              int n3, int c2, const int64_t* nn3, const float* feat2,
                              const int64_t* rows_o2, int n_o2, int n_o2_pad,
                              const float* feat3, const int64_t* rows_m3, const float* gate,
                              const int64_t* rows_m2, int n_mix, int n_mix_pad,
                              float* out /* [rows, c3 + c2] */, msmd_stream_t stream);
int msmd_gma_assemble_bwd_f32(const float* d_out, int n_o3, int c3, int n3, int c2,
                              const int64_t* nn3, const float* feat2, const int64_t* rows_o2,
                              int n_o2, int n_o2_pad, const int64_t* rows_m2, int n_mix,
                              int n_mix_pad, float* d_conv3 /* [n_o3,c3] */,
                              float* d_cross_gate /* [n3+1,c2] or NULL */,
                              float* d_gate /* [n_mix,c2] or NULL */,
                              const int64_t* order /* [n_o2] or NULL */,
                              const int64_t* starts /* [n3+2] or NULL */,
                              float* workspace /* ..._workspace_floats(c2) floats, with order */,
                              msmd_stream_t stream);
size_t msmd_gma_assemble_bwd_workspace_floats(int c2);

/* a16  the stage's gate tables: gate_control / cross_gate_control =
 * nn.Sequential(nn.Linear(c3, 64), nn.ReLU()) applied to the rows of the 3-D voxels
 * replaces: torch's Linear + ReLU (hipBLASLt GEMM + elementwise) at
 *           sparse_multimodal_encoder_painting.py:83-96 (definition), :398-417 (use)
 * y[n + n_tail, c_out] = relu?(cat(x, x_tail) w^T + b), w = nn.Linear's [c_out, c_in]
 * (x_tail: the dummy embedding row the reference concatenates, no copy of x needed);
 * backward: dw = g^T x, db = column sums of g, g = dy where y > 0 -- per-block partials
 * added in block order (deterministic).  c_out in {32, 64, 128}, c_in % 4 = 0
 * (msmd_rows_linear_supported). */
int msmd_rows_linear_supported(int c_in, int c_out);
int msmd_rows_linear_fwd_f32(const float* x, int n, const float* x_tail, int n_tail, int c_in,
                             const float* w, const float* b /* or NULL */, int c_out, int relu,
                             float* y, msmd_stream_t stream);
size_t msmd_rows_linear_bwd_workspace_bytes(int n_total, int c_in, int c_out);
int msmd_rows_linear_bwd_f32(const float* x, int n, const float* x_tail, int n_tail, int c_in,
                             const float* y, const float* dy, int c_out, int relu,
                             float* dw /* [c_out,c_in] */, float* db /* [c_out] or NULL */,
                             void* workspace, size_t workspace_bytes, msmd_stream_t stream);

/* ------------------------------------------------------------------------ *
 * a14  voxel_modality_split: LiDAR voxels vs virtual-point voxels
 * replaces: MSMDFusionDetector.voxel_modality_split + numba type_assign
 *           mmdet3d/models/detectors/MSMDFusion.py:27-45,251-325
 * mix3d[i]/mix2d[j] = 1 when the voxel exists in both sets (same b,z,y,x).
 * pair_3d/pair_2d list the matched rows, aligned, in ascending linear id;
 * *n_mixed (device) is their count.  Keys are exact integers (the reference's
 * float32 keys alias, SURVEY Appendix B.3 -- deliberate, documented fix).
 * ------------------------------------------------------------------------ */
size_t msmd_modality_split_workspace_bytes(int batch_size,
                                           const int* spatial_shape);

int msmd_modality_split(const int32_t* idx_3d /* [n3,4] */, int n3,
                        const int32_t* idx_2d /* [n2,4] */, int n2,
                        int batch_size, const int* spatial_shape,
                        int32_t* mix3d /* [n3] */, int32_t* mix2d /* [n2] */,
                        int32_t* pair_3d /* [min(n3,n2)] */,
                        int32_t* pair_2d /* [min(n3,n2)] */,
                        int32_t* n_mixed /* [1] */, void* workspace,
                        size_t workspace_bytes, msmd_stream_t stream);
/* The same, plus per-sample row counts in the same pass (what the detector and
 * the fusion stack otherwise obtain from boolean masks with a host round trip
 * each: only_3D / only_2D selections MSMDFusion.py:251-325 callers,
 * pad_missing_batch_id sparse_multimodal_encoder_painting.py:208-225, the
 * per-sample split of fps_NN_fast :349-369).
 * sample_stats[4*batch_size] = rows per sample of [3D unmatched | 3D matched |
 * 2D unmatched | 2D matched]; rows of a sample need not be contiguous. */
int msmd_modality_split_stats(const int32_t* idx_3d, int n3,
                              const int32_t* idx_2d, int n2, int batch_size,
                              const int* spatial_shape, int32_t* mix3d,
                              int32_t* mix2d, int32_t* pair_3d, int32_t* pair_2d,
                              int32_t* n_mixed, int32_t* sample_stats,
                              void* workspace, size_t workspace_bytes,
                              msmd_stream_t stream);

/* a14 with the REFERENCE's float32 keys (`reference_quirks=True` of the detector): key =
 * fl(fl(fl(z) * 1e6 + fl(y) * 1e3) + fl(x)) (MSMDFusion.py:271-272: int tensor * python float
 * promotes to float32), both sets sorted per sample, the r-th occurrence of a key in one list
 * matched with the r-th in the other (type_assign's two-pointer walk, :27-45; ties in row
 * order).  Keys alias for z >= 17 (2^24 < 17e6) and for x >= 1000: voxels that merely share a
 * rounded key are marked mixed -- what a checkpoint trained with the reference has seen.
 * pair_3d / pair_2d: matched rows in key order per sample, samples concatenated (:262-318).
 * reference_offsets != 0: pair rows numbered as the reference does (:288-289,313-314:
 * position in the sample + the PREVIOUS sample's count only; identical to global rows for
 * batch <= 2; needs each set's rows grouped by sample, ascending).  sample_stats (or NULL) as
 * msmd_modality_split_stats.  Grids whose largest key (in the kernel's own float32
 * arithmetic) reaches 2^26, or batch_size > 64: MSMD_ERR_RANGE.  Checked on the device, as the
 * rows are keyed: a row outside the grid or the batch (negative coordinates included), and --
 * with reference_offsets -- a row whose sample id is below its predecessor's; either sets
 * *n_mixed = -1 (every other output is then unspecified) instead of a count. */
size_t msmd_modality_split_float_keys_workspace_bytes(int n3, int n2, int batch_size);
int msmd_modality_split_float_keys(const int32_t* idx_3d, int n3, const int32_t* idx_2d, int n2,
                                   int batch_size, const int* spatial_shape, int32_t* mix3d,
                                   int32_t* mix2d, int32_t* pair_3d, int32_t* pair_2d,
                                   int32_t* n_mixed, int32_t* sample_stats /* or NULL */,
                                   int reference_offsets, void* workspace,
                                   size_t workspace_bytes, msmd_stream_t stream);

/* rows[r] = the r-th i, ascending, with flags[i * stride] == value (at most `capacity` are
 * written; total, if not NULL, receives the count): the row lists `mask.nonzero()` gives the
 * reference (sparse_multimodal_encoder_painting.py:332-340: only_3D / only_2D masks) without
 * the host waiting for the size -- the caller has it from msmd_modality_split_stats.
 * rows[count .. capacity) are set to -1 (torch.nonzero_static's padding). */
size_t msmd_rows_where_workspace_bytes(int n);
int msmd_rows_where_eq(const int32_t* flags, int stride, int n, int value, int64_t* rows,
                       int capacity, int32_t* total, void* workspace, size_t workspace_bytes,
                       msmd_stream_t stream);
/* The same for n_lists (flag vector, value) pairs in one scan: the only-3D and only-2D row lists
 * of all four image scales of a step (host arrays of n_lists pointers / ints).  Per list as
 * msmd_rows_where_eq, the -1 tail included; a list of capacity 0 is skipped. */
size_t msmd_rows_where_eq_many_workspace_bytes(const int* lens, int n_lists);
int msmd_rows_where_eq_many(const int32_t* const* flags, const int* strides, const int* lens,
                            const int* values, int64_t* const* rows, const int* capacities,
                            int n_lists, void* workspace, size_t workspace_bytes,
                            msmd_stream_t stream);

/* ------------------------------------------------------------------------ *
 * a15  GMA-Conv neighbour search helpers
 * replaces: furthest_point_sample_ext.furthest_point_sampling_wrapper
 *           mmdet3d/ops/furthest_point_sample/src/furthest_point_sample_cuda.cu:25-141
 *           ball_query_ext.ball_query_wrapper
 *           mmdet3d/ops/ball_query/src/ball_query_cuda.cu:11-54
 *           and the dense torch.norm/min/index_put_ glue of fps_NN_fast
 *           sparse_multimodal_encoder_painting.py:276-323
 * FPS reproduces the reference block reduction's tie order exactly.
 * ------------------------------------------------------------------------ */
int msmd_furthest_point_sample(const float* xyz /* [b,n,3] */, int b, int n,
                               int m, float* temp /* [b,n] scratch */,
                               int32_t* idx /* [b,m] */, msmd_stream_t stream);

/* Ragged batch: element i owns points [offsets[i], offsets[i+1]) of xyz
 * ([total,3]); idx[i, :] are indices local to the element; n_max = the largest
 * element (host).  One workgroup per element, all elements run concurrently. */
int msmd_furthest_point_sample_ragged(const float* xyz, const int32_t* offsets /* [b+1] device */,
                                      int b, int n_max, int m, float* temp /* [total] */,
                                      int32_t* idx /* [b,m] */, msmd_stream_t stream);

int msmd_ball_query(const float* center_xyz /* [b,m,3] */,
                    const float* xyz /* [b,n,3] */, int b, int n, int m,
                    float min_radius, float max_radius, int nsample,
                    int32_t* idx /* [b,m,nsample] */, msmd_stream_t stream);

/* nearest key per query: out_idx[q] = argmin_k |query_q - key_k| (lowest k on
 * ties) when sqrt(d2) < dist_thresh, else -1.  Integer voxel coordinates.
 * scratch: nq * 8 bytes. */
int msmd_nn_search(const int32_t* query_zyx /* [nq,3] */, int nq,
                   const int32_t* key_zyx /* [nk,3] */, int nk,
                   float dist_thresh, int32_t* out_idx /* [nq] */,
                   void* scratch, msmd_stream_t stream);

/* fps_NN_fast's last step: every query inside the ball of a valid
 * representative inherits that representative's nearest key; when several
 * balls cover a query the highest representative index wins (the order a
 * sequential index_put_ leaves, sparse_multimodal_encoder_painting.py:321). */
int msmd_nn_assign(const int32_t* group_idx /* [m,nsample] ball-query out */,
                   const int32_t* rep_nn /* [m] nearest key or -1 */, int m,
                   int nsample, int nq, int32_t* query_nn /* [nq], -1 init by callee */,
                   int32_t* scratch /* [nq] */, msmd_stream_t stream);

/* ------------------------------------------------------------------------ *
 * f3 (training half)  TransFusionHead.loss: target assignment and heat-map loss
 * replaces: iou3d_cuda.boxes_overlap_bev_gpu(boxes_a, boxes_b, ans_overlap)
 *           (mmdet3d/ops/iou3d/src/iou3d.cpp:70-98, kernel iou3d_kernel.cu:36-264) and
 *           its caller BaseInstance3DBoxes.overlaps
 *           (mmdet3d/core/bbox/structures/base_box3d.py:384-438), reached from
 *           HungarianAssigner3D.assign (core/bbox/assigners/hungarian_assigner.py:125)
 * ------------------------------------------------------------------------ */
/* Overlap AREA of rotated BEV rectangles (x1, y1, x2, y2, angle), out[na, nb]. */
int msmd_boxes_overlap_bev_f32(const float* boxes_a /* [na,5] */, int na,
                               const float* boxes_b /* [nb,5] */, int nb,
                               float* out /* [na,nb] */, msmd_stream_t stream);
/* 3-D IoU (mode 0) / IoF over boxes_a (mode 1) of LiDAR boxes
 * (x, y, z_bottom, dx, dy, dz, yaw, ...) for a whole batch: sample s compares its na
 * rows of boxes_a with its first nb_valid[s] rows of boxes_b (nb_valid NULL: all nb);
 * columns past nb_valid[s] are written as 0.  out[batch, na, nb]. */
int msmd_boxes_iou3d_f32(const float* boxes_a /* [batch,na,lda] */, int lda,
                         const float* boxes_b /* [batch,nb,ldb] */, int ldb,
                         const int32_t* nb_valid /* [batch] or NULL */, int batch,
                         int na, int nb, int mode, float* out, msmd_stream_t stream);

/* replaces: the per-box loop gaussian_radius -> draw_heatmap_gaussian of
 *           TransFusionHead.get_targets_single
 *           (mmdet3d/models/dense_heads/transfusion_head.py:1186-1210;
 *           mmdet3d/core/utils/gaussian.py:5-53)
 * heatmap[planes, h, w] (caller zero-fills; plane = sample * classes + class) takes the
 * maximum with exp(-(dx^2 + dy^2) / (2 (d/6)^2)), d = 2 radius + 1, evaluated in double
 * and rounded to float, clipped to the map.  plane < 0 or radius < 0: box skipped. */
int msmd_heatmap_gaussian_f32(const int32_t* plane /* [n] */,
                              const int32_t* center_x /* [n] */,
                              const int32_t* center_y /* [n] */,
                              const int32_t* radius /* [n] */, int n, int planes, int h,
                              int w, float* heatmap, msmd_stream_t stream);

/* replaces: clip_sigmoid (mmdet3d/models/utils/clip_sigmoid.py) + mmdet's
 *           GaussianFocalLoss(alpha=2, gamma=4) as called at transfusion_head.py:1247-1249.
 * p = clamp(sigmoid(logit), clip, 1 - clip);
 * loss_i = -log(p + 1e-12) (1-p)^2 [target == 1] - log(1 - p + 1e-12) p^2 (1-target)^4.
 * sums[0] = sum of loss_i, sums[1] = number of cells with target == 1 (the avg_factor the
 * reference reads back with .item()); grad (optional) = d loss_i / d logit_i.
 * Deterministic: per-block partials in double, summed in block order. */
size_t msmd_gaussian_focal_workspace_bytes(int64_t n);
int msmd_gaussian_focal_f32(const float* logits, const float* target, int64_t n,
                            float clip, float* grad /* [n] or NULL */,
                            float* sums /* [2] */, void* workspace,
                            size_t workspace_bytes, msmd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MSMD_HIP_H_ */
