// gma.hip -- the feature assembly of one GMA-Conv stage as ONE launch each way.
//
// SparseMultiModalEncoderPaint.grouped_sparse_conv
// (mmdet3d/models/middle_encoders/sparse_encoder_multimodal_encoderpaint_double_aware.py:325-430)
// builds the unified voxel set's features from three row groups:
//     only-3D rows:  [ conv3D(only-3D voxels)            | 0 (c2) ]
//     only-2D rows:  [ 0 (c3) | cross_gate[nearest 3D voxel] * feat2D[row]   ]
//     mixed rows:    [ feat3D[row3] | gate(feat3D[row3]) * feat2D[row2]      ]
// -- in the reference (and in this repo until round 3) a dozen indexing / cat / pad / mul
// launches per stage and twice that in backward (three index_add_ with their zero fills:
// 0.63 ms of the 14.5 ms LC step).  The two Linear+ReLU gates stay GEMMs; everything around
// them is HBM-bound row copying: one thread per 16-byte piece of an output row.
//
// backward: d conv3D = the left block of the only-3D rows; d gate = right block of the mixed
// rows * feat2D; d cross_gate[t] = sum over the only-2D rows whose nearest voxel is t of
// right block * feat2D -- float atomics, as torch's index_add_ (the one it replaces).
// feat3D / feat2D get no gradient (frozen LiDAR encoder, raw virtual-point voxels): the
// Python wrapper falls back to the unfused ops when they ask for one.
#include "common.hpp"

namespace msmd {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct GmaArgs {
  const float* conv3;     // [n_o3, c3]
  const float* cross;     // [n3 + 1, c2]   (row n3 = the dummy embedding's gate)
  const float* feat2;     // [n2, c2]
  const float* feat3;     // [n3, c3]
  const float* gate;      // [n_mix, c2]
  const int64_t* nn3;     // [n_o2]  nearest 3D voxel of an only-2D row, -1 = none
  const int64_t* rows_o2; // [n_o2]  row of feat2
  const int64_t* rows_m3; // [n_mix] row of feat3
  const int64_t* rows_m2; // [n_mix] row of feat2
  int n_o3, n_o2, n_o2_pad, n_mix, n_mix_pad, n3, c3, c2;
};

__global__ __launch_bounds__(256) void gma_assemble_fwd_kernel(GmaArgs A, float* __restrict__ out) {
  const int c = A.c3 + A.c2, c4 = c >> 2, k3 = A.c3 >> 2;
  const long rows = (long)A.n_o3 + A.n_o2 + A.n_o2_pad + A.n_mix + A.n_mix_pad;
  const long total = rows * c4;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long row = e / c4;
    const int q = (int)(e - row * c4);        // 16-byte piece inside the row
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < A.n_o3) {
      if (q < k3) v = ((const f32x4*)(A.conv3 + (size_t)row * A.c3))[q];
    } else if (row < (long)A.n_o3 + A.n_o2) {
      if (q >= k3) {
        const long i = row - A.n_o3;
        long t = A.nn3[i];
        t = t >= 0 ? t : A.n3;
        v = ((const f32x4*)(A.cross + (size_t)t * A.c2))[q - k3] *
            ((const f32x4*)(A.feat2 + (size_t)A.rows_o2[i] * A.c2))[q - k3];
      }
    } else if (row >= (long)A.n_o3 + A.n_o2 + A.n_o2_pad &&
               row < (long)A.n_o3 + A.n_o2 + A.n_o2_pad + A.n_mix) {
      const long i = row - A.n_o3 - A.n_o2 - A.n_o2_pad;
      if (q < k3)
        v = ((const f32x4*)(A.feat3 + (size_t)A.rows_m3[i] * A.c3))[q];
      else
        v = ((const f32x4*)(A.gate + (size_t)i * A.c2))[q - k3] *
            ((const f32x4*)(A.feat2 + (size_t)A.rows_m2[i] * A.c2))[q - k3];
    }
    ((f32x4*)out)[e] = v;
  }
}

__global__ __launch_bounds__(256) void gma_assemble_bwd_kernel(GmaArgs A,
                                                               const float* __restrict__ dout,
                                                               float* __restrict__ d_conv3,
                                                               float* __restrict__ d_cross,
                                                               float* __restrict__ d_gate) {
  const int c = A.c3 + A.c2, c4 = c >> 2, k3 = A.c3 >> 2;
  const long rows = (long)A.n_o3 + A.n_o2 + A.n_o2_pad + A.n_mix + A.n_mix_pad;
  const long total = rows * c4;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long row = e / c4;
    const int q = (int)(e - row * c4);
    if (row < A.n_o3) {
      if (q < k3) ((f32x4*)(d_conv3 + (size_t)row * A.c3))[q] = ((const f32x4*)dout)[e];
    } else if (row < (long)A.n_o3 + A.n_o2) {
      if (q >= k3 && d_cross) {   // (no sorted row lists: float atomics)
        const long i = row - A.n_o3;
        long t = A.nn3[i];
        t = t >= 0 ? t : A.n3;
        const f32x4 g = ((const f32x4*)dout)[e] *
                        ((const f32x4*)(A.feat2 + (size_t)A.rows_o2[i] * A.c2))[q - k3];
        float* d = d_cross + (size_t)t * A.c2 + 4 * (q - k3);
#pragma unroll
        for (int s = 0; s < 4; ++s) unsafeAtomicAdd(d + s, g[s]);   // hardware float add (atomicAdd is a CAS loop without -munsafe-fp-atomics)
      }
    } else if (row >= (long)A.n_o3 + A.n_o2 + A.n_o2_pad &&
               row < (long)A.n_o3 + A.n_o2 + A.n_o2_pad + A.n_mix) {
      if (q >= k3 && d_gate) {
        const long i = row - A.n_o3 - A.n_o2 - A.n_o2_pad;
        ((f32x4*)(d_gate + (size_t)i * A.c2))[q - k3] =
            ((const f32x4*)dout)[e] *
            ((const f32x4*)(A.feat2 + (size_t)A.rows_m2[i] * A.c2))[q - k3];
      }
    }
  }
}

// d cross_gate without atomics: the only-2D rows sorted by their nearest voxel (order[],
// starts[t] .. starts[t + 1] = the rows of target t; built once per batch by the index pass).
// Targets t < n3: one wave each -- its four 16-lane groups take every fourth row of the
// segment (a lane = one 16-byte piece of the c2-wide row), the four partial sums are added
// in group order.  Target n3 -- the dummy embedding's row, which gates EVERY only-2D voxel
// without a LiDAR neighbour: thousands of rows, 1.1 ms when one wave walked them alone -- is
// spread over kDummyBlocks extra workgroups of 16 groups (block partials in group order,
// then gma_cross_dummy_kernel adds the block partials in block order).  Fixed order
// throughout: deterministic, unlike the index_add_ this replaces; every target row is
// written (no zero fill).  c2 <= 64.
constexpr int kDummyBlocks = 64;
__device__ __forceinline__ f32x4 cross_term(const GmaArgs& A, const float* __restrict__ dout,
                                            long i, int c, int l) {
  return ((const f32x4*)(dout + (size_t)(A.n_o3 + i) * c + A.c3))[l] *
         ((const f32x4*)(A.feat2 + (size_t)A.rows_o2[i] * A.c2))[l];
}
__global__ __launch_bounds__(256) void gma_cross_grad_kernel(GmaArgs A, const float* __restrict__ dout,
                                                             const int64_t* __restrict__ order,
                                                             const int64_t* __restrict__ starts,
                                                             float* __restrict__ d_cross,
                                                             float* __restrict__ dummy_part,
                                                             int target_blocks) {
  const int lane = threadIdx.x & 63, g = lane >> 4, l = lane & 15, wave = threadIdx.x >> 6;
  const int c = A.c3 + A.c2, q2 = A.c2 >> 2;
  if ((int)blockIdx.x >= target_blocks) {   // a slice of the dummy row's segment
    __shared__ f32x4 sm[16][16];
    const int db = blockIdx.x - target_blocks, grp = wave * 4 + g;
    const long s0 = starts[A.n3], s1 = starts[A.n3 + 1];
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (l < q2)
      for (long p = s0 + db * 16 + grp; p < s1; p += 16 * kDummyBlocks)
        acc += cross_term(A, dout, order[p], c, l);
    sm[grp][l] = acc;
    __syncthreads();
    if (threadIdx.x < q2) {
      f32x4 r = sm[0][threadIdx.x];
      for (int k = 1; k < 16; ++k) r += sm[k][threadIdx.x];
      ((f32x4*)(dummy_part + (size_t)db * A.c2))[threadIdx.x] = r;
    }
    return;
  }
  const long t = (long)blockIdx.x * 4 + wave;
  if (t >= A.n3) return;
  const long s0 = starts[t], s1 = starts[t + 1];
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (l < q2)
    for (long p = s0 + g; p < s1; p += 4) acc += cross_term(A, dout, order[p], c, l);
  f32x4 r = acc;   // groups 1..3 -> group 0, in group order
#pragma unroll
  for (int k = 1; k < 4; ++k) {
    f32x4 o;
#pragma unroll
    for (int s = 0; s < 4; ++s) o[s] = __shfl(acc[s], l + 16 * k, 64);
    r += o;
  }
  if (g == 0 && l < q2) ((f32x4*)(d_cross + (size_t)t * A.c2))[l] = r;
}
__global__ __launch_bounds__(64) void gma_cross_dummy_kernel(const float* __restrict__ dummy_part,
                                                             int c2, float* __restrict__ d_row) {
  const int l = threadIdx.x;
  if (l >= (c2 >> 2)) return;
  f32x4 r = ((const f32x4*)dummy_part)[l];
  for (int k = 1; k < kDummyBlocks; ++k) r += ((const f32x4*)(dummy_part + (size_t)k * c2))[l];
  ((f32x4*)d_row)[l] = r;
}

inline int grid_for(long total) {
  long b = (total + 255) / 256;
  return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

bool fill_args(GmaArgs& A, const float* conv3, int n_o3, int c3, const float* cross, int n3,
               int c2, const int64_t* nn3, const float* feat2, const int64_t* rows_o2, int n_o2,
               int n_o2_pad, const float* feat3, const int64_t* rows_m3, const float* gate,
               const int64_t* rows_m2, int n_mix, int n_mix_pad) {
  if (c3 < 4 || c2 < 4 || (c3 & 3) || (c2 & 3) || n_o3 < 0 || n_o2 < 0 || n_mix < 0 ||
      n_o2_pad < 0 || n_mix_pad < 0 || n3 < 0)
    return false;
  if ((n_o3 && !conv3) || (n_o2 && (!cross || !nn3 || !feat2 || !rows_o2)) ||
      (n_mix && (!feat3 || !rows_m3 || !gate || !rows_m2 || !feat2)))
    return false;
  A = GmaArgs{conv3, cross, feat2, feat3, gate, nn3, rows_o2, rows_m3, rows_m2,
              n_o3,  n_o2,  n_o2_pad, n_mix, n_mix_pad, n3, c3, c2};
  return true;
}

}  // namespace
}  // namespace msmd

using namespace msmd;

MSMD_EXPORT int msmd_gma_assemble_fwd_f32(const float* conv3, int n_o3, int c3,
                                          const float* cross_gate, int n3, int c2,
                                          const int64_t* nn3, const float* feat2,
                                          const int64_t* rows_o2, int n_o2, int n_o2_pad,
                                          const float* feat3, const int64_t* rows_m3,
                                          const float* gate, const int64_t* rows_m2, int n_mix,
                                          int n_mix_pad, float* out, msmd_stream_t stream) {
  GmaArgs A;
  if (!fill_args(A, conv3, n_o3, c3, cross_gate, n3, c2, nn3, feat2, rows_o2, n_o2, n_o2_pad,
                 feat3, rows_m3, gate, rows_m2, n_mix, n_mix_pad))
    return MSMD_ERR_INVALID_ARG;
  const long rows = (long)n_o3 + n_o2 + n_o2_pad + n_mix + n_mix_pad;
  if (rows == 0) return MSMD_OK;
  if (!out) return MSMD_ERR_INVALID_ARG;
  MSMD_LAUNCH(gma_assemble_fwd_kernel, dim3(grid_for(rows * ((c3 + c2) >> 2))), dim3(256), 0,
              (hipStream_t)stream, A, out);
  return launch_status();
}

MSMD_EXPORT size_t msmd_gma_assemble_bwd_workspace_floats(int c2) {
  return (size_t)kDummyBlocks * (c2 > 0 ? c2 : 0);
}

MSMD_EXPORT int msmd_gma_assemble_bwd_f32(const float* d_out, int n_o3, int c3, int n3, int c2,
                                          const int64_t* nn3, const float* feat2,
                                          const int64_t* rows_o2, int n_o2, int n_o2_pad,
                                          const int64_t* rows_m2, int n_mix, int n_mix_pad,
                                          float* d_conv3 /* [n_o3,c3] */,
                                          float* d_cross_gate /* [n3+1,c2] or NULL */,
                                          float* d_gate /* [n_mix,c2] or NULL */,
                                          const int64_t* order /* [n_o2] or NULL */,
                                          const int64_t* starts /* [n3+2] or NULL */,
                                          float* workspace /* msmd_gma_assemble_bwd_workspace_floats */,
                                          msmd_stream_t stream) {
  GmaArgs A;
  // (the forward-only operands are not read in backward)
  if (!fill_args(A, d_conv3, n_o3, c3, d_cross_gate ? d_cross_gate : (const float*)d_out, n3, c2,
                 nn3, feat2, rows_o2, n_o2, n_o2_pad, d_out, rows_m2, d_out, rows_m2, n_mix,
                 n_mix_pad))
    return MSMD_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  const bool csr = d_cross_gate && order && starts && workspace && c2 <= 64;
  if (d_cross_gate && !csr)
    hipMemsetAsync(d_cross_gate, 0, sizeof(float) * ((size_t)n3 + 1) * c2, st);
  const long rows = (long)n_o3 + n_o2 + n_o2_pad + n_mix + n_mix_pad;
  if (!d_out && rows) return MSMD_ERR_INVALID_ARG;
  if (csr) {
    const int tb = ceil_div(n3 > 0 ? n3 : 1, 4);
    MSMD_LAUNCH(gma_cross_grad_kernel, dim3(tb + kDummyBlocks), dim3(256), 0, st, A, d_out, order,
                starts, d_cross_gate, workspace, tb);
    MSMD_LAUNCH(gma_cross_dummy_kernel, dim3(1), dim3(64), 0, st, (const float*)workspace, c2,
                d_cross_gate + (size_t)n3 * c2);
  }
  if (rows == 0) return launch_status();
  if (n_o3 && !d_conv3) return MSMD_ERR_INVALID_ARG;
  MSMD_LAUNCH(gma_assemble_bwd_kernel, dim3(grid_for(rows * ((c3 + c2) >> 2))), dim3(256), 0, st,
              A, d_out, d_conv3, csr ? (float*)nullptr : d_cross_gate, d_gate);
  return launch_status();
}
