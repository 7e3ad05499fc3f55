// dense.hip -- SparseConvTensor.dense() (BEV scatter), sparse_add and the
// LiDAR / virtual-point modality split.  HBM-bound permutation / set kernels.
#include "common.hpp"
#include "scan.hpp"

namespace msmd {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t cell_of(int4 r, const int* s) {
  return (((uint32_t)r.x * s[0] + r.y) * s[1] + r.z) * s[2] + r.w;
}
struct Shape3 {
  int s[3];
};

// ---------------------------------------------------------------- dense ----
// structure.py:55-64 does zero-fill + scatter to [B,D,H,W,C] + a full permute
// copy to [B,C,D,H,W].  Here: one memset + one kernel that writes channels-first
// directly.  A block stages 64 rows x C through LDS (row reads are coalesced
// 16-B loads) and writes with the ROW index fastest: rows are in ascending
// linear id after a strided conv, so neighbouring rows are neighbouring x and
// the stores of one channel coalesce.
constexpr int kDenseRows = 64;

template <bool SCATTER>
__global__ __launch_bounds__(256) void dense_kernel(float* __restrict__ feat,
                                                    const int32_t* __restrict__ idx, int n, int c,
                                                    Shape3 sh, float* __restrict__ dense) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* tile = (float*)smem_raw;                       // [64][c+1]
  long* cells = (long*)(tile + kDenseRows * (c + 1) + ((kDenseRows * (c + 1)) & 1));
  const int r0 = blockIdx.x * kDenseRows;
  const int nr = (n - r0) < kDenseRows ? (n - r0) : kDenseRows;
  const long vol = (long)sh.s[0] * sh.s[1] * sh.s[2];
  if (threadIdx.x < nr) {
    int4 r = ((const int4*)idx)[r0 + threadIdx.x];
    long cell = ((long)r.y * sh.s[1] + r.z) * sh.s[2] + r.w;
    cells[threadIdx.x] = (long)r.x * c * vol + cell;    // offset of channel 0
  }
  if (SCATTER) {
    for (int e = threadIdx.x; e < nr * c; e += 256) {
      int rr = e / c, cc = e - rr * c;
      tile[rr * (c + 1) + cc] = feat[(size_t)(r0 + rr) * c + cc];
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < kDenseRows * c; e += 256) {
    int cc = e / kDenseRows, rr = e - cc * kDenseRows;   // row fastest
    if (rr < nr) {
      float* p = dense + cells[rr] + (long)cc * vol;
      if (SCATTER) *p = tile[rr * (c + 1) + cc]; else tile[rr * (c + 1) + cc] = *p;
    }
  }
  if (!SCATTER) {
    __syncthreads();
    for (int e = threadIdx.x; e < nr * c; e += 256) {
      int rr = e / c, cc = e - rr * c;
      feat[(size_t)(r0 + rr) * c + cc] = tile[rr * (c + 1) + cc];
    }
  }
}

size_t dense_smem(int c) {
  size_t fl = (size_t)kDenseRows * (c + 1);
  fl += fl & 1;
  return fl * sizeof(float) + kDenseRows * sizeof(long);
}

// ----------------------------------------------------------- sparse_add ----
__global__ __launch_bounds__(256) void mark_rows(const int32_t* __restrict__ idx, int n, Shape3 sh,
                                                 uint32_t* bits) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) bitmap_set(bits, cell_of(((const int4*)idx)[i], sh.s));
}

__global__ __launch_bounds__(256) void add_rows(const float* __restrict__ feat,
                                                const int32_t* __restrict__ idx, int n, int c,
                                                Shape3 sh, const uint32_t* __restrict__ bits,
                                                const int* __restrict__ prefix, int n_out,
                                                int32_t* __restrict__ out_idx,
                                                float* __restrict__ out_feat,
                                                int32_t* __restrict__ map) {
  // one wave-quarter (16 lanes) per row: coalesced feature reads
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int i = (int)(t >> 4), sub = (int)(t & 15);
  if (i >= n) return;
  int4 r = ((const int4*)idx)[i];
  int o = bitmap_rank(bits, prefix, cell_of(r, sh.s));
  if (o >= n_out) return;
  if (sub == 0) {
    ((int4*)out_idx)[o] = r;
    if (map) map[i] = o;
  }
  for (int ch = sub; ch < c; ch += 16)
    unsafeAtomicAdd(&out_feat[(size_t)o * c + ch], feat[(size_t)i * c + ch]);
}

// the feature half alone, rows routed by the maps of an earlier index pass
__global__ __launch_bounds__(256) void add_mapped_rows(const float* __restrict__ feat,
                                                       const int32_t* __restrict__ map, int n,
                                                       int c, int n_out,
                                                       float* __restrict__ out_feat) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int i = (int)(t >> 4), sub = (int)(t & 15);
  if (i >= n) return;
  const int o = map[i];
  if (o < 0 || o >= n_out) return;
  for (int ch = sub; ch < c; ch += 16)
    unsafeAtomicAdd(&out_feat[(size_t)o * c + ch], feat[(size_t)i * c + ch]);
}

// ------------------------------------------------------- modality split ----
__global__ __launch_bounds__(256) void and_words(uint32_t* a, const uint32_t* __restrict__ b,
                                                 size_t words) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < words) a[i] &= b[i];
}
// stats (optional, zeroed by the caller): rows per sample of this voxel set, split by the
// mix flag -- stats[(base + (flag ? batch : 0)) + b].  One atomic per (wave, sample).
__global__ __launch_bounds__(256) void split_rows(const int32_t* __restrict__ idx, int n,
                                                  Shape3 sh, const uint32_t* __restrict__ both,
                                                  const int* __restrict__ prefix, int cap,
                                                  int32_t* __restrict__ mix,
                                                  int32_t* __restrict__ pair,
                                                  int32_t* __restrict__ stats, int batch,
                                                  int mixed_offset) {
  int i = blockIdx.x * 256 + threadIdx.x;
  int m = 0, b = -1;
  if (i < n) {
    const int4 r = ((const int4*)idx)[i];
    uint32_t cell = cell_of(r, sh.s);
    m = bitmap_test(both, cell);
    b = r.x;
    mix[i] = m;
    if (m) {
      int rk = bitmap_rank(both, prefix, cell);
      if (rk < cap) pair[rk] = i;
    }
  }
  if (stats) {
    for (int s = 0; s < batch; ++s) {
      const unsigned long long plain = __ballot(b == s && !m), mixed = __ballot(b == s && m);
      if ((threadIdx.x & 63) == 0) {
        if (plain) atomicAdd(&stats[s], __popcll(plain));
        if (mixed) atomicAdd(&stats[mixed_offset + s], __popcll(mixed));
      }
    }
  }
}

struct SetWs {
  uint32_t *bits, *bits2;
  int *prefix, *tiles;
  size_t words;
};
template <typename A>
void carve_set(A& a, SetWs* w, int batch, const int* shape, bool two) {
  size_t cells = (size_t)batch * shape[0] * shape[1] * shape[2];
  size_t words = (cells + 31) / 32;
  uint32_t* b0 = a.template take<uint32_t>(words);
  uint32_t* b1 = two ? a.template take<uint32_t>(words) : nullptr;
  int* pf = a.template take<int>(words);
  int* tl = a.template take<int>(scan_num_tiles((long)words) + 1);
  if (w) *w = SetWs{b0, b1, pf, tl, words};
}

int check_grid(int batch, const int* shape, Shape3* sh) {
  if (batch < 1 || !shape) return MSMD_ERR_INVALID_ARG;
  double cells = batch;
  for (int i = 0; i < 3; ++i) {
    if (shape[i] < 1) return MSMD_ERR_INVALID_ARG;
    sh->s[i] = shape[i];
    cells *= shape[i];
  }
  return cells >= 4294967295.0 ? MSMD_ERR_RANGE : MSMD_OK;
}

}  // namespace
}  // namespace msmd

using namespace msmd;

MSMD_EXPORT int msmd_dense_scatter_f32(const float* feat, const int32_t* indices, int n, int c,
                                       int batch_size, const int* spatial_shape, float* out,
                                       msmd_stream_t stream) {
  Shape3 sh;
  int rc = check_grid(batch_size, spatial_shape, &sh);
  if (rc) return rc;
  if (n < 0 || c < 1 || !out || (n > 0 && (!feat || !indices))) return MSMD_ERR_INVALID_ARG;
  size_t smem = dense_smem(c);
  if (smem > 160 * 1024) return MSMD_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  size_t bytes = sizeof(float) * (size_t)batch_size * c * sh.s[0] * sh.s[1] * sh.s[2];
  hipMemsetAsync(out, 0, bytes, st);
  if (n > 0) {
    static LdsGrant granted;
    rc = optin_dynamic_lds((const void*)dense_kernel<true>, smem, granted);
    if (rc != MSMD_OK) return rc;
    MSMD_LAUNCH(dense_kernel<true>, dim3(ceil_div(n, kDenseRows)), dim3(256), smem, st,
                       const_cast<float*>(feat), indices, n, c, sh, out);
  }
  return launch_status();
}

MSMD_EXPORT int msmd_dense_gather_f32(const float* dense, const int32_t* indices, int n, int c,
                                      int batch_size, const int* spatial_shape, float* feat,
                                      msmd_stream_t stream) {
  Shape3 sh;
  int rc = check_grid(batch_size, spatial_shape, &sh);
  if (rc) return rc;
  if (n < 0 || c < 1 || (n > 0 && (!feat || !indices || !dense))) return MSMD_ERR_INVALID_ARG;
  if (n == 0) return MSMD_OK;
  size_t smem = dense_smem(c);
  if (smem > 160 * 1024) return MSMD_ERR_UNSUPPORTED;
  static LdsGrant granted;
  rc = optin_dynamic_lds((const void*)dense_kernel<false>, smem, granted);
  if (rc != MSMD_OK) return rc;
  MSMD_LAUNCH(dense_kernel<false>, dim3(ceil_div(n, kDenseRows)), dim3(256), smem,
                     (hipStream_t)stream, feat, indices, n, c, sh, const_cast<float*>(dense));
  return launch_status();
}

MSMD_EXPORT size_t msmd_sparse_add_workspace_bytes(int batch_size, const int* spatial_shape) {
  ArenaSize a;
  carve_set(a, (SetWs*)nullptr, batch_size, spatial_shape, false);
  return a.off;
}

MSMD_EXPORT int msmd_sparse_add_count(const int32_t* idx_a, int n_a, const int32_t* idx_b,
                                      int n_b, int batch_size, const int* spatial_shape,
                                      int32_t* n_out, void* workspace, size_t workspace_bytes,
                                      msmd_stream_t stream) {
  Shape3 sh;
  int rc = check_grid(batch_size, spatial_shape, &sh);
  if (rc) return rc;
  if (n_a < 0 || n_b < 0 || !n_out) return MSMD_ERR_INVALID_ARG;
  Arena a(workspace, workspace_bytes);
  SetWs w;
  carve_set(a, &w, batch_size, spatial_shape, false);
  if (!a.ok()) return MSMD_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  hipMemsetAsync(w.bits, 0, sizeof(uint32_t) * w.words, st);
  if (n_a > 0)
    MSMD_LAUNCH(mark_rows, dim3(ceil_div(n_a, 256)), dim3(256), 0, st, idx_a, n_a, sh,
                       w.bits);
  if (n_b > 0)
    MSMD_LAUNCH(mark_rows, dim3(ceil_div(n_b, 256)), dim3(256), 0, st, idx_b, n_b, sh,
                       w.bits);
  device_scan(PopcCount{w.bits}, StorePrefix{w.prefix}, (int)w.words, w.tiles, n_out, -1, st);
  return launch_status();
}

MSMD_EXPORT int msmd_sparse_add_fill(const float* feat_a, const int32_t* idx_a, int n_a,
                                     const float* feat_b, const int32_t* idx_b, int n_b, int c,
                                     int batch_size, const int* spatial_shape, int n_out,
                                     int32_t* out_indices, float* out_feat, int32_t* map_a,
                                     int32_t* map_b, void* workspace, size_t workspace_bytes,
                                     msmd_stream_t stream) {
  Shape3 sh;
  int rc = check_grid(batch_size, spatial_shape, &sh);
  if (rc) return rc;
  // c == 0: index-only pass (out_indices + maps; no feature pointer is touched)
  if (n_a < 0 || n_b < 0 || c < 0 || n_out < 0 ||
      (n_out > 0 && (!out_indices || (c > 0 && !out_feat))))
    return MSMD_ERR_INVALID_ARG;
  Arena a(workspace, workspace_bytes);
  SetWs w;
  carve_set(a, &w, batch_size, spatial_shape, false);
  if (!a.ok()) return MSMD_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  if (n_out > 0 && c > 0) hipMemsetAsync(out_feat, 0, sizeof(float) * (size_t)n_out * c, st);
  if (n_a > 0)
    MSMD_LAUNCH(add_rows, dim3(ceil_div((long)n_a * 16, 256)), dim3(256), 0, st, feat_a,
                       idx_a, n_a, c, sh, w.bits, w.prefix, n_out, out_indices, out_feat, map_a);
  if (n_b > 0)
    MSMD_LAUNCH(add_rows, dim3(ceil_div((long)n_b * 16, 256)), dim3(256), 0, st, feat_b,
                       idx_b, n_b, c, sh, w.bits, w.prefix, n_out, out_indices, out_feat, map_b);
  return launch_status();
}

MSMD_EXPORT int msmd_sparse_add_rows(const float* feat_a, const int32_t* map_a, int n_a,
                                     const float* feat_b, const int32_t* map_b, int n_b, int c,
                                     int n_out, float* out_feat, msmd_stream_t stream) {
  if (n_a < 0 || n_b < 0 || c < 1 || n_out < 0 || (n_out > 0 && !out_feat) ||
      (n_a > 0 && (!feat_a || !map_a)) || (n_b > 0 && (!feat_b || !map_b)))
    return MSMD_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (n_out > 0) hipMemsetAsync(out_feat, 0, sizeof(float) * (size_t)n_out * c, st);
  if (n_a > 0)
    MSMD_LAUNCH(add_mapped_rows, dim3(ceil_div((long)n_a * 16, 256)), dim3(256), 0, st, feat_a,
                       map_a, n_a, c, n_out, out_feat);
  if (n_b > 0)
    MSMD_LAUNCH(add_mapped_rows, dim3(ceil_div((long)n_b * 16, 256)), dim3(256), 0, st, feat_b,
                       map_b, n_b, c, n_out, out_feat);
  return launch_status();
}

MSMD_EXPORT size_t msmd_modality_split_workspace_bytes(int batch_size, const int* spatial_shape) {
  ArenaSize a;
  carve_set(a, (SetWs*)nullptr, batch_size, spatial_shape, true);
  return a.off;
}

static int modality_split_impl(const int32_t* idx_3d, int n3, const int32_t* idx_2d, int n2,
                               int batch_size, const int* spatial_shape, int32_t* mix3d,
                               int32_t* mix2d, int32_t* pair_3d, int32_t* pair_2d,
                               int32_t* n_mixed, int32_t* stats, void* workspace,
                               size_t workspace_bytes, msmd_stream_t stream) {
  Shape3 sh;
  int rc = check_grid(batch_size, spatial_shape, &sh);
  if (rc) return rc;
  if (n3 < 0 || n2 < 0 || !n_mixed) return MSMD_ERR_INVALID_ARG;
  Arena a(workspace, workspace_bytes);
  SetWs w;
  carve_set(a, &w, batch_size, spatial_shape, true);
  if (!a.ok()) return MSMD_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  // (the two bitmaps lie next to each other in the arena: one fill)
  hipMemsetAsync(w.bits, 0, (size_t)((char*)(w.bits2 + w.words) - (char*)w.bits), st);
  if (stats) hipMemsetAsync(stats, 0, sizeof(int32_t) * 4 * batch_size, st);
  if (n3 > 0)
    MSMD_LAUNCH(mark_rows, dim3(ceil_div(n3, 256)), dim3(256), 0, st, idx_3d, n3, sh,
                       w.bits);
  if (n2 > 0)
    MSMD_LAUNCH(mark_rows, dim3(ceil_div(n2, 256)), dim3(256), 0, st, idx_2d, n2, sh,
                       w.bits2);
  MSMD_LAUNCH(and_words, dim3(ceil_div((long)w.words, 256)), dim3(256), 0, st, w.bits,
                     w.bits2, w.words);
  device_scan(PopcCount{w.bits}, StorePrefix{w.prefix}, (int)w.words, w.tiles, n_mixed, -1, st);
  const int cap = n3 < n2 ? n3 : n2;
  // stats = [3D plain | 3D mixed | 2D plain | 2D mixed], batch_size entries each
  if (n3 > 0)
    MSMD_LAUNCH(split_rows, dim3(ceil_div(n3, 256)), dim3(256), 0, st, idx_3d, n3, sh,
                       w.bits, w.prefix, cap, mix3d, pair_3d, stats, batch_size, batch_size);
  if (n2 > 0)
    MSMD_LAUNCH(split_rows, dim3(ceil_div(n2, 256)), dim3(256), 0, st, idx_2d, n2, sh,
                       w.bits, w.prefix, cap, mix2d, pair_2d,
                       stats ? stats + 2 * batch_size : nullptr, batch_size, batch_size);
  return launch_status();
}

MSMD_EXPORT int msmd_modality_split(const int32_t* idx_3d, int n3, const int32_t* idx_2d, int n2,
                                    int batch_size, const int* spatial_shape, int32_t* mix3d,
                                    int32_t* mix2d, int32_t* pair_3d, int32_t* pair_2d,
                                    int32_t* n_mixed, void* workspace, size_t workspace_bytes,
                                    msmd_stream_t stream) {
  return modality_split_impl(idx_3d, n3, idx_2d, n2, batch_size, spatial_shape, mix3d, mix2d,
                             pair_3d, pair_2d, n_mixed, nullptr, workspace, workspace_bytes,
                             stream);
}

MSMD_EXPORT int msmd_modality_split_stats(const int32_t* idx_3d, int n3, const int32_t* idx_2d,
                                          int n2, int batch_size, const int* spatial_shape,
                                          int32_t* mix3d, int32_t* mix2d, int32_t* pair_3d,
                                          int32_t* pair_2d, int32_t* n_mixed,
                                          int32_t* sample_stats, void* workspace,
                                          size_t workspace_bytes, msmd_stream_t stream) {
  if (!sample_stats) return MSMD_ERR_INVALID_ARG;
  return modality_split_impl(idx_3d, n3, idx_2d, n2, batch_size, spatial_shape, mix3d, mix2d,
                             pair_3d, pair_2d, n_mixed, sample_stats, workspace, workspace_bytes,
                             stream);
}

// ------------------------------------------------- sparse_add rows, no atomics --
// The feature half of sparse_add as a GATHER: every output row is written once,
// out[j] = a[inv_a[j]] + b[inv_b[j]] (a missing side adds nothing), instead of a zero fill
// and two passes of per-element float atomics (33 us each on a 60 k x 192 stage; the three
// adds of an LC step were 0.3 ms of the feature queue).  inv_x[j] = the LAST row of x that
// maps to output row j (msmd_rows_inverse, built once per batch by the index pass).  Rows
// that share their coordinates with a later row of the same tensor -- sparse_add sums them
// too -- are added by a second, nearly empty pass (fix-up: rows i with inv[map[i]] != i);
// without such rows the result is deterministic and bit-identical to a + b in that order.
namespace msmd {
namespace {
__global__ __launch_bounds__(256) void rows_inverse_kernel(const int32_t* __restrict__ map, int n,
                                                           int n_out, int32_t* inv) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int o = map[i];
  if (o >= 0 && o < n_out) atomicMax(&inv[o], i);
}
__global__ __launch_bounds__(256) void add_gather_rows(const float* __restrict__ fa,
                                                       const int32_t* __restrict__ inv_a,
                                                       const float* __restrict__ fb,
                                                       const int32_t* __restrict__ inv_b, int c4,
                                                       long total4, float* __restrict__ out) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total4; e += (long)gridDim.x * 256) {
    const int j = (int)(e / c4), q = (int)(e - (long)j * c4);
    const int ia = inv_a[j], ib = inv_b[j];
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (ia >= 0) v = ((const f32x4*)fa)[(size_t)ia * c4 + q];
    if (ib >= 0) v += ((const f32x4*)fb)[(size_t)ib * c4 + q];
    ((f32x4*)out)[e] = v;
  }
}
__global__ __launch_bounds__(256) void add_fixup_rows(const float* __restrict__ feat,
                                                      const int32_t* __restrict__ map,
                                                      const int32_t* __restrict__ inv, int n,
                                                      int c, int n_out, float* __restrict__ out) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int i = (int)(t >> 4), sub = (int)(t & 15);
  if (i >= n) return;
  const int o = map[i];
  if (o < 0 || o >= n_out || inv[o] == i) return;      // (the gather took this row)
  for (int ch = sub; ch < c; ch += 16)
    unsafeAtomicAdd(&out[(size_t)o * c + ch], feat[(size_t)i * c + ch]);
}
}  // namespace
}  // namespace msmd

MSMD_EXPORT int msmd_rows_inverse(const int32_t* map, int n, int n_out, int32_t* inv,
                                  msmd_stream_t stream) {
  if (n < 0 || n_out < 0 || (n > 0 && !map) || (n_out > 0 && !inv)) return MSMD_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (n_out > 0) hipMemsetAsync(inv, 0xFF, sizeof(int32_t) * (size_t)n_out, st);
  if (n > 0 && n_out > 0)
    MSMD_LAUNCH(rows_inverse_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, map, n, n_out, inv);
  return launch_status();
}

MSMD_EXPORT int msmd_sparse_add_rows_gather(const float* feat_a, const int32_t* map_a,
                                            const int32_t* inv_a, int n_a, const float* feat_b,
                                            const int32_t* map_b, const int32_t* inv_b, int n_b,
                                            int c, int n_out, float* out_feat,
                                            msmd_stream_t stream) {
  if (n_a < 0 || n_b < 0 || c < 4 || (c & 3) || n_out < 0 ||
      (n_out > 0 && (!out_feat || !inv_a || !inv_b)) ||
      (n_a > 0 && (!feat_a || !map_a)) || (n_b > 0 && (!feat_b || !map_b)))
    return MSMD_ERR_INVALID_ARG;
  if (n_out == 0) return MSMD_OK;
  hipStream_t st = (hipStream_t)stream;
  const long total4 = (long)n_out * (c >> 2);
  long blocks = (total4 + 255) / 256;
  blocks = blocks > 8192 ? 8192 : blocks;
  MSMD_LAUNCH(add_gather_rows, dim3((int)blocks), dim3(256), 0, st, feat_a, inv_a, feat_b, inv_b,
              c >> 2, total4, out_feat);
  if (n_a > 0)
    MSMD_LAUNCH(add_fixup_rows, dim3(ceil_div((long)n_a * 16, 256)), dim3(256), 0, st, feat_a,
                map_a, inv_a, n_a, c, n_out, out_feat);
  if (n_b > 0)
    MSMD_LAUNCH(add_fixup_rows, dim3(ceil_div((long)n_b * 16, 256)), dim3(256), 0, st, feat_b,
                map_b, inv_b, n_b, c, n_out, out_feat);
  return launch_status();
}

// ------------------------------------------------------------------ rows_where
// rows[r] = the r-th i (ascending) with flags[i * stride] == value: mask.nonzero() for a mask
// whose count the host already knows (msmd_modality_split_stats brought it), as two launches
// of the scan (torch.nonzero_static: six, plus the compare that makes the mask).
namespace msmd {
namespace {
struct FlagEq {
  const int32_t* f;
  int stride, v;
  __device__ int operator()(int i) const { return f[(size_t)i * stride] == v; }
};
struct EmitRow {
  int64_t* out;
  int cap;
  __device__ void operator()(int i, int p, int c) const {
    if (c && p < cap) out[p] = i;
  }
};
// rows[total .. cap) <- -1 (what torch.nonzero_static pads with): a caller whose host-side count
// is larger than the real one (stale statistics) gets a row id that index_select rejects instead
// of whatever the allocation held.  Normally total == cap and the block leaves at once.
__global__ __launch_bounds__(256) void rows_tail_fill(int64_t* __restrict__ rows, int cap,
                                                      const int* __restrict__ total) {
  for (int i = *total + (int)threadIdx.x; i < cap; i += 256) rows[i] = -1;
}
}  // namespace
}  // namespace msmd

MSMD_EXPORT size_t msmd_rows_where_workspace_bytes(int n) {
  return align_up(sizeof(int) * ((size_t)scan_num_tiles(n > 0 ? n : 1) + 1)) + 256;
}

MSMD_EXPORT int msmd_rows_where_eq(const int32_t* flags, int stride, int n, int value,
                                   int64_t* rows, int capacity, int32_t* total, void* workspace,
                                   size_t workspace_bytes, msmd_stream_t stream) {
  if (n < 0 || stride < 1 || capacity < 0 || (n > 0 && !flags) || (capacity > 0 && !rows))
    return MSMD_ERR_INVALID_ARG;
  if (workspace_bytes < msmd_rows_where_workspace_bytes(n) || ((uintptr_t)workspace & 255))
    return MSMD_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  int* tiles = (int*)workspace;
  int* tot = total ? total : (int*)((char*)workspace + align_up(sizeof(int) * ((size_t)scan_num_tiles(n > 0 ? n : 1) + 1)));
  device_scan(FlagEq{flags, stride, value}, EmitRow{rows, capacity}, n, tiles, tot, -1, st);
  if (capacity > 0) MSMD_LAUNCH(rows_tail_fill, dim3(1), dim3(256), 0, st, rows, capacity, tot);
  return launch_status();
}

// Several lists in ONE scan (the only-3D / only-2D row lists of the four image scales of an LC
// step: 8 calls = 16 scan launches + 8 tail fills before): blocks find their list from tile
// prefixes, the scan restarts per list, the block holding a list's last tile fills its tail.
namespace msmd {
namespace {
constexpr int kRowsMany = 16;
struct RowsTab {
  int n;
  int tile0[kRowsMany + 1];
  const int32_t* flags[kRowsMany];
  int64_t* rows[kRowsMany];
  int stride[kRowsMany], len[kRowsMany], value[kRowsMany], cap[kRowsMany];
};
__device__ __forceinline__ int rows_list_of(const int* __restrict__ first, int n, int b) {
  int s = 0;
  while (s + 1 < n && b >= first[s + 1]) ++s;
  return s;
}
__global__ __launch_bounds__(kScanBlock) void rows_sums_many(const RowsTab tab,
                                                            int* __restrict__ tile_sums) {
  __shared__ int smem[kScanBlock / 64];
  const int s = rows_list_of(tab.tile0, tab.n, blockIdx.x);
  const FlagEq count{tab.flags[s], tab.stride[s], tab.value[s]};
  const int base = (blockIdx.x - tab.tile0[s]) * kScanTile, n = tab.len[s];
  int c = 0;
#pragma unroll
  for (int q = 0; q < kScanItems; ++q) {
    const int i = base + q * kScanBlock + threadIdx.x;
    if (i < n) c += count(i);
  }
  c = wave_sum(c);
  if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int i = 0; i < kScanBlock / 64; ++i) t += smem[i];
    tile_sums[blockIdx.x] = t;
  }
}
__global__ __launch_bounds__(kScanBlock) void rows_emit_many(const RowsTab tab,
                                                            const int* __restrict__ tile_sums) {
  __shared__ int smem[kScanSmem];
  const int s = rows_list_of(tab.tile0, tab.n, blockIdx.x);
  const FlagEq count{tab.flags[s], tab.stride[s], tab.value[s]};
  const EmitRow emit{tab.rows[s], tab.cap[s]};
  const int base = (blockIdx.x - tab.tile0[s]) * kScanTile, n = tab.len[s];
  int v[kScanItems], ex[kScanItems];
#pragma unroll
  for (int q = 0; q < kScanItems; ++q) {
    const int i = base + q * kScanBlock + threadIdx.x;
    v[q] = i < n ? count(i) : 0;
  }
  int carry = block_range_sum<kScanBlock>(tile_sums, tab.tile0[s], (int)blockIdx.x, smem);
  const int tot = tile_excl_scan(v, ex, smem);
#pragma unroll
  for (int q = 0; q < kScanItems; ++q) {
    const int i = base + q * kScanBlock + threadIdx.x;
    if (i < n) emit(i, carry + ex[q], v[q]);
  }
  if ((int)blockIdx.x == tab.tile0[s + 1] - 1)       // the list's last tile: -1 past its total
    for (int i = carry + tot + (int)threadIdx.x; i < tab.cap[s]; i += kScanBlock) tab.rows[s][i] = -1;
}
}  // namespace
}  // namespace msmd

MSMD_EXPORT size_t msmd_rows_where_eq_many_workspace_bytes(const int* lens, int n_lists) {
  if (!lens || n_lists < 1) return 0;
  long tiles = 0;
  for (int i = 0; i < n_lists; ++i) tiles += scan_num_tiles(lens[i] > 0 ? lens[i] : 1);
  return align_up(sizeof(int) * (size_t)(tiles + 1)) + 256;
}

MSMD_EXPORT int msmd_rows_where_eq_many(const int32_t* const* flags, const int* strides,
                                        const int* lens, const int* values,
                                        int64_t* const* rows, const int* capacities, int n_lists,
                                        void* workspace, size_t workspace_bytes,
                                        msmd_stream_t stream) {
  if (!flags || !strides || !lens || !values || !rows || !capacities || n_lists < 1)
    return MSMD_ERR_INVALID_ARG;
  if (workspace_bytes < msmd_rows_where_eq_many_workspace_bytes(lens, n_lists) ||
      ((uintptr_t)workspace & 255))
    return MSMD_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  for (int g0 = 0; g0 < n_lists; g0 += kRowsMany) {
    RowsTab tab;
    tab.n = 0;
    tab.tile0[0] = 0;
    for (int i = g0; i < n_lists && i < g0 + kRowsMany; ++i) {
      if (lens[i] < 0 || strides[i] < 1 || capacities[i] < 0 || (lens[i] > 0 && !flags[i]) ||
          (capacities[i] > 0 && !rows[i]))
        return MSMD_ERR_INVALID_ARG;
      if (capacities[i] == 0) continue;
      const int s = tab.n++;
      tab.flags[s] = flags[i];
      tab.rows[s] = rows[i];
      tab.stride[s] = strides[i];
      tab.len[s] = lens[i];
      tab.value[s] = values[i];
      tab.cap[s] = capacities[i];
      // (an empty flag vector still owns one tile: its block writes the -1 tail)
      tab.tile0[s + 1] = tab.tile0[s] + scan_num_tiles(lens[i] > 0 ? lens[i] : 1);
    }
    if (tab.n == 0) continue;
    for (int s = tab.n + 1; s <= kRowsMany; ++s) tab.tile0[s] = 0x7fffffff;
    int* tile_sums = (int*)workspace;
    MSMD_LAUNCH(rows_sums_many, dim3(tab.tile0[tab.n]), dim3(kScanBlock), 0, st, tab, tile_sums);
    MSMD_LAUNCH(rows_emit_many, dim3(tab.tile0[tab.n]), dim3(kScanBlock), 0, st, tab,
                (const int*)tile_sums);
  }
  return launch_status();
}
