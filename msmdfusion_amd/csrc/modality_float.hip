// modality_float.hip -- voxel_modality_split with the REFERENCE's float32 keys.
//
// mmdet3d/models/detectors/MSMDFusion.py:271-272 keys a voxel as `z*1e6 + y*1e3 + x` on an
// int tensor: the Python floats promote every step to float32, so the key is
//   k = fl(fl(fl(z) * 1e6f + fl(y) * 1e3f) + fl(x))
// -- integer-valued, but not injective: above 2^24 (z >= 17 on the 0.075 m grid) the float
// spacing is 2 (4 above 2^25) and neighbouring x share a key; x >= 1000 runs into the next y.
// The reference then sorts both sets per sample and walks them with two pointers
// (type_assign, :27-45): the r-th occurrence of a key in one list is paired with the r-th
// occurrence in the other.  A checkpoint trained with the reference has seen those false
// "mixed" voxels on every frame; `reference_quirks=True` of the detector reproduces them
// through this entry point (the default, msmd_modality_split, keys voxels exactly).
//
// GPU form, all samples at once: key32 = sample << 26 | (uint)k, stable radix sort of
// (key32, row) per set (ties keep row order -- the reference's torch.sort leaves tie order
// unspecified; the oracle sorts stably too), then per sorted element two binary searches
// into the other list give "occurrence number < count over there" = matched and the partner,
// a scan over the 3D list's flags emits the pair lists in key order = the reference's
// per-sample sorted order, samples concatenated (:262-318).
//
// reference_offsets != 0: rows in the pair lists are numbered as the reference numbers them
// (:288-289,313-314): position inside the sample + the PREVIOUS sample's row count only (not
// the cumulative count).  Identical to global row numbers for batch <= 2 -- the only batch
// sizes the reference is correct for; for larger batches it reproduces the reference's (wrong)
// rows.  Needs each set's rows grouped by sample in ascending order (voxelize() gives that).
#include <hipcub/hipcub.hpp>

#include "common.hpp"
#include "scan.hpp"

namespace msmd {
namespace {

constexpr int kKeyBits = 26;

__device__ __forceinline__ uint32_t float_key(int z, int y, int x) {
  // -ffp-contract=off (Makefile): every step rounded to float32, as torch does
  float k = (float)z * 1e6f;
  k = k + (float)y * 1e3f;
  k = k + (float)x;
  return (uint32_t)k;     // integer-valued, < 2^26 (checked on the host from the grid shape)
}

// `bad` (one int, zeroed by the entry point) is raised for a row outside the grid / the batch
// (its key would spill into the sample bits; a negative float -> unsigned is undefined) and,
// with `grouped`, for a row whose sample id is below its predecessor's (reference_offsets
// finds each sample's first row by binary search).  Such a row gets key 0; the entry point's
// last kernel then reports the call as failed through *n_mixed = -1.
__global__ __launch_bounds__(256) void fkey_kernel(const int32_t* __restrict__ idx, int n,
                                                   int batch, int sz, int sy, int sx, int grouped,
                                                   uint32_t* __restrict__ keys,
                                                   int32_t* __restrict__ rows,
                                                   int32_t* __restrict__ bad) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int4 r = ((const int4*)idx)[i];
  const bool in = (unsigned)r.x < (unsigned)batch && (unsigned)r.y < (unsigned)sz &&
                  (unsigned)r.z < (unsigned)sy && (unsigned)r.w < (unsigned)sx;
  const bool order_ok = !grouped || i == 0 || idx[(size_t)(i - 1) * 4] <= r.x;
  if (!in || !order_ok) *bad = 1;      // (same value from every writer)
  keys[i] = in ? ((uint32_t)r.x << kKeyBits) | float_key(r.y, r.z, r.w) : 0u;
  rows[i] = i;
}
__global__ void fkey_report_kernel(const int32_t* __restrict__ bad, int32_t* __restrict__ n_mixed) {
  if (*bad) *n_mixed = -1;
}

// first row of every sample (rows grouped by sample, ascending): start[b], b = 0 .. batch
__global__ void sample_starts_kernel(const int32_t* __restrict__ idx, int n, int batch,
                                     int32_t* __restrict__ start) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b > batch) return;
  int lo = 0, hi = n;       // first row with sample id >= b
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (idx[(size_t)mid * 4] < b) lo = mid + 1; else hi = mid;
  }
  start[b] = lo;
}

__device__ __forceinline__ int lower_bound_u32(const uint32_t* __restrict__ a, int n, uint32_t v) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] < v) lo = mid + 1; else hi = mid;
  }
  return lo;
}
__device__ __forceinline__ int upper_bound_u32(const uint32_t* __restrict__ a, int n, uint32_t v) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] <= v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// Element i of this set's sorted list: matched when its occurrence number among equal keys is
// below the key's count in the other list.  mix[row] <- matched; flag / partner (optional,
// the 3D side): for the pair scan.  stats: rows per sample [plain | mixed] of this set.
__global__ __launch_bounds__(256) void match_kernel(const uint32_t* __restrict__ own_keys,
                                                    const int32_t* __restrict__ own_rows, int n,
                                                    const uint32_t* __restrict__ other_keys,
                                                    const int32_t* __restrict__ other_rows,
                                                    int n_other, int32_t* __restrict__ mix,
                                                    int32_t* __restrict__ flag,
                                                    int32_t* __restrict__ partner,
                                                    int32_t* __restrict__ stats, int batch) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  int m = 0, b = -1;
  if (i < n) {
    const uint32_t key = own_keys[i];
    const int occ = i - lower_bound_u32(own_keys, n, key);
    const int lo = lower_bound_u32(other_keys, n_other, key);
    const int hi = upper_bound_u32(other_keys, n_other, key);
    m = occ < hi - lo;
    b = (int)(key >> kKeyBits);
    mix[own_rows[i]] = m;
    if (flag) {
      flag[i] = m;
      partner[i] = m ? other_rows[lo + occ] : -1;
    }
  }
  if (stats) {
    for (int s = 0; s < batch; ++s) {
      const unsigned long long plain = __ballot(b == s && !m), mixed = __ballot(b == s && m);
      if ((threadIdx.x & 63) == 0) {
        if (plain) atomicAdd(&stats[s], __popcll(plain));
        if (mixed) atomicAdd(&stats[batch + s], __popcll(mixed));
      }
    }
  }
}

struct FlagCount {
  const int32_t* f;
  __device__ int operator()(int i) const { return f[i]; }
};
struct EmitPair {
  const uint32_t* keys;
  const int32_t *rows, *partner, *start3, *start2;   // start*: NULL = global row numbers
  int32_t *pair3, *pair2;
  int cap;
  __device__ void operator()(int i, int p, int c) const {
    if (!c || p >= cap) return;
    int r3 = rows[i], r2 = partner[i];
    if (start3) {     // position in the sample + the previous sample's count (MSMDFusion.py:288-289)
      const int b = (int)(keys[i] >> kKeyBits);
      if (b > 0) {
        r3 -= start3[b - 1];
        r2 -= start2[b - 1];
      }
    }
    pair3[p] = r3;
    pair2[p] = r2;
  }
};

struct FkWs {
  uint32_t *k3, *k3s, *k2, *k2s;
  int32_t *r3, *r3s, *r2, *r2s, *flag, *partner, *start3, *start2, *bad;
  int* tiles;
  void* cub;
  size_t cub_bytes;
};
template <typename A>
void carve_fk(A& a, FkWs* w, int n3, int n2, int batch) {
  const int m3 = n3 > 0 ? n3 : 1, m2 = n2 > 0 ? n2 : 1;
  size_t b3 = 0, b2 = 0;
  hipcub::DeviceRadixSort::SortPairs(nullptr, b3, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                     (int32_t*)nullptr, (int32_t*)nullptr, m3);
  hipcub::DeviceRadixSort::SortPairs(nullptr, b2, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                     (int32_t*)nullptr, (int32_t*)nullptr, m2);
  FkWs v;
  v.k3 = a.template take<uint32_t>(m3);
  v.k3s = a.template take<uint32_t>(m3);
  v.k2 = a.template take<uint32_t>(m2);
  v.k2s = a.template take<uint32_t>(m2);
  v.r3 = a.template take<int32_t>(m3);
  v.r3s = a.template take<int32_t>(m3);
  v.r2 = a.template take<int32_t>(m2);
  v.r2s = a.template take<int32_t>(m2);
  v.flag = a.template take<int32_t>(m3);
  v.partner = a.template take<int32_t>(m3);
  v.start3 = a.template take<int32_t>(batch + 1);
  v.start2 = a.template take<int32_t>(batch + 1);
  v.bad = a.template take<int32_t>(1);
  v.tiles = a.template take<int>(scan_num_tiles(m3) + 1);
  v.cub_bytes = b3 > b2 ? b3 : b2;
  v.cub = a.template take<char>(v.cub_bytes);
  if (w) *w = v;
}

}  // namespace
}  // namespace msmd

using namespace msmd;

MSMD_EXPORT size_t msmd_modality_split_float_keys_workspace_bytes(int n3, int n2, int batch_size) {
  if (n3 < 0 || n2 < 0 || batch_size < 1) return 0;
  ArenaSize a;
  carve_fk(a, (FkWs*)nullptr, n3, n2, batch_size);
  return a.off;
}

MSMD_EXPORT int msmd_modality_split_float_keys(const int32_t* idx_3d, int n3,
                                               const int32_t* idx_2d, int n2, int batch_size,
                                               const int* spatial_shape, int32_t* mix3d,
                                               int32_t* mix2d, int32_t* pair_3d,
                                               int32_t* pair_2d, int32_t* n_mixed,
                                               int32_t* sample_stats, int reference_offsets,
                                               void* workspace, size_t workspace_bytes,
                                               msmd_stream_t stream) {
  if (n3 < 0 || n2 < 0 || batch_size < 1 || !spatial_shape || !n_mixed) return MSMD_ERR_INVALID_ARG;
  if ((n3 > 0 && (!idx_3d || !mix3d)) || (n2 > 0 && (!idx_2d || !mix2d))) return MSMD_ERR_INVALID_ARG;
  if (n3 > 0 && n2 > 0 && (!pair_3d || !pair_2d)) return MSMD_ERR_INVALID_ARG;
  for (int d = 0; d < 3; ++d)
    if (spatial_shape[d] < 1) return MSMD_ERR_INVALID_ARG;
  // largest key of the grid below 2^26 (sample id in the 6 bits above it), in the SAME float32
  // arithmetic the kernel keys with: a bound evaluated in double can sit just under 2^26 while
  // the float32 sum (spacing 4 up there) rounds up to it.  volatile: no contraction, no
  // excess precision on the host.
  volatile float kmax = (float)(spatial_shape[0] - 1) * 1e6f;
  kmax = kmax + (float)(spatial_shape[1] - 1) * 1e3f;
  kmax = kmax + (float)(spatial_shape[2] - 1);
  if (!(kmax < (float)(1u << kKeyBits)) || batch_size > 64) return MSMD_ERR_RANGE;
  Arena a(workspace, workspace_bytes);
  FkWs w;
  carve_fk(a, &w, n3, n2, batch_size);
  if (!a.ok()) return MSMD_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  hipMemsetAsync(w.bad, 0, sizeof(int32_t), st);
  if (sample_stats) hipMemsetAsync(sample_stats, 0, sizeof(int32_t) * 4 * batch_size, st);
  if (n3 == 0 || n2 == 0) {        // nothing can match: flags 0, no pairs, plain counts only
    hipMemsetAsync(n_mixed, 0, sizeof(int32_t), st);
    if (n3 > 0) hipMemsetAsync(mix3d, 0, sizeof(int32_t) * (size_t)n3, st);
    if (n2 > 0) hipMemsetAsync(mix2d, 0, sizeof(int32_t) * (size_t)n2, st);
  }
  int end_bit = kKeyBits;
  while ((1 << (end_bit - kKeyBits)) < batch_size) ++end_bit;
  if (n3 > 0) {
    MSMD_LAUNCH(fkey_kernel, dim3(ceil_div(n3, 256)), dim3(256), 0, st, idx_3d, n3, batch_size,
                spatial_shape[0], spatial_shape[1], spatial_shape[2], reference_offsets ? 1 : 0,
                w.k3, w.r3, w.bad);
    size_t cb = w.cub_bytes;
    if (hipcub::DeviceRadixSort::SortPairs(w.cub, cb, w.k3, w.k3s, w.r3, w.r3s, n3, 0, end_bit,
                                           st) != hipSuccess)
      return MSMD_ERR_LAUNCH;
  }
  if (n2 > 0) {
    MSMD_LAUNCH(fkey_kernel, dim3(ceil_div(n2, 256)), dim3(256), 0, st, idx_2d, n2, batch_size,
                spatial_shape[0], spatial_shape[1], spatial_shape[2], reference_offsets ? 1 : 0,
                w.k2, w.r2, w.bad);
    size_t cb = w.cub_bytes;
    if (hipcub::DeviceRadixSort::SortPairs(w.cub, cb, w.k2, w.k2s, w.r2, w.r2s, n2, 0, end_bit,
                                           st) != hipSuccess)
      return MSMD_ERR_LAUNCH;
  }
  const bool both = n3 > 0 && n2 > 0;
  if (n3 > 0)
    MSMD_LAUNCH(match_kernel, dim3(ceil_div(n3, 256)), dim3(256), 0, st, (const uint32_t*)w.k3s,
                (const int32_t*)w.r3s, n3, (const uint32_t*)w.k2s, (const int32_t*)w.r2s, n2,
                mix3d, both ? w.flag : nullptr, w.partner, sample_stats, batch_size);
  if (n2 > 0)
    MSMD_LAUNCH(match_kernel, dim3(ceil_div(n2, 256)), dim3(256), 0, st, (const uint32_t*)w.k2s,
                (const int32_t*)w.r2s, n2, (const uint32_t*)w.k3s, (const int32_t*)w.r3s, n3,
                mix2d, (int32_t*)nullptr, (int32_t*)nullptr,
                sample_stats ? sample_stats + 2 * batch_size : nullptr, batch_size);
  if (both) {
    const int32_t *s3 = nullptr, *s2 = nullptr;
    if (reference_offsets) {
      MSMD_LAUNCH(sample_starts_kernel, dim3(1), dim3(128), 0, st, idx_3d, n3, batch_size, w.start3);
      MSMD_LAUNCH(sample_starts_kernel, dim3(1), dim3(128), 0, st, idx_2d, n2, batch_size, w.start2);
      s3 = w.start3;
      s2 = w.start2;
    }
    const int cap = n3 < n2 ? n3 : n2;
    device_scan(FlagCount{w.flag},
                EmitPair{w.k3s, w.r3s, w.partner, s3, s2, pair_3d, pair_2d, cap}, n3, w.tiles,
                n_mixed, -1, st);
  }
  // a row outside the grid / batch, or (reference_offsets) rows not grouped by sample
  MSMD_LAUNCH(fkey_report_kernel, dim3(1), dim3(1), 0, st, (const int32_t*)w.bad, n_mixed);
  return launch_status();
}
