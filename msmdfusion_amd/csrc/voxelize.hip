// voxelize.hip -- hard voxelization (+ fused mean VFE) for gfx950.
//
// Replaces hard_voxelize_gpu (mmdet3d/ops/voxel/src/voxelization_cuda.cu:184-326:
// an O(N^2) predecessor scan, a <<<1,1>>> serial loop and four device syncs)
// with an order-exact parallel formulation of hard_voxelize_cpu's loop
// (mmdet3d/ops/voxel/src/voxelization_cpu.cpp:68-96):
//
//   1. key[i]   = linear cell of point i (float32 floor((p-min)/size), same
//                 operation order as voxelization_cpu.cpp:23), inserted into a
//                 64-bit-slot hash table keeping min(point index) per cell
//                 -> the voxel's first point.
//   2. flag[i]  = point i is the first point of its voxel; an exclusive scan
//                 of the flags in point order is the first-appearance voxel id.
//   3. i*       = the first point whose voxel id == max_voxels: the reference
//                 loop breaks there, so every point >= i* is dropped.
//   4. slot of a point = number of earlier same-voxel points: slot 0 is the
//                 first point; slot r is found by round r of "smallest point
//                 index above the previous winner" (atomicMin per voxel),
//                 max_points-1 rounds of a streaming pass.
//   5. gather: voxel-stationary, writes rows [0, voxel_num) only, padding
//                 slots zero-filled in the same pass (the reference memsets
//                 max_voxels*max_points*C floats per call, voxelize.py:46-50),
//                 optionally reducing straight to the mean (HardSimpleVFE).
//
// All passes are HBM-bound integer work with coalesced reads; no host sync.
#include "common.hpp"
#include "scan.hpp"

namespace msmd {
namespace {

constexpr uint32_t kNoCell = 0xFFFFFFFFu;
constexpr uint32_t kNoPoint = 0xFFFFFFFFu;  // empty slot (point indices compare as unsigned)

struct VoxGeom {
  float vs[3], lo[3];
  int grid[3];  // x,y,z
};

__global__ __launch_bounds__(256) void vox_insert(const float* __restrict__ points, int n, int c,
                                                  VoxGeom g, uint32_t* __restrict__ key,
                                                  uint32_t* __restrict__ slot,
                                                  unsigned long long* table, int bits,
                                                  int* __restrict__ istar) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (i == 0) *istar = n;      // "no point dropped" until the scan finds i* (a launch of its own before)
  const float* p = points + (size_t)i * c;
  int q[3];
  bool ok = true;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    int v = (int)floorf((p[j] - g.lo[j]) / g.vs[j]);
    if (v < 0 || v >= g.grid[j]) ok = false;
    q[j] = v;
  }
  uint32_t k = kNoCell, s = 0;
  if (ok) {
    k = ((uint32_t)q[2] * g.grid[1] + q[1]) * g.grid[0] + q[0];
    s = hash_insert<false>(table, bits, k, (uint32_t)i);
  }
  key[i] = k;
  slot[i] = s;
}

struct FirstFlag {  // 1 when point i is the first point of its cell
  const uint32_t* key;
  const uint32_t* slot;
  const unsigned long long* table;
  __device__ int operator()(int i) const {
    return key[i] != kNoCell && (uint32_t)table[slot[i]] == (uint32_t)i;
  }
};
struct RankEmit {  // rank[i] = voxel id if first point; records i* (step 3)
  int* rank;
  int* istar;
  int max_voxels;
  uint32_t* win;      // the first point of voxel p also clears the voxel's slot row: only the
  int max_points;     // rows in use are initialised (a fill of max_voxels rows went first:
                      // 48 MB per cloud at the stress size, 4.8 MB at the nominal one)
  __device__ void operator()(int i, int p, int v) const {
    rank[i] = p;
    if (!v) return;
    if (p == max_voxels) *istar = i;
    if (p < max_voxels) {
      uint32_t* w = win + (size_t)p * max_points;
      for (int s = 0; s < max_points; ++s) w[s] = kNoPoint;
    }
  }
};

// The atomicMin cascade of step 4 for one point.  Slot values only ever DECREASE, so a plain
// (possibly stale, i.e. too high) read that is already below the value carried in decides
// what the atomic would: the slot keeps its value and the carried one moves on -- no atomic.
// A voxel of a coarse virtual-point scale holds hundreds of points: without the reads every
// one of them did max_points same-line atomics (191 us for the 10 clouds of an LC step).
__device__ __forceinline__ void slot_cascade(uint32_t* w, int max_points, uint32_t carry) {
  const volatile uint32_t* wv = w;
  const uint32_t last = wv[max_points - 1];
  if (last != kNoPoint && last < carry) return;     // ten smaller indices are in already
  for (int r = 0; r < max_points; ++r) {
    const uint32_t cur = wv[r];
    if (cur != kNoPoint && cur < carry) continue;   // would lose here: next slot
    const uint32_t old = atomicMin(&w[r], carry);
    if (old == kNoPoint) break;          // the slot was empty: nothing displaced
    carry = old > carry ? old : carry;
  }
}

// step 4 + coordinates: every kept point learns its voxel id and inserts its index
// into the voxel's `max_points` slots, which end up holding the smallest indices in
// ascending order -- one pass.  Slot r is an atomicMin cell: a point (or the value it
// displaced) that loses at slot r carries max(own, old) on to slot r + 1.  Whatever the
// interleaving, slot r receives the multiset that entered slot r - 1 minus its minimum,
// so slot r = the (r+1)-th smallest index: exactly the reference loop's "first
// max_points points of the voxel in point order" (voxelization_cpu.cpp:84-95).
__global__ __launch_bounds__(256) void vox_assign(int n, VoxGeom g,
                                                  const uint32_t* __restrict__ key,
                                                  const uint32_t* __restrict__ slot,
                                                  const unsigned long long* __restrict__ table,
                                                  const int* __restrict__ rank,
                                                  const int* __restrict__ istar,
                                                  uint32_t* __restrict__ win, int max_points,
                                                  int32_t* __restrict__ coors) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t k = key[i];
  if (k == kNoCell || i >= *istar) return;
  const int first = (int)(uint32_t)table[slot[i]];
  const int v = rank[first];
  if (first == i) {
    int x = k % g.grid[0];
    int y = (k / g.grid[0]) % g.grid[1];
    int z = k / (g.grid[0] * g.grid[1]);
    coors[(size_t)v * 3 + 0] = z;
    coors[(size_t)v * 3 + 1] = y;
    coors[(size_t)v * 3 + 2] = x;
  }
  uint32_t* w = win + (size_t)v * max_points;
  slot_cascade(w, max_points, (uint32_t)i);
}

// step 5: one thread per (voxel, channel); slots walked in order.
__global__ __launch_bounds__(256) void vox_gather(const float* __restrict__ points, int c,
                                                  const uint32_t* __restrict__ win, int max_points,
                                                  const int* __restrict__ voxel_num,
                                                  float* __restrict__ voxels,
                                                  int32_t* __restrict__ num_points,
                                                  float* __restrict__ mean) {
  const long total = (long)(*voxel_num) * c;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    int v = (int)(t / c), ch = (int)(t % c);
    const uint32_t* w = win + (size_t)v * max_points;
    float sum = 0.f;
    int cnt = 0;
    for (int s = 0; s < max_points; ++s) {
      const uint32_t src = w[s];
      float val = 0.f;
      if (src != kNoPoint) {
        val = points[(size_t)src * c + ch];
        ++cnt;
      }
      if (voxels) voxels[((size_t)v * max_points + s) * c + ch] = val;
      sum += val;
    }
    if (mean) mean[(size_t)v * c + ch] = sum / (float)cnt;
    if (ch == 0) num_points[v] = cnt;
  }
}

__global__ void vox_init_scalars(int* istar, int n) { *istar = n; }

struct VoxWs {
  uint32_t *key, *slot, *win;
  int *rank, *tiles, *istar;
  unsigned long long* table;
  int bits;
};

template <typename A>
void carve(A& a, VoxWs* w, int n, int max_voxels, int max_points) {
  int bits = next_pow2_bits(2L * (n > 0 ? n : 1));
  if (bits < 6) bits = 6;
  if (w) w->bits = bits;
#define TAKE(field, T, cnt)              \
  {                                      \
    T* p_ = a.template take<T>(cnt);     \
    if (w) w->field = p_;                \
  }
  TAKE(key, uint32_t, n);
  TAKE(slot, uint32_t, n);
  TAKE(rank, int, n);
  TAKE(tiles, int, scan_num_tiles(n) + 1);
  TAKE(istar, int, 64);
  TAKE(win, uint32_t, (size_t)max_voxels * max_points);
  TAKE(table, unsigned long long, (size_t)1 << bits);
#undef TAKE
}

}  // namespace
}  // namespace msmd

using namespace msmd;

MSMD_EXPORT size_t msmd_voxelize_workspace_bytes(int num_points, int max_voxels, int max_points) {
  ArenaSize a;
  carve(a, (VoxWs*)nullptr, num_points, max_voxels, max_points);
  return a.off;
}

MSMD_EXPORT int msmd_hard_voxelize(const float* points, int num_points, int num_features,
                                   const float* voxel_size, const float* coors_range,
                                   int max_points, int max_voxels, float* voxels,
                                   int32_t* coors, int32_t* num_points_per_voxel,
                                   float* voxel_mean, int32_t* voxel_num, void* workspace,
                                   size_t workspace_bytes, msmd_stream_t stream) {
  if (num_points < 0 || num_features < 3 || max_points < 1 || max_voxels < 1 || !coors ||
      !num_points_per_voxel || !voxel_num || !voxel_size || !coors_range ||
      (num_points > 0 && !points))
    return MSMD_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  VoxGeom g;
  double cells = 1;
  for (int i = 0; i < 3; ++i) {
    g.vs[i] = voxel_size[i];
    g.lo[i] = coors_range[i];
    // voxelization_cpu.cpp:119-122 -- round() of the float quotient
    g.grid[i] = (int)roundf((coors_range[3 + i] - coors_range[i]) / voxel_size[i]);
    if (g.grid[i] < 1) return MSMD_ERR_INVALID_ARG;
    cells *= g.grid[i];
  }
  if (cells >= 4294967295.0) return MSMD_ERR_RANGE;
  Arena a(workspace, workspace_bytes);
  VoxWs w;
  carve(a, &w, num_points, max_voxels, max_points);
  if (!a.ok()) return MSMD_ERR_WORKSPACE;

  const int n = num_points;
  hipMemsetAsync(w.table, 0xFF, sizeof(unsigned long long) << w.bits, st);
  const int nb = ceil_div(n, 256);
  if (n > 0)
    MSMD_LAUNCH(vox_insert, dim3(nb), dim3(256), 0, st, points, n, num_features, g, w.key,
                       w.slot, w.table, w.bits, w.istar);
  else
    MSMD_LAUNCH(vox_init_scalars, dim3(1), dim3(1), 0, st, w.istar, n);
  device_scan(FirstFlag{w.key, w.slot, w.table},
              RankEmit{w.rank, w.istar, max_voxels, w.win, max_points}, n, w.tiles, voxel_num,
              max_voxels, st);
  if (n > 0) {
    MSMD_LAUNCH(vox_assign, dim3(nb), dim3(256), 0, st, n, g, w.key, w.slot, w.table,
                       w.rank, w.istar, w.win, max_points, coors);
  }
  long work = (long)max_voxels * num_features;
  int gb = ceil_div(work, 256);
  if (gb > 4096) gb = 4096;
  MSMD_LAUNCH(vox_gather, dim3(gb), dim3(256), 0, st, points, num_features, w.win,
                     max_points, voxel_num, voxels, num_points_per_voxel, voxel_mean);
  return launch_status();
}

// ------------------------------------------------------------ many clouds --
// The same five passes for SEVERAL clouds per launch (the LiDAR sweeps of a batch and the
// virtual points of its four image scales: 10 clouds per LC step, each its own voxel size and
// channel count): a cloud's passes are a chain of short kernels -- 290 k points are ~1100
// blocks, one wave round -- so clouds voxelized one after the other are bound by launch
// latency (10 x 6 launches + 10 fills per step; 4 x 290 k points at the stress size: 545 us
// for 212 MB = 0.05 of HBM).  Blocks find their cloud from the table's block prefixes, the
// scan restarts at every cloud (a cloud owns whole scan tiles), one fill covers every cloud's
// slot and hash tables.
namespace msmd {
namespace {

constexpr int kVoxMany = 12;      // clouds per launch set (the table travels as a kernel argument)

struct VoxJob {
  const float* points;
  uint32_t *key, *slot, *win;
  int *rank, *istar;
  unsigned long long* table;
  float *voxels, *mean;
  int32_t *coors, *npv, *voxel_num;
  VoxGeom g;
  int n, c, bits, max_points, max_voxels;
};
struct VoxTab {
  int n;
  int blk0[kVoxMany + 1];     // first 256-point block of cloud s
  int tile0[kVoxMany + 1];    // first scan tile
  int gblk0[kVoxMany + 1];    // first block of the gather pass
  VoxJob j[kVoxMany];
};

__device__ __forceinline__ int vox_job_of(const int* __restrict__ first, int n, int b) {
  int s = 0;
  while (s + 1 < n && b >= first[s + 1]) ++s;
  return s;
}

__global__ __launch_bounds__(256) void vox_insert_many(const VoxTab tab) {
  const int s = vox_job_of(tab.blk0, tab.n, blockIdx.x);
  const VoxJob& J = tab.j[s];
  const int i = (blockIdx.x - tab.blk0[s]) * 256 + threadIdx.x;
  if (i >= J.n) return;
  if (i == 0) *J.istar = J.n;
  const float* p = J.points + (size_t)i * J.c;
  int q[3];
  bool ok = true;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    int v = (int)floorf((p[d] - J.g.lo[d]) / J.g.vs[d]);
    if (v < 0 || v >= J.g.grid[d]) ok = false;
    q[d] = v;
  }
  uint32_t k = kNoCell, sl = 0;
  if (ok) {
    k = ((uint32_t)q[2] * J.g.grid[1] + q[1]) * J.g.grid[0] + q[0];
    sl = hash_insert<false>(J.table, J.bits, k, (uint32_t)i);
  }
  J.key[i] = k;
  J.slot[i] = sl;
}

__global__ __launch_bounds__(kScanBlock) void vox_sums_many(const VoxTab tab,
                                                           int* __restrict__ tile_sums) {
  __shared__ int smem[kScanBlock / 64];
  const int s = vox_job_of(tab.tile0, tab.n, blockIdx.x);
  const VoxJob& J = tab.j[s];
  const FirstFlag count{J.key, J.slot, J.table};
  const int base = (blockIdx.x - tab.tile0[s]) * kScanTile;
  int c = 0;
#pragma unroll
  for (int q = 0; q < kScanItems; ++q) {
    const int i = base + q * kScanBlock + threadIdx.x;
    if (i < J.n) c += count(i);
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

__global__ __launch_bounds__(kScanBlock) void vox_rank_many(const VoxTab tab,
                                                           const int* __restrict__ tile_sums) {
  __shared__ int smem[kScanSmem];
  const int s = vox_job_of(tab.tile0, tab.n, blockIdx.x);
  const VoxJob& J = tab.j[s];
  const FirstFlag count{J.key, J.slot, J.table};
  const RankEmit emit{J.rank, J.istar, J.max_voxels, J.win, J.max_points};
  const int base = (blockIdx.x - tab.tile0[s]) * kScanTile;
  int v[kScanItems], ex[kScanItems];
#pragma unroll
  for (int q = 0; q < kScanItems; ++q) {
    const int i = base + q * kScanBlock + threadIdx.x;
    v[q] = i < J.n ? count(i) : 0;
  }
  int carry = block_range_sum<kScanBlock>(tile_sums, tab.tile0[s], (int)blockIdx.x, smem);
  const int tot = tile_excl_scan(v, ex, smem);
#pragma unroll
  for (int q = 0; q < kScanItems; ++q) {
    const int i = base + q * kScanBlock + threadIdx.x;
    if (i < J.n) emit(i, carry + ex[q], v[q]);
  }
  carry += tot;
  if ((int)blockIdx.x == tab.tile0[s + 1] - 1 && threadIdx.x == 0)
    *J.voxel_num = carry > J.max_voxels ? J.max_voxels : carry;
}

__global__ __launch_bounds__(256) void vox_assign_many(const VoxTab tab) {
  const int s = vox_job_of(tab.blk0, tab.n, blockIdx.x);
  const VoxJob& J = tab.j[s];
  const int i = (blockIdx.x - tab.blk0[s]) * 256 + threadIdx.x;
  if (i >= J.n) return;
  const uint32_t k = J.key[i];
  if (k == kNoCell || i >= *J.istar) return;
  const int first = (int)(uint32_t)J.table[J.slot[i]];
  const int v = J.rank[first];
  if (first == i) {
    const int x = k % J.g.grid[0];
    const int y = (k / J.g.grid[0]) % J.g.grid[1];
    const int z = k / (J.g.grid[0] * J.g.grid[1]);
    J.coors[(size_t)v * 3 + 0] = z;
    J.coors[(size_t)v * 3 + 1] = y;
    J.coors[(size_t)v * 3 + 2] = x;
  }
  slot_cascade(J.win + (size_t)v * J.max_points, J.max_points, (uint32_t)i);
}

__global__ __launch_bounds__(256) void vox_gather_many(const VoxTab tab) {
  const int s = vox_job_of(tab.gblk0, tab.n, blockIdx.x);
  const VoxJob& J = tab.j[s];
  const int c = J.c, max_points = J.max_points;
  const long total = (long)(*J.voxel_num) * c;
  const long stride = (long)(tab.gblk0[s + 1] - tab.gblk0[s]) * 256;
  for (long t = (long)(blockIdx.x - tab.gblk0[s]) * 256 + threadIdx.x; t < total; t += stride) {
    const int v = (int)(t / c), ch = (int)(t % c);
    const uint32_t* w = J.win + (size_t)v * max_points;
    float sum = 0.f;
    int cnt = 0;
    for (int sl = 0; sl < max_points; ++sl) {
      const uint32_t src = w[sl];
      float val = 0.f;
      if (src != kNoPoint) {
        val = J.points[(size_t)src * c + ch];
        ++cnt;
      }
      if (J.voxels) J.voxels[((size_t)v * max_points + sl) * c + ch] = val;
      sum += val;
    }
    if (J.mean) J.mean[(size_t)v * c + ch] = sum / (float)cnt;
    if (ch == 0) J.npv[v] = cnt;
  }
}

struct VoxManyWs {
  char* ones;
  size_t ones_bytes;
  int* tile_sums;
};

// jobs == nullptr: sizes only.  Jobs with no points are skipped (the caller routes them to the
// single-cloud entry point).
template <typename A>
void carve_vox_many(A& a, const msmd_voxelize_desc* d, int n_desc, VoxJob* jobs, VoxManyWs* w) {
  char* ones = (char*)a.template take<char>(0);
  const size_t o0 = a.off;
  for (int i = 0; i < n_desc; ++i) {
    if (d[i].num_points <= 0) continue;
    int bits = next_pow2_bits(2L * d[i].num_points);
    if (bits < 6) bits = 6;
    auto* tab = a.template take<unsigned long long>((size_t)1 << bits);
    if (jobs) jobs[i].table = tab, jobs[i].bits = bits;
  }
  const size_t o1 = a.off;
  for (int i = 0; i < n_desc; ++i) {      // slot rows: cleared voxel by voxel by the rank pass
    if (d[i].num_points <= 0) continue;
    auto* win = a.template take<uint32_t>((size_t)d[i].max_voxels * d[i].max_points);
    if (jobs) jobs[i].win = win;
  }
  long tiles = 0;
  for (int i = 0; i < n_desc; ++i) {
    const int n = d[i].num_points;
    if (n <= 0) continue;
    auto* key = a.template take<uint32_t>(n);
    auto* slot = a.template take<uint32_t>(n);
    auto* rank = a.template take<int>(n);
    auto* istar = a.template take<int>(64);
    if (jobs) jobs[i].key = key, jobs[i].slot = slot, jobs[i].rank = rank, jobs[i].istar = istar;
    tiles += scan_num_tiles(n);
  }
  int* ts = a.template take<int>(tiles + 1);
  if (w) *w = VoxManyWs{ones, o1 - o0, ts};
}

}  // namespace
}  // namespace msmd

using namespace msmd;

MSMD_EXPORT size_t msmd_hard_voxelize_many_workspace_bytes(const msmd_voxelize_desc* descs,
                                                           int n_desc) {
  if (!descs || n_desc < 1) return 0;
  size_t need = 0;
  for (int s0 = 0; s0 < n_desc; s0 += kVoxMany) {     // one launch set at a time reuses it
    ArenaSize a;
    const int cnt = n_desc - s0 < kVoxMany ? n_desc - s0 : kVoxMany;
    carve_vox_many(a, descs + s0, cnt, (VoxJob*)nullptr, (VoxManyWs*)nullptr);
    need = a.off > need ? a.off : need;
  }
  return need;
}

MSMD_EXPORT int msmd_hard_voxelize_many(const msmd_voxelize_desc* descs, int n_desc,
                                        void* workspace, size_t workspace_bytes,
                                        msmd_stream_t stream) {
  if (!descs || n_desc < 1) return MSMD_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  for (int s0 = 0; s0 < n_desc; s0 += kVoxMany) {
    const int cnt = n_desc - s0 < kVoxMany ? n_desc - s0 : kVoxMany;
    const msmd_voxelize_desc* d = descs + s0;
    VoxJob jobs[kVoxMany] = {};
    VoxTab tab;
    tab.n = 0;
    tab.blk0[0] = tab.tile0[0] = tab.gblk0[0] = 0;
    for (int i = 0; i < cnt; ++i) {
      if (d[i].num_points < 0 || d[i].num_features < 3 || d[i].max_points < 1 ||
          d[i].max_voxels < 1 || !d[i].coors || !d[i].num_points_per_voxel || !d[i].voxel_num ||
          (d[i].num_points > 0 && !d[i].points))
        return MSMD_ERR_INVALID_ARG;
      double cells = 1;
      for (int k = 0; k < 3; ++k) {
        jobs[i].g.vs[k] = d[i].voxel_size[k];
        jobs[i].g.lo[k] = d[i].coors_range[k];
        jobs[i].g.grid[k] =
            (int)roundf((d[i].coors_range[3 + k] - d[i].coors_range[k]) / d[i].voxel_size[k]);
        if (jobs[i].g.grid[k] < 1) return MSMD_ERR_INVALID_ARG;
        cells *= jobs[i].g.grid[k];
      }
      if (cells >= 4294967295.0) return MSMD_ERR_RANGE;
    }
    Arena a(workspace, workspace_bytes);
    VoxManyWs w;
    carve_vox_many(a, d, cnt, jobs, &w);
    if (!a.ok()) return MSMD_ERR_WORKSPACE;
    for (int i = 0; i < cnt; ++i) {
      if (d[i].num_points <= 0) {        // nothing to scan: no voxel
        hipMemsetAsync(d[i].voxel_num, 0, sizeof(int32_t), st);
        continue;
      }
      VoxJob& J = jobs[i];
      J.points = d[i].points;
      J.n = d[i].num_points;
      J.c = d[i].num_features;
      J.max_points = d[i].max_points;
      J.max_voxels = d[i].max_voxels;
      J.voxels = d[i].voxels;
      J.mean = d[i].voxel_mean;
      J.coors = d[i].coors;
      J.npv = d[i].num_points_per_voxel;
      J.voxel_num = d[i].voxel_num;
      const int s = tab.n++;
      tab.j[s] = J;
      tab.blk0[s + 1] = tab.blk0[s] + ceil_div(J.n, 256);
      tab.tile0[s + 1] = tab.tile0[s] + scan_num_tiles(J.n);
      long gb = ceil_div((long)J.max_voxels * J.c, 256);
      gb = gb > 2048 ? 2048 : gb;
      tab.gblk0[s + 1] = tab.gblk0[s] + (int)gb;
    }
    if (tab.n == 0) continue;
    for (int s = tab.n + 1; s <= kVoxMany; ++s)
      tab.blk0[s] = tab.tile0[s] = tab.gblk0[s] = 0x7fffffff;
    if (w.ones_bytes) hipMemsetAsync(w.ones, 0xFF, w.ones_bytes, st);
    MSMD_LAUNCH(vox_insert_many, dim3(tab.blk0[tab.n]), dim3(256), 0, st, tab);
    MSMD_LAUNCH(vox_sums_many, dim3(tab.tile0[tab.n]), dim3(kScanBlock), 0, st, tab, w.tile_sums);
    MSMD_LAUNCH(vox_rank_many, dim3(tab.tile0[tab.n]), dim3(kScanBlock), 0, st, tab,
                (const int*)w.tile_sums);
    MSMD_LAUNCH(vox_assign_many, dim3(tab.blk0[tab.n]), dim3(256), 0, st, tab);
    MSMD_LAUNCH(vox_gather_many, dim3(tab.gblk0[tab.n]), dim3(256), 0, st, tab);
  }
  return launch_status();
}

namespace {
__global__ __launch_bounds__(256) void voxel_mean_kernel(const float* __restrict__ voxels,
                                                         const int32_t* __restrict__ npv, int m,
                                                         int mp, int c, int oc,
                                                         float* __restrict__ out) {
  long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)m * oc) return;
  int v = (int)(t / oc), ch = (int)(t % oc);
  float s = 0.f;
  for (int p = 0; p < mp; ++p) s += voxels[((size_t)v * mp + p) * c + ch];
  out[t] = s / (float)npv[v];
}
}  // namespace

MSMD_EXPORT int msmd_voxel_mean(const float* voxels, const int32_t* num_points_per_voxel,
                                int num_voxels, int max_points, int num_features,
                                int out_features, float* out, msmd_stream_t stream) {
  if (num_voxels < 0 || out_features > num_features || out_features < 1)
    return MSMD_ERR_INVALID_ARG;
  if (num_voxels == 0) return MSMD_OK;
  MSMD_LAUNCH(voxel_mean_kernel, dim3(ceil_div((long)num_voxels * out_features, 256)),
                     dim3(256), 0, (hipStream_t)stream, voxels, num_points_per_voxel, num_voxels,
                     max_points, num_features, out_features, out);
  return launch_status();
}
