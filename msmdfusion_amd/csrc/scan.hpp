// scan.hpp -- device-wide exclusive prefix sum in two small kernels (tile sums -> apply),
// generic over where the per-element count comes from and what is done with the prefix.
// (Round 3: the single-block scan of the tile sums between the two is gone -- every apply
// block adds up the tile sums in front of it itself, a few KB from L2.  The index pass of
// an LC step runs 44 scans on a stream whose launches queue behind the feature pass's
// persistent kernels: a launch there costs far more than the few microseconds of its work.)
//   Count: int operator()(int i) const          -- value of element i
//   Emit : void operator()(int i, int prefix, int value) const
// Used for: popcount ranks of occupancy bitmaps (strided rulebook,
// sparse_add, modality split), first-point flags (voxelization) and per-offset
// pair compaction.  All HBM-bound streaming passes with coalesced access.
#pragma once
#include "common.hpp"

namespace msmd {

constexpr int kScanBlock = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanBlock * kScanItems;

inline int scan_num_tiles(long n) { return ceil_div(n, kScanTile); }

template <typename Count>
__global__ __launch_bounds__(kScanBlock) void scan_tile_sums(Count count, int n,
                                                             int* __restrict__ tile_sums) {
  __shared__ int smem[kScanBlock / 64];
  const int base = blockIdx.x * kScanTile;
  int s = 0;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    int i = base + j * kScanBlock + threadIdx.x;  // coalesced
    if (i < n) s += count(i);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int i = 0; i < kScanBlock / 64; ++i) t += smem[i];
    tile_sums[blockIdx.x] = t;
  }
}

// Sum of a[lo .. hi) by the whole block (same value in every thread).  smem: BLOCK/64 ints.
template <int BLOCK>
__device__ __forceinline__ int block_range_sum(const int* __restrict__ a, int lo, int hi,
                                               int* smem) {
  int s = 0;
  for (int i = lo + (int)threadIdx.x; i < hi; i += BLOCK) s += a[i];
  s = wave_sum(s);
  __syncthreads();   // smem may still be read from a previous use
  if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = s;
  __syncthreads();
  int t = 0;
#pragma unroll
  for (int i = 0; i < BLOCK / 64; ++i) t += smem[i];
  // ... and the next user of smem (block_excl_scan writes it without a barrier in front) must
  // not overtake a wave still reading here.  Without this barrier a block now and then took
  // a wrong carry: garbage positions in pair lists / row ranks, only under load (round 3:
  // memory faults in one bench leg out of ~6 -- found by bisecting, a bisection script of round 3, since pruned: git history)
  __syncthreads();
  return t;
}

// Exclusive prefixes of a tile: kScanItems values per thread, item-major (item j of every
// thread precedes item j + 1 of any thread, the order of the coalesced loads), across the
// block with ONE barrier pair for all items -- the values are loaded up front (kScanItems
// independent loads in flight instead of a load -> two barriers -> load chain per item: the
// index pass runs ~50 of these scans per LC step as short kernels whose time is that chain).
// smem: kScanItems * kScanBlock / 64 ints.  ex[j] <- prefix inside the tile; returns the total.
constexpr int kScanSmem = kScanItems * (kScanBlock / 64);
__device__ __forceinline__ int tile_excl_scan(const int (&v)[kScanItems], int (&ex)[kScanItems],
                                              int* smem) {
  constexpr int NW = kScanBlock / 64;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    ex[j] = wave_excl_scan(v[j], lane);
    if (lane == 63) smem[j * NW + w] = ex[j] + v[j];
  }
  __syncthreads();
  int run = 0;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int s = smem[j * NW + i];
      base += i < w ? s : 0;
      tot += s;
    }
    ex[j] += run + base;
    run += tot;
  }
  __syncthreads();      // smem may be written again
  return run;
}

// total <- 0 for an empty input
static __global__ void scan_empty_total(int* __restrict__ total) {
  if (total) *total = 0;
}

template <typename Count, typename Emit>
__global__ __launch_bounds__(kScanBlock) void scan_apply(Count count, Emit emit, int n,
                                                         const int* __restrict__ tile_sums,
                                                         int* __restrict__ total, int clamp) {
  __shared__ int smem[kScanSmem];
  const int base = blockIdx.x * kScanTile;
  int v[kScanItems], ex[kScanItems];
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    const int i = base + j * kScanBlock + threadIdx.x;
    v[j] = i < n ? count(i) : 0;
  }
  int carry = block_range_sum<kScanBlock>(tile_sums, 0, blockIdx.x, smem);
  const int tot = tile_excl_scan(v, ex, smem);
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    const int i = base + j * kScanBlock + threadIdx.x;
    if (i < n) emit(i, carry + ex[j], v[j]);
  }
  carry += tot;
  if (total && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0)
    *total = (clamp >= 0 && carry > clamp) ? clamp : carry;
}

// Host driver.  tile_sums: scratch of scan_num_tiles(n) ints (raw tile sums afterwards).
template <typename Count, typename Emit>
inline void device_scan(Count count, Emit emit, int n, int* tile_sums, int* total, int clamp,
                        hipStream_t st) {
  const int nt = scan_num_tiles(n);
  if (nt == 0) {
    MSMD_LAUNCH(scan_empty_total, dim3(1), dim3(1), 0, st, total);
    return;
  }
  MSMD_LAUNCH(scan_tile_sums<Count>, dim3(nt), dim3(kScanBlock), 0, st, count, n,
                     tile_sums);
  MSMD_LAUNCH((scan_apply<Count, Emit>), dim3(nt), dim3(kScanBlock), 0, st, count, emit, n,
                     (const int*)tile_sums, total, clamp);
}

// ---- occupancy bitmap + popcount rank ---------------------------------------
// A set of 32-bit cell ids in [0, cells) is a bitmap of ceil(cells/32) words;
// after device_scan(PopcCount, StorePrefix) the row of a present cell, in
// ascending cell order, is prefix[cell>>5] + popc(bits[cell>>5] & low mask).
struct PopcCount {
  const uint32_t* bits;
  __device__ int operator()(int i) const { return __popc(bits[i]); }
};
struct StorePrefix {
  int* prefix;
  __device__ void operator()(int i, int p, int) const { prefix[i] = p; }
};
__device__ __forceinline__ void bitmap_set(uint32_t* bits, uint32_t cell) {
  atomicOr(&bits[cell >> 5], 1u << (cell & 31));
}
__device__ __forceinline__ bool bitmap_test(const uint32_t* __restrict__ bits, uint32_t cell) {
  return (bits[cell >> 5] >> (cell & 31)) & 1u;
}
__device__ __forceinline__ int bitmap_rank(const uint32_t* __restrict__ bits,
                                           const int* __restrict__ prefix, uint32_t cell) {
  return prefix[cell >> 5] + __popc(bits[cell >> 5] & ((1u << (cell & 31)) - 1u));
}

}  // namespace msmd
