// plan_many.hip -- msmd_rulebook_plan for ALL tables of an index pass in one launch set.
//
// One LC step plans ~21 neighbour tables (tiling order, table in tile order, stream-K
// prefix, pair lists, wgrad segment table).  Table by table that is ~13 launches each --
// the 27-bit radix sort alone 7-8 -- of kernels that finish in 5-20 us on 20k-170k rows:
// ~280 launches and 3.2 ms of the index queue's 8.5 ms (profiles/r04_lc_stream_summary.txt),
// and the index queue's length IS the step.  Nothing in the index chain
// reads a plan (only the feature pass does), so the plans can wait until every table
// exists and run together:
//   keys of all rows (table id above the mask key) -> ONE radix sort -> order + tiled
//   tables -> tile weights -> prefixes -> pair tile sums -> pair lists -> segment tables,
// 7 kernels + one sort whatever the number of tables.  Each table's results are exactly
// msmd_rulebook_plan's (the sort is stable and the table id is the key's top bits, so a
// table's rows keep the order its own sort gives them).  Integer work, exact.
#include <hipcub/hipcub.hpp>

#include "common.hpp"
#include "scan.hpp"
#include "tiling_key.hpp"

#include <stdlib.h>

namespace msmd {
int stream_k_c1();                                   // spconv_split.hip
size_t wgrad_segment_table_ints(int kvol, int nchunk);   // spconv_wgrad_block.hip

namespace {

constexpr int kPlanMax = 32;      // tables per launch set (5 key bits)
constexpr int kPlanMaxK = 32;

struct PlanTab {
  int n;
  int row0[kPlanMax + 1];    // first row of table s in the concatenation of all rows
  int blk0[kPlanMax + 1];    // ... first 256-row block
  int tile0[kPlanMax + 1];   // ... first tile-weight block (128-row tiles, then 256-row ones)
  int pblk0[kPlanMax + 1];   // ... first pair-scan block (kvol * tiles-per-offset each)
  msmd_plan_desc d[kPlanMax];
};

__device__ __forceinline__ int table_of(const int* __restrict__ first, int n, int b) {
  int s = 0;
  while (s + 1 < n && b >= first[s + 1]) ++s;     // uniform: scalar loads of kernel arguments
  return s;
}

__global__ __launch_bounds__(256) void keys_many_kernel(const PlanTab tab,
                                                        uint32_t* __restrict__ keys,
                                                        int32_t* __restrict__ vals) {
  const int s = table_of(tab.blk0, tab.n, blockIdx.x);
  const msmd_plan_desc& d = tab.d[s];
  const int o = (blockIdx.x - tab.blk0[s]) * 256 + threadIdx.x;
  if (o >= d.n_rows) return;
  const int g = tab.row0[s] + o;
  keys[g] = ((uint32_t)s << 27) | row_key32(d.nbr, d.kvol, (size_t)d.n_rows, o);
  vals[g] = o;
}

__global__ __launch_bounds__(256) void finish_many_kernel(const PlanTab tab,
                                                          const int32_t* __restrict__ sorted) {
  const int s = table_of(tab.blk0, tab.n, blockIdx.x);
  const msmd_plan_desc& d = tab.d[s];
  const int p = (blockIdx.x - tab.blk0[s]) * 256 + threadIdx.x;
  const int n = d.n_rows;
  if (p >= n) return;
  const int row = sorted[tab.row0[s] + p];
  d.order[p] = row;
  if (d.tiled) {
    const int32_t* __restrict__ nbr = d.nbr;
    int32_t* __restrict__ tiled = d.tiled;
    for (int k = 0; k < d.kvol; ++k) tiled[(size_t)k * n + p] = nbr[(size_t)k * n + row];
  }
}

// spconv_split.hip: tile_weight_kernel for every (table, tile height, tile) -- from the SORTED
// KEYS instead of the tile-ordered table: position p of table s holds the row whose key is
// keys[row0[s] + p], and the key's low bits are the row's offset mask (ranked for 3x3x3: a
// permutation of the bits, which the weight -- a sum over the offsets -- does not see).  4
// bytes per row instead of 4 K: weight = c1 * |union of the tile's masks| + sum over its
// 32-row groups of |union of the group's masks|, rows past the table's end standing for the
// last row exactly as the table-reading kernel has it.  One wave per tile.
__global__ __launch_bounds__(256) void tile_weight_many_kernel(const PlanTab tab, int c1,
                                                               const uint32_t* __restrict__ keys,
                                                               int n_tiles) {
  const int vt = blockIdx.x * 4 + (threadIdx.x >> 6);        // tile of this wave
  if (vt >= n_tiles) return;
  const int lane = threadIdx.x & 63;
  const int s = table_of(tab.tile0, tab.n, vt);
  const msmd_plan_desc& d = tab.d[s];
  const int n = d.n_rows, kvol = d.kvol;
  int t = vt - tab.tile0[s];
  const int t128 = d.prefix128 ? (n + 127) / 128 : 0;
  int rows = 128;
  int32_t* weight = d.prefix128;
  if (t >= t128) {
    t -= t128;
    rows = 256;
    weight = d.prefix256;
  }
  const uint32_t low = kvol >= 32 ? 0xffffffffu : ((1u << kvol) - 1u);
  const uint32_t* __restrict__ k = keys + tab.row0[s];
  uint32_t all = 0;
  int groups = 0;
  for (int g0 = 0; g0 < rows; g0 += 64) {                    // two 32-row groups per pass
    int p = t * rows + g0 + lane;
    p = p < n ? p : n - 1;
    uint32_t m = k[p] & low;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) m |= __shfl_xor(m, o);  // OR inside each 32-lane half
    const uint32_t m0 = __shfl(m, 0), m1 = __shfl(m, 32);
    all |= m0 | m1;
    groups += __popc(m0) + __popc(m1);
  }
  if (lane == 0) {
    const int w = c1 * __popc(all) + groups;
    weight[t] = w > 0 ? w : 1;
  }
}

// in place: a[0..n) weights -> a[0..n] exclusive prefix; block 2s = table s's 128-row
// prefix, 2s + 1 its 256-row one
__global__ __launch_bounds__(1024) void tile_prefix_many_kernel(const PlanTab tab) {
  __shared__ int part[1024];
  const msmd_plan_desc& d = tab.d[blockIdx.x >> 1];
  const int rows = (blockIdx.x & 1) ? 256 : 128;
  int32_t* __restrict__ a = (blockIdx.x & 1) ? d.prefix256 : d.prefix128;
  if (!a) return;
  const int n = (d.n_rows + rows - 1) / rows;
  const int per = (n + 1023) / 1024, t = threadIdx.x;
  const int b = t * per, e = b + per < n ? b + per : n;
  int s = 0;
  for (int i = b; i < e; ++i) s += a[i];
  part[t] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int v = t >= o ? part[t - o] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int run = part[t] - s;
  for (int i = b; i < e; ++i) {
    const int w = a[i];
    a[i] = run;
    run += w;
  }
  if (t == 1023) a[n] = part[1023];
}

inline int rows_padded(int n_rows) { return scan_num_tiles(n_rows > 0 ? n_rows : 1) * kScanTile; }

// rulebook.hip: scan_tile_sums<PairCount> of every table with pair lists
__global__ __launch_bounds__(kScanBlock) void pair_sums_many_kernel(const PlanTab tab,
                                                                    int* __restrict__ tile_sums) {
  __shared__ int smem[kScanBlock / 64];
  const int s = table_of(tab.pblk0, tab.n, blockIdx.x);
  const msmd_plan_desc& d = tab.d[s];
  const int n = d.n_rows;
  const int tpk = (n + kScanTile - 1) / kScanTile;
  const int b = blockIdx.x - tab.pblk0[s];
  const int k = b / tpk, base = (b - k * tpk) * kScanTile;
  const int32_t* __restrict__ nbr = d.nbr + (size_t)k * n;
  int c = 0;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    const int o = base + j * kScanBlock + threadIdx.x;
    if (o < n) c += nbr[o] >= 0;
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

// rulebook.hip: pairs_apply_kernel of every table with pair lists; the blocks of an offset
// also share the -1 fill of the lists' entries past the padded table (ld > rows_pad: the
// strided convs, whose input side is the longer one -- was a memset of the whole tensor)
__global__ __launch_bounds__(kScanBlock) void pairs_apply_many_kernel(
    const PlanTab tab, const int* __restrict__ tile_sums_all) {
  __shared__ int smem[kScanSmem];
  const int s = table_of(tab.pblk0, tab.n, blockIdx.x);
  const msmd_plan_desc& d = tab.d[s];
  const int n_rows = d.n_rows, ld = d.ld;
  const int tpk = (n_rows + kScanTile - 1) / kScanTile, rows_pad = tpk * kScanTile;
  const int* __restrict__ tile_sums = tile_sums_all + tab.pblk0[s];
  const int b = blockIdx.x - tab.pblk0[s];
  const int k = b / tpk, tile_in_k = b - k * tpk;
  int carry = block_range_sum<kScanBlock>(tile_sums, k * tpk, b, smem);
  const int num_k = carry + block_range_sum<kScanBlock>(tile_sums, b, (k + 1) * tpk, smem);
  if (tile_in_k == 0 && threadIdx.x == 0) d.indice_num[k] = num_k;
  int32_t* __restrict__ pin = d.indice_pairs + ((size_t)k * 2 + 0) * ld;
  int32_t* __restrict__ pout = pin + ld;
  const int32_t* __restrict__ nbr = d.nbr + (size_t)k * n_rows;
  const int base = tile_in_k * kScanTile;
  int src[kScanItems], v[kScanItems], ex[kScanItems];
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    const int o = base + j * kScanBlock + threadIdx.x;
    src[j] = o < n_rows ? nbr[o] : -1;
    v[j] = src[j] >= 0;
  }
  tile_excl_scan(v, ex, smem);
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    const int o = base + j * kScanBlock + threadIdx.x;
    const int pos = carry + ex[j];
    if (v[j]) {
      if (pos < ld) {
        pin[pos] = src[j];
        pout[pos] = o;
      }
    } else {
      const int tail = num_k + (o - pos);
      if (tail < ld) {
        pin[tail] = -1;
        pout[tail] = -1;
      }
    }
  }
  if (ld > rows_pad) {
    const int per = (ld - rows_pad + tpk - 1) / tpk;
    const int lo = rows_pad + tile_in_k * per, hi = lo + per < ld ? lo + per : ld;
    for (int i = lo + threadIdx.x; i < hi; i += kScanBlock) {
      pin[i] = -1;
      pout[i] = -1;
    }
  }
}

// spconv_wgrad_block.hip: pair_segments_kernel for one chunk (= the whole pair list of every
// offset): prefix[K + 1] | p0[K] = 0 | cnt[K] = num.  One wave per table.
__global__ __launch_bounds__(64) void segtab_many_kernel(const PlanTab tab) {
  const msmd_plan_desc& d = tab.d[blockIdx.x];
  if (!d.segtab) return;
  const int kvol = d.kvol, k = threadIdx.x;
  const int n = k < kvol ? d.indice_num[k] : 0;
  const int steps = (n + 31) >> 5;
  int incl = steps;                                 // wave inclusive scan
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o);
    if (k >= o) incl += v;
  }
  int32_t* __restrict__ t = d.segtab;
  if (k < kvol) {
    t[k] = incl - steps;
    t[kvol + 1 + k] = 0;
    t[2 * kvol + 1 + k] = n;
  }
  if (k == kvol - 1) t[kvol] = incl;
}

struct ManyWs {
  uint32_t *keys, *keys_out;
  int32_t *vals, *sorted;
  int* tile_sums;
  void* cub;
  size_t cub_bytes;
};

template <typename A>
void carve_many(A& a, ManyWs* w, long rows, long pair_blocks) {
  size_t cb = 0;
  const int n = rows > 0 ? (int)rows : 1;
  hipcub::DeviceRadixSort::SortPairs(nullptr, cb, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                     (int32_t*)nullptr, (int32_t*)nullptr, n);
  uint32_t* keys = a.template take<uint32_t>(n);
  uint32_t* keys_out = a.template take<uint32_t>(n);
  int32_t* vals = a.template take<int32_t>(n);
  int32_t* sorted = a.template take<int32_t>(n);
  int* sums = a.template take<int>(pair_blocks + 1);
  void* cub = a.template take<char>(cb);
  if (w) *w = ManyWs{keys, keys_out, vals, sorted, sums, cub, cb};
}

inline bool lpt_mode() {
  static const int lpt = [] { const char* e = getenv("MSMD_TILE_LPT"); return e ? atoi(e) : 0; }();
  return lpt != 0;
}

// tables the launch set takes (the others go through msmd_rulebook_plan one by one)
inline bool batchable(const msmd_plan_desc& d) {
  return d.n_rows > 0 && d.kvol >= 1 && d.kvol <= kPlanMaxK && row_key_bits(d.kvol) <= 27 &&
         !lpt_mode();
}

inline int check_desc(const msmd_plan_desc& d) {
  if (d.kvol < 1 || d.kvol > 31) return MSMD_ERR_UNSUPPORTED;
  if (d.n_rows < 0) return MSMD_ERR_INVALID_ARG;
  if (d.n_rows > 0 && (!d.nbr || !d.order)) return MSMD_ERR_INVALID_ARG;
  if ((d.prefix128 || d.prefix256) && d.n_rows > 0 && !d.tiled) return MSMD_ERR_INVALID_ARG;
  if (d.indice_pairs && (!d.indice_num || d.ld < d.n_rows)) return MSMD_ERR_INVALID_ARG;
  if (d.segtab && !d.indice_pairs) return MSMD_ERR_INVALID_ARG;
  if ((double)d.kvol * rows_padded(d.n_rows) >= 2147483647.0) return MSMD_ERR_RANGE;
  return MSMD_OK;
}

size_t single_bytes(const msmd_plan_desc& d) {
  return msmd_rulebook_plan_workspace_bytes(d.kvol, d.n_rows, 128);
}

// msmd_rulebook_plan + the segment table for one table (not batchable, or LPT mode)
int plan_single(const msmd_plan_desc& d, void* ws, size_t ws_bytes, hipStream_t st) {
  hipStream_t stream = st;
  if (d.n_rows == 0) {
    if (d.prefix128) hipMemsetAsync(d.prefix128, 0, sizeof(int32_t), stream);
    if (d.prefix256) hipMemsetAsync(d.prefix256, 0, sizeof(int32_t), stream);
    if (d.indice_pairs) {
      if (d.ld > 0)
        hipMemsetAsync(d.indice_pairs, 0xFF, sizeof(int32_t) * (size_t)d.kvol * 2 * d.ld, stream);
      hipMemsetAsync(d.indice_num, 0, sizeof(int32_t) * d.kvol, stream);
    }
  } else {
    int rc;
    if (d.tiled) {
      rc = msmd_rulebook_plan(d.nbr, d.kvol, d.n_rows, 128, d.order, d.tiled, d.prefix128,
                              d.prefix256, d.indice_pairs, d.ld, d.indice_num, ws, ws_bytes,
                              (msmd_stream_t)stream);
    } else {
      rc = msmd_rulebook_tiling(d.nbr, d.kvol, d.n_rows, 128, d.order, nullptr, ws,
                                msmd_rulebook_tiling_workspace_bytes(d.n_rows, 128),
                                (msmd_stream_t)stream);
      if (rc == MSMD_OK && d.indice_pairs)
        rc = msmd_rulebook_pairs(d.nbr, d.kvol, d.n_rows, d.indice_pairs, d.ld, d.indice_num, ws,
                                 ws_bytes, (msmd_stream_t)stream);
    }
    if (rc != MSMD_OK) return rc;
  }
  if (d.segtab)
    return msmd_rulebook_pair_segments(d.indice_pairs, d.indice_num, d.ld, d.kvol,
                                       d.ld > 0 ? d.ld : 1, 1, d.segtab, (msmd_stream_t)stream);
  return launch_status();
}

// The next launch set: batchable tables from descs[*i] on, at most kPlanMax of them and fewer
// than 2^31 rows in all (*i moves past the ones taken or skipped).  tab may be null (sizes
// only).  Returns the number of tables taken.
int next_group(const msmd_plan_desc* descs, int n_desc, int* i, PlanTab* tab, long* rows_out,
               long* pblk_out) {
  int n = 0;
  long rows = 0, pblk = 0, blk = 0, tile = 0;
  if (tab) tab->row0[0] = tab->blk0[0] = tab->tile0[0] = tab->pblk0[0] = 0;
  for (; *i < n_desc && n < kPlanMax; ++*i) {
    const msmd_plan_desc& d = descs[*i];
    if (!batchable(d)) continue;
    const long t = (d.prefix128 ? ceil_div(d.n_rows, 128) : 0) +
                   (d.prefix256 ? ceil_div(d.n_rows, 256) : 0);
    const long pb = d.indice_pairs ? (long)d.kvol * scan_num_tiles(d.n_rows) : 0;
    if (n > 0 && (rows + d.n_rows >= 2147483647L || pblk + pb >= 2147483647L)) break;
    rows += d.n_rows;
    blk += ceil_div(d.n_rows, 256);
    tile += t;
    pblk += pb;
    if (tab) {
      tab->d[n] = d;
      tab->row0[n + 1] = (int)rows;
      tab->blk0[n + 1] = (int)blk;
      tab->tile0[n + 1] = (int)tile;
      tab->pblk0[n + 1] = (int)pblk;
    }
    ++n;
  }
  if (tab) {
    tab->n = n;
    for (int s = n + 1; s <= kPlanMax; ++s)
      tab->row0[s] = tab->blk0[s] = tab->tile0[s] = tab->pblk0[s] = 0x7fffffff;
  }
  *rows_out = rows;
  *pblk_out = pblk;
  return n;
}

}  // namespace
}  // namespace msmd

using namespace msmd;

MSMD_EXPORT size_t msmd_rulebook_plan_many_workspace_bytes(const msmd_plan_desc* descs,
                                                           int n_desc) {
  if (!descs || n_desc < 0) return 0;
  size_t most = 0;
  for (int i = 0; i < n_desc; ++i)
    if (!batchable(descs[i])) {
      const size_t b = single_bytes(descs[i]);
      most = b > most ? b : most;
    }
  int i = 0;
  long rows, pblk;
  while (next_group(descs, n_desc, &i, nullptr, &rows, &pblk) > 0) {   // the sets plan_many runs
    ArenaSize a;
    carve_many(a, (ManyWs*)nullptr, rows, pblk);
    most = a.off > most ? a.off : most;
  }
  return most;
}

MSMD_EXPORT int msmd_rulebook_plan_many(const msmd_plan_desc* descs, int n_desc, void* workspace,
                                        size_t workspace_bytes, msmd_stream_t stream) {
  if (n_desc < 0 || (n_desc > 0 && !descs)) return MSMD_ERR_INVALID_ARG;
  if (n_desc == 0) return MSMD_OK;
  if (!workspace || ((uintptr_t)workspace & 255)) return MSMD_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  for (int i = 0; i < n_desc; ++i) {
    const int rc = check_desc(descs[i]);
    if (rc != MSMD_OK) return rc;
  }
  if (workspace_bytes < msmd_rulebook_plan_many_workspace_bytes(descs, n_desc))
    return MSMD_ERR_WORKSPACE;
  for (int i = 0; i < n_desc; ++i)
    if (!batchable(descs[i])) {
      const int rc = plan_single(descs[i], workspace, workspace_bytes, st);
      if (rc != MSMD_OK) return rc;
    }
  int i = 0;
  long rows, pblk;
  PlanTab tab;
  while (next_group(descs, n_desc, &i, &tab, &rows, &pblk) > 0) {
    if (rows >= 2147483647L) return MSMD_ERR_RANGE;      // (one table of 2^31 rows)
    bool any_tiles = false, any_pairs = false, any_seg = false, any_prefix = false;
    for (int s = 0; s < tab.n; ++s) {
      const msmd_plan_desc& d = tab.d[s];
      any_tiles |= tab.tile0[s + 1] > tab.tile0[s];
      any_prefix |= d.prefix128 || d.prefix256;
      any_pairs |= d.indice_pairs != nullptr;
      any_seg |= d.segtab != nullptr;
    }
    Arena a(workspace, workspace_bytes);
    ManyWs w;
    carve_many(a, &w, rows, tab.pblk0[tab.n]);
    if (!a.ok()) return MSMD_ERR_WORKSPACE;
    const int nblk = tab.blk0[tab.n];
    MSMD_LAUNCH(keys_many_kernel, dim3(nblk), dim3(256), 0, st, tab, w.keys, w.vals);
    int seg_bits = 0;
    while ((1 << seg_bits) < tab.n) ++seg_bits;
    size_t cb = w.cub_bytes;
    if (hipcub::DeviceRadixSort::SortPairs(w.cub, cb, w.keys, w.keys_out, w.vals, w.sorted,
                                           (int)rows, 0, 27 + seg_bits, st) != hipSuccess)
      return MSMD_ERR_LAUNCH;
    MSMD_LAUNCH(finish_many_kernel, dim3(nblk), dim3(256), 0, st, tab, (const int32_t*)w.sorted);
    if (any_tiles)
      MSMD_LAUNCH(tile_weight_many_kernel, dim3(ceil_div(tab.tile0[tab.n], 4)), dim3(256), 0, st,
                  tab, stream_k_c1(), (const uint32_t*)w.keys_out, tab.tile0[tab.n]);
    if (any_prefix)
      MSMD_LAUNCH(tile_prefix_many_kernel, dim3(2 * tab.n), dim3(1024), 0, st, tab);
    if (any_pairs) {
      const int pb = tab.pblk0[tab.n];
      MSMD_LAUNCH(pair_sums_many_kernel, dim3(pb), dim3(kScanBlock), 0, st, tab, w.tile_sums);
      MSMD_LAUNCH(pairs_apply_many_kernel, dim3(pb), dim3(kScanBlock), 0, st, tab,
                  (const int*)w.tile_sums);
    }
    if (any_seg) MSMD_LAUNCH(segtab_many_kernel, dim3(tab.n), dim3(64), 0, st, tab);
  }
  return launch_status();
}
