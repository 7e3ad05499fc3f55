// tiling.hip -- the tiling order of a rulebook in ONE call.
//
// The conv kernels tile the output rows of a neighbour table in an order that (a) puts
// rows with similar 27-bit offset masks into the same wave / 128-row tile and (b)
// sequences whole tiles heaviest first for the persistent scheduler (spconv.hip:
// row_mask_kernel / tile_cost_kernel have the reasoning and the measured effect).
// Done through torch this was 8 host-side ops per rulebook -- mask keys, a merge
// sort, tile costs, a second sort, index_select, cat, the column permutation -- and
// the LC step builds ~24 such orders: the index pass's host time and its merge-sort
// launches became the longest stage of the step pipeline.  Here: masks -> radix sort
// (hipCUB, only the key's significant bits) -> tile costs -> radix sort of the tiles
// -> one kernel that writes the final order and the table in tile order.
// Integer work, exact; any order gives the same conv results.
#include <hipcub/hipcub.hpp>

#include "common.hpp"

#include <stdlib.h>

namespace msmd {
// defined in spconv.hip
void launch_row_keys(const int32_t* nbr, int kvol, int n, uint32_t* keys, int32_t* row_ids,
                     int* key_bits, hipStream_t st);
void launch_tile_costs(const int32_t* nbr, int kvol, int n, const int32_t* order, int rows,
                       int32_t* cost, int32_t* tile_ids, hipStream_t st);

namespace {

// order[p] = sorted[tile_seq[t] * rows + p % rows] for the full tiles, the partial last
// tile stays last; tiled[k][p] = nbr[k][order[p]].
__global__ __launch_bounds__(256) void finish_kernel(const int32_t* __restrict__ nbr, int kvol,
                                                     int n, const int32_t* __restrict__ sorted,
                                                     const int32_t* __restrict__ tile_seq,
                                                     int full, int rows,
                                                     int32_t* __restrict__ order,
                                                     int32_t* __restrict__ tiled) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const int t = p / rows;
  // (tried: a zigzag sequence -- position 2i the i-th heaviest tile, 2i+1 the i-th lightest
  // -- so that every stream-K segment holds dense and light-mask tiles alike: no gain, 266
  // against 263 us on 128->128, and the dynamic scheduler lost its heaviest-first order)
  const int src = t < full ? tile_seq[t] * rows + (p - t * rows) : p;
  const int row = sorted[src];
  order[p] = row;
  if (tiled)
    for (int k = 0; k < kvol; ++k) tiled[(size_t)k * n + p] = nbr[(size_t)k * n + row];
}

struct TilingWs {
  uint32_t *keys, *keys_out;
  int32_t *vals, *sorted, *cost, *cost_out, *tile_ids, *tile_seq;
  void* cub;
  size_t cub_bytes;
};

template <typename A>
void carve(A& a, TilingWs* w, int n, int rows) {
  const int nt = ceil_div(n > 0 ? n : 1, rows);
  size_t b1 = 0, b2 = 0;
  hipcub::DeviceRadixSort::SortPairs(nullptr, b1, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                     (int32_t*)nullptr, (int32_t*)nullptr, n > 0 ? n : 1);
  hipcub::DeviceRadixSort::SortPairs(nullptr, b2, (int32_t*)nullptr, (int32_t*)nullptr,
                                     (int32_t*)nullptr, (int32_t*)nullptr, nt);
  const size_t cb = b1 > b2 ? b1 : b2;
  uint32_t* keys = a.template take<uint32_t>(n);
  uint32_t* keys_out = a.template take<uint32_t>(n);
  int32_t* vals = a.template take<int32_t>(n);
  int32_t* sorted = a.template take<int32_t>(n);
  int32_t* cost = a.template take<int32_t>(nt);
  int32_t* cost_out = a.template take<int32_t>(nt);
  int32_t* tile_ids = a.template take<int32_t>(nt);
  int32_t* tile_seq = a.template take<int32_t>(nt);
  void* cub = a.template take<char>(cb);
  if (w) *w = TilingWs{keys, keys_out, vals, sorted, cost, cost_out, tile_ids, tile_seq, cub, cb};
}

}  // namespace
}  // namespace msmd

using namespace msmd;

MSMD_EXPORT size_t msmd_rulebook_tiling_workspace_bytes(int n_rows, int rows_per_tile) {
  if (n_rows < 0 || rows_per_tile < 1) return 0;
  ArenaSize a;
  carve(a, (TilingWs*)nullptr, n_rows, rows_per_tile);
  return a.off;
}

MSMD_EXPORT int msmd_rulebook_tiling(const int32_t* nbr, int kernel_volume, int n_rows,
                                     int rows_per_tile, int32_t* order, int32_t* tiled,
                                     void* workspace, size_t workspace_bytes,
                                     msmd_stream_t stream) {
  if (kernel_volume < 1 || kernel_volume > 31) return MSMD_ERR_UNSUPPORTED;
  if (n_rows < 0 || rows_per_tile < 1 || (n_rows > 0 && (!nbr || !order)))
    return MSMD_ERR_INVALID_ARG;
  if (n_rows == 0) return MSMD_OK;
  Arena a(workspace, workspace_bytes);
  TilingWs w;
  carve(a, &w, n_rows, rows_per_tile);
  if (!a.ok()) return MSMD_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int n = n_rows, full = n / rows_per_tile;
  int key_bits = 32;
  launch_row_keys(nbr, kernel_volume, n, w.keys, w.vals, &key_bits, st);   // keys + row ids
  size_t cb = w.cub_bytes;
  if (hipcub::DeviceRadixSort::SortPairs(w.cub, cb, w.keys, w.keys_out, w.vals, w.sorted, n, 0,
                                         key_bits, st) != hipSuccess)
    return MSMD_ERR_LAUNCH;
  // Heaviest-first sequencing of whole tiles is what the DYNAMIC tile scheduler wants (LPT
  // list scheduling); stream-K -- the split kernels' schedule -- cuts the sequence into
  // equal cost shares wherever the tiles lie, so the cost pass and the second sort (5 of a
  // tiling's ~12 launches on the index stream) buy nothing there.  MSMD_TILE_LPT=1 restores
  // them (needed with MSMD_STREAMK=0).
  static const int lpt = [] { const char* e = getenv("MSMD_TILE_LPT"); return e ? atoi(e) : 0; }();
  if (full > 1 && lpt) {
    launch_tile_costs(nbr, kernel_volume, n, w.sorted, rows_per_tile, w.cost, w.tile_ids, st);
    cb = w.cub_bytes;
    if (hipcub::DeviceRadixSort::SortPairs(w.cub, cb, w.cost, w.cost_out, w.tile_ids, w.tile_seq,
                                           full, 0, 6, st) != hipSuccess)   // cost <= 31
      return MSMD_ERR_LAUNCH;
  }
  MSMD_LAUNCH(finish_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, nbr, kernel_volume, n,
              w.sorted, w.tile_seq, full > 1 && lpt ? full : 0, rows_per_tile, order, tiled);
  return launch_status();
}


// ---------------------------------------------------------------------------------------
// Everything the conv kernels want derived from ONE neighbour table, in one call: tiling
// order + table in tile order (msmd_rulebook_tiling), the stream-K prefixes for the tile
// heights in use (msmd_rulebook_tile_prefix: 128 and/or 256 rows), the reference pair lists
// for wgrad (msmd_rulebook_pairs).  Same results as the separate entry points; what it
// saves is host time in the index pass (the LC step is bound by the interpreter: four
// Python-level calls and a dozen allocations per table become one call).
MSMD_EXPORT size_t msmd_rulebook_plan_workspace_bytes(int kernel_volume, int n_rows,
                                                      int rows_per_tile) {
  return align_up(msmd_rulebook_tiling_workspace_bytes(n_rows, rows_per_tile)) +
         align_up(msmd_rulebook_pairs_workspace_bytes(kernel_volume, n_rows));
}

MSMD_EXPORT int msmd_rulebook_plan(const int32_t* nbr, int kernel_volume, int n_rows,
                                   int rows_per_tile, int32_t* order, int32_t* tiled,
                                   int32_t* prefix128, int32_t* prefix256, int32_t* indice_pairs,
                                   int ld, int32_t* indice_num, void* workspace,
                                   size_t workspace_bytes, msmd_stream_t stream) {
  if (!nbr || kernel_volume < 1 || n_rows < 1 || !order || !tiled) return MSMD_ERR_INVALID_ARG;
  const size_t t_bytes = align_up(msmd_rulebook_tiling_workspace_bytes(n_rows, rows_per_tile));
  const size_t p_bytes = align_up(msmd_rulebook_pairs_workspace_bytes(kernel_volume, n_rows));
  if (workspace_bytes < t_bytes + (indice_pairs ? p_bytes : 0) || ((uintptr_t)workspace & 255))
    return MSMD_ERR_WORKSPACE;
  int rc = msmd_rulebook_tiling(nbr, kernel_volume, n_rows, rows_per_tile, order, tiled,
                                workspace, t_bytes, stream);
  if (rc) return rc;
  if (prefix128) {
    rc = msmd_rulebook_tile_prefix(tiled, kernel_volume, n_rows, n_rows, 128, prefix128, stream);
    if (rc) return rc;
  }
  if (prefix256) {
    rc = msmd_rulebook_tile_prefix(tiled, kernel_volume, n_rows, n_rows, 256, prefix256, stream);
    if (rc) return rc;
  }
  if (indice_pairs) {
    if (!indice_num || ld < n_rows) return MSMD_ERR_INVALID_ARG;
    rc = msmd_rulebook_pairs(nbr, kernel_volume, n_rows, indice_pairs, ld, indice_num,
                             (char*)workspace + t_bytes, p_bytes, stream);
  }
  return rc;
}
