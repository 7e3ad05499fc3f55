// rulebook.hip -- voxel index + rulebook builders for gfx950.
//
// Native product: an OUTPUT-STATIONARY neighbour table nbr[K, n_out]
// (nbr[k][o] = input row that reaches output row o through kernel offset k,
// or -1).  That is the layout the implicit-GEMM kernels (spconv.hip) consume:
// one coalesced int32 load per (tile row, offset), no atomics, no scatter.
// msmd_rulebook_pairs() compacts it to the reference's indicePairs/indiceNum
// (spconv_ops.h:55-59) for the weight-gradient kernel and for parity checks.
//
// SubM (geometry.h:247-297; CUDA indice.cu.h:148-203 uses a dense int32 grid
// of B*D*H*W cells = 340 MB/sample at 41x1440x1440): a 64-bit-slot hash table
// of 2N slots (key = linear cell id, value = row), one probe sequence per
// (output row, offset), writes coalesced along the row axis.
//
// Strided (geometry.h:144-194; CUDA indice.cu.h:22-65,112-145 + torch::_unique
// sort): an occupancy bitmap of the OUTPUT grid (1 bit/cell: 1.4 MB/sample at
// 21x720x720) + popcount prefix scan.  The rank of a set bit IS the output row
// in ascending linear id -- the order the reference's CUDA path gets from its
// sort -- so no sort, no hash and no dedup pass are needed.
//
// Everything here is HBM/L2-bound integer work; only dilation 1 is built (all
// the reference configs use it), anything else returns MSMD_ERR_UNSUPPORTED.
#include "common.hpp"
#include "scan.hpp"

#include <stdlib.h>

namespace msmd {
namespace {

struct Geom {
  int shape[3];   // grid the table/bitmap indexes (z,y,x)
  int ks[3], st[3], pd[3];
  int kvol;
};

__device__ __forceinline__ uint32_t cell_id(int b, int z, int y, int x, const int* s) {
  return (((uint32_t)b * s[0] + z) * s[1] + y) * s[2] + x;
}

// ---------------------------------------------------------------- SubM ----
__global__ __launch_bounds__(256) void subm_insert(const int32_t* __restrict__ idx, int n,
                                                   Geom g, unsigned long long* table, int bits) {
  int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  int4 r = ((const int4*)idx)[j];
  // duplicate coordinates: the CPU reference's grid keeps the LAST row
  // (geometry.h:277-282) -> keep max(row)
  hash_insert<true>(table, bits, cell_id(r.x, r.y, r.z, r.w, g.shape), (uint32_t)j);
}

// thread (o, k): blockIdx.y = kernel offset -> stores of one block are
// contiguous in nbr[k][*]; the 16-B index row load is coalesced.
__global__ __launch_bounds__(256) void subm_lookup(const int32_t* __restrict__ idx, int n, Geom g,
                                                   const unsigned long long* __restrict__ table,
                                                   int bits, int32_t* __restrict__ nbr) {
  int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= n) return;
  const int k = blockIdx.y;
  const int kx = k % g.ks[2], ky = (k / g.ks[2]) % g.ks[1], kz = k / (g.ks[2] * g.ks[1]);
  int4 r = ((const int4*)idx)[o];
  // pair (in, out) sits under offset k when out = in + pad - k (stride 1,
  // geometry.h:40-44,69): the input of output o is at o - pad + k.
  int z = r.y - g.pd[0] + kz, y = r.z - g.pd[1] + ky, x = r.w - g.pd[2] + kx;
  int v = -1;
  if (z >= 0 && z < g.shape[0] && y >= 0 && y < g.shape[1] && x >= 0 && x < g.shape[2])
    v = hash_find(table, bits, cell_id(r.x, z, y, x, g.shape));
  nbr[(size_t)k * n + o] = v;
}

// SubM through an occupancy bitmap of the INPUT grid (large voxel sets).  The hash above
// costs one random 8-byte slot read per (row, offset): 27 sector fetches per voxel, 3.5 % of
// HBM peak on algorithmic bytes at 720 k voxels.  With one bit per cell the three
// x-neighbours of a (z, y) line come from ONE 32-bit word (two when they straddle a word),
// words of neighbouring voxels coincide, and a miss -- 24 of 27 probes on LiDAR data --
// needs nothing else.  A hit needs the cell's rank among the occupied cells: a prefix per
// BLOCK of 8 words (one 32-byte sector of the bitmap, 256 cells) plus the popcounts of the
// words before it in that sector, then rank -> row through one more table.  The bitmap costs
// a clear and one counting pass proportional to the GRID (1/8 byte per cell each; the prefix
// scan runs over 1/256 of the cells), so the caller picks it by size
// (kernels.rulebook_subm).
//
// Round 4: the grid-proportional part no longer touches the bitmap.  One BYTE per block
// (`used`, 1/256 byte per cell: 3 MB for the 765 M-cell stress grid against the fine bitmap's
// 95.6 MB) says whether the block holds a voxel.  Only `used` is cleared; every voxel stores
// 1 into its block's byte and zeros into the block's 32-byte sector (plain idempotent
// stores, subm_bm_mark_blocks), a second pass sets the fine bits, the counting passes read a
// sector only where the byte is set and store a prefix only for occupied blocks, and a
// look-up tests the byte first.  At the stress size that removes ~300 MB of traffic (clear +
// two counting reads of the bitmap + the prefix of every block) from a job whose algorithmic
// bytes are 106 MB.  (One BIT per block, set with atomicOr by the first voxel to arrive, was
// built first: LiDAR voxels come in scan order, hundreds of consecutive rows hit one word,
// and the same-address atomics made the stress case 2-8x SLOWER than clearing the bitmap.)
constexpr int kBmBlockWords = 8;

// Cell numbering of the bitmap: a block (= one 32-byte sector = 256 cells) is a 4 x 8 x 8
// (z, y, x) brick of the grid, bricks in (b, z, y, x) order, cells inside a brick in (z, y, x)
// order -- bit = (z & 3) << 6 | (y & 7) << 3 | (x & 7), so the cells of an x line inside a
// brick share one word.  A voxel's 3x3x3 neighbourhood lies in 2.3 bricks on average (at most
// 8) instead of in 9 different sectors of a row-major bitmap, and one thread resolves all of
// a voxel's offsets out of them (subm_bm_lookup).  The rank order is the brick order: any
// fixed order serves, rank -> row goes through rank2row.
struct BmDims {
  int tz, ty, tx;
};
__host__ __device__ __forceinline__ BmDims bm_dims(const int* shape) {
  return BmDims{(shape[0] + 3) >> 2, (shape[1] + 7) >> 3, (shape[2] + 7) >> 3};
}
inline size_t bm_blocks(int batch, const int* shape) {
  const BmDims d = bm_dims(shape);
  return (size_t)batch * d.tz * d.ty * d.tx;
}
__device__ __forceinline__ uint32_t bm_line(int b, int z, int y, const BmDims& d, uint32_t* bit) {
  *bit = ((uint32_t)(z & 3) << 6) | ((uint32_t)(y & 7) << 3);
  return (((uint32_t)b * d.tz + (z >> 2)) * d.ty + (y >> 3)) * d.tx;
}
__device__ __forceinline__ uint32_t bm_cell(int b, int z, int y, int x, const int* shape) {
  const BmDims d = bm_dims(shape);
  uint32_t bit;
  const uint32_t blk = bm_line(b, z, y, d, &bit) + (x >> 3);
  return (blk << 8) | bit | (uint32_t)(x & 7);
}

// used == nullptr: the bitmap was cleared as a whole (grids small against the voxel set: the
// per-voxel sector clears of the flag scheme cost more than clearing and counting it all)
__device__ __forceinline__ bool bm_block_used(const uint8_t* __restrict__ used, uint32_t blk) {
  return !used || used[blk] != 0;
}
// Flags when the fine bitmap has more than this many bytes per voxel (the stress grid: 133;
// the LC step's stage grids: 8-36)
constexpr size_t kBmFlagBytesPerVoxel = 64;
inline bool bm_use_flags(size_t nblocks, int n) {
  return nblocks * kBmBlockWords * sizeof(uint32_t) > kBmFlagBytesPerVoxel * (size_t)(n > 0 ? n : 1);
}

struct BmBlockCount {
  const uint32_t* bits;
  const uint8_t* used;
  __device__ int operator()(int i) const {
    if (!bm_block_used(used, (uint32_t)i)) return 0;
    const uint4* p = (const uint4*)(bits + (size_t)i * kBmBlockWords);
    int c = 0;
#pragma unroll
    for (int q = 0; q < kBmBlockWords / 4; ++q) {
      const uint4 v = p[q];
      c += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
    }
    return c;
  }
};
// prefix of the occupied blocks only (nothing reads the others')
struct StoreUsedPrefix {
  int* prefix;
  __device__ void operator()(int i, int p, int v) const {
    if (v) prefix[i] = p;
  }
};

__device__ __forceinline__ int bm_rank(const uint32_t* __restrict__ bits,
                                       const int* __restrict__ block_prefix, uint32_t cell,
                                       uint32_t word /* bits[cell >> 5], already loaded */) {
  const uint32_t wi = cell >> 5, blk = wi / kBmBlockWords, in_blk = wi % kBmBlockWords;
  int r = block_prefix[blk] + __popc(word & ((1u << (cell & 31)) - 1u));
  const uint4* line = (const uint4*)(bits + (size_t)blk * kBmBlockWords);   // one 32-byte sector
  const uint4 lo = line[0], hi = line[1];
  const uint32_t ws[kBmBlockWords] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
  for (uint32_t w = 0; w < (uint32_t)kBmBlockWords; ++w) r += w < in_blk ? __popc(ws[w]) : 0;
  return r;
}

// pass 1: the block's byte and a cleared sector (every voxel of a block stores the same values)
__global__ __launch_bounds__(256) void subm_bm_mark_blocks(const int32_t* __restrict__ idx, int n,
                                                           Geom g, uint8_t* used, uint32_t* bits) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const int4 r = ((const int4*)idx)[j];
  const uint32_t blk = bm_cell(r.x, r.y, r.z, r.w, g.shape) >> 8;
  used[blk] = 1;
  uint4* line = (uint4*)(bits + (size_t)blk * kBmBlockWords);
  line[0] = make_uint4(0, 0, 0, 0);
  line[1] = make_uint4(0, 0, 0, 0);
}
// pass 2 (a launch of its own: every sector is cleared before any fine bit is set)
__global__ __launch_bounds__(256) void subm_bm_mark(const int32_t* __restrict__ idx, int n, Geom g,
                                                    uint32_t* bits) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const int4 r = ((const int4*)idx)[j];
  bitmap_set(bits, bm_cell(r.x, r.y, r.z, r.w, g.shape));
}

// rank -> row; duplicate coordinates keep the LAST row, as the hash and the CPU grid do
__global__ __launch_bounds__(256) void subm_bm_rows(const int32_t* __restrict__ idx, int n, Geom g,
                                                    const uint32_t* __restrict__ bits,
                                                    const int* __restrict__ block_prefix,
                                                    int32_t* rank2row) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const int4 r = ((const int4*)idx)[j];
  const uint32_t c = bm_cell(r.x, r.y, r.z, r.w, g.shape);
  atomicMax(&rank2row[bm_rank(bits, block_prefix, c, bits[c >> 5])], j);
}

// All offsets of output row o by one thread (the bricks of its neighbourhood stay in the
// CU's L1 between the lines; the 16-byte index row is read once): for every (kz, ky) line the
// kx cells come from one word per brick touched.
// One kz plane of output row o: KY lines of KX cells (compile-time: 3 x 3, the case every
// config uses; bm_lookup_plane_any below for the rest).  A line's cells lie in at most two
// bricks and, inside a brick, in ONE word; the flag bytes of all lines are loaded first, then
// all words -- two memory latencies for the plane instead of two per line -- and only hits
// go on to the rank (prefix + sector) and rank2row: 70 -> 59.5 us at the stress size.
// (Loading the whole sector of every brick touched up front, so that the ranks come out of
// registers and the prefixes / rank2row entries of all hits load together: 67 us -- six
// 32-byte loads per thread cost more than the 1.5 hits' dependent chains.)
template <int KY, int KX>
__device__ __forceinline__ void bm_lookup_plane(const int4 r, int o, int n, const Geom& g,
                                                const uint32_t* __restrict__ bits,
                                                const uint8_t* __restrict__ used,
                                                const int* __restrict__ block_prefix,
                                                const int32_t* __restrict__ rank2row,
                                                int32_t* __restrict__ nbr, int kz) {
  const BmDims d = bm_dims(g.shape);
  const int z = r.y - g.pd[0] + kz, x0 = r.w - g.pd[2];
  const bool zok = z >= 0 && z < g.shape[0];
  const int bx0 = x0 >> 3, bx1 = (x0 + KX - 1) >> 3;      // (x0 = -1: brick -1, never valid)
  uint32_t blk[KY][2], lbit[KY];
  bool ok[KY][2];
#pragma unroll
  for (int ky = 0; ky < KY; ++ky) {
    const int y = r.z - g.pd[1] + ky;
    const bool line = zok && y >= 0 && y < g.shape[1];
    uint32_t lb = 0;
    const uint32_t lblk = line ? bm_line(r.x, z, y, d, &lb) : 0u;
    lbit[ky] = lb;
    blk[ky][0] = lblk + (uint32_t)bx0;
    blk[ky][1] = lblk + (uint32_t)bx1;
    ok[ky][0] = line && bx0 >= 0 && bx0 < d.tx;
    ok[ky][1] = line && bx1 != bx0 && bx1 < d.tx;
  }
  if (used) {
    uint8_t u[KY][2];
#pragma unroll
    for (int ky = 0; ky < KY; ++ky) {
      u[ky][0] = ok[ky][0] ? used[blk[ky][0]] : (uint8_t)0;
      u[ky][1] = ok[ky][1] ? used[blk[ky][1]] : (uint8_t)0;
    }
#pragma unroll
    for (int ky = 0; ky < KY; ++ky) {
      ok[ky][0] = u[ky][0] != 0;
      ok[ky][1] = u[ky][1] != 0;
    }
  }
  uint32_t wrd[KY][2];
#pragma unroll
  for (int ky = 0; ky < KY; ++ky) {
    wrd[ky][0] = ok[ky][0] ? bits[(blk[ky][0] << 3) | (lbit[ky] >> 5)] : 0u;
    wrd[ky][1] = ok[ky][1] ? bits[(blk[ky][1] << 3) | (lbit[ky] >> 5)] : 0u;
  }
  int32_t* out = nbr + (size_t)kz * KY * KX * n + o;
#pragma unroll
  for (int ky = 0; ky < KY; ++ky) {
#pragma unroll
    for (int kx = 0; kx < KX; ++kx) {
      const int x = x0 + kx;
      const bool right = (x >> 3) != bx0;
      const uint32_t w = right ? wrd[ky][1] : wrd[ky][0];
      const uint32_t bit = lbit[ky] | (uint32_t)(x & 7);
      int v = -1;
      if (x >= 0 && x < g.shape[2] && ((w >> (bit & 31)) & 1u)) {
        const uint32_t c = ((right ? blk[ky][1] : blk[ky][0]) << 8) | bit;
        v = rank2row[bm_rank(bits, block_prefix, c, w)];
      }
      *out = v;
      out += n;
    }
  }
}

// kz planes [kz0, kz1) of the row, any kernel size
__device__ __forceinline__ void bm_lookup_plane_any(const int4 r, int o, int n, const Geom& g,
                                                    const uint32_t* __restrict__ bits,
                                                    const uint8_t* __restrict__ used,
                                                    const int* __restrict__ block_prefix,
                                                    const int32_t* __restrict__ rank2row,
                                                    int32_t* __restrict__ nbr, int kz) {
  const BmDims d = bm_dims(g.shape);
  const int x0 = r.w - g.pd[2];
  int32_t* out = nbr + (size_t)kz * g.ks[1] * g.ks[2] * n + o;
  const int z = r.y - g.pd[0] + kz;
  for (int ky = 0; ky < g.ks[1]; ++ky) {
    const int y = r.z - g.pd[1] + ky;
    const bool line = z >= 0 && z < g.shape[0] && y >= 0 && y < g.shape[1];
    uint32_t lbit = 0;
    const uint32_t lblk = line ? bm_line(r.x, z, y, d, &lbit) : 0u;
    uint32_t have = 0xffffffffu, w = 0;
    for (int kx = 0; kx < g.ks[2]; ++kx) {
      const int x = x0 + kx;
      int v = -1;
      if (line && x >= 0 && x < g.shape[2]) {
        const uint32_t c = ((lblk + (uint32_t)(x >> 3)) << 8) | lbit | (uint32_t)(x & 7);
        if ((c >> 8) != have) {
          have = c >> 8;
          w = bm_block_used(used, have) ? bits[c >> 5] : 0u;
        }
        if ((w >> (c & 31)) & 1u) v = rank2row[bm_rank(bits, block_prefix, c, w)];
      }
      *out = v;
      out += n;
    }
  }
}

__device__ __forceinline__ void bm_lookup_row(const int4 r, int o, int n, const Geom& g,
                                              const uint32_t* __restrict__ bits,
                                              const uint8_t* __restrict__ used,
                                              const int* __restrict__ block_prefix,
                                              const int32_t* __restrict__ rank2row,
                                              int32_t* __restrict__ nbr, int kz) {
  if (g.ks[1] == 3 && g.ks[2] == 3)
    bm_lookup_plane<3, 3>(r, o, n, g, bits, used, block_prefix, rank2row, nbr, kz);
  else
    bm_lookup_plane_any(r, o, n, g, bits, used, block_prefix, rank2row, nbr, kz);
}

__global__ __launch_bounds__(256) void subm_bm_lookup(const int32_t* __restrict__ idx, int n, Geom g,
                                                      const uint32_t* __restrict__ bits,
                                                      const uint8_t* __restrict__ coarse,
                                                      const int* __restrict__ block_prefix,
                                                      const int32_t* __restrict__ rank2row,
                                                      int32_t* __restrict__ nbr) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= n) return;
  // blockIdx.y = the kz plane
  bm_lookup_row(((const int4*)idx)[o], o, n, g, bits, coarse, block_prefix, rank2row, nbr,
                blockIdx.y);
}

// ------------------------------------------------------------- strided ----
// Output position reached from input coordinate c through offset component kc
// (dilation 1): val = (c + pad - kc) / stride when divisible and in range.
__device__ __forceinline__ bool out_coord(int c, int kc, int pad, int stride, int lim, int* val) {
  int t = c + pad - kc;
  if (t < 0) return false;
  int q = t / stride;
  if (q * stride != t || q >= lim) return false;
  *val = q;
  return true;
}

__global__ __launch_bounds__(256) void conv_mark(const int32_t* __restrict__ idx, int n, Geom g,
                                                 uint32_t* bits) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int k = blockIdx.y;
  const int kx = k % g.ks[2], ky = (k / g.ks[2]) % g.ks[1], kz = k / (g.ks[2] * g.ks[1]);
  int4 r = ((const int4*)idx)[i];
  int z, y, x;
  if (out_coord(r.y, kz, g.pd[0], g.st[0], g.shape[0], &z) &&
      out_coord(r.z, ky, g.pd[1], g.st[1], g.shape[1], &y) &&
      out_coord(r.w, kx, g.pd[2], g.st[2], g.shape[2], &x))
    bitmap_set(bits, cell_id(r.x, z, y, x, g.shape));
}

// The same marking pass with the INPUT set given as the occupancy bitmap of its grid (the
// previous strided conv's output bitmap) instead of a row list: what lets a chain of strided
// convs count all its output sets back to back, without the host reading one count before
// the next level can start (msmd_rulebook_conv3d_count_chain).
__global__ __launch_bounds__(256) void conv_mark_from_bits(const uint32_t* __restrict__ in_bits,
                                                           long in_words, int in_d, int in_h,
                                                           int in_w, Geom g, uint32_t* bits) {
  const long wi = (long)blockIdx.x * 256 + threadIdx.x;
  if (wi >= in_words) return;
  uint32_t word = in_bits[wi];
  while (word) {
    const int bit = __ffs(word) - 1;
    word &= word - 1;
    uint32_t c = (uint32_t)(wi * 32 + bit);
    const int x = c % in_w;
    c /= in_w;
    const int y = c % in_h;
    c /= in_h;
    const int z = c % in_d, b = c / in_d;
    for (int kz = 0; kz < g.ks[0]; ++kz) {
      int oz;
      if (!out_coord(z, kz, g.pd[0], g.st[0], g.shape[0], &oz)) continue;
      for (int ky = 0; ky < g.ks[1]; ++ky) {
        int oy;
        if (!out_coord(y, ky, g.pd[1], g.st[1], g.shape[1], &oy)) continue;
        for (int kx = 0; kx < g.ks[2]; ++kx) {
          int ox;
          if (out_coord(x, kx, g.pd[2], g.st[2], g.shape[2], &ox))
            bitmap_set(bits, cell_id(b, oz, oy, ox, g.shape));
        }
      }
    }
  }
}

// conv_mark_from_bits the other way round, for output grids of moderate size: one thread
// per OUTPUT cell tests the (at most kvol) input cells that reach it -- in = out * stride -
// pad + k -- and a wave's 64 answers become two words of the output bitmap by ballot: no
// atomics, no serial walk over a word's set bits (LiDAR ground lines are runs of 32 set bits:
// a thread of the scatter form did up to 32 x 27 dependent iterations while its wave waited;
// 106 us per call on the LC grids, 6 calls per step = the second largest item on the index
// queue).  Work ~ output cells x kvol bit tests, whatever the occupancy: the host takes this
// form up to kConvGatherCells output cells and the scatter form above that.
constexpr long kConvGatherCells = 16L << 20;
__global__ __launch_bounds__(256) void conv_mark_gather(const uint32_t* __restrict__ in_bits,
                                                        int in_d, int in_h, int in_w, Geom g,
                                                        long out_cells,
                                                        uint32_t* __restrict__ out_bits) {
  const long c = (long)blockIdx.x * 256 + threadIdx.x;
  bool hit = false;
  if (c < out_cells) {
    long r = c;
    const int x = (int)(r % g.shape[2]);
    r /= g.shape[2];
    const int y = (int)(r % g.shape[1]);
    r /= g.shape[1];
    const int z = (int)(r % g.shape[0]);
    const int b = (int)(r / g.shape[0]);
    const int x0 = x * g.st[2] - g.pd[2];
    for (int kz = 0; kz < g.ks[0] && !hit; ++kz) {
      const int iz = z * g.st[0] - g.pd[0] + kz;
      if (iz < 0 || iz >= in_d) continue;
      for (int ky = 0; ky < g.ks[1] && !hit; ++ky) {
        const int iy = y * g.st[1] - g.pd[1] + ky;
        if (iy < 0 || iy >= in_h) continue;
        const uint32_t base = (((uint32_t)b * in_d + iz) * in_h + iy) * in_w;
        for (int kx = 0; kx < g.ks[2]; ++kx) {
          const int ix = x0 + kx;
          if (ix < 0 || ix >= in_w) continue;
          const uint32_t cell = base + (uint32_t)ix;
          hit = hit || ((in_bits[cell >> 5] >> (cell & 31)) & 1u);
        }
      }
    }
  }
  const unsigned long long m = __ballot(hit);
  const int lane = threadIdx.x & 63;
  const long w0 = (c - lane) >> 5;           // the wave's first cell is a multiple of 64
  const long words = (out_cells + 31) >> 5;
  if (lane == 0 && w0 < words) out_bits[w0] = (uint32_t)m;
  if (lane == 32 && w0 + 1 < words) out_bits[w0 + 1] = (uint32_t)(m >> 32);
}

__global__ __launch_bounds__(256) void conv_fill(const int32_t* __restrict__ idx, int n, Geom g,
                                                 const uint32_t* __restrict__ bits,
                                                 const int* __restrict__ prefix, int n_out,
                                                 int32_t* __restrict__ out_idx,
                                                 int32_t* __restrict__ nbr_fwd,
                                                 int32_t* __restrict__ nbr_bwd) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int k = blockIdx.y;
  const int kx = k % g.ks[2], ky = (k / g.ks[2]) % g.ks[1], kz = k / (g.ks[2] * g.ks[1]);
  int4 r = ((const int4*)idx)[i];
  int z, y, x, o = -1;
  if (out_coord(r.y, kz, g.pd[0], g.st[0], g.shape[0], &z) &&
      out_coord(r.z, ky, g.pd[1], g.st[1], g.shape[1], &y) &&
      out_coord(r.w, kx, g.pd[2], g.st[2], g.shape[2], &x)) {
    o = bitmap_rank(bits, prefix, cell_id(r.x, z, y, x, g.shape));
    if (o < n_out) {
      // (k,o) has exactly one source COORDINATE; when the input set repeats a coordinate
      // (reference_quirks: a false "mixed" voxel on an only-3D voxel's cell) its LAST row
      // feeds the output, as in every SubM look-up -- a max instead of a plain store, so the
      // table does not depend on which of the two threads writes last (result unused: a
      // fire-and-forget atomic to an address no other thread of the launch touches otherwise)
      atomicMax(&nbr_fwd[(size_t)k * n_out + o], i);
      ((int4*)out_idx)[o] = make_int4(r.x, z, y, x);  // same value from every writer
    } else {
      o = -1;
    }
  }
  if (nbr_bwd) nbr_bwd[(size_t)k * n + i] = o;
}

// ------------------------------------------------- nbr table -> pair list --
struct NbrValid {
  const int32_t* nbr;  // one offset's row
  __device__ int operator()(int i) const { return nbr[i] >= 0; }
};
// Compaction of all K offsets in one launch set: element e = k*n_rows + o.
// Count/Emit see flat element ids; tiles never straddle offsets because each
// offset is padded to whole tiles (rows_pad).
struct PairCount {
  const int32_t* nbr;
  int n_rows, rows_pad;
  __device__ int operator()(int e) const {
    int k = e / rows_pad, o = e - k * rows_pad;
    return o < n_rows && nbr[(size_t)k * n_rows + o] >= 0;
  }
};
// Compaction of one offset's neighbours into the reference's pair lists, with the tail of
// both rows filled with -1 (the reference allocates indicePairs as full(-1),
// spconv_ops.h:55-59) by the table's EMPTY entries: entry o of offset k without a
// neighbour, the r-th such, writes position num_k + r.  No 0xFF fill of the whole
// [K,2,ld] tensor in front (19 MB per table at 90k rows, 12 tables per LC step), no scan of
// the tile sums, no separate count kernel: a block adds up the tile sums it needs.
__global__ __launch_bounds__(kScanBlock) void pairs_apply_kernel(
    const int32_t* __restrict__ nbr, int n_rows, int rows_pad, int ld, int kvol,
    const int* __restrict__ tile_sums, int32_t* __restrict__ pairs, int32_t* __restrict__ num) {
  __shared__ int smem[kScanSmem];
  const int tpk = rows_pad / kScanTile;
  const int k = blockIdx.x / tpk, tile_in_k = blockIdx.x - k * tpk;
  // pairs of this offset in front of this tile, and in the whole offset
  int carry = block_range_sum<kScanBlock>(tile_sums, k * tpk, (int)blockIdx.x, smem);
  const int num_k = carry + block_range_sum<kScanBlock>(tile_sums, (int)blockIdx.x, (k + 1) * tpk, smem);
  if (tile_in_k == 0 && threadIdx.x == 0) num[k] = num_k;
  int32_t* pin = pairs + ((size_t)k * 2 + 0) * ld;
  int32_t* pout = pin + ld;
  const int base = tile_in_k * kScanTile;
  int src[kScanItems], v[kScanItems], ex[kScanItems];
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    const int o = base + j * kScanBlock + threadIdx.x;
    src[j] = o < n_rows ? nbr[(size_t)k * n_rows + o] : -1;
    v[j] = src[j] >= 0;
  }
  tile_excl_scan(v, ex, smem);
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    const int o = base + j * kScanBlock + threadIdx.x;
    const int pos = carry + ex[j];         // pairs of offset k in front of entry o
    if (v[j]) {
      if (pos < ld) {
        pin[pos] = src[j];
        pout[pos] = o;
      }
    } else {
      const int tail = num_k + (o - pos);  // o - pos = empty entries in front of o
      if (tail < ld) {
        pin[tail] = -1;
        pout[tail] = -1;
      }
    }
  }
}

int check_geom(const int* shape, const int* ks, const int* st, const int* pd, int batch,
               Geom* g) {
  if (!shape || !ks || batch < 1) return MSMD_ERR_INVALID_ARG;
  double cells = batch;
  g->kvol = 1;
  for (int i = 0; i < 3; ++i) {
    g->shape[i] = shape[i];
    g->ks[i] = ks[i];
    g->st[i] = st ? st[i] : 1;
    g->pd[i] = pd ? pd[i] : ks[i] / 2;  // SubM: spconv_ops.h:76-79
    if (shape[i] < 1 || ks[i] < 1 || g->st[i] < 1 || g->pd[i] < 0) return MSMD_ERR_INVALID_ARG;
    cells *= shape[i];
    g->kvol *= ks[i];
  }
  if (g->kvol > 4096) return MSMD_ERR_UNSUPPORTED;  // spconv_ops.h:51
  if (cells >= 4294967295.0) return MSMD_ERR_RANGE;
  return MSMD_OK;
}

}  // namespace
}  // namespace msmd

using namespace msmd;

// ------------------------------------------------------------------ SubM ---
MSMD_EXPORT size_t msmd_rulebook_subm_workspace_bytes(int n) {
  int bits = next_pow2_bits(2L * (n > 0 ? n : 1));
  if (bits < 6) bits = 6;
  return align_up(sizeof(unsigned long long) << bits);
}

MSMD_EXPORT int msmd_rulebook_subm3d(const int32_t* indices, int n, int batch_size,
                                     const int* spatial_shape, const int* ksize, int32_t* nbr,
                                     void* workspace, size_t workspace_bytes,
                                     msmd_stream_t stream) {
  Geom g;
  int rc = check_geom(spatial_shape, ksize, nullptr, nullptr, batch_size, &g);
  if (rc) return rc;
  if (n < 0 || (n > 0 && (!indices || !nbr))) return MSMD_ERR_INVALID_ARG;
  if (n == 0) return MSMD_OK;
  int bits = next_pow2_bits(2L * n);
  if (bits < 6) bits = 6;
  if (workspace_bytes < (sizeof(unsigned long long) << bits) || ((uintptr_t)workspace & 255))
    return MSMD_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  auto* table = (unsigned long long*)workspace;
  hipMemsetAsync(table, 0xFF, sizeof(unsigned long long) << bits, st);
  const int nb = ceil_div(n, 256);
  MSMD_LAUNCH(subm_insert, dim3(nb), dim3(256), 0, st, indices, n, g, table, bits);
  MSMD_LAUNCH(subm_lookup, dim3(nb, g.kvol), dim3(256), 0, st, indices, n, g, table, bits,
                     nbr);
  return launch_status();
}

namespace {
struct SubmBmWs {
  uint32_t* bits;
  uint8_t* coarse;              // one byte per block
  size_t coarse_bytes;
  int* block_prefix;
  int* tiles;
  int* total;
  int32_t* rank2row;
  size_t nwords, nblocks;
};
template <typename A>
void carve_subm_bm(A& a, SubmBmWs* w, int n, int batch, const int* shape) {
  const size_t nblocks = bm_blocks(batch, shape);
  const size_t nwords = nblocks * kBmBlockWords;
  uint32_t* bits = a.template take<uint32_t>(nwords);
  const size_t cw = nblocks;
  uint8_t* coarse = a.template take<uint8_t>(cw);
  int* prefix = a.template take<int>(nblocks);
  int* tiles = a.template take<int>(scan_num_tiles((long)nblocks) + 1);
  int* total = a.template take<int>(1);
  int32_t* rows = a.template take<int32_t>(n > 0 ? n : 1);
  if (w) *w = SubmBmWs{bits, coarse, cw, prefix, tiles, total, rows, nwords, nblocks};
}
}  // namespace

MSMD_EXPORT size_t msmd_rulebook_subm_bitmap_workspace_bytes(int n, int batch_size,
                                                             const int* spatial_shape) {
  if (n < 0 || batch_size < 1 || !spatial_shape) return 0;
  ArenaSize a;
  carve_subm_bm(a, (SubmBmWs*)nullptr, n, batch_size, spatial_shape);
  return a.off;
}

MSMD_EXPORT int msmd_rulebook_subm3d_bitmap(const int32_t* indices, int n, int batch_size,
                                            const int* spatial_shape, const int* ksize,
                                            int32_t* nbr, void* workspace,
                                            size_t workspace_bytes, msmd_stream_t stream) {
  Geom g;
  int rc = check_geom(spatial_shape, ksize, nullptr, nullptr, batch_size, &g);
  if (rc) return rc;
  if (n < 0 || (n > 0 && (!indices || !nbr))) return MSMD_ERR_INVALID_ARG;
  if (n == 0) return MSMD_OK;
  if (bm_blocks(batch_size, spatial_shape) >= (1u << 24)) return MSMD_ERR_RANGE;  // brick << 8 | bit
  Arena a(workspace, workspace_bytes);
  SubmBmWs w;
  carve_subm_bm(a, &w, n, batch_size, spatial_shape);
  if (!a.ok()) return MSMD_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const bool flags = bm_use_flags(w.nblocks, n);
  const uint8_t* used = flags ? w.coarse : nullptr;
  if (flags)
    hipMemsetAsync(w.coarse, 0, w.coarse_bytes, st);
  else
    hipMemsetAsync(w.bits, 0, sizeof(uint32_t) * w.nwords, st);
  hipMemsetAsync(w.rank2row, 0xFF, sizeof(int32_t) * (size_t)n, st);
  const int nb = ceil_div(n, 256);
  if (flags)
    MSMD_LAUNCH(subm_bm_mark_blocks, dim3(nb), dim3(256), 0, st, indices, n, g, w.coarse, w.bits);
  MSMD_LAUNCH(subm_bm_mark, dim3(nb), dim3(256), 0, st, indices, n, g, w.bits);
  device_scan(BmBlockCount{w.bits, used}, StoreUsedPrefix{w.block_prefix}, (int)w.nblocks,
              w.tiles, w.total, -1, st);
  MSMD_LAUNCH(subm_bm_rows, dim3(nb), dim3(256), 0, st, indices, n, g, (const uint32_t*)w.bits,
              (const int*)w.block_prefix, w.rank2row);
  // a thread per (row, kz plane): 152 against 163 us at the stress size, 46 against 55 at the
  // nominal one, with one thread per row (27 dependent look-ups in a row)
  MSMD_LAUNCH(subm_bm_lookup, dim3(nb, g.ks[0]), dim3(256), 0, st, indices, n, g,
              (const uint32_t*)w.bits, used, (const int*)w.block_prefix,
              (const int32_t*)w.rank2row, nbr);
  return launch_status();
}

// ------------------------------------------------- SubM, many tables --------
// The SubM tables of an index pass in one launch set (msmd_rulebook_subm3d_many).  Like the
// plans (plan_many.hip) nothing in the index chain reads a SubM table -- the feature pass and
// the planning do -- so the ~12 tables of an LC step are built together at the end of
// prepare(): 2 fills + 6 kernels instead of 3 (hash index) or 10 (bitmap index) launches
// each.  Per table the same index structure and the same kernels' arithmetic as the single
// calls; results identical.
namespace {

constexpr int kSubmMax = 16;      // tables per launch set

struct SubmJob {
  const int32_t* idx;
  int32_t* nbr;
  unsigned long long* table;    // hash index (or null)
  uint32_t* bits;               // bitmap index (or null)
  uint8_t* coarse;              // one byte per 256-cell block of `bits`
  int* block_prefix;
  int32_t* rank2row;
  Geom g;
  int n, hbits, nblocks, pad;
};
struct SubmTab {
  int n;
  int blk0[kSubmMax + 1];       // first 256-row block of table s
  int sblk0[kSubmMax + 1];      // first scan tile of table s's bitmap blocks
  SubmJob j[kSubmMax];
};

__device__ __forceinline__ int subm_table_of(const int* __restrict__ first, int n, int b) {
  int s = 0;
  while (s + 1 < n && b >= first[s + 1]) ++s;
  return s;
}

__global__ __launch_bounds__(256) void subm_insert_many(const SubmTab tab) {
  const int s = subm_table_of(tab.blk0, tab.n, blockIdx.x);
  const SubmJob& J = tab.j[s];
  const int j = (blockIdx.x - tab.blk0[s]) * 256 + threadIdx.x;
  if (j >= J.n) return;
  const int4 r = ((const int4*)J.idx)[j];
  if (J.table) {
    hash_insert<true>(J.table, J.hbits, cell_id(r.x, r.y, r.z, r.w, J.g.shape), (uint32_t)j);
    return;
  }
  const uint32_t c = bm_cell(r.x, r.y, r.z, r.w, J.g.shape);
  if (!J.coarse) {                // bitmap cleared as a whole: set the bit now
    bitmap_set(J.bits, c);
    return;
  }
  const uint32_t blk = c >> 8;    // (subm_bm_mark_blocks)
  J.coarse[blk] = 1;
  uint4* line = (uint4*)(J.bits + (size_t)blk * kBmBlockWords);
  line[0] = make_uint4(0, 0, 0, 0);
  line[1] = make_uint4(0, 0, 0, 0);
}

__global__ __launch_bounds__(256) void subm_bm_mark_many(const SubmTab tab) {
  const int s = subm_table_of(tab.blk0, tab.n, blockIdx.x);
  const SubmJob& J = tab.j[s];
  if (!J.coarse) return;          // (hash index, or bits set by the first pass)
  const int j = (blockIdx.x - tab.blk0[s]) * 256 + threadIdx.x;
  if (j >= J.n) return;
  const int4 r = ((const int4*)J.idx)[j];
  bitmap_set(J.bits, bm_cell(r.x, r.y, r.z, r.w, J.g.shape));
}

__global__ __launch_bounds__(kScanBlock) void subm_bm_sums_many(const SubmTab tab,
                                                               int* __restrict__ tile_sums) {
  __shared__ int smem[kScanBlock / 64];
  const int s = subm_table_of(tab.sblk0, tab.n, blockIdx.x);
  const SubmJob& J = tab.j[s];
  const BmBlockCount count{J.bits, J.coarse};
  const int base = (blockIdx.x - tab.sblk0[s]) * kScanTile;
  int c = 0;
#pragma unroll
  for (int q = 0; q < kScanItems; ++q) {
    const int i = base + q * kScanBlock + threadIdx.x;
    if (i < J.nblocks) c += count(i);
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

__global__ __launch_bounds__(kScanBlock) void subm_bm_prefix_many(
    const SubmTab tab, const int* __restrict__ tile_sums) {
  __shared__ int smem[kScanSmem];
  const int s = subm_table_of(tab.sblk0, tab.n, blockIdx.x);
  const SubmJob& J = tab.j[s];
  const BmBlockCount count{J.bits, J.coarse};
  const int base = (blockIdx.x - tab.sblk0[s]) * kScanTile;
  int v[kScanItems], ex[kScanItems];
#pragma unroll
  for (int q = 0; q < kScanItems; ++q) {
    const int i = base + q * kScanBlock + threadIdx.x;
    v[q] = i < J.nblocks ? count(i) : 0;
  }
  const int carry = block_range_sum<kScanBlock>(tile_sums, tab.sblk0[s], (int)blockIdx.x, smem);
  tile_excl_scan(v, ex, smem);
#pragma unroll
  for (int q = 0; q < kScanItems; ++q) {
    const int i = base + q * kScanBlock + threadIdx.x;
    if (v[q]) J.block_prefix[i] = carry + ex[q];       // (occupied blocks only)
  }
}

__global__ __launch_bounds__(256) void subm_bm_rows_many(const SubmTab tab) {
  const int s = subm_table_of(tab.blk0, tab.n, blockIdx.x);
  const SubmJob& J = tab.j[s];
  if (!J.bits) return;
  const int j = (blockIdx.x - tab.blk0[s]) * 256 + threadIdx.x;
  if (j >= J.n) return;
  const int4 r = ((const int4*)J.idx)[j];
  const uint32_t c = bm_cell(r.x, r.y, r.z, r.w, J.g.shape);
  atomicMax(&J.rank2row[bm_rank(J.bits, J.block_prefix, c, J.bits[c >> 5])], j);
}

// blockIdx.y: the offset k (hash index) or the kz plane (bitmap index)
__global__ __launch_bounds__(256) void subm_lookup_many(const SubmTab tab) {
  const int s = subm_table_of(tab.blk0, tab.n, blockIdx.x);
  const SubmJob& J = tab.j[s];
  const Geom& g = J.g;
  const int o = (blockIdx.x - tab.blk0[s]) * 256 + threadIdx.x;
  const int n = J.n;
  if (o >= n || (!J.table && (int)blockIdx.y >= g.ks[0])) return;   // bitmap index: y = kz plane
  const int4 r = ((const int4*)J.idx)[o];
  if (J.table) {
    const int k = blockIdx.y;
    if (k >= g.kvol) return;
    const int kx = k % g.ks[2], ky = (k / g.ks[2]) % g.ks[1], kz = k / (g.ks[2] * g.ks[1]);
    const int z = r.y - g.pd[0] + kz, y = r.z - g.pd[1] + ky, x = r.w - g.pd[2] + kx;
    int v = -1;
    if (z >= 0 && z < g.shape[0] && y >= 0 && y < g.shape[1] && x >= 0 && x < g.shape[2])
      v = hash_find(J.table, J.hbits, cell_id(r.x, z, y, x, g.shape));
    J.nbr[(size_t)k * n + o] = v;
    return;
  }
  bm_lookup_row(r, o, n, g, J.bits, J.coarse, J.block_prefix, J.rank2row, J.nbr, blockIdx.y);
}

struct SubmManyWs {
  char *ones, *zeros;                 // the 0xFF-filled and the zero-filled region
  size_t ones_bytes, zeros_bytes;
  int* tile_sums;
};

// one pass for the sizes, one for the pointers (jobs == nullptr: sizes only)
template <typename A>
int carve_subm_many(A& a, const msmd_subm_desc* d, int n_desc, SubmJob* jobs, SubmManyWs* w) {
  // 0xFF region: hash tables, rank -> row arrays
  char* ones = (char*)a.template take<char>(0);
  const size_t o0 = a.off;
  for (int i = 0; i < n_desc; ++i) {
    if (d[i].n <= 0) continue;
    if (d[i].method == 0) {
      int bits = next_pow2_bits(2L * d[i].n);
      if (bits < 6) bits = 6;
      auto* t = a.template take<unsigned long long>((size_t)1 << bits);
      if (jobs) jobs[i].table = t, jobs[i].hbits = bits;
    } else {
      auto* r = a.template take<int32_t>(d[i].n);
      if (jobs) jobs[i].rank2row = r;
    }
  }
  const size_t o1 = a.off;
  // zero region: the block flags of the sets that use them, the whole bitmap of the others
  char* zeros = (char*)a.template take<char>(0);
  long scan_tiles = 0;
  for (int i = 0; i < n_desc; ++i) {
    if (d[i].n <= 0 || d[i].method == 0) continue;
    const size_t nblocks = bm_blocks(d[i].batch_size, d[i].spatial_shape);
    if (nblocks >= (1u << 24)) return MSMD_ERR_RANGE;
    if (bm_use_flags(nblocks, d[i].n)) {
      auto* c = a.template take<uint8_t>(nblocks);
      if (jobs) jobs[i].coarse = c;
    } else {
      auto* b = a.template take<uint32_t>(nblocks * kBmBlockWords);
      if (jobs) jobs[i].bits = b;
    }
    if (jobs) jobs[i].nblocks = (int)nblocks;
    scan_tiles += scan_num_tiles((long)nblocks);
  }
  const size_t o2 = a.off;
  for (int i = 0; i < n_desc; ++i) {
    if (d[i].n <= 0 || d[i].method == 0) continue;
    const size_t nblocks = bm_blocks(d[i].batch_size, d[i].spatial_shape);
    if (bm_use_flags(nblocks, d[i].n)) {
      auto* b = a.template take<uint32_t>(nblocks * kBmBlockWords);
      if (jobs) jobs[i].bits = b;
    }
    auto* p = a.template take<int>(nblocks);
    if (jobs) jobs[i].block_prefix = p;
  }
  int* sums = a.template take<int>(scan_tiles + 1);
  if (w) *w = SubmManyWs{ones, zeros, o1 - o0, o2 - o1, sums};
  return MSMD_OK;
}

}  // namespace

MSMD_EXPORT size_t msmd_rulebook_subm3d_many_workspace_bytes(const msmd_subm_desc* descs,
                                                             int n_desc) {
  if (!descs || n_desc < 0) return 0;
  size_t most = 0;
  for (int g0 = 0; g0 < n_desc; g0 += kSubmMax) {
    const int cnt = n_desc - g0 < kSubmMax ? n_desc - g0 : kSubmMax;
    ArenaSize a;
    if (carve_subm_many(a, descs + g0, cnt, (SubmJob*)nullptr, (SubmManyWs*)nullptr) != MSMD_OK)
      return 0;
    most = a.off > most ? a.off : most;
  }
  return most;
}

MSMD_EXPORT int msmd_rulebook_subm3d_many(const msmd_subm_desc* descs, int n_desc, void* workspace,
                                          size_t workspace_bytes, msmd_stream_t stream) {
  if (n_desc < 0 || (n_desc > 0 && !descs)) return MSMD_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  for (int g0 = 0; g0 < n_desc; g0 += kSubmMax) {
    const int cnt = n_desc - g0 < kSubmMax ? n_desc - g0 : kSubmMax;
    const msmd_subm_desc* d = descs + g0;
    SubmTab tab;
    SubmJob jobs[kSubmMax];
    for (int i = 0; i < cnt; ++i) {
      jobs[i] = SubmJob{};
      if (d[i].method != 0 && d[i].method != 1) return MSMD_ERR_INVALID_ARG;
      const int rc = check_geom(d[i].spatial_shape, d[i].ksize, nullptr, nullptr, d[i].batch_size,
                                &jobs[i].g);
      if (rc) return rc;
      if (d[i].n < 0 || (d[i].n > 0 && (!d[i].indices || !d[i].nbr))) return MSMD_ERR_INVALID_ARG;
      jobs[i].idx = d[i].indices;
      jobs[i].nbr = d[i].nbr;
      jobs[i].n = d[i].n;
    }
    Arena a(workspace, workspace_bytes);
    SubmManyWs w;
    const int rc = carve_subm_many(a, d, cnt, jobs, &w);
    if (rc) return rc;
    if (!a.ok()) return MSMD_ERR_WORKSPACE;
    tab.n = 0;
    tab.blk0[0] = tab.sblk0[0] = 0;
    int max_y = 1;
    bool any_bitmap = false, any_flags = false;
    for (int i = 0; i < cnt; ++i) {
      if (jobs[i].n <= 0) continue;
      const int s = tab.n++;
      tab.j[s] = jobs[i];
      tab.blk0[s + 1] = tab.blk0[s] + ceil_div(jobs[i].n, 256);
      tab.sblk0[s + 1] = tab.sblk0[s] + (jobs[i].bits ? scan_num_tiles((long)jobs[i].nblocks) : 0);
      const Geom& g = jobs[i].g;
      const int y = jobs[i].bits ? g.ks[0] : g.kvol;
      max_y = y > max_y ? y : max_y;
      any_bitmap |= jobs[i].bits != nullptr;
      any_flags |= jobs[i].coarse != nullptr;
    }
    if (tab.n == 0) continue;
    for (int s = tab.n + 1; s <= kSubmMax; ++s) tab.blk0[s] = tab.sblk0[s] = 0x7fffffff;
    if (w.ones_bytes) hipMemsetAsync(w.ones, 0xFF, w.ones_bytes, st);
    if (w.zeros_bytes) hipMemsetAsync(w.zeros, 0, w.zeros_bytes, st);
    const int nblk = tab.blk0[tab.n];
    MSMD_LAUNCH(subm_insert_many, dim3(nblk), dim3(256), 0, st, tab);
    if (any_bitmap) {
      if (any_flags) MSMD_LAUNCH(subm_bm_mark_many, dim3(nblk), dim3(256), 0, st, tab);
      const int nt = tab.sblk0[tab.n];
      MSMD_LAUNCH(subm_bm_sums_many, dim3(nt), dim3(kScanBlock), 0, st, tab, w.tile_sums);
      MSMD_LAUNCH(subm_bm_prefix_many, dim3(nt), dim3(kScanBlock), 0, st, tab,
                  (const int*)w.tile_sums);
      MSMD_LAUNCH(subm_bm_rows_many, dim3(nblk), dim3(256), 0, st, tab);
    }
    MSMD_LAUNCH(subm_lookup_many, dim3(nblk, max_y), dim3(256), 0, st, tab);
  }
  return launch_status();
}

// --------------------------------------------------------------- strided ---
namespace {
struct ConvWs {
  uint32_t* bits;
  int* prefix;
  int* tiles;
  size_t words;
};
template <typename A>
void carve_conv(A& a, ConvWs* w, int batch, const int* out_shape) {
  size_t cells = (size_t)batch * out_shape[0] * out_shape[1] * out_shape[2];
  size_t words = (cells + 31) / 32;
  uint32_t* b = a.template take<uint32_t>(words);
  int* p = a.template take<int>(words);
  int* t = a.template take<int>(scan_num_tiles((long)words) + 1);
  if (w) *w = ConvWs{b, p, t, words};
}
}  // namespace

MSMD_EXPORT size_t msmd_rulebook_conv_workspace_bytes(int batch_size, const int* out_shape) {
  ArenaSize a;
  carve_conv(a, (ConvWs*)nullptr, batch_size, out_shape);
  return a.off;
}

MSMD_EXPORT int msmd_rulebook_conv3d_count(const int32_t* indices, int n, int batch_size,
                                           const int* out_shape, const int* ksize,
                                           const int* stride, const int* padding,
                                           int32_t* n_out, void* workspace,
                                           size_t workspace_bytes, msmd_stream_t stream) {
  Geom g;
  if (!stride || !padding || !n_out) return MSMD_ERR_INVALID_ARG;
  int rc = check_geom(out_shape, ksize, stride, padding, batch_size, &g);
  if (rc) return rc;
  if (n < 0 || (n > 0 && !indices)) return MSMD_ERR_INVALID_ARG;
  Arena a(workspace, workspace_bytes);
  ConvWs w;
  carve_conv(a, &w, batch_size, out_shape);
  if (!a.ok()) return MSMD_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  hipMemsetAsync(w.bits, 0, sizeof(uint32_t) * w.words, st);
  if (n > 0)
    MSMD_LAUNCH(conv_mark, dim3(ceil_div(n, 256), g.kvol), dim3(256), 0, st, indices, n, g,
                       w.bits);
  device_scan(PopcCount{w.bits}, StorePrefix{w.prefix}, (int)w.words, w.tiles, n_out, -1, st);
  return launch_status();
}

// A chain of strided convs (each one's input set = the previous one's output set, as in
// SparseEncoder where only set-preserving SubM convs sit between them): the output sets of
// ALL levels are marked and counted back to back -- level l+1 from level l's bitmap -- and
// the host reads the `levels` counts once instead of once per level.  The workspace is the
// concatenation of the per-level msmd_rulebook_conv_workspace_bytes blocks (256-byte
// aligned), so msmd_rulebook_conv3d_fill runs per level on its own block afterwards.
MSMD_EXPORT size_t msmd_rulebook_conv_chain_workspace_bytes(int batch_size, int levels,
                                                            const int* out_shapes) {
  if (levels < 1 || !out_shapes) return 0;
  size_t total = 0;
  for (int l = 0; l < levels; ++l)
    total += align_up(msmd_rulebook_conv_workspace_bytes(batch_size, out_shapes + 3 * l));
  return total;
}

MSMD_EXPORT int msmd_rulebook_conv3d_count_chain(const int32_t* indices, int n, int batch_size,
                                                 int levels, const int* out_shapes,
                                                 const int* ksizes, const int* strides,
                                                 const int* paddings, int32_t* n_out,
                                                 void* workspace, size_t workspace_bytes,
                                                 msmd_stream_t stream) {
  if (levels < 1 || !out_shapes || !ksizes || !strides || !paddings || !n_out)
    return MSMD_ERR_INVALID_ARG;
  if (n < 0 || (n > 0 && !indices)) return MSMD_ERR_INVALID_ARG;
  if (workspace_bytes < msmd_rulebook_conv_chain_workspace_bytes(batch_size, levels, out_shapes) ||
      ((uintptr_t)workspace & 255))
    return MSMD_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  char* base = (char*)workspace;
  ConvWs prev{};
  const int* prev_shape = nullptr;
  for (int l = 0; l < levels; ++l) {
    Geom g;
    int rc = check_geom(out_shapes + 3 * l, ksizes + 3 * l, strides + 3 * l, paddings + 3 * l,
                        batch_size, &g);
    if (rc) return rc;
    const size_t bytes = align_up(msmd_rulebook_conv_workspace_bytes(batch_size, out_shapes + 3 * l));
    Arena a(base, bytes);
    ConvWs w;
    carve_conv(a, &w, batch_size, out_shapes + 3 * l);
    if (!a.ok()) return MSMD_ERR_WORKSPACE;
    hipMemsetAsync(w.bits, 0, sizeof(uint32_t) * w.words, st);
    if (l == 0) {
      if (n > 0)
        MSMD_LAUNCH(conv_mark, dim3(ceil_div(n, 256), g.kvol), dim3(256), 0, st, indices, n, g,
                    w.bits);
    } else {
      const long oc = (long)batch_size * g.shape[0] * g.shape[1] * g.shape[2];
      if (oc <= kConvGatherCells)
        MSMD_LAUNCH(conv_mark_gather, dim3(ceil_div(oc, 256)), dim3(256), 0, st,
                    (const uint32_t*)prev.bits, prev_shape[0], prev_shape[1], prev_shape[2], g, oc,
                    w.bits);
      else
        MSMD_LAUNCH(conv_mark_from_bits, dim3(ceil_div((long)prev.words, 256)), dim3(256), 0, st,
                    (const uint32_t*)prev.bits, (long)prev.words, prev_shape[0], prev_shape[1],
                    prev_shape[2], g, w.bits);
    }
    device_scan(PopcCount{w.bits}, StorePrefix{w.prefix}, (int)w.words, w.tiles, n_out + l, -1, st);
    prev = w;
    prev_shape = out_shapes + 3 * l;
    base += bytes;
  }
  return launch_status();
}

MSMD_EXPORT int msmd_rulebook_conv3d_fill(const int32_t* indices, int n, int batch_size,
                                          const int* out_shape, const int* ksize,
                                          const int* stride, const int* padding, int n_out,
                                          int32_t* out_indices, int32_t* nbr_fwd,
                                          int32_t* nbr_bwd, void* workspace,
                                          size_t workspace_bytes, msmd_stream_t stream) {
  Geom g;
  if (!stride || !padding) return MSMD_ERR_INVALID_ARG;
  int rc = check_geom(out_shape, ksize, stride, padding, batch_size, &g);
  if (rc) return rc;
  if (n < 0 || n_out < 0 || (n_out > 0 && (!out_indices || !nbr_fwd)))
    return MSMD_ERR_INVALID_ARG;
  Arena a(workspace, workspace_bytes);
  ConvWs w;
  carve_conv(a, &w, batch_size, out_shape);
  if (!a.ok()) return MSMD_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  if (n_out > 0) hipMemsetAsync(nbr_fwd, 0xFF, sizeof(int32_t) * (size_t)g.kvol * n_out, st);
  if (n > 0)
    MSMD_LAUNCH(conv_fill, dim3(ceil_div(n, 256), g.kvol), dim3(256), 0, st, indices, n, g,
                       w.bits, w.prefix, n_out, out_indices, nbr_fwd, nbr_bwd);
  return launch_status();
}

// ------------------------------------------------- union + strided chain ---
// The fusion stack's stage chain (sparse_multimodal_encoder_painting.py:413-428): the input
// set of stage l's strided conv is the UNION (sparse_add) of the stage's own voxel set and the
// previous stage's conv output set.  The own sets are known up front, so the whole chain --
// |union_l| and |out_l| of every stage -- is counted on the device back to back and read
// ONCE (7 dependent host reads per LC step before: each waited for its counting kernels'
// turn next to the feature pass).  The fills run after the read through the existing entry
// points (msmd_sparse_add_fill, msmd_rulebook_conv3d_fill) on this call's per-level
// workspace regions, whose layouts are theirs: level l >= 1 has a union region (grid
// in_shapes[l]) followed by its conv region (grid out_shapes[l]), level 0 (whose input set is
// extra[0] as given) a conv region only; in_shapes[l] == out_shapes[l - 1].
namespace {
__global__ __launch_bounds__(256) void rows_mark(const int32_t* __restrict__ idx, int n, Geom g,
                                                 uint32_t* bits) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int4 r = ((const int4*)idx)[i];
  bitmap_set(bits, cell_id(r.x, r.y, r.z, r.w, g.shape));
}
inline size_t region_bytes(int batch, const int* shape) {
  return align_up(msmd_rulebook_conv_workspace_bytes(batch, shape));
}
}  // namespace

MSMD_EXPORT size_t msmd_rulebook_add_conv_chain_workspace_bytes(int batch_size, int levels,
                                                                const int* in_shapes,
                                                                const int* out_shapes) {
  if (levels < 1 || !in_shapes || !out_shapes) return 0;
  size_t total = 0;
  for (int l = 0; l < levels; ++l)   // (level 0 has no union region: its input set is given)
    total += (l ? region_bytes(batch_size, in_shapes + 3 * l) : 0) +
             region_bytes(batch_size, out_shapes + 3 * l);
  return total;
}

MSMD_EXPORT int msmd_rulebook_add_conv_count_chain(const int32_t* const* extra, const int* n_extra,
                                                   int batch_size, int levels,
                                                   const int* in_shapes, const int* out_shapes,
                                                   const int* ksizes, const int* strides,
                                                   const int* paddings, int32_t* counts,
                                                   void* workspace, size_t workspace_bytes,
                                                   msmd_stream_t stream) {
  if (levels < 1 || !extra || !n_extra || !in_shapes || !out_shapes || !ksizes || !strides ||
      !paddings || !counts)
    return MSMD_ERR_INVALID_ARG;
  if (workspace_bytes < msmd_rulebook_add_conv_chain_workspace_bytes(batch_size, levels, in_shapes,
                                                                     out_shapes) ||
      ((uintptr_t)workspace & 255))
    return MSMD_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  char* base = (char*)workspace;
  ConvWs prev{};
  for (int l = 0; l < levels; ++l) {
    if (n_extra[l] < 0 || (n_extra[l] > 0 && !extra[l])) return MSMD_ERR_INVALID_ARG;
    Geom g;      // conv geometry over the OUTPUT grid; gi: the input grid (union bitmap)
    int rc = check_geom(out_shapes + 3 * l, ksizes + 3 * l, strides + 3 * l, paddings + 3 * l,
                        batch_size, &g);
    if (rc) return rc;
    const int* ish = in_shapes + 3 * l;
    Geom gi = g;
    double cells = batch_size;
    for (int i = 0; i < 3; ++i) {
      if (ish[i] < 1) return MSMD_ERR_INVALID_ARG;
      gi.shape[i] = ish[i];
      cells *= ish[i];
    }
    if (cells >= 4294967295.0) return MSMD_ERR_RANGE;
    if (l > 0)
      for (int i = 0; i < 3; ++i)
        if (ish[i] != out_shapes[3 * (l - 1) + i]) return MSMD_ERR_INVALID_ARG;
    // union region (levels >= 1)
    const size_t ub = l ? region_bytes(batch_size, ish) : 0;
    ConvWs u{};
    if (l > 0) {
      Arena ua(base, ub);
      carve_conv(ua, &u, batch_size, ish);
      if (!ua.ok()) return MSMD_ERR_WORKSPACE;
      hipMemcpyAsync(u.bits, prev.bits, sizeof(uint32_t) * u.words, hipMemcpyDeviceToDevice, st);
      if (n_extra[l] > 0)
        MSMD_LAUNCH(rows_mark, dim3(ceil_div(n_extra[l], 256)), dim3(256), 0, st, extra[l],
                    n_extra[l], gi, u.bits);
      device_scan(PopcCount{u.bits}, StorePrefix{u.prefix}, (int)u.words, u.tiles, counts + 2 * l,
                  -1, st);
    }
    base += ub;
    // conv region
    const size_t cb = region_bytes(batch_size, out_shapes + 3 * l);
    Arena ca(base, cb);
    ConvWs w;
    carve_conv(ca, &w, batch_size, out_shapes + 3 * l);
    if (!ca.ok()) return MSMD_ERR_WORKSPACE;
    hipMemsetAsync(w.bits, 0, sizeof(uint32_t) * w.words, st);
    if (l == 0) {
      if (n_extra[0] > 0)
        MSMD_LAUNCH(conv_mark, dim3(ceil_div(n_extra[0], 256), g.kvol), dim3(256), 0, st, extra[0],
                    n_extra[0], g, w.bits);
    } else {
      const long oc = (long)batch_size * g.shape[0] * g.shape[1] * g.shape[2];
      if (oc <= kConvGatherCells)
        MSMD_LAUNCH(conv_mark_gather, dim3(ceil_div(oc, 256)), dim3(256), 0, st,
                    (const uint32_t*)u.bits, ish[0], ish[1], ish[2], g, oc, w.bits);
      else
        MSMD_LAUNCH(conv_mark_from_bits, dim3(ceil_div((long)u.words, 256)), dim3(256), 0, st,
                    (const uint32_t*)u.bits, (long)u.words, ish[0], ish[1], ish[2], g, w.bits);
    }
    device_scan(PopcCount{w.bits}, StorePrefix{w.prefix}, (int)w.words, w.tiles,
                counts + 2 * l + 1, -1, st);
    prev = w;
    base += cb;
  }
  return launch_status();
}

// ------------------------------------------------------------ pair lists ---
namespace {
inline int rows_padded(int n_rows) { return scan_num_tiles(n_rows > 0 ? n_rows : 1) * kScanTile; }
}

MSMD_EXPORT size_t msmd_rulebook_pairs_workspace_bytes(int kernel_volume, int n_rows) {
  ArenaSize a;
  a.take<int>((size_t)kernel_volume * (rows_padded(n_rows) / kScanTile) + 1);
  a.take<int>(64);
  return a.off;
}

MSMD_EXPORT int msmd_rulebook_pairs(const int32_t* nbr, int kernel_volume, int n_rows,
                                    int32_t* indice_pairs, int ld, int32_t* indice_num,
                                    void* workspace, size_t workspace_bytes,
                                    msmd_stream_t stream) {
  if (kernel_volume < 1 || n_rows < 0 || ld < 0 || !indice_num ||
      (n_rows > 0 && (!nbr || !indice_pairs)))
    return MSMD_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int rp = rows_padded(n_rows), tpk = rp / kScanTile;
  if ((double)kernel_volume * rp >= 2147483647.0) return MSMD_ERR_RANGE;
  Arena a(workspace, workspace_bytes);
  int* tiles = a.take<int>((size_t)kernel_volume * tpk + 1);
  int* total = a.take<int>(64);
  if (!a.ok()) return MSMD_ERR_WORKSPACE;
  // (entries past rows_pad -- ld larger than the padded table -- are not reached by the
  // empty entries' -1 writes)
  if (ld > rp)
    hipMemsetAsync(indice_pairs, 0xFF, sizeof(int32_t) * (size_t)kernel_volume * 2 * ld, st);
  if (n_rows == 0) {
    if (ld > 0 && ld <= rp)
      hipMemsetAsync(indice_pairs, 0xFF, sizeof(int32_t) * (size_t)kernel_volume * 2 * ld, st);
    hipMemsetAsync(indice_num, 0, sizeof(int32_t) * kernel_volume, st);
    return launch_status();
  }
  (void)total;
  MSMD_LAUNCH(scan_tile_sums<PairCount>, dim3(kernel_volume * tpk), dim3(kScanBlock), 0, st,
              PairCount{nbr, n_rows, rp}, kernel_volume * rp, tiles);
  MSMD_LAUNCH(pairs_apply_kernel, dim3(kernel_volume * tpk), dim3(kScanBlock), 0, st, nbr, n_rows,
              rp, ld, kernel_volume, (const int*)tiles, indice_pairs, indice_num);
  return launch_status();
}
