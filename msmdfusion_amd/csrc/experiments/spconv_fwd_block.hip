// EXPERIMENT, NOT BUILT INTO libmsmd_hip.so.  Correct (it passed every split-conv parity test
// behind msmd_spconv_fwd_split) and SLOWER than spconv_fwd_split_kernel: 329 against 249 us on
// the 128 -> 128 bench layer -- measurements and the ablation that explains them in
// profiles/r03_fwd_block_ablation.txt, discussion in DESIGN.md 8.3.  It needs the per-tile
// activity bytes (a tile_weight_kernel extension: bit r of byte k = 16-row group r of the tile
// is connected through offset k) and a dispatch hook in spconv_split.hip's dispatch_fwd_split;
// kept as the starting point for a version that reads pre-split bf16 planes.
//
// spconv_fwd_block.hip -- forward / dgrad of the sparse convolution with producer and
// consumer waves (round 3; the wide layers: 5..8 sixteen-channel output tiles per pass).
//
//   out[o][:] = sum over offsets k of  in[nbr[k][o]][:] . W[k]        (nbr = -1: no term)
//
// (the reference: mmdet3d/ops/spconv/include/spconv/spconv_ops.h:163-252 -- gather the
// offset's input rows, torch::mm with W[k], scatter-add; dgrad is the same walk with the
// backward table and W[k]^T, spconv_ops.h:363-456.)  Arithmetic as spconv_split.hip: every
// fp32 operand is the exact sum of three bf16 planes, six bf16 MFMA products accumulated in
// fp32; same packed weight image, same row tiling, same stream-K work sequence and tile
// exchange as spconv_fwd_split_kernel -- what changes is WHO does what inside a workgroup.
//
// In spconv_fwd_split_kernel every wave gathers and converts its own 32 rows, waits for the
// workgroup's weight DMA at a barrier whose queue drain stalls its own MFMA stream, and
// reads all of the unit's weights from LDS: with the gathers and the conversion removed it
// still ran at half the matrix pipe's rate.  Here (as in
// spconv_wgrad_block.hip) one workgroup of 12 waves owns a CU:
//   * waves 4-11 PRODUCE.  Producer p owns row tile p (16 of the tile's 128 rows).  Per
//     unit (offset k, 32-channel k-block kb): the row index (4 bytes, three units ahead),
//     two 16-byte buffer loads of the row's 8 channels (two units ahead; "no neighbour" is
//     an out-of-range offset: zeros, no traffic), the wave's share of the unit's packed
//     weights (register loads, same queue: everything in order, nothing drained), fp32 -> 3
//     bf16 planes, ds_write_b128 into an LDS ring in MFMA operand order;
//   * waves 0-3 CONSUME: wave (rh, ch) owns 4 row tiles x NT/2 output tiles (64
//     accumulator registers at NT = 8), reads its operands from the ring (24 ds_read_b128
//     per 96 MFMAs) and does nothing else; a row tile no row of which is connected through
//     the offset is skipped at 16-row granularity (the activity bytes of the tiling plan);
//   * ring of 3 units, ONE s_barrier per unit, consumers two units behind the producers;
//     units stream across tile boundaries (the pipeline is filled once per launch, not once
//     per tile);
//   * stream-K as before: the launch's (tile, offset) cost sequence is cut into equal
//     ranges, one per workgroup (ticket order); a tile cut by a range boundary is summed
//     through the exchange buffer by the workgroup that holds its last offset.
#include "common.hpp"

#include <stdlib.h>

#include <type_traits>

namespace msmd {
namespace {

typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// products kept for NP planes as (weight plane, row plane), smallest terms first
template <int NP>
struct Prod;
template <>
struct Prod<1> {
  static constexpr int n = 1;
  static constexpr int w[1] = {0};
  static constexpr int r[1] = {0};
};
template <>
struct Prod<2> {
  static constexpr int n = 3;
  static constexpr int w[3] = {1, 0, 0};
  static constexpr int r[3] = {0, 1, 0};
};
template <>
struct Prod<3> {
  static constexpr int n = 6;
  static constexpr int w[6] = {2, 0, 1, 1, 0, 0};
  static constexpr int r[6] = {0, 2, 1, 0, 1, 0};
};

__device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

constexpr unsigned kOob = 0xffffff00u;   // byte offset no buffer covers: loads return 0
constexpr int kRows = 128;               // rows of a tile (8 row tiles of 16)
constexpr int kConsumers = 4;
constexpr int kSkMinRanks = 64;          // as spconv_split.hip

template <int V>
using ic = std::integral_constant<int, V>;

struct FwdArgs {
  const float* in;
  const u32x4* wp;
  const int32_t* nbr;        // [kvol][ld], tile order when `order` is given
  const int32_t* order;      // output row of tiled position p (nullptr: p itself)
  float* out;                // first channel of this pass
  f32x4* scratch;            // exchange buffer: one tile's accumulators per ticket
  int* flags;                // one per ticket
  int* ticket;
  const int32_t* tile_start; // [n_tiles + 1] cost prefix (msmd_rulebook_tile_prefix)
  const uint8_t* act;        // [n_tiles][32]: bit r of byte k = row tile r connected through k
  unsigned wp_bytes;
  int n_in, cin, ld, n_out, kvol, flip, ldo, width, nt_total, mt0, c0, c1, n_tiles;
  int dbg;   // -DMSMD_FWD_BLOCK_DBG builds (experiments): 1 no row gathers, 2 no conversion,
             // 4 no weight loads, 8 no LDS writes, 16 no MFMAs, 32 no LDS reads, 64 rows folded
};

__device__ __forceinline__ void wait_lds() { __builtin_amdgcn_s_waitcnt(0xc07f); }

// entries of a[0..n) (+ c0 * index) that are <= x, 64-ary search; same value in every lane
__device__ __forceinline__ int count_le(const int32_t* __restrict__ a, int n, int x, int c0,
                                        int lane) {
  int lo = 0, hi = n;
  while (hi - lo > 64) {
    const int step = (hi - lo + 63) >> 6;
    const int idx = lo + lane * step;
    const int v = idx < hi ? a[idx] + c0 * idx : 0x7fffffff;
    const int c = __builtin_popcountll(__ballot(v <= x));
    if (c == 0) {
      hi = lo;
    } else {
      const int nlo = lo + (c - 1) * step + 1;
      const int nhi = lo + c * step < hi ? lo + c * step : hi;
      lo = nlo;
      hi = nhi;
    }
  }
  const int idx = lo + lane;
  const int v = idx < hi ? a[idx] + c0 * idx : 0x7fffffff;
  return lo + __builtin_popcountll(__ballot(v <= x));
}

// ---------------------------------------------------------------- work walk --
// One wave's position in its workgroup's range of the stream-K sequence: tiles from the
// highest down, the workgroup's offsets of the tile ascending, k-blocks ascending.  Every
// wave of the workgroup walks the same sequence on its own (scalar registers + one lane per
// offset); nothing is shared, nothing is staged.
struct Walk {
  int g0, g1, S, lo_tile, kbt;
  int tile, ts, ts_hi, sk_ts, lo, hi;
  unsigned mask;      // this workgroup's offsets of the tile not yet finished (bit k)
  unsigned all;       // the tile's active offsets
  int k, kb;
  unsigned a8;        // activity byte of (tile, k)
  unsigned av;        // lane k: activity byte of offset k of this tile
  unsigned av_next;   // ... of tile - 1 (loaded a tile ahead)
  bool owner;

  __device__ __forceinline__ bool valid() const { return U(tile) >= U(lo_tile); }
  static __device__ __forceinline__ int U(int v) { return __builtin_amdgcn_readfirstlane(v); }

  __device__ __forceinline__ unsigned load_av(const FwdArgs& A, int t, int lane) const {
    const int tt = t < 0 ? 0 : t;
    return A.act[(size_t)tt * 32 + (lane & 31)];
  }
  // lane k: stream-K cost of offset k of this tile and the costs before it
  __device__ __forceinline__ void costs(const FwdArgs& A, int lane, int& cost, int& pre) const {
    const unsigned a = lane < A.kvol ? av : 0u;
    const int groups = __popc((a | (a >> 1)) & 0x55u);   // 32-row groups, as the plan counts
    cost = a ? A.c1 + groups : 0;
    pre = wave_excl_scan(cost, lane);
  }
  // tile, ts_hi and av are set: the workgroup's share of the tile
  __device__ __forceinline__ void enter(const FwdArgs& A, int lane) {
    int cost, pre;
    costs(A, lane, cost, pre);
    const int total = __builtin_amdgcn_readlane(pre, 63);   // (lane 63's own cost is 0)
    const int tw = total > 0 ? total : 1;
    ts = ts_hi - tw;
    sk_ts = ts + A.c0 * (tile + 1);
    lo = g0 > sk_ts ? g0 - sk_ts : 0;
    hi = g1 - sk_ts < tw ? g1 - sk_ts : tw;
    all = (unsigned)__ballot(cost > 0);
    mask = (unsigned)__ballot(cost > 0 && pre >= lo && pre < hi);
    if (all) {
      const int p_last = __builtin_amdgcn_readlane(pre, 31 - __builtin_clz(all));
      owner = p_last >= lo && p_last < hi;
    } else {
      owner = lo == 0 && hi > 0;
    }
    // (all of this is wave-uniform by construction; say so, or the compiler keeps the walk in
    // vector registers and turns its branches into exec-mask code with full queue drains)
    ts = U(ts);
    sk_ts = U(sk_ts);
    lo = U(lo);
    hi = U(hi);
    all = (unsigned)U((int)all);
    mask = (unsigned)U((int)mask);
    owner = U((int)owner) != 0;
    kb = 0;
    k = mask ? __builtin_ctz(mask) : 0;
    a8 = (unsigned)__builtin_amdgcn_readlane((int)av, k);
  }
  // false: the workgroup has nothing to do
  __device__ __forceinline__ bool init(const FwdArgs& A, int seg, int grid, int lane) {
    kbt = (A.cin + 31) >> 5;
    const int W = __builtin_amdgcn_readfirstlane(A.tile_start[A.n_tiles]) + A.c0 * A.n_tiles;
    S = (W + grid - 1) / grid;
    S = S < A.c0 + kSkMinRanks ? A.c0 + kSkMinRanks : S;
    g0 = seg * S;
    g1 = g0 + S < W ? g0 + S : W;
    if (g0 >= W) return false;
    lo_tile = U(count_le(A.tile_start, A.n_tiles, g0, A.c0, lane)) - 1;
    tile = U(count_le(A.tile_start, A.n_tiles, g1 - 1, A.c0, lane)) - 1;
    // a range that ends inside its highest tile's overhead zone does not visit that tile
    if (g1 <= __builtin_amdgcn_readfirstlane(A.tile_start[tile]) + A.c0 * (tile + 1)) --tile;
    if (tile < lo_tile) return false;
    ts_hi = __builtin_amdgcn_readfirstlane(A.tile_start[tile + 1]);
    av = load_av(A, tile, lane);
    av_next = load_av(A, tile - 1, lane);
    enter(A, lane);
    return true;
  }
  __device__ __forceinline__ void next_tile(const FwdArgs& A, int lane) {
    tile = U(tile) - 1;
    if (tile < lo_tile) return;
    ts_hi = ts;
    av = av_next;
    av_next = load_av(A, tile - 1, lane);
    enter(A, lane);
  }
  // the next unit of the tile; false: the tile's units are exhausted
  __device__ __forceinline__ bool next_unit() {
    kb = U(kb) + 1;
    if (kb < kbt) return true;
    kb = 0;
    mask = (unsigned)U((int)(mask & (mask - 1)));
    if (!mask) return false;
    k = __builtin_ctz(mask);
    a8 = (unsigned)__builtin_amdgcn_readlane((int)av, k);
    return true;
  }
  __device__ __forceinline__ bool last_unit_of_tile() const {
    const unsigned m = (unsigned)U((int)mask);
    return U(kb) + 1 == kbt && (m & (m - 1)) == 0u;
  }
  __device__ __forceinline__ bool none() const { return U((int)mask) == 0; }   // no units here
};

// ------------------------------------------------------------------ producer --
// 8 fp32 (two 16-byte pieces of one row) -> NP operands of 8 bf16
template <int NP>
__device__ __forceinline__ void split_row8(const u32x4 (&raw)[2], u32x4 (&pl)[NP]) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const unsigned e0 = raw[t >> 1][2 * (t & 1)], e1 = raw[t >> 1][2 * (t & 1) + 1];
    float v0 = __uint_as_float(e0), v1 = __uint_as_float(e1);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      unsigned hi;
      asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(v0), "v"(v1));
      pl[p][t] = hi;
      if (p + 1 < NP) {   // exact residuals
        v0 = v0 - __uint_as_float(hi << 16);
        v1 = v1 - __uint_as_float(hi & 0xffff0000u);
      }
    }
  }
}

// Iteration n of a producer: the offsets of unit n + 2's rows (their indices arrived during
// the last iteration), the index loads of unit n + 3, the row and weight loads of unit
// n + 2, then unit n (loaded two iterations ago) is converted and written to ring slot
// n % 3.  Buffer loads return in order: the index loads go out BEFORE the iteration's row
// loads, so the wait for them one iteration later leaves those in flight.
// PW producer waves per workgroup (8: one row tile each; 4: two each).
template <int NT, int NP, int PW>
__device__ __forceinline__ void produce(const FwdArgs& A, Walk w, int p, u32x4* ring,
                                        int lane) {
  constexpr int RT = 8 / PW;                                     // row tiles per wave
  constexpr int kPw = (NP * NT + PW - 1) / PW;                   // weight pieces per wave
  constexpr int slot_u = (NP * NT + 8 * NP) * 64;                // u32x4 units per ring slot
  const int j = lane & 15, q = lane >> 4;
  const __amdgpu_buffer_rsrc_t rs_rows = __builtin_amdgcn_make_buffer_rsrc(
      (void*)A.in, 0, (int)((unsigned)A.n_in * (unsigned)A.cin * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_idx = __builtin_amdgcn_make_buffer_rsrc(
      (void*)A.nbr, 0, (int)((unsigned)A.kvol * (unsigned)A.ld * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w =
      __builtin_amdgcn_make_buffer_rsrc((void*)A.wp, 0, (int)A.wp_bytes, 0x00020000);
  const unsigned row_bytes = (unsigned)A.cin * 4u;
  const int lane_pos = 16 * RT * p + j;    // row tiles RT p .. RT p + RT - 1

  unsigned wsrc[kPw];   // byte offset of the wave's pieces inside a unit's weight image
  int wdst[kPw];        // ... and where they go in a ring slot
#pragma unroll
  for (int i = 0; i < kPw; ++i) {
    int piece = p + PW * i;
    if (piece >= NP * NT) piece = 0;     // a short last round repeats piece 0 (same bytes)
    const int pl = piece / NT, t = piece - pl * NT;
    int st = A.mt0 + t;                  // tiles past the packed image repeat the last one
    st = st < A.nt_total ? st : A.nt_total - 1;   // (computed, never stored)
    wsrc[i] = (unsigned)((pl * A.nt_total + st) * 64 + lane) * 16u;
    wdst[i] = (pl * NT + t) * 64 + lane;
  }
  u32x4* const rows_dst = ring + NP * NT * 64 + RT * p * NP * 64 + lane;

  while (w.valid() && w.none()) w.next_tile(A, lane);   // tiles with nothing for us

  // The walk's head runs five units ahead of the unit being written: a unit's row indices
  // are loaded three iterations before its rows are (one iteration was not enough: every
  // iteration then waited a full memory latency for them -- 0.5 us per unit with nothing
  // else to do).  Slot u % 3 holds unit u's indices and descriptor.
  unsigned idxq[3][RT];
  int dq_valid[3];                // the unit: inside the range?
  unsigned dq_col[3], dq_wbase[3];
  int count = 0;                  // units seen by the head
  u32x4 raw[3][RT][2], wr[3][kPw];
  unsigned off[RT], g_wbase = 0;

  auto load_idx = [&](auto slot) {
    constexpr int Sl = decltype(slot)::value;
    const bool v = w.valid();
    const int t = v ? Walk::U(w.tile) : 0, k = Walk::U(w.k), kb = Walk::U(w.kb);
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      int pos = t * kRows + lane_pos + 16 * r;
      pos = pos < A.n_out ? pos : A.n_out - 1;
      idxq[Sl][r] = __builtin_amdgcn_raw_buffer_load_b32(rs_idx, pos * 4, (k * A.ld) * 4, 0);
    }
    dq_valid[Sl] = v;
    dq_col[Sl] = (unsigned)kb * 128u;
    const int kw = A.flip ? A.kvol - 1 - k : k;
    dq_wbase[Sl] = (unsigned)((kw * w.kbt + kb) * NP * A.nt_total) * 1024u;
    if (v) {
      ++count;
      if (!w.next_unit()) {
        do w.next_tile(A, lane);
        while (w.valid() && w.none());
      }
    }
  };
  auto make_offset = [&](auto slot) {
    constexpr int Sl = decltype(slot)::value;
    const unsigned chan0 = (dq_col[Sl] >> 2) + 8u * q;
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      const bool ok = dq_valid[Sl] && (int)idxq[Sl][r] >= 0 && chan0 < (unsigned)A.cin;
      unsigned ri = idxq[Sl][r];
#ifdef MSMD_FWD_BLOCK_DBG
      if (A.dbg & 64) ri &= 4095u;
#endif
      off[r] = ok ? __umul24(ri, row_bytes) + dq_col[Sl] + 32u * q : kOob;
#ifdef MSMD_FWD_BLOCK_DBG
      if (A.dbg & 1) off[r] = kOob;
#endif
    }
    g_wbase = dq_wbase[Sl];
  };
  auto issue = [&](auto slot) {
    constexpr int Sl = decltype(slot)::value;
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      raw[Sl][r][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_rows, (int)off[r], 0, 0);
      raw[Sl][r][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_rows, (int)(off[r] + 16u), 0, 0);
    }
#ifdef MSMD_FWD_BLOCK_DBG
    if (A.dbg & 4) return;
#endif
#pragma unroll
    for (int i = 0; i < kPw; ++i)
      wr[Sl][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)wsrc[i], (int)g_wbase, 0);
  };
  auto finish = [&](auto slot) {
    constexpr int Sl = decltype(slot)::value;
    u32x4* d = ring + Sl * slot_u;
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      u32x4 pl[NP];
#ifdef MSMD_FWD_BLOCK_DBG
      if (A.dbg & 2) {
#pragma unroll
        for (int x = 0; x < NP; ++x) pl[x] = raw[Sl][r][x & 1];
      } else
#endif
      split_row8<NP>(raw[Sl][r], pl);
#ifdef MSMD_FWD_BLOCK_DBG
      if (A.dbg & 8) {
#pragma unroll
        for (int x = 0; x < NP; ++x) asm volatile("" :: "v"(pl[x]));
        continue;
      }
#endif
#pragma unroll
      for (int x = 0; x < NP; ++x) rows_dst[Sl * slot_u + (r * NP + x) * 64] = pl[x];
    }
#ifdef MSMD_FWD_BLOCK_DBG
    if (A.dbg & 8) {
#pragma unroll
      for (int i = 0; i < kPw; ++i) asm volatile("" :: "v"(wr[Sl][i]));
      return;
    }
#endif
#pragma unroll
    for (int i = 0; i < kPw; ++i) d[wdst[i]] = wr[Sl][i];
  };
  auto iter = [&](auto slot) {
    constexpr int Sl = decltype(slot)::value;
    make_offset(ic<(Sl + 2) % 3>{});       // unit n + 2
    __builtin_amdgcn_sched_barrier(0);
    load_idx(ic<(Sl + 2) % 3>{});          // unit n + 5, into the slot just read
    __builtin_amdgcn_sched_barrier(0);
    issue(ic<(Sl + 2) % 3>{});             // unit n + 2
    finish(slot);                          // unit n
    wait_lds();
    asm volatile("s_barrier" ::: "memory");
  };
  // prologue: units 0 and 1 in flight, the indices of units 2, 3, 4 loaded
  load_idx(ic<0>{});
  load_idx(ic<1>{});
  load_idx(ic<2>{});
  make_offset(ic<0>{});
  __builtin_amdgcn_sched_barrier(0);
  load_idx(ic<0>{});
  __builtin_amdgcn_sched_barrier(0);
  issue(ic<0>{});
  make_offset(ic<1>{});
  __builtin_amdgcn_sched_barrier(0);
  load_idx(ic<1>{});
  __builtin_amdgcn_sched_barrier(0);
  issue(ic<1>{});
  int n = 0;
  do {
    iter(ic<0>{});
    iter(ic<1>{});
    iter(ic<2>{});
    n += 3;
  } while (n < count);
  asm volatile("s_barrier" ::: "memory");   // the consumers are two units behind
  asm volatile("s_barrier" ::: "memory");
}

// ------------------------------------------------------------------ consumer --
// Wave (rh, ch): row tiles 4 rh .. 4 rh + 3 (operands X, three single-tile slots, each
// read two phases before its use) x output tiles ch NB .. ch NB + NB - 1 (the unit's
// weights, in two halves Y, Z).  A unit is 8 phases of one row tile x one weight half:
//     pass 1 (Y):  row tiles 0, 1, 2, 3        (the second weight half -> Z meanwhile)
//     pass 2 (Z):  row tiles 3, 2, 1, 0        (the next unit's first half -> Y)
// -- the schedule of spconv_wgrad_block.hip's consumer (see there for the slot rotation).
template <int NT, int NP>
__device__ __forceinline__ void consume(const FwdArgs& A, Walk w, int c, int seg,
                                        const u32x4* ring, int* orow_all, int lane) {
  using P = Prod<NP>;
  constexpr int NA = 4, NB = NT / 2;
  constexpr int LB = (NB + 1) / 2, HB = NB - LB;
  constexpr int NL = 2 * NA - 1;
  constexpr int slot_u = (NP * NT + 8 * NP) * 64;
  constexpr int kSlotU = 8 * NT * 64;   // one tile's accumulators, f32x4 units
  const int rh = c >> 1, ch = c & 1;
  const int j = lane & 15, q = lane >> 4;
  const u32x4* r_src = ring + NP * NT * 64 + rh * NA * NP * 64 + lane;
  const u32x4* w_src = ring + ch * NB * 64 + lane;

  f32x4 acc[NA][NB];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  u32x4 X[3][NP], Y[LB][NP], Z[HB > 0 ? HB : 1][NP];
#ifdef MSMD_FWD_BLOCK_DBG
  if (A.dbg & 32) {   // operands never read: any defined value
    const u32x4 v = {(unsigned)lane, 1u, 2u, 3u};
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
      X[0][pl] = X[1][pl] = X[2][pl] = v;
#pragma unroll
      for (int t = 0; t < LB; ++t) Y[t][pl] = v;
#pragma unroll
      for (int t = 0; t < (HB > 0 ? HB : 1); ++t) Z[t][pl] = v;
    }
  }
#endif
  // (output rows of this lane's row j in the wave's four row tiles, -1: none, are kept in
  // LDS -- orow_all -- between a tile's entry and its stores)

  auto rd_a = [&](auto xs, int slot, int t) {
    constexpr int Xs = decltype(xs)::value;
#ifdef MSMD_FWD_BLOCK_DBG
    if (A.dbg & 32) return;
#endif
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) X[Xs][pl] = r_src[slot * slot_u + (t * NP + pl) * 64];
  };
  auto rd_b = [&](auto& dst, int slot, int t0, auto nn) {
    constexpr int Nn = decltype(nn)::value;
#ifdef MSMD_FWD_BLOCK_DBG
    if (A.dbg & 32) return;
#endif
#pragma unroll
    for (int t = 0; t < Nn; ++t)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) dst[t][pl] = w_src[slot * slot_u + (pl * NT + t0 + t) * 64];
  };
  // (the tile-change code below takes the lane id as an OPAQUE value `ln`: what it derives
  // from it -- addresses, lane masks of the cost scan -- is then computed there, a few times
  // per workgroup, instead of being hoisted out of the unit loop and held in ~60 registers
  // that the loop's own operands need)
  auto opaque_lane = [&]() {
    int ln = lane;
    asm volatile("" : "+v"(ln));
    return ln;
  };
  auto tile_rows = [&](int ln) {   // on entering a tile
    int* orow = orow_all + c * (NA * 64) + ln;
    const int j = ln & 15;
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      const int pos = Walk::U(w.tile) * kRows + (rh * NA + a) * 16 + j;
      const int pc = pos < A.n_out ? pos : A.n_out - 1;
      const int r = A.order ? A.order[pc] : pc;
      orow[a * 64] = pos < A.n_out ? r : -1;
    }
  };
  // the tile is finished (for this workgroup): exchange or store, accumulators <- 0
  auto flush = [&](int ln) {
    const int lane = ln, q = ln >> 4;
    int* orow = orow_all + c * (NA * 64) + ln;
    if (!w.owner) {
      // a piece of a tile another workgroup owns: accumulators -> scratch[ticket], signal.
      // Agent-scope accesses to the scratch lines and the flag only: coherent across the
      // XCDs' L2s without a fence (spconv_split.hip).
      unsigned long long* sp = (unsigned long long*)(A.scratch + (size_t)seg * kSlotU + lane);
      // (opaque: the 2 x 16 piece addresses would otherwise be computed once, outside the
      // unit loop, and held in 64 registers for the whole kernel)
      unsigned long long* d = sp + (rh * NA * NT + ch * NB) * 128;
#pragma unroll
      for (int a = 0; a < NA; ++a) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          // (one running address, opaque: 2 x 16 precomputed ones cost 64 registers)
          asm volatile("" : "+v"(d));
          const u64x2 v = __builtin_bit_cast(u64x2, acc[a][b]);
          __hip_atomic_store(d, v[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(d + 1, v[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          d += 128;
        }
        d += (NT - NB) * 128;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // written through before the signal
      if (lane == 0)
        __hip_atomic_fetch_add(&A.flags[seg], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (w.lo > 0) {   // the lower tickets that hold the tile's first offsets, in order
        int cost, pre;
        w.costs(A, lane, cost, pre);
        for (int c2 = w.sk_ts / w.S; c2 < seg; ++c2) {
          const int rlo = c2 * w.S - w.sk_ts;
          if ((unsigned)__ballot(cost > 0 && pre >= rlo && pre < rlo + w.S) == 0u) continue;
          while (__hip_atomic_load(&A.flags[c2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <
                 kConsumers)
            __builtin_amdgcn_s_sleep(4);
          asm volatile("" ::: "memory");
          unsigned long long* sp = (unsigned long long*)(A.scratch + (size_t)c2 * kSlotU + lane);
          unsigned long long* d = sp + (rh * NA * NT + ch * NB) * 128;
#pragma unroll
          for (int a = 0; a < NA; ++a) {
            u64x2 v[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
              asm volatile("" : "+v"(d));
              v[b][0] = __hip_atomic_load(d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              v[b][1] = __hip_atomic_load(d + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              d += 128;
            }
            d += (NT - NB) * 128;
#pragma unroll
            for (int b = 0; b < NB; ++b) acc[a][b] += __builtin_bit_cast(f32x4, v[b]);
            __builtin_amdgcn_sched_barrier(0);   // (a row tile's pieces at a time: few temporaries)
          }
          if (lane == 0) {   // the last of the reading waves re-arms the flag
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (__hip_atomic_fetch_add(&A.flags[c2], 1, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT) == 2 * kConsumers - 1)
              __hip_atomic_store(&A.flags[c2], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
      // lane (j, q) holds out[row j][16 t + 4 q .. + 3] of every (row tile, output tile)
#pragma unroll
      for (int a = 0; a < NA; ++a) {
        const int row = orow[a * 64];
        if (row < 0) continue;
        float* o = A.out + (size_t)row * A.ldo + 4 * q;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const int ct = ch * NB + b;
          if (16 * ct + 4 * q < A.width) *(f32x4*)(o + 16 * ct) = acc[a][b];
        }
      }
    }
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  // to the next tile with units for this workgroup; a tile it owns without any (no row of
  // it is connected at all) is stored as zeros on the way
  auto next_tile = [&](int ln) {
    for (;;) {
      w.next_tile(A, ln);
      if (!w.valid()) return;
      if (w.none() && !w.owner) continue;
      tile_rows(ln);
      if (!w.none()) return;
      flush(ln);
    }
  };
  {
    const int ln = opaque_lane();
    if (w.none() && !w.owner) {
      next_tile(ln);
    } else {
      tile_rows(ln);
      if (w.none()) {
        flush(ln);
        next_tile(ln);
      }
    }
  }

  int slot = 0;
  auto issue = [&](auto beta, auto lc, int cur_slot, int next_slot) {
    constexpr int Bt = decltype(beta)::value, L = decltype(lc)::value;
    constexpr int l = L >= NL ? L - NL : L;
    constexpr int tile = l < NA ? l : 2 * NA - 2 - l;
    rd_a(ic<(Bt + L) % 3>{}, L >= NL ? next_slot : cur_slot, tile);
  };
  // One unit.  LAST = the tile's last unit for this workgroup: it reads nothing of the next
  // unit -- nothing but the accumulators is live across the exchange / store code that
  // follows it (with the 48 operand registers of the next unit live through that code the
  // register allocator spilled them for the whole loop).
  auto unit = [&](auto beta, auto last, int nslot) {
    constexpr int Bt = decltype(beta)::value;
    constexpr bool LAST = decltype(last)::value != 0;
    unsigned am = ((unsigned)Walk::U((int)w.a8) >> (4 * rh)) & 15u;   // which of the four row tiles are connected
#ifdef MSMD_FWD_BLOCK_DBG
    if (A.dbg & 16) am = 0;
    if (A.dbg & 128) am = 15u;
#endif
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      __builtin_amdgcn_sched_barrier(0);
      if (a == 0 && HB > 0) rd_b(Z, slot, LB, ic<HB>{});
      if (a == 0) issue(beta, ic<2>{}, slot, nslot);
      if (a == 1) issue(beta, ic<3>{}, slot, nslot);
      if (a == 2) issue(beta, ic<4>{}, slot, nslot);
      if (a == 3) issue(beta, ic<5>{}, slot, nslot);
      __builtin_amdgcn_sched_barrier(0);
      if ((am >> a) & 1u) {
#pragma unroll
        for (int t = 0; t < P::n; ++t)
#pragma unroll
          for (int b = 0; b < LB; ++b)
            acc[a][b] = mfma_bf16(Y[b][P::w[t]], X[(Bt + a) % 3][P::r[t]], acc[a][b]);
      }
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int a = NA - 1 - i;
      __builtin_amdgcn_sched_barrier(0);
      if (i == 0 && !LAST) rd_b(Y, nslot, 0, ic<LB>{});   // unit m + 1 was complete at the last barrier
      if (i == 1) issue(beta, ic<NA + 2>{}, slot, nslot);
      if (i == 2 && !LAST) issue(beta, ic<NA + 3>{}, slot, nslot);
      if (i == 3 && !LAST) issue(beta, ic<NA + 4>{}, slot, nslot);
      __builtin_amdgcn_sched_barrier(0);
      if (HB > 0 && ((am >> a) & 1u)) {
#pragma unroll
        for (int t = 0; t < P::n; ++t)
#pragma unroll
          for (int b = 0; b < HB; ++b)
            acc[a][LB + b] = mfma_bf16(Z[b][P::w[t]],
                                       X[(Bt + (i == 0 ? NA - 1 : NA - 1 + i)) % 3][P::r[t]],
                                       acc[a][LB + b]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!LAST) w.next_unit();
    slot = nslot;
  };
  static_assert(NL % 3 == 1, "the slot rotation below assumes betas 0, 1, 2 in turn");
  __builtin_amdgcn_s_setprio(3);
  asm volatile("s_barrier" ::: "memory");
  asm volatile("s_barrier" ::: "memory");
  // Tile by tile.  The inner loop holds only whole units (no exchange, no stores, no walk
  // to another tile): what is live across it is the accumulators, the operands and a few
  // scalars.  The tile's last unit reads nothing ahead; after the flush the next tile's
  // first operands are read afresh (one exposed LDS latency per tile) and the slot
  // rotation restarts at 0.
  int done = 0;   // units consumed = barriers passed after the first two
  while (w.valid()) {
    const int n_units = __builtin_popcount((unsigned)Walk::U((int)w.mask)) * w.kbt;
    rd_a(ic<0>{}, slot, 0);
    rd_a(ic<1>{}, slot, 1);
    rd_b(Y, slot, 0, ic<LB>{});
    int u = 0;
    for (; u + 3 <= n_units - 1; u += 3) {
      unit(ic<0>{}, ic<0>{}, slot == 2 ? 0 : slot + 1);
      asm volatile("s_barrier" ::: "memory");
      unit(ic<1>{}, ic<0>{}, slot == 2 ? 0 : slot + 1);
      asm volatile("s_barrier" ::: "memory");
      unit(ic<2>{}, ic<0>{}, slot == 2 ? 0 : slot + 1);
      asm volatile("s_barrier" ::: "memory");
    }
    const int r = n_units - 1 - u;   // 0, 1 or 2 more whole units, then the tile's last
    if (r >= 1) {
      unit(ic<0>{}, ic<0>{}, slot == 2 ? 0 : slot + 1);
      asm volatile("s_barrier" ::: "memory");
    }
    if (r >= 2) {
      unit(ic<1>{}, ic<0>{}, slot == 2 ? 0 : slot + 1);
      asm volatile("s_barrier" ::: "memory");
    }
    if (r == 0) unit(ic<0>{}, ic<1>{}, slot == 2 ? 0 : slot + 1);
    else if (r == 1) unit(ic<1>{}, ic<1>{}, slot == 2 ? 0 : slot + 1);
    else unit(ic<2>{}, ic<1>{}, slot == 2 ? 0 : slot + 1);
    done += n_units;
    {
      const int ln = opaque_lane();
      flush(ln);
      next_tile(ln);
    }
    asm volatile("s_barrier" ::: "memory");
  }
  // the producers run whole triples of units (at least one)
  int rest = done == 0 ? 3 : (3 - done % 3) % 3;
  for (; rest > 0; --rest) asm volatile("s_barrier" ::: "memory");
}

template <int NT, int NP, int PW>
__global__ __launch_bounds__(64 * (kConsumers + PW)) void spconv_fwd_block_kernel(FwdArgs A) {
  extern __shared__ __attribute__((aligned(16))) u32x4 ring[];
  __shared__ int s_ticket;
  __shared__ int s_orow[kConsumers * 4 * 64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid == 0) {
    // tickets in the order workgroups become resident: an owner only ever waits for lower
    // tickets, which are running by construction (spconv_split.hip)
    const int t = atomicAdd(A.ticket, 1);
    if (t == (int)gridDim.x - 1) *A.ticket = 0;
    s_ticket = t;
  }
  __syncthreads();
  const int seg = __builtin_amdgcn_readfirstlane(s_ticket);
  Walk w;
  if (!w.init(A, seg, (int)gridDim.x, lane)) return;
  if (wave >= kConsumers) produce<NT, NP, PW>(A, w, wave - kConsumers, ring, lane);
  else consume<NT, NP>(A, w, wave, seg, ring, s_orow, lane);
}

int cu_count() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1)
      return 256;
    return v;
  }();
  return n;
}

template <int NT, int NP, int PW>
int launch(const FwdArgs& A, int grid, hipStream_t st) {
  const size_t smem = sizeof(u32x4) * 3 * (NP * NT + 8 * NP) * 64;
  auto kern = spconv_fwd_block_kernel<NT, NP, PW>;
  static bool attr = false;   // per instantiation
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)smem);
    attr = true;
  }
  MSMD_LAUNCH(kern, dim3(grid), dim3(64 * (kConsumers + PW)), smem, st, A);
  return launch_status();
}

}  // namespace

// workgroups of a launch over `row_tiles` 128-row tiles (never more than the exchange
// buffer has slots for: spconv_split.hip sizes it for at least one per CU)
int fwd_block_grid(int row_tiles, int kvol) {
  const long ranks_max = (long)row_tiles * kvol;
  const long cus = cu_count();
  return (int)(ranks_max < cus ? ranks_max : cus);
}

// one pass (<= 8 output tiles) of the forward / dgrad; `tiles` = output tiles of the pass
int fwd_block(const float* in, int n_in, int cin, const void* wp, const int32_t* nbr, int ld,
              int n_out, int kvol, int flip, const int32_t* order, int* ticket, float* out,
              int ldo, int width, int nt_total, int mt0, int tiles, void* scratch, int* flags,
              const int32_t* tile_start, const uint8_t* act, int c0, int c1, int np,
              hipStream_t st) {
  FwdArgs A;
  A.in = in;
  A.wp = (const u32x4*)wp;
  A.nbr = nbr;
  A.order = order;
  A.out = out;
  A.scratch = (f32x4*)scratch;
  A.flags = flags;
  A.ticket = ticket;
  A.tile_start = tile_start;
  A.act = act;
  A.n_in = n_in;
  A.cin = cin;
  A.ld = ld;
  A.n_out = n_out;
  A.kvol = kvol;
  A.flip = flip;
  A.ldo = ldo;
  A.width = width;
  A.nt_total = nt_total;
  A.mt0 = mt0;
  A.c0 = c0;
  A.c1 = c1;
  A.n_tiles = ceil_div(n_out, kRows);
  A.wp_bytes = (unsigned)((size_t)kvol * ((cin + 31) / 32) * np * nt_total * 1024);
  static const int dbg = [] { const char* e = getenv("MSMD_FWD_DBG"); return e ? atoi(e) : 0; }();
  A.dbg = dbg;
  const int grid = fwd_block_grid(A.n_tiles, kvol);
  // producer waves per workgroup: MSMD_FWD_PW=4 / 8 (experiments)
  static const int pw = [] { const char* e = getenv("MSMD_FWD_PW"); return e ? atoi(e) : 4; }();
#define MSMD_GO(NT_, PW_)                                          \
  (np == 3 ? launch<NT_, 3, PW_>(A, grid, st)                      \
           : np == 2 ? launch<NT_, 2, PW_>(A, grid, st) : launch<NT_, 1, PW_>(A, grid, st))
  if (pw == 8) {
    if (tiles > 6) return MSMD_GO(8, 8);
    if (tiles > 4) return MSMD_GO(6, 8);
    return MSMD_GO(4, 8);
  }
  if (tiles > 6) return MSMD_GO(8, 4);
  if (tiles > 4) return MSMD_GO(6, 4);
  return MSMD_GO(4, 4);
#undef MSMD_GO
}

}  // namespace msmd
