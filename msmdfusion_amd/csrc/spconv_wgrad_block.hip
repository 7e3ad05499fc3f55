// spconv_wgrad_block.hip -- weight gradient of the sparse convolution, whole c_in x c_out
// block per workgroup (round 3).
//
//   dW[k][ci][co] = sum over the pairs p of offset k of  in[i_p][ci] * dout[o_p][co]
//
// (the reference: the per-offset loop of mmdet3d/ops/spconv/include/spconv/spconv_ops.h:363-456
// -- gather both sides, torch::mm(input^T, output_grad) per offset).  Arithmetic as
// spconv_split.hip: every fp32 operand is the exact sum of three bf16 planes, six bf16 MFMA
// products accumulated in fp32 (fp32-equivalent); the pair index is the MFMA's contraction
// (32 pairs per v_mfma_f32_16x16x32_bf16).
//
// What bounded the previous kernels (64x64 channel slabs, every wave gathers and converts
// its own operands): each side's rows were fetched and split into planes once
// per slab of the OTHER side (2048 B and two conversions per pair at 128 x 128), and the
// conversion alone was half the kernel's time.  Here
//   * one workgroup (8 waves, one per CU) owns the WHOLE block of up to 128 x 128 channels
//     (wider layers: equal blocks of 4..8 sixteen-channel tiles per side), so a pair's two
//     rows are fetched ONCE and split ONCE (1024 B per pair at 128 x 128);
//   * the waves have roles.  Waves 4-7 PRODUCE: 16-byte buffer loads of the fp32 rows (an
//     absent pair is an out-of-range offset: zeros, no traffic, no branch), fp32 -> 3 bf16
//     planes in registers, ds_write_b128 into an LDS ring in MFMA OPERAND ORDER (lane (i, g)
//     of a producer holds pairs 8g..8g+7 of channel-row i: exactly one lane's 16-byte operand
//     of a 16 x 16 x 32 tile, so images are lane-linear and conflict-free, nothing is
//     transposed).  Waves 0-3 CONSUME: each owns a quadrant of the block's tiles, reads its
//     A and B operands from the ring with ds_read_b128 and does nothing but MFMAs.  One
//     producer and one consumer share each SIMD: conversion VALU work runs beside the
//     matrix pipe instead of in front of it;
//   * ring of 3 steps (one step = 32 pairs = one contraction), ONE s_barrier per step:
//     iteration n: producers write step n, consumers multiply step n-2 and prefetch the
//     first operands of step n-1.  Row loads run two steps ahead of their conversion, pair
//     indices three (registers only; vmcnt never drained at a barrier);
//   * work = the sequence of all steps of all (block, offset) segments, cut into EQUAL
//     contiguous ranges, one per workgroup (persistent, balanced to one step).  A range
//     that crosses a segment boundary flushes its accumulators: partial slot (g + segment)
//     for workgroup g -- at most G + segments - 1 slots (18 MB at 128 x 128) instead of one
//     per (offset, 2048-pair chunk) (116 MB at 135 k rows); wgrad_block_reduce_kernel adds a
//     segment's slots in workgroup order (fixed: deterministic) and writes dW in the
//     caller's layout.
//   * round 4: the step sequence is ROW-CHUNK major.  A segment is (block, chunk of output
//     rows, offset): the pairs of offset k whose output row lies in [c R, (c + 1) R) -- a
//     contiguous piece of the offset's pair list, which is in ascending output row
//     (msmd_rulebook_pair_segments finds the boundaries once per rulebook).  In offset-major
//     order every offset's pass streamed the whole feature map past an L2 that had long
//     forgotten it: a row was fetched from HBM once per offset it takes part in (PMC, r03:
//     651 MB per launch at L2 hit 0.11 for 46 MB of features -- 5.7 TB/s, the kernel sat on
//     the bandwidth ceiling).  Now the 27 segments of a chunk are neighbours in the sequence,
//     run at the same time on consecutive workgroups, and workgroup <-> range is XCD-aware
//     (block b runs on XCD b % 8: it takes range (b % 8) * (G / 8) + b / 8, so one XCD's
//     L2 serves one contiguous eighth of the rows): dout rows of a chunk are fetched once for
//     its 27 offsets, the input rows once for the offsets that share them.
#include "common.hpp"

#include <stdlib.h>

#include <type_traits>

namespace msmd {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// products kept for NP planes as (A plane, B plane), smallest terms first (spconv_split.hip)
template <int NP>
struct Prod;
template <>
struct Prod<1> {
  static constexpr int n = 1;
  static constexpr int a[1] = {0};
  static constexpr int b[1] = {0};
};
template <>
struct Prod<2> {
  static constexpr int n = 3;
  static constexpr int a[3] = {1, 0, 0};
  static constexpr int b[3] = {0, 1, 0};
};
template <>
struct Prod<3> {
  static constexpr int n = 6;
  static constexpr int a[6] = {2, 0, 1, 1, 0, 0};
  static constexpr int b[6] = {0, 2, 1, 0, 1, 0};
};

__device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

constexpr unsigned kOob = 0xffffff00u;   // byte offset no buffer covers: loads return 0
constexpr int kRing = 3;                 // steps in the LDS ring
constexpr int kMaxTiles = 8;             // 16-channel tiles per side of a block
constexpr int kMinSteps = 8;             // smallest range worth a workgroup (256 pairs)
constexpr int kMaxKvol = 64;

template <int V>
using ic = std::integral_constant<int, V>;

// Phase timing of one producer and one consumer wave (build with `make PROF=1`; never in
// the shipped library): s_memtime deltas summed per phase over the first 16 workgroups.
#ifdef MSMD_KERNEL_PROF
__device__ unsigned long long g_wbprof[16];
#define WB_BEGIN() unsigned long long wb_t = __builtin_amdgcn_s_memtime()
#define WB_MARK(i)                                                                    \
  {                                                                                   \
    const unsigned long long wb_n = __builtin_amdgcn_s_memtime();                     \
    if (lane == 0 && blockIdx.x < 16) atomicAdd(&g_wbprof[i], wb_n - wb_t);           \
    wb_t = wb_n;                                                                      \
  }
#else
#define WB_BEGIN()
#define WB_MARK(i)
#endif

// Geometry of a side: T tiles per block, the first producer wave (and consumer row /
// column 0) takes n0 = ceil(T / 2) of them, the second the rest.  Tile (a0 + a), row i of
// a group of n tiles starting at tile a0  <->  channel 16 a0 + n i + a of the block: a
// lane's n channels are consecutive floats (one load), a pair's 16 lanes read one
// contiguous piece of the row.
__host__ __device__ inline int side_n0(int T) { return (T + 1) / 2; }

// position in the step sequence: segment = blk * NS + loc (loc = chunk * kvol + k), step j of
// `steps`; the segment's pairs are entries [p0, p0 + num) of offset k's pair list
struct Cursor {
  int seg, loc, k, blk, j, steps, num, p0;
};

// The segment table of one block (global memory): prefix[NS + 1] = steps before each segment,
// p0[NS], cnt[NS].  Every index is uniform; the reads are SCALAR loads issued from assembly
// (left to the compiler they became vector loads, whose waits drain the producers' row loads
// in flight at every segment change).
__device__ __forceinline__ int sload1(const int32_t* base, int idx) {
  int v;
  asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)"
               : "=s"(v)
               : "s"(base), "s"(idx * 4)
               : "memory");
  return v;
}
struct SegTab {
  const int32_t* prefix;
  const int32_t* p0;
  const int32_t* cnt;
  int NS, kvol, nseg;
  __device__ __forceinline__ void load(Cursor& c) const {
    if (c.seg < nseg) {
      unsigned long long pp;
      int pv, cv;
      asm volatile("s_load_dwordx2 %0, %3, %6\n\ts_load_dword %1, %4, %6\n\t"
                   "s_load_dword %2, %5, %6\n\ts_waitcnt lgkmcnt(0)"
                   : "=&s"(pp), "=&s"(pv), "=&s"(cv)
                   : "s"(prefix), "s"(p0), "s"(cnt), "s"(c.loc * 4)
                   : "memory");
      c.steps = (int)(pp >> 32) - (int)(pp & 0xffffffffu);
      c.num = cv;
      c.p0 = pv;
      c.k = c.loc % kvol;
    } else {
      c.steps = 0x7fffffff;
      c.num = 0;
      c.p0 = 0;
      c.k = 0;
    }
  }
  // the step at global index s (s < total steps); Sk = steps of one block's NS segments
  __device__ __forceinline__ Cursor at(int s, int Sk) const {
    Cursor c;
    c.blk = s / Sk;
    const int r = s - c.blk * Sk;
    int lo = 0, hi = NS;          // largest loc with prefix[loc] <= r (its segment is not empty)
    if (NS < 64) {                // one chunk (NS = kvol): the whole prefix in one wave-wide load
      const int lane = threadIdx.x & 63;
      const int v = lane < NS ? prefix[lane] : 0x7fffffff;
      lo = __builtin_popcountll(__ballot(v <= r)) - 1;
      lo = __builtin_amdgcn_readfirstlane(lo);
    } else {
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (sload1(prefix, mid) <= r) lo = mid; else hi = mid;
      }
    }
    c.loc = lo;
    c.seg = c.blk * NS + lo;
    c.j = r - sload1(prefix, lo);
    load(c);
    return c;
  }
  __device__ __forceinline__ void advance(Cursor& c) const {
    if (++c.j < c.steps) return;
    c.j = 0;
    do {
      ++c.seg;
      if (++c.loc == NS) {
        c.loc = 0;
        ++c.blk;
      }
      load(c);
    } while (c.seg < nseg && c.steps == 0);
  }
};

struct BlockArgs {
  const float* in;
  const float* dout;
  const int32_t* pairs;
  const int32_t* num;
  const int32_t* segtab;  // prefix[NS + 1] | p0[NS] | cnt[NS], NS = nchunk * kvol
  float* partial;
  int cin, cout, ld, kvol, nchunk;
  int TA, TB, nba, nbb;   // tiles per block and blocks per side
  int min_steps;
  int dbg;                // experiments: 1 rows folded onto 4096 (cache hits), 2 no row loads
};

// s_waitcnt lgkmcnt(0) as an INSTRUCTION the compiler's wait-count pass sees (vmcnt 63,
// expcnt 7: not waited for)
__device__ __forceinline__ void wait_lds() { __builtin_amdgcn_s_waitcnt(0xc07f); }

// ------------------------------------------------------------------ producer --
// One producer wave = one side (in / dout), N of its tiles.  Per step and lane: 8 pairs x N
// channels.
template <int N>
__device__ __forceinline__ void load_row(__amdgpu_buffer_rsrc_t rs, unsigned off,
                                         unsigned (&out)[N]) {
  if constexpr (N == 4) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
    out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; out[3] = v[3];
  } else if constexpr (N == 3) {
    const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(rs, (int)off, 0, 0);
    out[0] = v[0]; out[1] = v[1]; out[2] = v[2];
  } else if constexpr (N == 2) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, 0);
    out[0] = v[0]; out[1] = v[1];
  } else {
    out[0] = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0);
  }
}

// raw [8 pairs][N channels] -> op[tile = channel][plane] (the lane's 8 contraction slots).
// v_cvt_pk_bf16_f32 of (pair 2t, pair 2t+1) is dword t of the operand: the transposition
// the MFMA wants costs nothing.  Scalar residuals (a v_pk_add_f32 beside MFMAs costs more
// than the two v_sub_f32 it replaces: the file is built with -fno-slp-vectorize).
// (Tried: the residual v - float(bf16) as ONE v_dot2c_f32_bf16 -- 7 VALU operations per
// operand dword instead of 11.  Measured on MI355X: the conversion phase of a producer wave
// got SLOWER (1465 -> 1622 cycles per 32-pair step at 128 x 128: the dot instruction is
// multi-pass) and the accumulate form did not reproduce the subtraction bit for bit.)
template <int NP, int N, int T0, int T1>   // operand dwords [T0, T1) of 4 (pairs 2t, 2t + 1)
__device__ __forceinline__ void split_rows(const unsigned (&r)[8][N], u32x4 (&op)[N][NP]) {
#pragma unroll
  for (int t = T0; t < T1; ++t)
#pragma unroll
    for (int a = 0; a < N; ++a) {
      float v0 = __uint_as_float(r[2 * t][a]), v1 = __uint_as_float(r[2 * t + 1][a]);
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) {
        unsigned hi;
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(v0), "v"(v1));
        op[a][pl][t] = hi;
        if (pl + 1 < NP) {   // exact residuals
          v0 = v0 - __uint_as_float(hi << 16);
          v1 = v1 - __uint_as_float(hi & 0xffff0000u);
        }
      }
    }
}

// Eight producer waves (two per SIMD beside one consumer): wave (side, half h, parity)
// converts the steps of its parity -- one wave alone issues ~350 instructions per step and
// cannot keep up with the consumer's 96 MFMAs (measured: 3100 cycles of producer work per
// 1536-cycle step); two waves taking alternate steps have two steps' time for each.
//
// An OWN step spans two iterations (= two barriers): part 1 (iteration s - 1) issues the
// index load of the own step after next, the row loads of the next own step, and converts
// the first half of step s into registers; part 2 (iteration s) converts the rest and
// writes the step to ring slot s % 3 -- the slot is free only once barrier s - 1 has been
// passed (the consumers read step s - 3 during iteration s - 1).  Buffer loads return in
// order, so a wait for index registers drains every OLDER load: an own step's index load is
// issued BEFORE its row loads; the wait for it one own step later leaves those 8 row loads
// in flight, and they have two iterations to land.
template <int NP, int N, int PAR>
__device__ __forceinline__ void produce(const BlockArgs& A, const SegTab& tab, Cursor cur,
                                        int nsteps, int q4, int side, int a0, u32x4* ring,
                                        int lane) {
  const int i = lane & 15, g = lane >> 4;
  const int T = side ? A.TB : A.TA, c = side ? A.cout : A.cin;
  const __amdgpu_buffer_rsrc_t rs_rows = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(side ? A.dout : A.in), 0, (int)kOob, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_idx = __builtin_amdgcn_make_buffer_rsrc(
      (void*)A.pairs, 0, (int)((unsigned)A.kvol * 2u * (unsigned)A.ld * 4u), 0x00020000);
  const unsigned row_bytes = (unsigned)c * 4u;
  const unsigned lane_col = (unsigned)(16 * a0 + N * i) * 4u;   // inside the block
  const int slot_u = (A.TA + A.TB) * NP * 64;                   // u32x4 units per ring slot
  u32x4* dst0 = ring + ((side ? A.TA : 0) + a0) * NP * 64 + lane;
#ifdef MSMD_WGRAD_BLOCK_DBG
  const bool fold = A.dbg & 1, norows = A.dbg & 2;
#endif

  u32x4 idx[2];             // pair indices of the next own step but one
  int idx_rem;              // pairs of that step from this lane's first one on (<= 0: none)
  unsigned idx_col;         // byte offset of the lane's channels in the row
  unsigned raw[2][8][N];    // rows of this own step and the next
  unsigned off[8];
  u32x4 op[N][NP];

  // per-segment values, recomputed only when the cursor enters another segment
  int seg_id = -1;
  unsigned seg_idx_off = 0;   // byte offset of the segment's pair list (this side)
  unsigned seg_col = 0;       // byte offset of the lane's channels in a row
  auto seg_update = [&]() {
    seg_id = cur.seg;
    seg_idx_off = (unsigned)((cur.k * 2 + side) * A.ld) * 4u;
    const int bs = side ? cur.blk % A.nbb : cur.blk / A.nbb;
    seg_col = (unsigned)(bs * T * 16) * 4u + lane_col;
  };
  if (PAR) tab.advance(cur);   // first own step = step PAR
  seg_update();
  const unsigned lane_idx_off = (unsigned)(8 * g) * 4u;
  int t_idx = PAR;            // step the next index load is for
  auto load_idx = [&]() {
    // past the range: the loads still go out (any address the descriptor covers), rem = 0
    const int pos = 32 * cur.j;       // inside the segment; its pairs start at entry p0
    const unsigned voff = seg_idx_off + (unsigned)(cur.p0 + pos) * 4u + lane_idx_off;
    idx[0] = __builtin_amdgcn_raw_buffer_load_b128(rs_idx, (int)voff, 0, 0);
    idx[1] = __builtin_amdgcn_raw_buffer_load_b128(rs_idx, (int)(voff + 16u), 0, 0);
    idx_rem = (t_idx < nsteps ? cur.num - pos : 0) - 8 * g;
    idx_col = seg_col;
    tab.advance(cur);          // the other parity's step
    tab.advance(cur);
    if (cur.seg != seg_id) seg_update();
    t_idx += 2;
  };
  auto make_offsets = [&]() {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      unsigned r = idx[e >> 2][e & 3];
#ifdef MSMD_WGRAD_BLOCK_DBG
      r = fold ? (r & 4095u) : r;
#endif
      const unsigned o = __umul24(r, row_bytes) + idx_col;   // rows < 2^24 (host check)
      off[e] = e < idx_rem ? o : kOob;
#ifdef MSMD_WGRAD_BLOCK_DBG
      off[e] = norows ? kOob : off[e];
#endif
    }
  };
  auto load_rows = [&](auto slot) {
    constexpr int S = decltype(slot)::value;
#pragma unroll
    for (int e = 0; e < 8; ++e) load_row<N>(rs_rows, off[e], raw[S][e]);
  };
  const bool wb_on = side == 0 && a0 == 0 && PAR == 0;   // (PROF builds: the wave that is timed)
  (void)wb_on;
  WB_BEGIN();
  int ring_slot = PAR;      // ring slot of the current own step (s % 3)
  // part 1 of the own step in raw slot S: loads for later own steps, first half converted
  auto part1 = [&](auto slot) {
    constexpr int S = decltype(slot)::value;
#ifdef MSMD_WGRAD_BLOCK_DBG
    if (A.dbg & 64) return;     // barriers only
    if (A.dbg & 128) { make_offsets(); load_idx(); return; }   // no row loads issued at all
#endif
    make_offsets();                      // next own step (indices loaded one own step ago)
    __builtin_amdgcn_sched_barrier(0);
    load_idx();                          // the own step after next
    __builtin_amdgcn_sched_barrier(0);
    load_rows(ic<S ^ 1>{});              // next own step
    if (wb_on) WB_MARK(0);
#ifdef MSMD_WGRAD_BLOCK_DBG
    if (A.dbg & 4) {   // no conversion: raw bits as operands (wrong results by design)
#pragma unroll
      for (int a = 0; a < N; ++a)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) op[a][pl] = (u32x4){raw[S][0][a], raw[S][1][a], raw[S][2][a], raw[S][3][a]};
    } else
#endif
    split_rows<NP, N, 0, 2>(raw[S], op);
    // (pin the half-converted operands here: the compiler otherwise sinks the conversion
    // behind the barrier, next to the LDS writes of part 2, and the two parts are uneven)
#pragma unroll
    for (int a = 0; a < N; ++a)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) asm volatile("" : "+v"(op[a][pl]));
    if (wb_on) WB_MARK(1);
  };
  auto part2 = [&](auto slot) {
    constexpr int S = decltype(slot)::value;
#ifdef MSMD_WGRAD_BLOCK_DBG
    if (A.dbg & (64 | 128)) return;
#endif
    // (and keep the second half of the conversion on this side of the barrier)
#pragma unroll
    for (int e = 4; e < 8; ++e)
#pragma unroll
      for (int a = 0; a < N; ++a) asm volatile("" : "+v"(raw[S][e][a]));
#ifdef MSMD_WGRAD_BLOCK_DBG
    if (!(A.dbg & 4))
#endif
    split_rows<NP, N, 2, 4>(raw[S], op);
    if (wb_on) WB_MARK(1);
    u32x4* d = dst0 + ring_slot * slot_u;
#ifdef MSMD_WGRAD_BLOCK_DBG
    if (!(A.dbg & 32))
#endif
#pragma unroll
    for (int a = 0; a < N; ++a)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) d[(a * NP + pl) * 64] = op[a][pl];
    ring_slot = ring_slot >= 1 ? ring_slot - 1 : ring_slot + 2;   // (s + 2) % 3
    wait_lds();
    if (wb_on) WB_MARK(2);
  };
  auto bar = [&]() {
    asm volatile("s_barrier" ::: "memory");
    if (wb_on) WB_MARK(3);
  };
  // prologue: rows of the first own step in flight, indices of the second loaded
  load_idx();
  make_offsets();
  __builtin_amdgcn_sched_barrier(0);
  load_idx();
  __builtin_amdgcn_sched_barrier(0);
  load_rows(ic<0>{});
  if (PAR == 0) {
    part1(ic<0>{});          // step 0 is due in iteration 0: its first half before the loop
    for (int it = 0; it < q4; ++it) {
      part2(ic<0>{}); bar();
      part1(ic<1>{}); bar();
      part2(ic<1>{}); bar();
      part1(ic<0>{}); bar();
    }
  } else {
    for (int it = 0; it < q4; ++it) {
      part1(ic<0>{}); bar();
      part2(ic<0>{}); bar();
      part1(ic<1>{}); bar();
      part2(ic<1>{}); bar();
    }
  }
  asm volatile("s_barrier" ::: "memory");   // the consumers are two steps behind
  asm volatile("s_barrier" ::: "memory");
}

// ------------------------------------------------------------------ consumer --
// One consumer wave = NA x NB tiles (a quadrant of the block), alone with its SIMD's matrix
// pipe.  With three waves per SIMD a wave has 168 registers: 64 accumulators, the two B
// halves of the step (Y, Z) and THREE single-tile A slots.  A step is 2 NA phases, each
// multiplying one A tile by one B half (12 MFMAs at NB = 4):
//     pass 1 (first B half, Y):   a = 0, 1, .., NA-1     (the second B half -> Z meanwhile)
//     pass 2 (second B half, Z):  a = NA-1, .., 1, 0     (the next step's first half -> Y)
// -- the turn reuses tile NA-1: 2 NA - 1 A reads per step, numbered l = 0 .. 2 NA - 2 (and
// on into the next step); read l goes to slot (beta + l) % 3 and is issued TWO phases
// before its use (one phase = 192 cycles was not enough: with the producers' 48 KB of
// writes and 132 KB of reads per step in the LDS queues a read takes longer than that --
// the MFMA stream ran at 62 % of its rate).  beta advances by 2 NA - 1 per step: the step
// code is instantiated for beta = 0, 1, 2.
// (Four half-slots of two tiles each -- every operand read once per step -- need 160 +
// registers and spilled.)
// Step m runs after barrier m + 1 (the producers finished it before barrier m).
template <int NP, int NA, int NB>
__device__ __forceinline__ void consume(const BlockArgs& A, const SegTab& tab, Cursor cur,
                                        int nsteps, int q4, int ta0, int tb0,
                                        const u32x4* ring, int wg, int lane) {
  using P = Prod<NP>;
  constexpr int LB = (NB + 1) / 2, HB = NB - LB;
  constexpr int NL = 2 * NA - 1;   // A reads per step
  const int slot_u = (A.TA + A.TB) * NP * 64;
  const u32x4* a_src = ring + ta0 * NP * 64 + lane;
  const u32x4* b_src = ring + (A.TA + tb0) * NP * 64 + lane;
  const size_t slot_elems = (size_t)A.TA * A.TB * 256;

  f32x4 acc[NA][NB];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  u32x4 X[3][NP], Y[LB][NP], Z[HB][NP];

  auto rd_a = [&](auto xs, int slot, int t) {   // A tile t of ring slot `slot` -> X[xs]
    constexpr int Xs = decltype(xs)::value;
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) X[Xs][pl] = a_src[slot * slot_u + (t * NP + pl) * 64];
  };
  auto rd_b = [&](auto& dst, int slot, int t0, auto n) {
    constexpr int Nn = decltype(n)::value;
#pragma unroll
    for (int t = 0; t < Nn; ++t)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) dst[t][pl] = b_src[slot * slot_u + ((t0 + t) * NP + pl) * 64];
  };
  auto flush = [&]() {   // accumulators -> partial slot (workgroup + segment), fragment order
    float* dst = A.partial + (size_t)(wg + cur.seg) * slot_elems;
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        *(f32x4*)(dst + ((size_t)((ta0 + a) * A.TB + tb0 + b) * 64 + lane) * 4) = acc[a][b];
        acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
  };
  int m = 0, slot = 0;
  const bool wb_on = ta0 == 0 && tb0 == 0;   // (PROF builds: the wave that is timed)
  (void)wb_on;
  WB_BEGIN();
  // A read number l (>= NL: of the next step) into its slot
  auto issue = [&](auto beta, auto lc, int cur_slot, int next_slot) {
    constexpr int Bt = decltype(beta)::value, L = decltype(lc)::value;
    constexpr int l = L >= NL ? L - NL : L;
    constexpr int tile = l < NA ? l : 2 * NA - 2 - l;
    rd_a(ic<(Bt + L) % 3>{}, L >= NL ? next_slot : cur_slot, tile);
  };
  // step m; its A reads 0 and 1 are in flight (slots beta, beta + 1), its first B half in Y
  auto step = [&](auto beta) {
    constexpr int Bt = decltype(beta)::value;
    if (wb_on) WB_MARK(7);
    if (m < nsteps) {
      const int nslot = slot == 2 ? 0 : slot + 1;
      // pass 1: phase p = a uses read p
#pragma unroll
      for (int a = 0; a < NA; ++a) {
        __builtin_amdgcn_sched_barrier(0);
        if (a == 0) rd_b(Z, slot, LB, ic<HB>{});
        if (a == 0) issue(beta, ic<2>{}, slot, nslot);
        if (a == 1) issue(beta, ic<3>{}, slot, nslot);
        if (a == 2) issue(beta, ic<4>{}, slot, nslot);
        if (a == 3) issue(beta, ic<5>{}, slot, nslot);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < P::n; ++t)
#pragma unroll
          for (int b = 0; b < LB; ++b)
            acc[a][b] = mfma_bf16(X[(Bt + a) % 3][P::a[t]], Y[b][P::b[t]], acc[a][b]);
      }
      if (wb_on) WB_MARK(4);
      // pass 2: phase p = NA + i uses read p - 1 (tile NA - 1 - i); the turn issues nothing
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int a = NA - 1 - i;
        __builtin_amdgcn_sched_barrier(0);
        if (i == 0) rd_b(Y, nslot, 0, ic<LB>{});  // step m + 1 was complete at the last barrier
        if (i == 1) issue(beta, ic<NA + 2>{}, slot, nslot);
        if (i == 2) issue(beta, ic<NA + 3>{}, slot, nslot);
        if (i == 3) issue(beta, ic<NA + 4>{}, slot, nslot);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < P::n; ++t)
#pragma unroll
          for (int b = 0; b < HB; ++b)
            acc[a][LB + b] = mfma_bf16(X[(Bt + (i == 0 ? NA - 1 : NA - 1 + i)) % 3][P::a[t]],
                                       Z[b][P::b[t]], acc[a][LB + b]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (wb_on) WB_MARK(5);
      // last step of its segment, or of this workgroup's range
      if (cur.j + 1 == cur.steps || m + 1 == nsteps) flush();
      tab.advance(cur);
      slot = nslot;
    }
    ++m;
    if (wb_on) WB_MARK(6);
    asm volatile("s_barrier" ::: "memory");
  };
  // the matrix pipe first: the producers take the issue slots the MFMA stream leaves
  __builtin_amdgcn_s_setprio(3);
  asm volatile("s_barrier" ::: "memory");
  asm volatile("s_barrier" ::: "memory");
  rd_a(ic<0>{}, 0, 0);
  rd_a(ic<1>{}, 0, NA > 1 ? 1 : 0);
  rd_b(Y, 0, 0, ic<LB>{});
  const int total = 4 * q4;
  int it = 0;
  for (; it + 3 <= total; it += 3) {
    step(ic<0>{});
    step(ic<NL % 3>{});
    step(ic<(2 * NL) % 3>{});
  }
  if (it < total) { step(ic<0>{}); ++it; }
  if (it < total) step(ic<NL % 3>{});
}

// total steps of one block's segments; S = nblk * Sk; range length per workgroup
__device__ __forceinline__ int range_len(int S, int G, int min_steps) {
  int L = (S + G - 1) / G;
  return L < min_steps ? min_steps : L;
}

// range <-> workgroup: block b runs on XCD b % 8 (observed placement, MI355X_MICROARCH.md);
// XCD x takes the x-th contiguous eighth of the ranges, so the chunks it works on -- and the
// rows they gather -- stay in ITS L2.  A bijection on [0, G) for any G (identity unless 8 | G).
__host__ __device__ inline int range_of_block(int b, int G) {
  return (G & 7) == 0 ? (b & 7) * (G >> 3) + (b >> 3) : b;
}

template <int NP>
__global__ __launch_bounds__(768) void spconv_wgrad_block_kernel(BlockArgs A) {
  __shared__ __attribute__((aligned(16))) u32x4 ring[kRing * 2 * kMaxTiles * NP * 64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int NS = A.nchunk * A.kvol;
  const int Sk = sload1(A.segtab, NS);
  const int nblk = A.nba * A.nbb, nseg = nblk * NS;
  const int S = Sk * nblk;
  const int L = range_len(S, (int)gridDim.x, A.min_steps);
  const int wg = range_of_block((int)blockIdx.x, (int)gridDim.x);
  const int s0 = wg * L;
  if (s0 >= S) return;
  const int s1 = s0 + L < S ? s0 + L : S;
  const int nsteps = s1 - s0, q4 = (nsteps + 3) / 4;
  SegTab tab{A.segtab, A.segtab + NS + 1, A.segtab + 2 * NS + 1, NS, A.kvol, nseg};
  const Cursor cur = tab.at(s0, Sk);

  const int n0a = side_n0(A.TA), n0b = side_n0(A.TB);
  if (wave >= 4) {
    const int pw = wave - 4, side = pw >> 2, h = (pw >> 1) & 1, par = pw & 1;
    const int T = side ? A.TB : A.TA, n0 = side ? n0b : n0a;
    const int n = h ? T - n0 : n0, a0 = h ? n0 : 0;
#define MSMD_PRODUCE(N_)                                                                    \
  case N_:                                                                                  \
    if (par) produce<NP, N_, 1>(A, tab, cur, nsteps, q4, side, a0, ring, lane);             \
    else produce<NP, N_, 0>(A, tab, cur, nsteps, q4, side, a0, ring, lane);                 \
    break
    switch (n) {
      MSMD_PRODUCE(4);
      MSMD_PRODUCE(3);
      MSMD_PRODUCE(2);
    }
#undef MSMD_PRODUCE
  } else {
    const int wa = wave >> 1, wb = wave & 1;
    const int na = wa ? A.TA - n0a : n0a, ta0 = wa ? n0a : 0;
    const int nb = wb ? A.TB - n0b : n0b, tb0 = wb ? n0b : 0;
#define MSMD_CONSUME(NA_, NB_)                                                              \
  case NA_ * 8 + NB_:                                                                       \
    consume<NP, NA_, NB_>(A, tab, cur, nsteps, q4, ta0, tb0, ring, wg, lane);               \
    break
    switch (na * 8 + nb) {
      MSMD_CONSUME(4, 4);
      MSMD_CONSUME(4, 3);
      MSMD_CONSUME(4, 2);
      MSMD_CONSUME(3, 4);
      MSMD_CONSUME(3, 3);
      MSMD_CONSUME(3, 2);
      MSMD_CONSUME(2, 4);
      MSMD_CONSUME(2, 3);
      MSMD_CONSUME(2, 2);
    }
#undef MSMD_CONSUME
  }
}

// dW = sum of the partial slots of an offset's segments -- chunk after chunk, a segment's
// slots in range order (fixed: deterministic); fragment order -> dW layout.  One thread = one
// 16-byte piece of a (block, offset) image: loads are lane-linear (coalesced, 8 in flight),
// the four sums go to four rows of dW.  grid = (pieces of one offset / 256, kvol).
__global__ __launch_bounds__(256) void wgrad_block_reduce_kernel(BlockArgs A, int G, int krsc,
                                                                 float* __restrict__ dw) {
  const int k = blockIdx.y;
  const int NS = A.nchunk * A.kvol;
  const int32_t* __restrict__ prefix = A.segtab;
  const int Sk = prefix[NS];
  const int nblk = A.nba * A.nbb;
  const int L = range_len(Sk * nblk, G, A.min_steps);
  const int CA = A.TA * 16, CB = A.TB * 16;
  const int n0a = side_n0(A.TA), n1a = A.TA - n0a, n0b = side_n0(A.TB), n1b = A.TB - n0b;
  const int pieces_blk = A.TA * A.TB * 64;          // f32x4 pieces of one block image
  const size_t slot_elems = (size_t)pieces_blk * 4;
  const size_t stride = slot_elems / 4;
  const int per_k = A.cin * A.cout;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < nblk * pieces_blk; e += gridDim.x * 256) {
    const int blk = e / pieces_blk, pc = e - blk * pieces_blk;
    const int tile = pc >> 6, ln = pc & 63;
    const int ta = tile / A.TB, tb = tile - ta * A.TB;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < A.nchunk; ++c) {
      const int loc = c * A.kvol + k;
      const int st0 = prefix[loc], steps_k = prefix[loc + 1] - st0;
      if (steps_k <= 0) continue;
      const int seg = blk * NS + loc;
      const int gs = blk * Sk + st0;
      const int g_lo = gs / L, g_hi = (gs + steps_k - 1) / L;
      const f32x4* src = (const f32x4*)(A.partial + (size_t)seg * slot_elems) + pc;
      int g = g_lo;
      for (; g + 8 <= g_hi + 1; g += 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(g + u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
      }
      for (; g <= g_hi; ++g) s += src[(size_t)g * stride];
    }
    // piece (tile ta, tb; lane (ib = ln & 15, gq = ln >> 4)): rows ia = 4 gq + r, column ib
    const int ba = blk / A.nbb, bb = blk - ba * A.nbb;
    const int ib = ln & 15, gq = ln >> 4;
    const int co = bb * CB + (tb < n0b ? n0b * ib + tb : 16 * n0b + n1b * ib + (tb - n0b));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ia = 4 * gq + r;
      const int ci = ba * CA + (ta < n0a ? n0a * ia + ta : 16 * n0a + n1a * ia + (ta - n0a));
      if (krsc) dw[((size_t)co * A.kvol + k) * A.cin + ci] = s[r];   // [c_out][K][c_in]
      else dw[(size_t)k * per_k + (size_t)ci * A.cout + co] = s[r];
    }
  }
}

// The segment table of a pair list: chunk c of offset k = the pairs whose OUTPUT row lies in
// [c R, (c + 1) R) -- entries [lower_bound(c R), lower_bound((c + 1) R)) of the offset's
// output-row list, which msmd_rulebook_pairs leaves in ascending order.  One workgroup:
// (nchunk + 1) * kvol binary searches, then the exclusive scan of the segments' step counts
// in (chunk, offset) order.  table = prefix[NS + 1] | p0[NS] | cnt[NS].
constexpr int kMaxChunks = 256;
__global__ __launch_bounds__(1024) void pair_segments_kernel(const int32_t* __restrict__ pairs,
                                                             const int32_t* __restrict__ num,
                                                             int ld, int kvol, int chunk_rows,
                                                             int nchunk,
                                                             int32_t* __restrict__ table) {
  extern __shared__ int sh[];
  int* cs = sh;                                   // [(nchunk + 1) * kvol] first pair of a chunk
  int* scan = sh + (nchunk + 1) * kvol;           // [1024 / 64] wave totals
  const int NS = nchunk * kvol;
  for (int e = threadIdx.x; e < (nchunk + 1) * kvol; e += 1024) {
    const int c = e / kvol, k = e - c * kvol;
    const int n = num[k];
    int v = 0;
    if (c == nchunk) {
      v = n;
    } else if (c > 0) {
      const int32_t* o = pairs + ((size_t)k * 2 + 1) * ld;
      const long key = (long)c * chunk_rows;
      int lo = 0, hi = n;                         // first entry with out row >= key
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (o[mid] < key) lo = mid + 1; else hi = mid;
      }
      v = lo;
    }
    cs[e] = v;
  }
  __syncthreads();
  int32_t* prefix = table;
  int32_t* p0 = table + NS + 1;
  int32_t* cnt = table + 2 * NS + 1;
  // exclusive scan of the step counts: thread t owns segments [t * per, (t + 1) * per)
  const int per = (NS + 1023) / 1024;
  const int b0 = threadIdx.x * per, b1 = b0 + per < NS ? b0 + per : NS;
  int sum = 0;
  for (int l = b0; l < b1; ++l) {
    const int n = cs[l + kvol] - cs[l];           // (chunk c + 1, k) - (chunk c, k)
    p0[l] = cs[l];
    cnt[l] = n;
    sum += (n + 31) >> 5;
  }
  int total;
  int run = block_excl_scan<1024>(sum, scan, &total);
  for (int l = b0; l < b1; ++l) {
    prefix[l] = run;
    run += (cs[l + kvol] - cs[l] + 31) >> 5;
  }
  if (threadIdx.x == 0) prefix[NS] = total;
}

// blocks of a side: the fewest equal blocks of 4..8 tiles; 0 = unsupported
int side_blocks(int c) {
  if (c < 64 || c % 16) return 0;
  const int T = c / 16;
  for (int nb = 1; nb <= T / 4; ++nb)
    if (T % nb == 0 && T / nb <= kMaxTiles && T / nb >= 4) return nb;
  return 0;
}

int cu_count() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1)
      return 256;
    // (MSMD_RESERVE_CUS: CUs left to the step's other queues, see spconv_split.hip)
    const char* e = getenv("MSMD_RESERVE_CUS");
    int r = e ? atoi(e) : 0;
    r = r < 0 ? 0 : r > v / 2 ? v / 2 : r;
    r &= ~7;                     // keep the grid a multiple of 8: the XCD-aware range mapping
    return v - r;
  }();
  return n;
}

}  // namespace

bool wgrad_block_supported(int c_in, int c_out, int kvol, int ld) {
  // (row indices below 2^24: the row offset is a 24-bit multiply)
  return side_blocks(c_in) > 0 && side_blocks(c_out) > 0 && kvol <= kMaxKvol &&
         ld < (1 << 24) && (double)kvol * 2.0 * ld * 4.0 < 4.0e9;
}

size_t wgrad_segment_table_ints(int kvol, int nchunk) { return 3 * (size_t)nchunk * kvol + 1; }
int wgrad_block_max_kvol() { return kMaxKvol; }

int wgrad_pair_segments(const int32_t* pairs, const int32_t* num, int ld, int kvol,
                        int chunk_rows, int nchunk, int32_t* table, hipStream_t st) {
  if (nchunk < 1 || nchunk > kMaxChunks || kvol < 1 || kvol > kMaxKvol || chunk_rows < 1)
    return MSMD_ERR_INVALID_ARG;
  const size_t smem = sizeof(int) * ((size_t)(nchunk + 1) * kvol + 16);
  static LdsGrant granted;
  const int rc = optin_dynamic_lds((const void*)pair_segments_kernel, smem, granted);
  if (rc != MSMD_OK) return rc;
  MSMD_LAUNCH(pair_segments_kernel, dim3(1), dim3(1024), smem, st, pairs, num, ld, kvol,
              chunk_rows, nchunk, table);
  return launch_status();
}

// partial slots: one per workgroup + segment; behind them room for a one-chunk segment table
// (callers that pass no table: the offset-major sequence of round 3)
size_t wgrad_block_workspace_bytes(int kvol, int c_in, int c_out, int nchunk) {
  const int nba = side_blocks(c_in), nbb = side_blocks(c_out);
  if (!nba || !nbb) return 0;
  if (nchunk < 1) nchunk = 1;
  const size_t slot = (size_t)(c_in / nba) * (c_out / nbb) * sizeof(float);
  return align_up(slot * ((size_t)cu_count() + (size_t)nba * nbb * kvol * nchunk)) +
         align_up(sizeof(int32_t) * wgrad_segment_table_ints(kvol, 1));
}

// partials + reduction; rows * channels * 4 < 4 GiB on both sides is the caller's check.
// segtab (msmd_rulebook_pair_segments) / nchunk: the row-chunk-major sequence; NULL: built
// here for one chunk = the whole pair list of every offset.
int wgrad_block(const float* in_feat, int c_in, const float* d_out, int c_out,
                const int32_t* pairs, const int32_t* num, int ld, int kvol, int np,
                float* d_weight, int krsc_out, float* ws, const int32_t* segtab, int nchunk,
                hipStream_t st) {
  BlockArgs A;
  A.in = in_feat;
  A.dout = d_out;
  A.pairs = pairs;
  A.num = num;
  A.partial = ws;
  A.cin = c_in;
  A.cout = c_out;
  A.ld = ld;
  A.kvol = kvol;
  A.nba = side_blocks(c_in);
  A.nbb = side_blocks(c_out);
  A.TA = c_in / 16 / A.nba;
  A.TB = c_out / 16 / A.nbb;
  A.min_steps = kMinSteps;
  if (!segtab) {
    nchunk = 1;
    const size_t slot = (size_t)(c_in / A.nba) * (c_out / A.nbb) * sizeof(float);
    int32_t* own = (int32_t*)((char*)ws + align_up(slot * ((size_t)cu_count() +
                                                          (size_t)A.nba * A.nbb * kvol)));
    const int rc = wgrad_pair_segments(pairs, num, ld, kvol, ld > 0 ? ld : 1, 1, own, st);
    if (rc != MSMD_OK) return rc;
    segtab = own;
  }
  A.segtab = segtab;
  A.nchunk = nchunk;
  static const int dbg = [] { const char* e = getenv("MSMD_WGRAD_DBG"); return e ? atoi(e) : 0; }();
  A.dbg = dbg;
  const int G = cu_count();
  if (np == 3) MSMD_LAUNCH(spconv_wgrad_block_kernel<3>, dim3(G), dim3(768), 0, st, A);
  else if (np == 2) MSMD_LAUNCH(spconv_wgrad_block_kernel<2>, dim3(G), dim3(768), 0, st, A);
  else MSMD_LAUNCH(spconv_wgrad_block_kernel<1>, dim3(G), dim3(768), 0, st, A);
  const int rb = ceil_div(c_in * c_out / 4, 256);   // one thread per 16-byte piece
  MSMD_LAUNCH(wgrad_block_reduce_kernel, dim3(rb, kvol), dim3(256), 0, st, A, G, krsc_out,
              d_weight);
  return launch_status();
}

}  // namespace msmd

#ifdef MSMD_KERNEL_PROF
// out[16] <- cycle sums since the last call (then cleared).  Producer wave (in side, first
// half): 0 offsets + index and row load issue, 1 conversion (incl. the wait for its rows),
// 2 LDS writes + their wait, 3 barrier.  Consumer wave (first quadrant): 4 first A half,
// 5 second A half, 6 flush / cursor, 7 barrier.
MSMD_EXPORT int msmd_debug_wbprof(unsigned long long* out) {
  hipDeviceSynchronize();
  unsigned long long z[16] = {0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(msmd::g_wbprof), sizeof(z)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(msmd::g_wbprof), z, sizeof(z)) != hipSuccess) return -1;
  return 0;
}
#endif
