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

// position in the step sequence: segment = blk * kvol + k, step j of `steps`
struct Cursor {
  int seg, k, blk, j, steps, num;
};

// One wave's view of the shared tables (LDS): steps and pair count per offset.
struct SegTab {
  const int* sstep;   // [kvol]
  const int* snum;    // [kvol]
  int kvol, nseg;
  __device__ __forceinline__ void load(Cursor& c) const {
    c.steps = c.seg < nseg ? __builtin_amdgcn_readfirstlane(sstep[c.k]) : 0x7fffffff;
    c.num = c.seg < nseg ? __builtin_amdgcn_readfirstlane(snum[c.k]) : 0;
  }
  // the step at global index s (s < total steps); Sk = steps of one block's kvol segments
  __device__ __forceinline__ Cursor at(int s, int Sk) const {
    Cursor c;
    c.blk = s / Sk;
    int r = s - c.blk * Sk, k = 0;
    for (; k < kvol; ++k) {
      const int st = __builtin_amdgcn_readfirstlane(sstep[k]);
      if (r < st) break;
      r -= st;
    }
    c.k = k;
    c.seg = c.blk * kvol + k;
    c.j = r;
    load(c);
    return c;
  }
  __device__ __forceinline__ void advance(Cursor& c) const {
    if (++c.j < c.steps) return;
    c.j = 0;
    do {
      ++c.seg;
      if (++c.k == kvol) {
        c.k = 0;
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
  float* partial;
  int cin, cout, ld, kvol;
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
// (MSMD_WGRAD_DOT2=1: the residual v - float(bf16) as ONE v_dot2c_f32_bf16 -- 7 VALU
// operations per operand dword instead of 11.  Measured on MI355X: the conversion phase of a
// producer wave got SLOWER (1465 -> 1622 cycles per 32-pair step at 128 x 128: the dot
// instruction is multi-pass) and the accumulate form did not reproduce the subtraction bit
// for bit (test_split_wgrad failed).  Off.)
#ifndef MSMD_WGRAD_DOT2
#define MSMD_WGRAD_DOT2 0
#endif
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
template <int NP, int N>
__device__ __forceinline__ void split_rows(const unsigned (&r)[8][N], u32x4 (&op)[N][NP],
                                           unsigned sel_lo, unsigned sel_hi) {
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int a = 0; a < N; ++a) {
      float v0 = __uint_as_float(r[2 * t][a]), v1 = __uint_as_float(r[2 * t + 1][a]);
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) {
        unsigned hi;
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(v0), "v"(v1));
        op[a][pl][t] = hi;
        if (pl + 1 < NP) {   // exact residuals
#if MSMD_WGRAD_DOT2
          const bf16x2 h2 = __builtin_bit_cast(bf16x2, hi);
          v0 = __builtin_amdgcn_fdot2_f32_bf16(h2, __builtin_bit_cast(bf16x2, sel_lo), v0, false);
          v1 = __builtin_amdgcn_fdot2_f32_bf16(h2, __builtin_bit_cast(bf16x2, sel_hi), v1, false);
#else
          v0 = v0 - __uint_as_float(hi << 16);
          v1 = v1 - __uint_as_float(hi & 0xffff0000u);
#endif
        }
      }
    }
}

// Iteration `it` (6 q of them, q = ceil(steps / 6): no exits inside the unrolled body, so
// the compiler's wait counts are exact) converts step it, has the rows of steps it + 1 and
// it + 2 in flight and the pair indices of steps it + 3 and it + 4.  Buffer loads return
// in order, so a wait for index registers drains every OLDER load: the index load of an
// iteration is issued BEFORE its row loads and is used two iterations later -- that wait
// leaves the 16 row loads issued since in flight.
template <int NP, int N>
__device__ __forceinline__ void produce(const BlockArgs& A, const SegTab& tab, Cursor cur,
                                        int nsteps, int q6, int side, int a0, u32x4* ring,
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

  u32x4 idx[2][2];          // pair indices, two steps in flight
  int idx_rem[2];           // pairs of that step from this lane's first one on (<= 0: none)
  unsigned idx_col[2];      // byte offset of the lane's channels in the row
  unsigned raw[kRing][8][N];
  unsigned off[8];

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
  seg_update();
  const unsigned lane_idx_off = (unsigned)(8 * g) * 4u;
  int t_idx = 0;            // step the next index load is for
  auto load_idx = [&](auto buf) {
    constexpr int B = decltype(buf)::value;
    // past the range: the loads still go out (any address the descriptor covers), rem = 0
    const int pos = 32 * cur.j;
    const unsigned voff = seg_idx_off + (unsigned)pos * 4u + lane_idx_off;
    idx[B][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_idx, (int)voff, 0, 0);
    idx[B][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_idx, (int)(voff + 16u), 0, 0);
    idx_rem[B] = (t_idx < nsteps ? cur.num - pos : 0) - 8 * g;
    idx_col[B] = seg_col;
    tab.advance(cur);
    if (cur.seg != seg_id) seg_update();
    ++t_idx;
  };
  auto make_offsets = [&](auto buf) {
    constexpr int B = decltype(buf)::value;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      unsigned r = idx[B][e >> 2][e & 3];
#ifdef MSMD_WGRAD_BLOCK_DBG
      r = fold ? (r & 4095u) : r;
#endif
      const unsigned o = __umul24(r, row_bytes) + idx_col[B];   // rows < 2^24 (host check)
      off[e] = e < idx_rem[B] ? o : kOob;
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
  // prologue: rows of steps 0 and 1 in flight, indices of steps 2 and 3 loaded
  load_idx(ic<0>{});
  load_idx(ic<1>{});
  make_offsets(ic<0>{});
  load_idx(ic<0>{});          // step 2
  load_rows(ic<0>{});
  make_offsets(ic<1>{});
  __builtin_amdgcn_sched_barrier(0);
  load_idx(ic<1>{});          // step 3
  __builtin_amdgcn_sched_barrier(0);
  load_rows(ic<1>{});

  unsigned sel_lo, sel_hi;   // bf16 pairs {-1, 0} and {0, -1}
  asm volatile("v_mov_b32 %0, 0xbf80\n\tv_mov_b32 %1, 0xbf800000" : "=v"(sel_lo), "=v"(sel_hi));
  const bool wb_on = side == 0 && a0 == 0;   // (PROF builds: the wave that is timed)
  (void)wb_on;
  WB_BEGIN();
  auto iter = [&](auto slot, auto buf) {   // iteration it: slot it % 3, index buffer it % 2
    constexpr int S = decltype(slot)::value;
    make_offsets(buf);                   // step it + 2 (indices loaded at iteration it - 2)
    __builtin_amdgcn_sched_barrier(0);
    load_idx(buf);                       // step it + 4
    __builtin_amdgcn_sched_barrier(0);
    load_rows(ic<(S + 2) % kRing>{});    // step it + 2
    if (wb_on) WB_MARK(0);
    u32x4 op[N][NP];
    split_rows<NP, N>(raw[S], op, sel_lo, sel_hi);   // step it (zeros past the range)
    if (wb_on) WB_MARK(1);
    u32x4* d = dst0 + S * slot_u;
#pragma unroll
    for (int a = 0; a < N; ++a)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) d[(a * NP + pl) * 64] = op[a][pl];
    wait_lds();
    if (wb_on) WB_MARK(2);
    asm volatile("s_barrier" ::: "memory");
    if (wb_on) WB_MARK(3);
  };
  for (int it = 0; it < q6; ++it) {
    iter(ic<0>{}, ic<0>{});
    iter(ic<1>{}, ic<1>{});
    iter(ic<2>{}, ic<0>{});
    iter(ic<0>{}, ic<1>{});
    iter(ic<1>{}, ic<0>{});
    iter(ic<2>{}, ic<1>{});
  }
  asm volatile("s_barrier" ::: "memory");   // the consumers are two steps behind
  asm volatile("s_barrier" ::: "memory");
}

// ------------------------------------------------------------------ consumer --
// One consumer wave = NA x NB tiles (a quadrant of the block).  Per step: A rows in two
// halves (the second half is read from the ring under the first half's MFMAs), the next
// step's B operands and first A half under the second half's MFMAs.  Step m runs after
// barrier m + 2 (the producers finished it before barrier m + 1).
template <int NP, int NA, int NB>
__device__ __forceinline__ void consume(const BlockArgs& A, const SegTab& tab, Cursor cur,
                                        int nsteps, int q6, int ta0, int tb0,
                                        const u32x4* ring, int wg, int lane) {
  using P = Prod<NP>;
  constexpr int LO = (NA + 1) / 2, HI = NA - LO;
  const int slot_u = (A.TA + A.TB) * NP * 64;
  const u32x4* a_src = ring + ta0 * NP * 64 + lane;
  const u32x4* b_src = ring + (A.TA + tb0) * NP * 64 + lane;
  const size_t slot_elems = (size_t)A.TA * A.TB * 256;

  f32x4 acc[NA][NB];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  u32x4 bf[2][NB][NP], alo[LO][NP], ahi[HI > 0 ? HI : 1][NP];

  auto read_b = [&](auto par, int slot) {
    constexpr int Q = decltype(par)::value;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) bf[Q][b][pl] = b_src[slot * slot_u + (b * NP + pl) * 64];
  };
  auto read_alo = [&](int slot) {
#pragma unroll
    for (int a = 0; a < LO; ++a)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) alo[a][pl] = a_src[slot * slot_u + (a * NP + pl) * 64];
  };
  auto read_ahi = [&](int slot) {
#pragma unroll
    for (int a = 0; a < HI; ++a)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl)
        ahi[a][pl] = a_src[slot * slot_u + ((LO + a) * NP + pl) * 64];
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
  int m = 0;
  const bool wb_on = ta0 == 0 && tb0 == 0;   // (PROF builds: the wave that is timed)
  (void)wb_on;
  WB_BEGIN();
  // step m (ring slot S, B registers of parity Q); its alo and bf[Q] are in registers
  auto step = [&](auto slot, auto par) {
    constexpr int S = decltype(slot)::value, Q = decltype(par)::value;
    if (wb_on) WB_MARK(7);
    if (m < nsteps) {
      __builtin_amdgcn_sched_barrier(0);
      read_ahi(S);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < P::n; ++t)
#pragma unroll
        for (int a = 0; a < LO; ++a)
#pragma unroll
          for (int b = 0; b < NB; ++b)
            acc[a][b] = mfma_bf16(alo[a][P::a[t]], bf[Q][b][P::b[t]], acc[a][b]);
      __builtin_amdgcn_sched_barrier(0);
      if (wb_on) WB_MARK(4);
      // (ahi landed long ago; said here so that no wait for it is placed AFTER the reads
      // below -- the 4-bit counter could only express that by draining them too)
      wait_lds();
      // step m + 1 was complete at the last barrier (past the range: stale bytes, unused)
      read_alo((S + 1) % kRing);
      read_b(ic<Q ^ 1>{}, (S + 1) % kRing);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < P::n; ++t)
#pragma unroll
        for (int a = 0; a < HI; ++a)
#pragma unroll
          for (int b = 0; b < NB; ++b)
            acc[LO + a][b] = mfma_bf16(ahi[a][P::a[t]], bf[Q][b][P::b[t]], acc[LO + a][b]);
      __builtin_amdgcn_sched_barrier(0);
      if (wb_on) WB_MARK(5);
      // last step of its segment, or of this workgroup's range
      if (cur.j + 1 == cur.steps || m + 1 == nsteps) flush();
      tab.advance(cur);
    }
    ++m;
    if (wb_on) WB_MARK(6);
    asm volatile("s_barrier" ::: "memory");
  };
  asm volatile("s_barrier" ::: "memory");
  asm volatile("s_barrier" ::: "memory");
  read_alo(0);
  read_b(ic<0>{}, 0);
  for (int it = 0; it < q6; ++it) {
    step(ic<0>{}, ic<0>{});
    step(ic<1>{}, ic<1>{});
    step(ic<2>{}, ic<0>{});
    step(ic<0>{}, ic<1>{});
    step(ic<1>{}, ic<0>{});
    step(ic<2>{}, ic<1>{});
  }
}

// total steps of one block's segments; S = nblk * Sk; range length per workgroup
__device__ __forceinline__ int range_len(int S, int G, int min_steps) {
  int L = (S + G - 1) / G;
  return L < min_steps ? min_steps : L;
}

template <int NP>
__global__ __launch_bounds__(512) void spconv_wgrad_block_kernel(BlockArgs A) {
  __shared__ __attribute__((aligned(16))) u32x4 ring[kRing * 2 * kMaxTiles * NP * 64];
  __shared__ int sstep[kMaxKvol], snum[kMaxKvol], s_total;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid < kMaxKvol) {
    const int n = tid < A.kvol ? A.num[tid] : 0;
    snum[tid] = n;
    const int st = (n + 31) >> 5;
    sstep[tid] = st;
    const int tot = wave_sum(st);
    if (tid == 0) s_total = tot;
  }
  __syncthreads();
  const int Sk = __builtin_amdgcn_readfirstlane(s_total);
  const int nblk = A.nba * A.nbb, nseg = nblk * A.kvol;
  const int S = Sk * nblk;
  const int L = range_len(S, (int)gridDim.x, A.min_steps);
  const int wg = blockIdx.x;
  const int s0 = wg * L;
  if (s0 >= S) return;
  const int s1 = s0 + L < S ? s0 + L : S;
  const int nsteps = s1 - s0, q6 = (nsteps + 5) / 6;
  SegTab tab{sstep, snum, A.kvol, nseg};
  const Cursor cur = tab.at(s0, Sk);

  const int n0a = side_n0(A.TA), n0b = side_n0(A.TB);
  if (wave >= 4) {
    const int pw = wave - 4, side = pw >> 1, h = pw & 1;
    const int T = side ? A.TB : A.TA, n0 = side ? n0b : n0a;
    const int n = h ? T - n0 : n0, a0 = h ? n0 : 0;
    switch (n) {
      case 4: produce<NP, 4>(A, tab, cur, nsteps, q6, side, a0, ring, lane); break;
      case 3: produce<NP, 3>(A, tab, cur, nsteps, q6, side, a0, ring, lane); break;
      default: produce<NP, 2>(A, tab, cur, nsteps, q6, side, a0, ring, lane); break;
    }
  } else {
    const int wa = wave >> 1, wb = wave & 1;
    const int na = wa ? A.TA - n0a : n0a, ta0 = wa ? n0a : 0;
    const int nb = wb ? A.TB - n0b : n0b, tb0 = wb ? n0b : 0;
#define MSMD_CONSUME(NA_, NB_)                                                              \
  case NA_ * 8 + NB_:                                                                       \
    consume<NP, NA_, NB_>(A, tab, cur, nsteps, q6, ta0, tb0, ring, wg, lane);           \
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

// dW = sum of a segment's partial slots in workgroup order; fragment order -> dW layout.
__global__ __launch_bounds__(256) void wgrad_block_reduce_kernel(BlockArgs A, int G, int krsc,
                                                                 float* __restrict__ dw) {
  const int k = blockIdx.y;
  int Sk = 0, start_k = 0, steps_k = 0;
  for (int q = 0; q < A.kvol; ++q) {
    const int st = (A.num[q] + 31) >> 5;
    if (q == k) { start_k = Sk; steps_k = st; }
    Sk += st;
  }
  const int nblk = A.nba * A.nbb;
  const int L = range_len(Sk * nblk, G, A.min_steps);
  const int CA = A.TA * 16, CB = A.TB * 16;
  const int n0a = side_n0(A.TA), n1a = A.TA - n0a, n0b = side_n0(A.TB), n1b = A.TB - n0b;
  const size_t slot_elems = (size_t)A.TA * A.TB * 256;
  const int per_k = A.cin * A.cout;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < per_k; e += gridDim.x * 256) {
    const int ci = e / A.cout, co = e - ci * A.cout;
    const int ba = ci / CA, cil = ci - ba * CA, bb = co / CB, col = co - bb * CB;
    int ta, ia, tb, ib;
    if (cil < 16 * n0a) { ia = cil / n0a; ta = cil - ia * n0a; }
    else { const int c2 = cil - 16 * n0a; ia = c2 / n1a; ta = n0a + c2 - ia * n1a; }
    if (col < 16 * n0b) { ib = col / n0b; tb = col - ib * n0b; }
    else { const int c2 = col - 16 * n0b; ib = c2 / n1b; tb = n0b + c2 - ib * n1b; }
    const size_t elem = ((size_t)(ta * A.TB + tb) * 64 + (ia >> 2) * 16 + ib) * 4 + (ia & 3);
    float s = 0.f;
    if (steps_k > 0) {
      const int blk = ba * A.nbb + bb, seg = blk * A.kvol + k;
      const int gs = blk * Sk + start_k;
      const int g_lo = gs / L, g_hi = (gs + steps_k - 1) / L;
      const float* src = A.partial + (size_t)seg * slot_elems + elem;
      for (int g = g_lo; g <= g_hi; ++g) s += src[(size_t)g * slot_elems];
    }
    if (krsc) dw[((size_t)co * A.kvol + k) * A.cin + ci] = s;   // [c_out][K][c_in]
    else dw[(size_t)k * per_k + e] = s;
  }
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
    return v;
  }();
  return n;
}

}  // namespace

bool wgrad_block_supported(int c_in, int c_out, int kvol, int ld) {
  // (row indices below 2^24: the row offset is a 24-bit multiply)
  return side_blocks(c_in) > 0 && side_blocks(c_out) > 0 && kvol <= kMaxKvol &&
         ld < (1 << 24) && (double)kvol * 2.0 * ld * 4.0 < 4.0e9;
}

size_t wgrad_block_workspace_bytes(int kvol, int c_in, int c_out) {
  const int nba = side_blocks(c_in), nbb = side_blocks(c_out);
  if (!nba || !nbb) return 0;
  const size_t slot = (size_t)(c_in / nba) * (c_out / nbb) * sizeof(float);
  return align_up(slot * ((size_t)cu_count() + (size_t)nba * nbb * kvol));
}

// partials + reduction; rows * channels * 4 < 4 GiB on both sides is the caller's check
int wgrad_block(const float* in_feat, int c_in, const float* d_out, int c_out,
                const int32_t* pairs, const int32_t* num, int ld, int kvol, int np,
                float* d_weight, int krsc_out, float* ws, hipStream_t st) {
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
  static const int dbg = [] { const char* e = getenv("MSMD_WGRAD_DBG"); return e ? atoi(e) : 0; }();
  A.dbg = dbg;
  const int G = cu_count();
  if (np == 3) MSMD_LAUNCH(spconv_wgrad_block_kernel<3>, dim3(G), dim3(512), 0, st, A);
  else if (np == 2) MSMD_LAUNCH(spconv_wgrad_block_kernel<2>, dim3(G), dim3(512), 0, st, A);
  else MSMD_LAUNCH(spconv_wgrad_block_kernel<1>, dim3(G), dim3(512), 0, st, A);
  int rb = ceil_div(c_in * c_out, 256);
  if (rb > 64) rb = 64;
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
