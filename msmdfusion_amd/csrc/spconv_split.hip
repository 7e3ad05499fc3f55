// spconv_split.hip -- the sparse convolution arithmetic at bf16 MFMA rate with
// fp32-equivalent results (error-free operand splitting).
//
// The fp32 matrix instruction (v_mfma_f32_16x16x4_f32, spconv.hip) runs at 1/16
// of the bf16 rate on gfx950.  Every fp32 value is the exact sum of three bf16
// values  x = h + m + l  (h = rn_bf16(x), m = rn_bf16(x - h), l = x - h - m: the
// residuals are exact in fp32 and l has <= 8 significant bits), so
//
//   a * b = ah*bh + (ah*bm + am*bh) + (ah*bl + al*bh + am*bm)  + O(2^-24 |ab|)
//
// -- six bf16 products (each exact: 8 x 8 significant bits), accumulated in the
// MFMA's fp32 accumulator.  The dropped terms (am*bl, al*bm, al*bl) are below
// fp32's own rounding of the product, so the result is fp32-equivalent
// (measured against fp64 in tests/test_gpu_kernels.py: same error as the fmaf
// chain), at 6 x v_mfma_f32_16x16x32_bf16 (96 cycles) per 8 x
// v_mfma_f32_16x16x4_f32 (256 cycles) for the same 16x16x32 block.
// NP (number of planes) = 3 is that mode; NP = 2 keeps the three leading
// products (relative error ~2^-17, still 50x tighter than TF32, which is what
// spconv-2.x runs on Ampere by default); NP = 1 is plain bf16 operands.
//
// Data: the features stay fp32 in HBM (the module's own tensor); a gathered row
// piece is split into planes in registers, between the MFMAs of the previous
// unit (44 VALU ops per 8 channels, amortised over all of c_out).  Weights are
// split once per call while being packed into MFMA fragment order.
//
// Forward / dgrad: output-stationary pipelined implicit GEMM (see the kernel).
#include "common.hpp"

#include <stdlib.h>
#include <type_traits>

namespace msmd {
namespace {

typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

constexpr int kMaxK = 32;

int env_int2(const char* name, int dflt) {
  const char* s = getenv(name);
  return s ? atoi(s) : dflt;
}

// Split 8 fp32 values into NP bf16 planes (round-to-nearest-even at every level).
template <int NP>
__device__ __forceinline__ void split8(const float (&x)[8], u32x4 (&p)[NP]) {
  float r[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) r[t] = x[t];
#pragma unroll
  for (int pl = 0; pl < NP; ++pl) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const f32x2 v = {r[2 * t], r[2 * t + 1]};
      const bf16x2 h = __builtin_convertvector(v, bf16x2);
      p[pl][t] = __builtin_bit_cast(unsigned int, h);
      if (pl + 1 < NP) {
        const f32x2 back = __builtin_convertvector(h, f32x2);
        r[2 * t] = v[0] - back[0];      // exact (Sterbenz-style: same binade or below)
        r[2 * t + 1] = v[1] - back[1];
      }
    }
  }
}

// packed[(((k*KB + kb)*NP + p)*NT + mt)*64 + lane] (16 B) = plane p of
//   W[k][c = 32kb + 8(lane>>4) + s][d = 16mt + (lane&15)],  s = 0..7
// (zero outside c_in x c_out); flags as msmd_spconv_pack_weight.
template <int NP>
__global__ __launch_bounds__(256) void pack_weight_split_kernel(const float* __restrict__ w,
                                                                int kvol, int cin, int cout,
                                                                int flags,
                                                                u32x4* __restrict__ packed_a,
                                                                u32x4* __restrict__ packed_b) {
  // packed_b (optional): the image of the opposite transposition, same launch (forward
  // and dgrad of one conv want both)
  const int krsc = flags & 2;
  const long total_a =
      (long)kvol * (((flags & 1 ? cout : cin) + 31) / 32) * (((flags & 1 ? cin : cout) + 15) / 16) * 64;
  const long total_b =
      packed_b ? (long)kvol * (((flags & 1 ? cin : cout) + 31) / 32) * (((flags & 1 ? cout : cin) + 15) / 16) * 64
               : 0;
  for (long e2 = (long)blockIdx.x * 256 + threadIdx.x; e2 < total_a + total_b;
       e2 += (long)gridDim.x * 256) {
    const bool second = e2 >= total_a;
    const long e = second ? e2 - total_a : e2;
    const int transpose = (flags & 1) ^ (second ? 1 : 0);
    u32x4* __restrict__ packed = second ? packed_b : packed_a;
    const int ci = transpose ? cout : cin, co = transpose ? cin : cout;
    const int KB = (ci + 31) / 32, NT = (co + 15) / 16;
    const int lane = e & 63;
    long t = e >> 6;
    const int mt = t % NT;
    t /= NT;
    const int kb = t % KB;
    const int k = t / KB;
    const int d = 16 * mt + (lane & 15);
    float x[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int c = 32 * kb + 8 * (lane >> 4) + s;
      float v = 0.f;
      if (c < ci && d < co) {
        const int wi = transpose ? d : c, wo = transpose ? c : d;
        v = krsc ? w[((size_t)wo * kvol + k) * cin + wi] : w[((size_t)k * cin + wi) * cout + wo];
      }
      x[s] = v;
    }
    u32x4 p[NP];
    split8<NP>(x, p);
#pragma unroll
    for (int pl = 0; pl < NP; ++pl)
      packed[((((size_t)k * KB + kb) * NP + pl) * NT + mt) * 64 + lane] = p[pl];
  }
}

// Several weights in ONE launch: the 16 trainable convs of an LC step were 15 launches of
// 6 us (+ the gap behind each) on the feature pass's queue.  `descs` lives in device memory;
// an element of the grid-stride loop finds its weight by the running element count.
struct PackDesc {          // mirrored by msmdfusion_amd/kernels.py (ctypes)
  const float* w;
  u32x4* packed_a;
  u32x4* packed_b;         // the opposite transposition, or NULL
  long start;              // first element (16-byte unit) of this weight in the launch
  int kvol, cin, cout, flags;
};
template <int NP>
__global__ __launch_bounds__(256) void pack_weight_split_many_kernel(
    const PackDesc* __restrict__ descs, int n_desc, long total) {
  for (long e2 = (long)blockIdx.x * 256 + threadIdx.x; e2 < total; e2 += (long)gridDim.x * 256) {
    int lo = 0, hi = n_desc;                 // last descriptor with start <= e2
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (descs[mid].start <= e2) lo = mid; else hi = mid;
    }
    const PackDesc d = descs[lo];
    const long el = e2 - d.start;
    const int krsc = d.flags & 2;
    const long total_a = (long)d.kvol * (((d.flags & 1 ? d.cout : d.cin) + 31) / 32) *
                         (((d.flags & 1 ? d.cin : d.cout) + 15) / 16) * 64;
    const bool second = el >= total_a;
    const long e = second ? el - total_a : el;
    const int transpose = (d.flags & 1) ^ (second ? 1 : 0);
    u32x4* __restrict__ packed = second ? d.packed_b : d.packed_a;
    const int ci = transpose ? d.cout : d.cin, co = transpose ? d.cin : d.cout;
    const int KB = (ci + 31) / 32, NT = (co + 15) / 16;
    const int lane = e & 63;
    long t = e >> 6;
    const int mt = t % NT;
    t /= NT;
    const int kb = t % KB;
    const int k = t / KB;
    const int dd = 16 * mt + (lane & 15);
    float x[8];
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) {
      const int c = 32 * kb + 8 * (lane >> 4) + s8;
      float v = 0.f;
      if (c < ci && dd < co) {
        const int wi = transpose ? dd : c, wo = transpose ? c : dd;
        v = krsc ? d.w[((size_t)wo * d.kvol + k) * d.cin + wi]
                 : d.w[((size_t)k * d.cin + wi) * d.cout + wo];
      }
      x[s8] = v;
    }
    u32x4 p[NP];
    split8<NP>(x, p);
#pragma unroll
    for (int pl = 0; pl < NP; ++pl)
      packed[((((size_t)k * KB + kb) * NP + pl) * NT + mt) * 64 + lane] = p[pl];
  }
}

// The products kept for NP planes, as (weight plane, activation plane).
template <int NP>
struct Products;
template <>
struct Products<1> {
  static constexpr int n = 1;
  static constexpr int a[1] = {0};
  static constexpr int b[1] = {0};
};
template <>
struct Products<2> {
  static constexpr int n = 3;
  static constexpr int a[3] = {1, 0, 0};
  static constexpr int b[3] = {0, 1, 0};
};
template <>
struct Products<3> {  // smallest terms first
  static constexpr int n = 6;
  static constexpr int a[6] = {2, 0, 1, 1, 0, 0};
  static constexpr int b[6] = {0, 2, 1, 0, 1, 0};
};

__device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

typedef int i32x4 __attribute__((ext_vector_type(4)));

// Raw buffer descriptor (stride 0, num_records bytes): loads whose byte offset
// is out of range return 0 without touching memory -- "no neighbour" costs no
// traffic and needs no predication or zeroing.
__device__ __forceinline__ i32x4 make_rsrc(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
  r[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}
constexpr unsigned kOobOffset = 0xffffff00u;

// The gathers are issued from inline asm: hipcc drains the whole VM queue
// (vmcnt(0)) at the first use of an ordinary load's result while an LDS-DMA is
// in flight, and at every __syncthreads(); hidden from it, the queue is
// counted by hand (all waits below are "allow the N newest ops").
// One gathered row piece = 8 fp32 channels = 2 x 16 bytes.  (s_nop: the
// descriptor may have been written by v_readfirstlane and the hazard recognizer
// does not look inside asm -- VALU-written SGPR -> VMEM needs 5 wait states.)
__device__ __forceinline__ void gather_row8(u32x4 (&raw)[2], unsigned off, i32x4 rs) {
  asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, 0 offen"
               : "=v"(raw[0])
               : "v"(off), "s"(rs));
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:16"
               : "=v"(raw[1])
               : "v"(off), "s"(rs));
}
// s_waitcnt vmcnt(N) that the raw registers depend on (so nothing reading them
// can be scheduled above it).
template <int N>
__device__ __forceinline__ void wait_rows(u32x4 (&raw)[2][2]) {
  asm volatile("s_waitcnt vmcnt(%4)"
               : "+v"(raw[0][0]), "+v"(raw[0][1]), "+v"(raw[1][0]), "+v"(raw[1][1])
               : "n"(N));
}
// 4 fp32 (one 16-byte piece) -> dwords [2h, 2h+1] of each bf16 plane.
// Scalar arithmetic on purpose: beside MFMAs a v_pk_add_f32 costs ~13 cycles more than the
// two v_sub_f32 it replaces (MI355X_MICROARCH.md, "price of one filler") -- with the
// residuals as f32x2 vectors this conversion added ~780 cycles to a unit's 1536 MFMA
// cycles (the file is built with -fno-slp-vectorize so the pairs are not re-packed).
template <int NP>
__device__ __forceinline__ void split_quarter(const u32x4& piece, int h, u32x4 (&planes)[NP]) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    // (copy the elements out first: __builtin_bit_cast applied directly to a vector
    // element lvalue reads element 0 whatever the index -- clang 20)
    const unsigned e0 = piece[2 * t], e1 = piece[2 * t + 1];
    float v0 = __builtin_bit_cast(float, e0), v1 = __builtin_bit_cast(float, e1);
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) {
      const f32x2 v = {v0, v1};
      const unsigned hi = __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2));
      planes[pl][2 * h + t] = hi;
      if (pl + 1 < NP) {   // exact residuals
        v0 = v0 - __builtin_bit_cast(float, hi << 16);
        v1 = v1 - __builtin_bit_cast(float, hi & 0xffff0000u);
      }
    }
  }
}

// Phase timing of one wave (build with `make PROF=1`; never in the shipped library):
// s_memtime deltas of wave 1 of the first 8 workgroups, summed per phase.
#ifdef MSMD_KERNEL_PROF
__device__ unsigned long long g_kprof[16];
// per-tile trace: {HW_ID, XCC_ID, block, tile, t_start, t_end (100 MHz wall clock), items, -}
constexpr int kTraceCap = 16384;
__device__ unsigned long long g_ktrace[kTraceCap * 8];
__device__ unsigned int g_ktrace_n;
#define KP_BEGIN() unsigned long long kp_t = __builtin_amdgcn_s_memtime()
#define KP_MARK(i)                                                          \
  {                                                                         \
    const unsigned long long kp_n = __builtin_amdgcn_s_memtime();           \
    if (lane == 0 && wave == 1 && blockIdx.x < 8) atomicAdd(&g_kprof[i], kp_n - kp_t); \
    kp_t = kp_n;                                                            \
  }
#else
#define KP_BEGIN()
#define KP_MARK(i)
#endif

// Number of entries of the ascending array a[0..n) that are <= x, by 64-ary search:
// the whole wave loads 64 probes at a time (2-3 dependent loads for n <= 8192 instead of
// 13 for a binary search).  Same result in every lane.
// (the array searched is a[idx] + c0 * idx: stream-K's per-tile overhead term)
__device__ __forceinline__ int count_le(const int32_t* __restrict__ a, int n, int x, int c0,
                                        int lane) {
  int lo = 0, hi = n;   // the answer (first index with value > x) lies in [lo, hi]
  while (hi - lo > 64) {
    const int step = (hi - lo + 63) >> 6;
    const int idx = lo + lane * step;
    const int v = idx < hi ? a[idx] + c0 * idx : 0x7fffffff;
    const int c = __builtin_popcountll(__ballot(v <= x));   // probes 0 .. c-1 are <= x
    if (c == 0) {
      hi = lo;          // a[lo] > x already
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
constexpr int kSkMinRanks = 64;  // stream-K: smallest segment (cost units) worth a workgroup
// Cost of one (tile, active offset) in stream-K's work sequence: sk_c1() + the number of the
// tile's 32-row groups (= waves) with a row connected through the offset.  tools/ktrace.py
// shows the workgroups of the dense tiles at 2.3 us per item and those of the light-mask
// tiles at 1.5, but weighting by the busy waves does not pay: C1 = 0 / 1 / 3 / 6 / 12 gave
// 298 / 278 / 268 / 261 / 263 us on the 128->128 layer -- the per-item time follows what
// the whole chip is doing at that moment (everybody starts in dense tiles), not the tile.
// 12 keeps the term as a tie-breaker.
constexpr int kSkC1Default = 12;
inline int sk_c1() { return kSkC1Default; }

// ------------------------------------------------------- forward / dgrad --
// One workgroup = 4 waves x 32 output rows (two 16-row MFMA groups per wave) x
// all of c_out; persistent, drawing 128-row tiles from a global counter.
//
// Work decomposition of a tile: UNITS = (active kernel offset k, 32-channel
// k-block kb) in ascending order, enumerated on the scalar unit from the tile's
// 27-bit offset mask (no LDS lists, no divisions); an ITEM = UB consecutive
// units = one weight buffer fill + one barrier (UB * NT = 8: 24 KiB at NP = 3).
//
// Pipeline (everything below overlaps the MFMAs of the current unit g):
//   * weights of item i+1: LDS-DMA (global_load_lds, lane-linear packed image);
//   * rows of unit g+2: 16-byte buffer loads of the fp32 features (the module's
//     own tensor: no split copy in HBM), out-of-range offset for "no neighbour"
//     (returns 0, no traffic), from row indices fetched from LDS a unit earlier;
//   * rows of unit g+1: fp32 -> bf16 planes in registers (cvt_pk / pk_add),
//     interleaved between the MFMAs;
//   * weight fragments of the next 32 output channels: LDS -> registers,
//     double-buffered, issued a third of the way into the current MFMA block;
//   * the NEXT tile's slice of the neighbour table + its output rows: LDS-DMA
//     into the other table buffer during item 1 (tile id from an atomic issued
//     at tile start) -- a tile switch costs one barrier, not 3 round trips.
// All barriers are raw s_barrier and all VM waits are counted by hand:
// __syncthreads() would drain the queue at every item.
// `nbr` must be in TILE order when `order` is given: column p of the table
// belongs to output row order[p] (msmd_rulebook_permute_cols), so a tile's
// slice is 512 contiguous bytes per offset.
// WV = waves per workgroup: 4 (128-row tiles, two workgroups per CU) or 8 (256-row tiles,
// one workgroup per CU: the same 2 waves per SIMD, but ONE weight stream per CU instead of
// two -- at ~17 B/clk the CU's memory pipeline could not feed two 24-KiB-per-item weight
// streams plus the gathers at MFMA rate; stream-K removed the reason tiles had to be small).
// PP (round 5): the PING-PONG schedule of the 8-wave / 256-row form.  What bounds the 4-wave
// kernel is the CU's vector-memory rate (~17-18 B/clk: two workgroups x one 18-24 KiB weight
// image per item + eight waves x 4 KiB of gathered rows per ~3000 matrix cycles = 26-30 B/clk
// wanted); one 8-wave workgroup per CU shares ONE weight stream (-40 % bytes per MFMA), but
// with a single barrier per item (round 2, round 4) the two waves of every SIMD meet at it,
// issue their memory work together and then queue their MFMA phases behind each other on the
// one matrix pipe.  Here the workgroup is two 4-wave groups -- rows 0-127 (waves 0-3) and rows
// 128-255 (waves 4-7), one wave of each per SIMD -- half an item apart: an item is a LOAD
// segment (weight DMA of the next item, gathers of the next unit, wait for this unit's rows,
// fp32 -> bf16 planes) and a MULTIPLY segment (the unit's MFMAs, weight fragments from LDS),
// a barrier after each, and group B runs one segment behind group A: at any moment one wave
// of a SIMD feeds the matrix pipe while its partner uses the vector-memory and VALU pipes.
// Three weight buffers: the image of item g is read in segments 2g+1 (A) and 2g+2 (B), so its
// buffer is free for item g+3's DMA from segment 2g+3 on (A issues it in 2g+4, B in 2g+5).
// TB = table buffers: 2 (the next tile's slice staged under the current tile's item 1), or 1
// for the 12-tile instantiation, whose three 36 KiB weight buffers leave room for one table
// only -- the next slice is staged at the tile switch (one exposed load per tile of ~100+
// items).
template <int NT, int UB, int NP, int WV, int NB = 2, bool PP = false, int TB = 2>
__global__ __launch_bounds__(WV * 64, WV == 4 ? 2 : 1) void spconv_fwd_split_kernel(
    const float* __restrict__ in, int n_in, int cin, const u32x4* __restrict__ wp,
    const int32_t* __restrict__ nbr, int ld, int n_out, int kvol, int flip,
    const int32_t* __restrict__ order, int* __restrict__ tile_counter, float* __restrict__ out,
    int ldo, int cout, int nt_total, int mt0, f32x4* __restrict__ scratch,
    int* __restrict__ flags, const int32_t* __restrict__ tile_start, int sk_c0, int sk_c1v,
    int dbg, float* __restrict__ bn_part, int n_ranges) {
  // Scheduling.  Without `tile_start`: persistent workgroups draw whole 128-row tiles from
  // a global counter (tiles arrive heaviest first: LPT list scheduling), every tile is one
  // unit and the result does not depend on the tiling order at all.  A tile whose rows are
  // connected through all 27 offsets is ~108 items of work -- about a workgroup slot's fair
  // share of the whole launch when there are fewer than two tiles per slot: the slots that
  // draw a second tile set the launch time and half the chip idles (tools/ktrace.py).
  // r01 cut the heaviest tiles into two units; stream-K below replaces that.
  // `out` points at this pass's first output channel (tile mt0 of nt_total in the
  // packed weights), rows are ldo floats apart, `cout` channels are stored.
  //
  // STREAM-K scheduling (`tile_start` given).  The launch's
  // work is the sequence of (row tile, active offset) RANKS, tile after tile in tiling
  // order; tile_start[t] = rank at which tile t begins (prefix sum of max(|offset mask of
  // the tile|, 1): msmd_rulebook_tile_prefix), tile_start[T] = W ranks in all.  Workgroup
  // number g (its ticket) takes the contiguous range [g S, (g+1) S), S = ceil(W / grid)
  // -- the same share for everybody, so nobody draws a second 108-item tile while half
  // the chip idles (r01: 34 % of the slot-time idle on the 128-channel layers) and no
  // heaviest-first order is needed.  A range cuts through tiles: the workgroup that holds
  // a tile's LAST rank owns it and stores its rows; every other workgroup touching the
  // tile leaves its accumulators in scratch[its ticket] and raises flags[its ticket].
  // A workgroup walks its range from the HIGHEST tile down: the one tile it can only
  // contribute to is its first piece of work (done long before the owner -- the next
  // ticket -- reaches that tile at the END of its own range), and an owner only ever
  // waits for LOWER tickets, which are resident and running by construction: no
  // deadlock whatever else occupies the chip.  Contributions are added in ticket order:
  // deterministic.  grid - 1 exchanges of one tile's accumulators per launch (32 MB at
  // 512 x 128 channels) instead of one per split tile (157 MB).
  // A tile visit also has a fixed cost (table staging, pipeline fill, the stores): each
  // tile is charged `sk_c0` extra units in FRONT of its ranks, i.e. tile t occupies
  // [tile_start[t] + c0 t, tile_start[t+1] + c0 (t+1)) of the sequence and its ranks start
  // c0 into that.  Without it the segments of the light-mask tiles (few ranks each, many
  // tiles) took 4x as long as those of the dense ones on the 32-channel layers.  A range
  // that only touches a tile's overhead zone does not visit the tile.
  constexpr int R = 2, kRows = WV * R * 16, kRowShift = WV == 4 ? 7 : 8;
  constexpr int kUnitU = NP * NT * 64;  // 16-byte units of one unit's weights (in LDS)
  constexpr int kWU = UB * kUnitU;
  constexpr int kGr = R * 2;            // gather loads per unit per lane
  constexpr int kPw = (NP * NT + WV - 1) / WV;  // weight DMA ops per unit per wave
  constexpr int kWp = UB * kPw;           // ... per item per wave
  constexpr int NS = NT / 2;            // fragment steps (pairs of 16-channel tiles) per unit
  static_assert(NT % 2 == 0, "pairs of output tiles");
  static_assert(!PP || (WV == 8 && NB == 3 && UB == 1), "ping-pong: 8 waves, 3 buffers, 1 unit");
  using P = Products<NP>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u32x4* wl = (u32x4*)smem;                       // [NB][kWU]
  int* nbt = (int*)(wl + NB * kWU);               // [2][kvol + 1][kRows]; row kvol = output rows
  const int tstride = (kvol + 1) * kRows;
  int* ctl = nbt + TB * tstride;                  // [0],[1]: offset masks; [2]: next tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, q = lane >> 4;
  const int kbt = (cin + 31) >> 5;      // the last k-block may be partial (c_in % 8 == 0)
  const int n_tiles = (n_out + kRows - 1) / kRows;
  // every workgroup draws 1 + (tiles it processed) tickets: the draw that returns
  // this value is the last one of the launch and puts the counter back to 0
  // (stream-K: one ticket per workgroup, the last one is number grid - 1)
  const int last_ticket = n_tiles + (int)gridDim.x - 1;   // (dynamic tile scheduler)
  int lr[R];
#pragma unroll
  for (int r = 0; r < R; ++r) lr[r] = (wave * R + r) * 16 + j;
  const i32x4 rs = make_rsrc(in, (unsigned)((size_t)n_in * cin * 4));
  const unsigned row_bytes = (unsigned)cin * 4u;

  // table slice of tile T -> buffer b, by LDS-DMA (4 bytes per lane).  Positions
  // past n_out are clamped: their results are computed and dropped.
  auto stage_table = [&](int T, int b) {
    int* dst = nbt + (TB == 1 ? 0 : b) * tstride;
    const int n_e = (kvol + (order ? 1 : 0)) * kRows;
    for (int e0 = wave * 64; e0 < n_e; e0 += WV * 64) {
      const int e = e0 + lane, k = e >> kRowShift;
      int p = T * kRows + (e & (kRows - 1));
      p = p < n_out ? p : n_out - 1;
      const int32_t* src = k < kvol ? nbr + (size_t)k * ld + p : order + p;
      __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(dst + e0), 4, 0, 0);
    }
  };

  int tb = 0;
  const bool sk = tile_start != nullptr;
  const int n_row_tiles = n_tiles;
  const int tile_lim = sk ? 0x40000000 : n_tiles;   // "no next tile" from here up
  // stream-K: `n_ranges` equal ranges of the rank sequence for gridDim.x persistent workgroups
  // (n_ranges >= gridDim.x).  A workgroup draws a range, walks it, draws the next: the ones
  // whose ranges ran fast (light-mask tiles; one workgroup left alone on its CU runs an item
  // in 1.5 us, two sharing a CU need 2.2 us each: tools/ktrace.py) take over work instead of
  // idling while the dense ranges finish.  Tickets are drawn in order and a ticket's holder is
  // running, so the exchange argument below is unchanged.
  int sk_S = 1, sk_nr = 0, sk_seg = 0, sk_g0 = 0, sk_g1 = 0, sk_lo_tile = 0;
  int tile = 0;
  for (;;) {      // stream-K: one range per pass; otherwise a single pass
  // the tile prefix goes to LDS in one coalesced pass (the weight buffers are idle until the
  // first item of a range), under the ticket's atomic: the two 64-ary searches below then cost
  // LDS reads instead of 4-6 dependent trips to L2 at the head of every range
  const int32_t* ts_search = tile_start;
  if (sk && n_tiles + 1 <= (int)(NB * kWU * 4)) {
    int* ts_lds = (int*)wl;
    for (int i = tid; i <= n_tiles; i += WV * 64) ts_lds[i] = tile_start[i];
    ts_search = ts_lds;
  }
  if (tid == 0) {
    ctl[0] = 0;
    ctl[1] = 0;
    ctl[2] = atomicAdd(tile_counter, 1);
  }
  __syncthreads();   // nothing in flight: the fence costs nothing here
  tile = __builtin_amdgcn_readfirstlane(ctl[2]);
  if (sk) {   // the ticket is a range of the rank sequence
    sk_seg = tile;
    const int W = ts_search[n_row_tiles] + sk_c0 * n_row_tiles;
    sk_S = (W + n_ranges - 1) / n_ranges;
    sk_S = sk_S < sk_c0 + kSkMinRanks ? sk_c0 + kSkMinRanks : sk_S;
    sk_nr = (W + sk_S - 1) / sk_S;              // ranges that hold work
    // every workgroup draws until a ticket is past the last range: sk_nr + gridDim.x draws in
    // all, and the one that returns the last value puts the counter back to 0 -- a workgroup
    // that becomes resident late (another kernel held the CU) can draw it right here
    if (tid == 0 && sk_seg == sk_nr + (int)gridDim.x - 1) *tile_counter = 0;
    if (sk_seg >= sk_nr) return;
    sk_g0 = sk_seg * sk_S;
    sk_g1 = sk_g0 + sk_S < W ? sk_g0 + sk_S : W;
    // tiles of the first and the last rank (64-ary searches, uniform across the block)
    sk_lo_tile = count_le(ts_search, n_row_tiles, sk_g0, sk_c0, lane) - 1;
    tile = count_le(ts_search, n_row_tiles, sk_g1 - 1, sk_c0, lane) - 1;
    // a range that ends inside its highest tile's overhead zone does not visit that tile
    // (the lowest tile's piece is never empty: the range starts before the tile's end)
    if (sk_g1 <= ts_search[tile] + sk_c0 * (tile + 1)) --tile;
    if (tile < sk_lo_tile) {      // nothing but overhead zones: next range
      __syncthreads();
      continue;
    }
  } else {
    if (tid == 0 && tile == last_ticket) *tile_counter = 0;
    if (tile >= n_tiles) return;
  }
  tb = 0;
  stage_table(tile, 0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  for (;;) {
    int nxt_v = 0;
    if (tid == 0)   // consumed during item 1
      nxt_v = sk ? (tile > sk_lo_tile ? tile - 1 : tile_lim) : atomicAdd(tile_counter, 1);
    int sk_ts = 0, sk_tw = 0;
    if (sk) {
      const int t0s = tile_start[tile];
      sk_tw = tile_start[tile + 1] - t0s;
      sk_ts = t0s + sk_c0 * (tile + 1);        // where the tile's ranks start
    }
    const int* tab = nbt + (TB == 1 ? 0 : tb) * tstride;
    int* cst = ctl + 4 + tb * 32;     // stream-K cost of every offset of this tile (0: none)
    {  // offsets any row of this tile is connected through, and how many waves each keeps busy
      unsigned m = 0;
      for (int k = wave; k < kvol; k += WV) {
        int nact = 0;
#pragma unroll
        for (int c = 0; c < kRows / 64; ++c) {
          const unsigned long long b = __ballot(tab[k * kRows + 64 * c + lane] >= 0);
          nact += ((unsigned)b != 0u) + ((unsigned)(b >> 32) != 0u);
        }
        if (nact) m |= 1u << k;
        if (lane == 0) cst[k] = nact ? sk_c1v + nact : 0;
      }
      if (lane == 0 && m) atomicOr((unsigned*)&ctl[tb], m);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (the tile-id atomic too)
    __builtin_amdgcn_s_barrier();
    unsigned mask = __builtin_amdgcn_readfirstlane(ctl[tb]);
    // stream-K: the offsets whose cost span STARTS in [sk_lo, sk_hi) of the tile's sk_tw
    // units are this workgroup's; lane k (and k + 32) holds offset k's cost and start
    const int sk_lo = sk ? (sk_g0 > sk_ts ? sk_g0 - sk_ts : 0) : 0;
    const int sk_hi = sk ? (sk_g1 - sk_ts < sk_tw ? sk_g1 - sk_ts : sk_tw) : 0;
    int sk_c = 0, sk_p = 0;
    bool sk_owner = true;
    if (sk) {
      for (int k = 0; k < kvol; ++k) {
        const int c = cst[k];               // (uniform address: one broadcast read)
        sk_p += k < (lane & 31) ? c : 0;
        sk_c = k == (lane & 31) ? c : sk_c;
      }
      // the workgroup that takes the tile's last offset owns the tile (an empty tile: the
      // one whose range holds its single unit)
      if (mask) {
        const int p_last = __builtin_amdgcn_readlane(sk_p, 31 - __builtin_clz(mask));
        sk_owner = p_last >= sk_lo && p_last < sk_hi;
      } else {
        sk_owner = sk_lo == 0 && sk_hi > 0;
      }
      mask = (unsigned)__ballot(sk_c > 0 && sk_p >= sk_lo && sk_p < sk_hi);
    }
    const int kb0 = 0, kb1 = kbt;
    const int n_units = __builtin_popcount(mask) * (kb1 - kb0);
    const int n_items = (n_units + UB - 1) / UB;

    f32x4 acc[R][NT];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[r][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // unit cursors (scalar): remaining-offset mask + k-block
    unsigned mw = mask, mg = mask, ms = mask;
    int kbw = kb0, kbg = kb0, kbs = kb0;
#define MSMD_ADV(M, KB)          \
  if (++(KB) == kb1) {           \
    (KB) = kb0;                  \
    (M) &= (M)-1;                \
  }
    int s_next[R];
    auto load_src = [&]() {  // row indices of cursor `s`, then advance it
      const int k = ms ? __builtin_ctz(ms) : 0;
#pragma unroll
      for (int r = 0; r < R; ++r) s_next[r] = ms ? tab[k * kRows + lr[r]] : -1;
      MSMD_ADV(ms, kbs);
    };
    auto issue_g = [&](u32x4 (&raw)[R][2], int& valid) {  // cursor `g`, rows s_next
      valid = -1;
      const int chan0 = kbg * 32 + q * 8;   // this lane's 8 channels; past c_in -> zeros
      const unsigned col = (unsigned)chan0 * 4u;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int src = s_next[r];
        valid = src > valid ? src : valid;
        // (dbg & 8, experiments only: fold the gathers onto 4096 rows -- all L2 hits)
        // (24-bit multiply, full rate: rows < 2^24 and row bytes < 2^24 -- the entry point's
        // range check; the plain product compiled to a quarter-rate 64-bit multiply-add)
        const unsigned rowb = __umul24((unsigned)((dbg & 8) ? (src & 4095) : src), row_bytes) + col;
        const unsigned off = (src < 0 || chan0 >= cin || (dbg & 2)) ? kOobOffset : rowb;
        gather_row8(raw[r], off, rs);
      }
      MSMD_ADV(mg, kbg);
    };
    auto issue_w = [&](int it) {  // UB units at cursor `w` -> buffer it&1
      u32x4* wb = wl + (NB == 2 ? (it & 1) : it % NB) * kWU;
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int k = mw ? __builtin_ctz(mw) : 0;
        const int kw = flip ? kvol - 1 - k : k;
        const u32x4* g = wp + ((size_t)kw * kbt + kbw) * (NP * nt_total * 64);
        // every wave issues the same number of ops (a short last round repeats
        // piece 0: same bytes to the same place) so the queue counts are static
#pragma unroll
        for (int pp = 0; pp < kPw; ++pp) {
          int piece = wave + WV * pp;
          if ((NP * NT) % WV != 0 && piece >= NP * NT) piece = 0;
          const int pl = piece / NT;
          int tile = mt0 + piece - pl * NT;          // tiles past the packed image repeat
          tile = tile < nt_total ? tile : nt_total - 1;   // the last one (never stored)
          __builtin_amdgcn_global_load_lds((glb_void*)(g + (pl * nt_total + tile) * 64 + lane),
                                           (lds_void*)(wb + u * kUnitU + piece * 64), 16, 0, 0);
        }
        MSMD_ADV(mw, kbw);
      }
    };
    // pieces [p0, p1) of the weight image of item `it` (UB == 1; the cursor moves with the
    // last piece): the ping-pong form may issue some of them from the multiply segment
    auto issue_w_part = [&](int it, int p0, int p1) {
      u32x4* wb = wl + (NB == 2 ? (it & 1) : it % NB) * kWU;
      const int k = mw ? __builtin_ctz(mw) : 0;
      const int kw = flip ? kvol - 1 - k : k;
      const u32x4* g = wp + ((size_t)kw * kbt + kbw) * (NP * nt_total * 64);
#pragma unroll
      for (int pp = p0; pp < p1; ++pp) {
        int piece = wave + WV * pp;
        if ((NP * NT) % WV != 0 && piece >= NP * NT) piece = 0;
        const int pl = piece / NT;
        int tile = mt0 + piece - pl * NT;
        tile = tile < nt_total ? tile : nt_total - 1;
        __builtin_amdgcn_global_load_lds((glb_void*)(g + (pl * nt_total + tile) * 64 + lane),
                                         (lds_void*)(wb + piece * 64), 16, 0, 0);
      }
      if (p1 == kPw) MSMD_ADV(mw, kbw);
    };
    auto split_all = [&](const u32x4 (&raw)[R][2], u32x4 (&cv)[R][NP]) {
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int h = 0; h < 2; ++h) split_quarter<NP>(raw[r][h], h, cv[r]);
    };
    // Multiply unit (it, u) from its converted rows `b`; meanwhile convert the
    // next unit's rows raw_n -> b_n, a quarter (4 channels of one row group) per
    // fragment step, interleaved with the MFMAs.
    // `during(st)`: memory-op issue work the caller wants done UNDER this unit's MFMAs
    // (called once per fragment step, between two MFMA groups)
    auto compute = [&](int it, int u, const u32x4 (&b)[R][NP], int valid,
                       const u32x4 (&raw_n)[R][2], u32x4 (&b_n)[R][NP], auto&& during) {
      if (!__any(valid >= 0) || (dbg & 4)) {
#pragma unroll
        for (int st = 0; st < NS; ++st) during(st);
        split_all(raw_n, b_n);
        return;
      }
      const u32x4* wb = wl + (NB == 2 ? (it & 1) : it % NB) * kWU + u * kUnitU + lane;
      u32x4 a[2][2][NP];
#pragma unroll
      for (int nn = 0; nn < 2; ++nn)
#pragma unroll
        for (int p = 0; p < NP; ++p) a[0][nn][p] = wb[(p * NT + nn) * 64];
      // The next step's fragments are requested a third of the way into this
      // step's MFMAs: when the first MFMA of the next step waits for them
      // (lgkmcnt(0): nothing newer is outstanding) they have had 16 MFMAs to land.
      constexpr int kHead = P::n >= 3 ? P::n / 3 : 0;
#pragma unroll
      for (int st = 0; st < NS; ++st) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < kHead; ++t)
#pragma unroll
          for (int nn = 0; nn < 2; ++nn)
#pragma unroll
            for (int r = 0; r < R; ++r)
              acc[r][2 * st + nn] =
                  mfma_bf16(a[st & 1][nn][P::a[t]], b[r][P::b[t]], acc[r][2 * st + nn]);
        __builtin_amdgcn_sched_barrier(0);
        if (st + 1 < NS) {
#pragma unroll
          for (int nn = 0; nn < 2; ++nn)
#pragma unroll
            for (int p = 0; p < NP; ++p)
              a[(st + 1) & 1][nn][p] = wb[(p * NT + 2 * (st + 1) + nn) * 64];
        }
        during(st);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int qi = st * 4 / NS; qi < (st + 1) * 4 / NS; ++qi)
          split_quarter<NP>(raw_n[qi >> 1][qi & 1], qi & 1, b_n[qi >> 1]);
#pragma unroll
        for (int t = kHead; t < P::n; ++t)
#pragma unroll
          for (int nn = 0; nn < 2; ++nn)
#pragma unroll
            for (int r = 0; r < R; ++r)
              acc[r][2 * st + nn] =
                  mfma_bf16(a[st & 1][nn][P::a[t]], b[r][P::b[t]], acc[r][2 * st + nn]);
        // one MFMA, then up to three conversion VALU ops in its shadow
#pragma unroll
        for (int i = 0; i < (P::n - kHead) * 2 * R; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    };

    u32x4 raw0[R][2], raw1[R][2];     // fp32 rows in flight (units g+1, g+2)
    int vr0 = -1, vr1 = -1;
    bool staged = false;
    int nxt = tile_lim;
    KP_BEGIN();
#ifdef MSMD_KERNEL_PROF
    const unsigned long long kt0 = wall_clock64();
#endif
    if constexpr (PP) {
      // ---- ping-pong item loop (see the kernel's header) ----
      u32x4 cv[R][NP];
      const bool grp_b = wave >= WV / 2;
      // the unit's MFMAs alone (its rows' planes `b` were made in the load segment)
      auto multiply = [&](int it, const u32x4 (&b)[R][NP], int valid) {
        if (!__any(valid >= 0) || (dbg & 4)) return;
        const u32x4* wb = wl + (it % NB) * kWU + lane;
        // fragments double-buffered per 16-channel TILE (12 registers a buffer; per PAIR of
        // tiles, as the 4-wave kernel does, costs 24 more and spills the 12-tile instantiation):
        // the next tile's three 16-byte reads are requested after the first third of this
        // tile's MFMAs and have the other two thirds (128 cycles) to land
        u32x4 a[2][NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) a[0][p] = wb[(p * NT) * 64];
        constexpr int kHead = P::n >= 3 ? P::n / 3 : 0;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < kHead; ++t)
#pragma unroll
            for (int r = 0; r < R; ++r)
              acc[r][n] = mfma_bf16(a[n & 1][P::a[t]], b[r][P::b[t]], acc[r][n]);
          __builtin_amdgcn_sched_barrier(0);
          if (n + 1 < NT) {
#pragma unroll
            for (int p = 0; p < NP; ++p) a[(n + 1) & 1][p] = wb[(p * NT + n + 1) * 64];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = kHead; t < P::n; ++t)
#pragma unroll
            for (int r = 0; r < R; ++r)
              acc[r][n] = mfma_bf16(a[n & 1][P::a[t]], b[r][P::b[t]], acc[r][n]);
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      // kWM of an item's kPw weight DMA pieces are issued at the head of the MULTIPLY segment
      // instead of in the load segment (experiment: -DMSMD_PP_WM=n): the load segment is the
      // longer one for 6 and 8 column tiles.  The pieces then are the newest operations when
      // the segment ends: its wait drains the queue.
#ifndef MSMD_PP_WM
#define MSMD_PP_WM 0
#endif
      constexpr int kWM = (MSMD_PP_WM) < kPw ? (MSMD_PP_WM) : kPw - 1;
      auto pin_planes = [&](u32x4 (&c)[R][NP]) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int p = 0; p < NP; ++p) asm volatile("" : "+v"(c[r][p]));
      };
      // prologue: weights of item 0, rows of unit 0, indices of unit 1.  Every wave has its
      // own pieces of the image landed before its first barrier (the 4 newest ops = the rows)
      issue_w(0);
      load_src();
      issue_g(raw0, vr0);
      load_src();
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kGr) : "memory");
      if (grp_b) __builtin_amdgcn_s_barrier();      // group B runs one segment behind
      // One item: LOAD segment | barrier | MULTIPLY segment | barrier.
      // Queue at the counted wait (oldest first): rows(g) [4] | weights(g+1) [kPw] | rows(g+1)
      // [4]: "all but the newest kPw + 4" = rows(g); after the MFMAs "all but the newest 4"
      // = this wave's pieces of weights(g+1), a multiply segment after their issue.
#define MSMD_PP_ITEM(G, RAW_C, V_C, RAW_N, V_N)                                          \
  {                                                                                     \
    if ((G) == 0 && tid == 0) {                                                         \
      ctl[2] = nxt_v;                                                                   \
      ctl[tb ^ 1] = 0;                                                                  \
      if (!sk && nxt_v == last_ticket) *tile_counter = 0;                               \
    }                                                                                   \
    if (TB == 2 && (G) == 1) { /* (both groups are past the barrier after A's first load) */ \
      nxt = __builtin_amdgcn_readfirstlane(ctl[2]);                                     \
      if (nxt < tile_lim) stage_table(nxt, tb ^ 1);                                     \
      staged = true;                                                                    \
    }                                                                                   \
    KP_MARK(6);                                                                         \
    issue_w_part((G) + 1, 0, kPw - kWM);                                                \
    KP_MARK(2);                                                                         \
    issue_g(RAW_N, V_N);                                                                \
    load_src();                                                                         \
    KP_MARK(3);                                                                         \
    wait_rows<kGr + kPw - kWM>(RAW_C);                                                  \
    KP_MARK(4);                                                                         \
    split_all(RAW_C, cv);                                                               \
    /* (the conversion belongs to THIS segment: left alone the optimiser sinks it below \
       the barrier into multiply()'s branch, where it competes with the MFMAs) */       \
    pin_planes(cv);                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                  \
    KP_MARK(8);                                                                         \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                  \
    __builtin_amdgcn_s_barrier();                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                  \
    KP_MARK(1);                                                                         \
    if (kWM > 0) issue_w_part((G) + 1, kPw - kWM, kPw);                                 \
    if (dbg & 32) __builtin_amdgcn_s_setprio(1);                                        \
    multiply((G), cv, V_C);                                                             \
    if (dbg & 32) __builtin_amdgcn_s_setprio(0);                                        \
    KP_MARK(5);                                                                         \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kWM > 0 ? 0 : kGr) : "memory");            \
    KP_MARK(0);                                                                         \
    __builtin_amdgcn_s_barrier();                                                       \
    KP_MARK(9);                                                                         \
  }
      for (int it = 0; it < n_items; it += 2) {
        MSMD_PP_ITEM(it, raw0, vr0, raw1, vr1);
        if (it + 1 < n_items) MSMD_PP_ITEM(it + 1, raw1, vr1, raw0, vr0);
      }
#undef MSMD_PP_ITEM
      if (!grp_b) __builtin_amdgcn_s_barrier();     // group A makes up B's head start
    } else {
    u32x4 cv0[R][NP], cv1[R][NP];     // bf16 planes (units g, g+1)
    // ---- prologue: weights of item 0, rows of units 0 and 1, indices of unit 2,
    // planes of unit 0
    issue_w(0);
    if (NB == 3) issue_w(1);
    load_src();
    issue_g(raw0, vr0);
    load_src();
    issue_g(raw1, vr1);
    load_src();
    wait_rows<kGr>(raw0);
    split_all(raw0, cv0);
    // One unit g (slot S = g % 2): start the gathers of g+2 into the raw slot g
    // vacated, fetch the indices of g+3, wait for the rows of g+1 (newer ops: the
    // gathers just issued and, for the first unit of an item, the weight DMA of
    // its top), multiply g while converting g+1.
#define MSMD_UNIT(IT, U, RAW_C, V_C, CV_C, RAW_N, V_N, CV_N)            \
  {                                                                     \
    const int v_cur = V_C;                                              \
    issue_g(RAW_C, V_C);                                                \
    load_src();                                                         \
    KP_MARK(3);                                                         \
    if ((U) == 0)                                                       \
      wait_rows<kGr + kWp>(RAW_N);                                      \
    else                                                                \
      wait_rows<kGr>(RAW_N);                                            \
    KP_MARK(4);                                                         \
    compute((IT), (U), CV_C, v_cur, RAW_N, CV_N, [](int) {});           \
    KP_MARK(5);                                                         \
  }
#define MSMD_SLOT_UNIT(IT, U, S)                                        \
  if ((S) % 2 == 0) MSMD_UNIT(IT, U, raw0, vr0, cv0, raw1, vr1, cv1)    \
  else MSMD_UNIT(IT, U, raw1, vr1, cv1, raw0, vr0, cv0)
    // One item.  Its top drains the VM queue: weights(it) (LDS-DMA, issued at the
    // previous item's top) and the previous item's last gathers, which the next two
    // units need anyway.  A counted wait ("all but the newest UB*kGr ops = the
    // gathers") would assume that an LDS-DMA retires in issue order relative to
    // younger buffer loads; it does not under memory pressure -- with a second
    // kernel (wgrad on its side stream) competing for the CUs, waves read weight
    // buffers that had not landed yet.  Counted waits remain only between
    // buffer loads (wait_rows), which do return in order.
#define MSMD_ITEM(IT, PH)                                                              \
  {                                                                                    \
    if ((IT) == 1 && tid == 0) {                                                       \
      ctl[2] = nxt_v;                                                                  \
      ctl[tb ^ 1] = 0;                                                                 \
      if (!sk && nxt_v == last_ticket) *tile_counter = 0;                              \
    }                                                                                  \
    KP_MARK(6);                                                                        \
    if (NB == 3 && (IT) >= 4)   /* weights two items ahead: those of IT + 1 stay in flight */ \
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(kWp + UB * kGr) : "memory");   \
    else if (NB == 2 && (dbg & 16) && (IT) >= 1)  /* experiment: the item's own gathers stay in flight */ \
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(UB * kGr) : "memory");         \
    else                                                                               \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                      \
    KP_MARK(0);                                                                        \
    __builtin_amdgcn_s_barrier();                                                      \
    KP_MARK(1);                                                                        \
    if ((IT) == 1) {                                                                   \
      nxt = __builtin_amdgcn_readfirstlane(ctl[2]);                                    \
      if (nxt < tile_lim) stage_table(nxt, tb ^ 1);                                    \
      staged = true;                                                                   \
    }                                                                                  \
    issue_w((IT) + NB - 1);                                                            \
    KP_MARK(2);                                                                        \
    MSMD_SLOT_UNIT(IT, 0, (PH)*UB + 0)                                                 \
    if (UB > 1) { MSMD_SLOT_UNIT(IT, 1, (PH)*UB + 1) }                                 \
    if (UB > 2) { MSMD_SLOT_UNIT(IT, 2, (PH)*UB + 2) }                                 \
    if (UB > 3) { MSMD_SLOT_UNIT(IT, 3, (PH)*UB + 3) }                                 \
  }
    // (Tried for UB == 1, the 96-128-channel layers, where a PROF build showed per item:
    // top wait 785, barrier 206, weight-DMA issue 496, gather issue + index fetch 534, wait
    // for rows 242, MFMAs + conversion 2047, glue 291 cycles: issuing the weight DMA and the
    // gathers BETWEEN the MFMA groups of the unit instead of in front of them.  Correct, and
    // no faster -- 256 against 258 us on the 128->128 layer in the same call: the issue cycles
    // moved under the MFMAs stretch the MFMA stream by as much.  Removed in round 3.)
    for (int it = 0; it < n_items; it += 2) {
      MSMD_ITEM(it, 0);
      if (it + 1 < n_items) MSMD_ITEM(it + 1, 1);
    }
    }   // (!PP)
#ifdef MSMD_KERNEL_PROF
    if (lane == 0 && wave == 1 && blockIdx.x < 8) atomicAdd(&g_kprof[7], (unsigned long long)n_items);
    if (tid == 0) {
      const unsigned e = atomicAdd(&g_ktrace_n, 1u);
      if (e < (unsigned)kTraceCap) {
        unsigned long long* t = g_ktrace + (size_t)e * 8;
        t[0] = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 4);    // HW_REG_HW_ID
        t[1] = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 20);    // HW_REG_XCC_ID
        t[2] = blockIdx.x;
        t[3] = (unsigned)tile;
        t[4] = kt0;
        t[5] = wall_clock64();
        t[6] = (unsigned)n_items;
        t[7] = mask;
      }
    }
#endif
#undef MSMD_ITEM
#undef MSMD_SLOT_UNIT
#undef MSMD_UNIT
#undef MSMD_ADV
    // The last units started gathers for units past the end (out-of-range: zeros).
    // The compiler does not know those asm loads are still in flight and reuses the
    // ring registers as epilogue temporaries: drain them first.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ---- epilogue: lane (j,q) holds out[row j][16n + 4q .. +3] ----
    const int rt = tile;
    constexpr int kSlotU = WV * R * NT * 64;   // one tile's accumulators, f32x4 units
    // The exchange goes through agent-scope (sc1) accesses to the scratch lines and the
    // flag only: coherent across the XCDs' L2s on their own.  Fences would do it too,
    // but an agent-scope release writes back the whole L2 and an acquire invalidates it
    // -- the weights and the table live there (measured: 62 -> 192 us on a 32-channel layer).
    if (sk && !sk_owner) {
      // a piece of a tile another workgroup owns: accumulators -> scratch[ticket], signal
      // (nothing at all when the range took none of the tile's offsets)
      if (mask != 0u) {
      unsigned long long* sp = (unsigned long long*)(scratch + (size_t)sk_seg * kSlotU + lane);
      // (opaque to the optimiser: the R * NT store addresses are loop-invariant, and hoisted
      // out of the tile loop they cost the NT = 8 instantiation 12 spilled registers)
      asm volatile("" : "+v"(sp));
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const u64x2 v = __builtin_bit_cast(u64x2, acc[r][n]);
          unsigned long long* d = sp + ((wave * R + r) * NT + n) * 128;
          __hip_atomic_store(d, v[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(d + 1, v[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // written through before the signal
      if (lane == 0)
        __hip_atomic_fetch_add(&flags[sk_seg], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      // owner (or a whole tile): first add the pieces of the lower tickets that hold the
      // tile's first ranks, in ticket order
      if (sk && sk_lo > 0) {
        for (int c = sk_ts / sk_S; c < sk_seg; ++c) {
          // (a lower ticket whose range took none of this tile's offsets left nothing)
          const int rlo = c * sk_S - sk_ts;
          if ((unsigned)__ballot(sk_c > 0 && sk_p >= rlo && sk_p < rlo + sk_S) == 0u) continue;
          while (__hip_atomic_load(&flags[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < WV)
            __builtin_amdgcn_s_sleep(4);
          asm volatile("" ::: "memory");
          unsigned long long* sp = (unsigned long long*)(scratch + (size_t)c * kSlotU + lane);
          asm volatile("" : "+v"(sp));
#pragma unroll
          for (int r = 0; r < R; ++r)
#pragma unroll
            for (int n = 0; n < NT; ++n) {
              unsigned long long* d = sp + ((wave * R + r) * NT + n) * 128;
              u64x2 v;
              v[0] = __hip_atomic_load(d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              v[1] = __hip_atomic_load(d + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              acc[r][n] += __builtin_bit_cast(f32x4, v);
            }
          if (lane == 0) {   // the last of the reading waves re-arms the flag
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (__hip_atomic_fetch_add(&flags[c], 1, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT) == 2 * WV - 1)
              __hip_atomic_store(&flags[c], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
      // The BatchNorm that follows wants the per-channel sum and sum of squares of these
      // rows: the tile's partials come straight from the accumulators (bn.hip's statistics
      // pass would read the whole output again) -- one [2][c_out] slot per row tile, rows and
      // waves in fixed order: deterministic.  The four waves' sums meet in the weight buffer,
      // which is idle between a tile's last unit and the next tile's first (a barrier makes
      // sure every wave has left that unit); LDS of its own for this -- 4 KB -- took the NT = 8
      // instantiation from two workgroups per CU to one (145 -> 185 us per launch), and one
      // slot per WAVE instead made the BN's finalize kernel read four times as many (7 -> 12 us).
      if (bn_part) {
        float* stat = (float*)wl;
        float* st_w = stat + wave * (2 * NT * 16);
        __builtin_amdgcn_s_barrier();          // nobody reads weights from wl any more
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const bool ok = rt * kRows + lr[r] < n_out;
            const f32x4 v = ok ? acc[r][n] : (f32x4){0.f, 0.f, 0.f, 0.f};
            s1 += v;
            s2 += v * v;
          }
#pragma unroll
          for (int m = 1; m < 16; m <<= 1)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              s1[e] += __shfl_xor(s1[e], m, 64);
              s2[e] += __shfl_xor(s2[e], m, 64);
            }
          if (j == 0) {
            *(f32x4*)(st_w + 16 * n + 4 * q) = s1;
            *(f32x4*)(st_w + NT * 16 + 16 * n + 4 * q) = s2;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (tid < 2 * NT * 16) {
          float v = 0.f;
#pragma unroll
          for (int w2 = 0; w2 < WV; ++w2) v += stat[w2 * (2 * NT * 16) + tid];
          const int which = tid >= NT * 16 ? 1 : 0, c = tid - which * NT * 16;
          if (c < cout) bn_part[(size_t)rt * 2 * ldo + (size_t)which * ldo + 16 * mt0 + c] = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // (before the next tile's weights land there)
      }
      // ---- lane (j,q) holds out[row j][16n + 4q .. +3] ----
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int p = rt * kRows + lr[r];
        if (p >= n_out) continue;
        const int row = order ? tab[kvol * kRows + lr[r]] : p;
        float* o = out + (size_t)row * ldo + 4 * q;
#pragma unroll
        for (int n = 0; n < NT; ++n)
          if (16 * n + 4 * q < cout) *(f32x4*)(o + 16 * n) = acc[r][n];
      }
    }
    // ---- next tile ----
    if (!staged) {
      if (tid == 0) {
        ctl[2] = nxt_v;
        ctl[tb ^ 1] = 0;
        if (!sk && nxt_v == last_ticket) *tile_counter = 0;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      nxt = __builtin_amdgcn_readfirstlane(ctl[2]);
      if (nxt < tile_lim) stage_table(nxt, tb ^ 1);
    }
    // drains the trailing (null-unit) gathers and weight DMA, lands the table
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (nxt >= tile_lim) break;
    __builtin_amdgcn_s_barrier();
    tile = nxt;
    tb ^= 1;
  }
  if (!sk) return;       // (the dynamic scheduler drew its tiles inside the loop)
  __syncthreads();       // every wave is out of the range's last tile: wl, ctl, tables are free
  }
}

int split_slots_per_cu() { return 2; }
// CUs the persistent conv kernels leave to the other queues of a step (the index pass of the
// next batch, the neighbour-search chains): a kernel of theirs that finds every CU holding two
// conv workgroups (156 of 160 KB of LDS, 476 of 512 registers per SIMD) waits for a conv
// workgroup to EXIT before it can start.  MSMD_RESERVE_CUS.
int reserved_cus() {
  static const int n = [] {
    const int v = env_int2("MSMD_RESERVE_CUS", 0);
    return v < 0 ? 0 : v > 128 ? 128 : v;
  }();
  return n;
}

template <int NT, int UB, int NP, int WV, int NB = 2, bool PP = false, int TB = 2>
int launch_fwd_split(const float* in, int n_in, int cin, const void* wp, const int32_t* nbr,
                     int ld, int n_out, int kvol, int flip, const int32_t* order,
                     int* tile_counter, float* out, int ldo, int cout, int nt_total, int mt0,
                     void* scratch, int* flags, const int32_t* tile_start, int sk_grid,
                     float* bn_part, hipStream_t st) {
  // stream-K: a tile visit's fixed cost in units (offset x k-block) of this instantiation,
  // charged per tile in ranks of ceil(cin / 32) units each
  // (swept 0..48 on the bench layers: 8 is within 2 % of the best for every width)
  const int ovh_units = 8;
  const int kbt = (cin + 31) / 32;
  const int sk_c0 = ((ovh_units + kbt - 1) / kbt) * (sk_c1() + 2);   // in cost units
  constexpr int kRows = WV * 32;
  const size_t smem = sizeof(u32x4) * NB * UB * NP * NT * 64 +
                      sizeof(int) * (TB * (size_t)(kvol + 1) * kRows + 72);
  const int n_tiles = ceil_div(n_out, kRows);
  int nblk = n_tiles;
  const int slots = (256 - reserved_cus()) * (WV == 4 ? split_slots_per_cu() : 1);
  if (nblk > slots) nblk = slots;
  // stream-K: sk_grid ranges for (at most) one workgroup per slot
  if (tile_start) nblk = sk_grid < slots ? sk_grid : slots;
  auto kern = spconv_fwd_split_kernel<NT, UB, NP, WV, NB, PP, TB>;
  static LdsGrant granted;  // per instantiation
  const int lds_rc = optin_dynamic_lds((const void*)kern, smem, granted);
  if (lds_rc != MSMD_OK) return lds_rc;
  MSMD_LAUNCH(kern, dim3(nblk), dim3(WV * 64), smem, st, in, n_in, cin, (const u32x4*)wp, nbr, ld,
              n_out, kvol, flip, order, tile_counter, out, ldo, cout, nt_total, mt0,
              (f32x4*)scratch, flags, tile_start, sk_c0, sk_c1(), env_int2("MSMD_DBG", 0), bn_part,
              sk_grid);
  return launch_status();
}

// Waves per workgroup / rows per tile of the split kernel (see the kernel).  Layers of
// 161..192 output channels (and wider) run the ping-pong form -- 8 waves, 256-row tiles, one
// workgroup and ONE weight stream per CU -- where 192 channels are ONE 12-tile pass: 506 us
// against 604 on 192 -> 192, 407 against 498 on 128 -> 192 (profiles/r05_pp_layers.txt).
// Narrower layers keep 4 x 128 rows, two workgroups per CU: layer by layer the ping-pong form
// is within +-5 % of it there (96 -> 128 +3.5 %, 128 -> 128 0, 80 -> 80 -7 %), and inside the
// LC step -- other streams' kernels wanting CUs that an 8-wave, 130 KiB workgroup holds whole
// -- its 8-tile instantiation ran 10 % slower (101-104 against 113-114 TF, same call).
// MSMD_FWD_PP_MIN moves the threshold (97: every layer above 96 channels), MSMD_FWD_PP=0
// keeps 4 waves everywhere (the round-1..4 kernel), for A/B runs.
// (History: plain 8-wave / 256-row instantiations -- one barrier per item -- measured within
// +-5 % on the 128-/192-channel layers and 25 % slower on the 80-channel ones in rounds 2 and
// 4: the two waves of a SIMD met at every barrier and serialised their MFMA phases.)
int fwd_pp() {
  static const int v = env_int2("MSMD_FWD_PP", 1);
  return v;
}
int fwd_waves(int cout) {
  static const int min_cout = env_int2("MSMD_FWD_PP_MIN", 161);
  return (fwd_pp() && cout >= min_cout) ? 8 : 4;
}
// Column passes: at most 8 tiles of 16 channels each -- except 161..192 channels in the
// ping-pong form, ONE pass of 12 tiles: a row piece is gathered and split into planes once
// for all of c_out (two 6-tile passes did that work twice: the load segment of an item --
// weight DMA, gathers, conversion, ~2400 cycles -- is longer than its 72 MFMAs' 1150).
int fwd_passes(int cout) {
  const int nt_total = (cout + 15) / 16;
  static const int one12 = env_int2("MSMD_FWD_NT12", 1);
  if (fwd_waves(cout) == 8 && one12 && (nt_total == 11 || nt_total == 12)) return 1;
  return (nt_total + 7) / 8;
}
// stream-K: workgroups (= segments = exchange slots) of a launch over `row_tiles` tiles,
// and the exchange buffer: one pass's accumulators of one tile per workgroup
// (ranges per workgroup slot: MSMD_SK_MULT)
int sk_ranges_per_slot() {
  static const int m = [] {
    const int v = env_int2("MSMD_SK_MULT", 1);
    return v < 1 ? 1 : v > 8 ? 8 : v;
  }();
  return m;
}
int sk_grid_size(int row_tiles, int kvol, int waves) {
  const long ranks_max = (long)row_tiles * kvol;
  const long ranges = (256L - reserved_cus()) * (waves == 4 ? split_slots_per_cu() : 1) *
                      sk_ranges_per_slot();
  return (int)(ranks_max < ranges ? ranks_max : ranges);
}
size_t fwd_sk_ws_bytes(int n_out, int kvol, int cout) {
  const int nt_total = (cout + 15) / 16;
  const int n_pass = fwd_passes(cout);
  int per = (nt_total + n_pass - 1) / n_pass;
  per = per > 8 ? 12 : per > 6 ? 8 : per > 4 ? 6 : per > 2 ? 4 : 2;      // the instantiation's NT
  if (fwd_waves(cout) == 8 && per < 6) per = 6;           // (ping-pong: NT = 6 or 8)
  size_t need = 0;
  for (int waves = fwd_waves(cout); waves <= fwd_waves(cout); waves += 4) {
    const int rows = waves * 32;
    const size_t b = (size_t)sk_grid_size(ceil_div(n_out > 0 ? n_out : 0, rows), kvol, waves) *
                     rows * 16 * per * sizeof(float);
    need = b > need ? b : need;
  }
  return need;
}

// c_out is covered in passes of at most 128 channels (8 tiles of 16; a pass's
// tile count is rounded up to even, the extra tile is computed and not stored).
template <int NP>
int dispatch_fwd_split(const float* in, int n_in, int cin, const void* wp, const int32_t* nbr,
                       int ld, int n_out, int kvol, int flip, const int32_t* order,
                       int* tile_counter, int sync_ints, float* out, int cout, void* ws,
                       size_t ws_bytes, const int32_t* tile_prefix, float* bn_part,
                       hipStream_t st) {
  const int nt_total = (cout + 15) / 16;
  const int n_pass = fwd_passes(cout);
  const int per = (nt_total + n_pass - 1) / n_pass;   // tiles per pass
  const int waves = fwd_waves(cout);
  const int row_tiles = ceil_div(n_out, waves * 32);
  // Stream-K (see the kernel) whenever the caller passed the tile prefix (computed for this
  // layer's tile size: msmd_spconv_fwd_split_tile_rows) and the exchange buffers cover one
  // slot / one flag per workgroup; MSMD_STREAMK=0 falls back to the dynamic tile scheduler.
  static const int sk_on = env_int2("MSMD_STREAMK", 1);
  const int sk_grid = sk_grid_size(row_tiles, kvol, waves);
  const int32_t* tile_start = nullptr;
  if (sk_on && tile_prefix && ws && ws_bytes >= fwd_sk_ws_bytes(n_out, kvol, cout) &&
      sync_ints >= 1 + sk_grid) {
    tile_start = tile_prefix;
  }
  int* flags = tile_counter + 1;
  for (int ps = 0; ps < n_pass; ++ps) {
    const int mt0 = ps * per;
    const int tiles = (mt0 + per <= nt_total) ? per : nt_total - mt0;
    const int width = (16 * (mt0 + tiles) <= cout ? 16 * tiles : cout - 16 * mt0);
    float* o = out + 16 * mt0;
    int rc;
#define MSMD_GO(NT_, UB_, WV_, NB_, PP_)                                                         \
  rc = launch_fwd_split<NT_, UB_, NP, WV_, NB_, PP_>(in, n_in, cin, wp, nbr, ld, n_out, kvol,    \
                                                     flip, order, tile_counter, o, cout, width,  \
                                                     nt_total, mt0, ws, flags, tile_start,       \
                                                     sk_grid, bn_part, st)
#define MSMD_GOPP(NT_, TB_)                                                                      \
  rc = launch_fwd_split<NT_, 1, NP, 8, 3, true, TB_>(in, n_in, cin, wp, nbr, ld, n_out, kvol,    \
                                                     flip, order, tile_counter, o, cout, width,  \
                                                     nt_total, mt0, ws, flags, tile_start,       \
                                                     sk_grid, bn_part, st)
    if (waves == 8) {          // ping-pong; a short last pass computes (and drops) spare tiles
      if (tiles > 8) { MSMD_GOPP(12, 1); }
      else if (tiles > 6) { MSMD_GOPP(8, 2); }
      else { MSMD_GOPP(6, 2); }
    }
    else if (tiles > 6) { MSMD_GO(8, 1, 4, 2, false); }
    else if (tiles > 4) { MSMD_GO(6, 1, 4, 2, false); }
    else if (tiles > 2) { MSMD_GO(4, 2, 4, 2, false); }
    else { MSMD_GO(2, 4, 4, 2, false); }
#undef MSMD_GO
#undef MSMD_GOPP
    if (rc != MSMD_OK) return rc;
  }
  return MSMD_OK;
}

// out[k][p] = nbr[k][order[p]]: the neighbour table in tile order.
__global__ __launch_bounds__(256) void permute_cols_kernel(const int32_t* __restrict__ nbr,
                                                           int kvol, int ld, int n,
                                                           const int32_t* __restrict__ order,
                                                           int32_t* __restrict__ out) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const int row = order[p];
  for (int k = 0; k < kvol; ++k) out[(size_t)k * n + p] = nbr[(size_t)k * ld + row];
}

// Stream-K work table: weight[t] = max(sum over the offsets of tile t's cost, 1), the cost of
// an offset = sk_c1() + the number of 32-row groups of the tile with a row connected through it
// (0 when none; exactly what the conv kernel computes from the tile's table slice, positions
// past n clamped to the last row as there).  Tile t = columns [t * rows, (t+1) * rows);
// prefix[t] = sum of the weights before t, prefix[n_tiles] = total.  One block of `rows`
// threads per tile, then a one-block scan.
__global__ __launch_bounds__(256) void tile_weight_kernel(const int32_t* __restrict__ nbr,
                                                          int kvol, int ld, int n, int rows,
                                                          int c1, int32_t* __restrict__ weight) {
  __shared__ int s_cnt[kMaxK];
  __shared__ int s_total;
  if (threadIdx.x < kMaxK) s_cnt[threadIdx.x] = 0;
  if (threadIdx.x == 0) s_total = 0;
  __syncthreads();
  int p = blockIdx.x * rows + threadIdx.x;
  p = p < n ? p : n - 1;
  for (int k = 0; k < kvol; ++k) {
    const unsigned long long b = __ballot(nbr[(size_t)k * ld + p] >= 0);
    const int groups = ((unsigned)b != 0u) + ((unsigned)(b >> 32) != 0u);
    if ((threadIdx.x & 63) == 0 && groups) atomicAdd(&s_cnt[k], groups);
  }
  __syncthreads();
  if (threadIdx.x < kvol && s_cnt[threadIdx.x]) atomicAdd(&s_total, c1 + s_cnt[threadIdx.x]);
  __syncthreads();
  if (threadIdx.x == 0) weight[blockIdx.x] = s_total > 0 ? s_total : 1;
}
__global__ __launch_bounds__(1024) void tile_prefix_kernel(int32_t* __restrict__ a, int n) {
  // in place: a[0..n) weights -> a[0..n] exclusive prefix (a has n + 1 entries)
  __shared__ int part[1024];
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
  int run = part[t] - s;    // exclusive prefix of this thread's slice
  for (int i = b; i < e; ++i) {
    const int w = a[i];
    a[i] = run;
    run += w;
  }
  if (t == 1023) a[n] = part[1023];
}

// ------------------------------------------------------------------ wgrad --
// dW[k] = sum_p in[i_p,:]^T (x) dout[o_p,:] with the pair index as the MFMA's contraction
// (32 pairs per v_mfma_f32_16x16x32_bf16).  The whole-block kernel of
// spconv_wgrad_block.hip takes every layer whose widths are multiples of 16; what is left
// here is the 64 x 64 SLAB kernel for the other widths (multiples of 4): workgroup =
// (2048-pair chunk, offset k, channel slab), its 4 waves take interleaved 32-pair steps,
// partials are reduced by wgrad_reduce (spconv.hip) in fixed order.
//
// Operands: lane (i, g) covers pairs 8g .. 8g+7 of the step and, on each side, W consecutive
// channels of the slab: ONE load per pair per side through a raw buffer descriptor with
// 32-bit byte offsets staged in LDS (an absent pair or channel is an out-of-range offset:
// zeros, no branch, no traffic).  The MFMA wants 8 consecutive contraction slots of ONE
// channel per lane -- the block the lane just loaded, transposed -- and v_cvt_pk_bf16_f32
// takes its two inputs from two registers: converting (pair 2t, pair 2t+1) of channel a
// together yields dword t of tile a's operand directly (tile a, row i <-> channel W i + a).
// Slabs are as wide as the layer's remainder needs: a remainder of 16 or 32 channels is 1
// or 2 tiles wide on that side (80 x 80 = 16 + 4 + 4 + 1 tiles instead of 4 x 16).
// History (all measured, all removed in round 3): the pointer-addressed
// first version, 128 x 64 slabs, a 128 x 128 slab with the conversion shared through LDS by
// all four waves, and pre-split bf16 plane tensors gathered by LDS-DMA.
template <int NP, int WA, int WB>
__device__ __forceinline__ void wgrad_var_body(
    const float* __restrict__ in, int cin, const float* __restrict__ dout, int cout,
    const unsigned* s_in, const unsigned* s_out, f32x4* red, int a0, int b0, int cnt,
    float* __restrict__ dst /* [cin][cout] partial of this (k, chunk) */, int dbg) {
  using P = Products<NP>;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const __amdgpu_buffer_rsrc_t rs_a =
      __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, (int)kOobOffset, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b =
      __builtin_amdgcn_make_buffer_rsrc((void*)dout, 0, (int)kOobOffset, 0x00020000);
  const unsigned cola = (unsigned)(a0 + WA * i) * 4u, colb = (unsigned)(b0 + WB * i) * 4u;
  const bool in_a = a0 + WA * i < cin, in_b = b0 + WB * i < cout;

  f32x4 acc[WA][WB];
#pragma unroll
  for (int a = 0; a < WA; ++a)
#pragma unroll
    for (int b = 0; b < WB; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  unsigned ra[8][WA], rb[8][WB];   // this lane's 8 pairs x W channels (fp32 bits)
  auto load_w = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned off, unsigned* out, auto w) {
    constexpr int W = decltype(w)::value;
    if constexpr (W == 4) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
      out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; out[3] = v[3];
    } else if constexpr (W == 2) {
      const auto v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, 0);
      out[0] = v[0]; out[1] = v[1];
    } else {
      out[0] = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)off, 0, 0);
    }
  };
  auto fetch = [&](int step) {
    const int e0 = 32 * step + 8 * g;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const u32x4 oa4 = *(const u32x4*)(s_in + e0 + 4 * h), ob4 = *(const u32x4*)(s_out + e0 + 4 * h);
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        const unsigned fa = (oa4[s2] == kOobOffset || !in_a) ? kOobOffset : oa4[s2] + cola;
        load_w(rs_a, fa, ra[4 * h + s2], std::integral_constant<int, WA>{});
        const unsigned fb = (ob4[s2] == kOobOffset || !in_b) ? kOobOffset : ob4[s2] + colb;
        load_w(rs_b, fb, rb[4 * h + s2], std::integral_constant<int, WB>{});
      }
    }
  };
  auto split_side = [&](auto& r, auto* op, auto w) {
    constexpr int W = decltype(w)::value;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int a = 0; a < W; ++a) {
        float v0 = __uint_as_float(r[2 * t][a]), v1 = __uint_as_float(r[2 * t + 1][a]);
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
          unsigned hi;
          asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(v0), "v"(v1));
          op[a][pl][t] = hi;
          if (pl + 1 < NP) {
            v0 = v0 - __uint_as_float(hi << 16);
            v1 = v1 - __uint_as_float(hi & 0xffff0000u);
          }
        }
      }
  };
  const int n_steps = (cnt + 31) / 32;
  if (wave < n_steps) fetch(wave);
  for (int step = wave; step < n_steps; step += 4) {
    u32x4 oa[WA][NP], ob[WB][NP];
    split_side(ra, oa, std::integral_constant<int, WA>{});
    split_side(rb, ob, std::integral_constant<int, WB>{});
    if (step + 4 < n_steps) fetch(step + 4);   // in flight under the MFMAs below
    if (dbg & 4) {
#pragma unroll
      for (int a = 0; a < WA; ++a)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) asm volatile("" ::"v"(oa[a][pl]));
#pragma unroll
      for (int b = 0; b < WB; ++b)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) asm volatile("" ::"v"(ob[b][pl]));
      continue;
    }
    __builtin_amdgcn_s_setprio(2);
#pragma unroll
    for (int t = 0; t < P::n; ++t)
#pragma unroll
      for (int a = 0; a < WA; ++a)
#pragma unroll
        for (int b = 0; b < WB; ++b)
          acc[a][b] = mfma_bf16(oa[a][P::a[t]], ob[b][P::b[t]], acc[a][b]);
    __builtin_amdgcn_s_setprio(0);
  }
  // cross-wave sum, fixed tree order (w0+w2) + (w1+w3): deterministic
  constexpr int T = WA * WB;
  __syncthreads();   // everyone is done with the index arrays (red aliases them)
  if (wave >= 2) {
#pragma unroll
    for (int a = 0; a < WA; ++a)
#pragma unroll
      for (int b = 0; b < WB; ++b) red[((wave - 2) * T + a * WB + b) * 64 + lane] = acc[a][b];
  }
  __syncthreads();
  if (wave < 2) {
#pragma unroll
    for (int a = 0; a < WA; ++a)
#pragma unroll
      for (int b = 0; b < WB; ++b) acc[a][b] += red[(wave * T + a * WB + b) * 64 + lane];
  }
  __syncthreads();
  if (wave == 1) {
#pragma unroll
    for (int a = 0; a < WA; ++a)
#pragma unroll
      for (int b = 0; b < WB; ++b) red[(a * WB + b) * 64 + lane] = acc[a][b];
  }
  __syncthreads();
  if (wave == 0) {
    const int cb = b0 + WB * i;
#pragma unroll
    for (int a = 0; a < WA; ++a) {
      f32x4 v[WB];
#pragma unroll
      for (int b = 0; b < WB; ++b) v[b] = acc[a][b] + red[(a * WB + b) * 64 + lane];
      // D of tile (a,b): lane (col n = i, g) reg r -> ci = a0 + WA(4g+r) + a, co = b0 + WB n + b
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ci = a0 + WA * (4 * g + r) + a;
        if (ci < cin && cb < cout) {
          float* o = dst + (size_t)ci * cout + cb;
          if constexpr (WB == 4) *(f32x4*)o = (f32x4){v[0][r], v[1][r], v[2][r], v[3][r]};
          else if constexpr (WB == 2) *(f32x2*)o = (f32x2){v[0][r], v[1][r]};
          else *o = v[0][r];
        }
      }
    }
  }
}

// slabs of a side: full 64-channel ones, then the remainder (16 -> W=1, 32 -> W=2, else a
// masked W=4 slab)
__host__ __device__ inline int wgrad_var_slabs(int c) { return (c + 63) / 64; }
__device__ __forceinline__ int wgrad_var_width(int c, int slab) {
  const int rem = c - 64 * slab;
  return rem == 16 ? 1 : rem == 32 ? 2 : 4;
}

template <int NP>
__global__ __launch_bounds__(256, 2) void spconv_wgrad_split_var_kernel(
    const float* __restrict__ in, int cin, const float* __restrict__ dout, int cout,
    const int32_t* __restrict__ pairs, const int32_t* __restrict__ num, int ld, int nchunks,
    int kvol, float* __restrict__ partial /* [K][nchunks][cin][cout] */, int dbg) {
  constexpr int CHUNK = kWgradSplitChunk;
  __shared__ __attribute__((aligned(16))) char lds_raw[2 * 16 * 64 * sizeof(f32x4)];
  unsigned* s_in = (unsigned*)lds_raw;       // BYTE OFFSETS of the rows (or the OOB offset)
  unsigned* s_out = s_in + CHUNK;
  static_assert(2 * CHUNK * sizeof(int) <= sizeof(lds_raw), "index arrays must fit");
  const int SB = wgrad_var_slabs(cout);
  int k, chunk, slab;
  if (!wgrad_work(nchunks, kvol, wgrad_var_slabs(cin) * SB, chunk, k, slab)) return;
  const int Pk = num[k];
  const int p_begin = chunk * CHUNK;
  if (p_begin >= Pk) return;
  const int cnt = (Pk - p_begin) < CHUNK ? (Pk - p_begin) : CHUNK;
  const int sa = slab / SB, sb = slab % SB;
  {
    const int32_t* pin = pairs + ((size_t)k * 2 + 0) * ld + p_begin;
    const int32_t* pout = pairs + ((size_t)k * 2 + 1) * ld + p_begin;
    const unsigned rowa = (unsigned)cin * 4u, rowb = (unsigned)cout * 4u;
    for (int e = threadIdx.x; e < CHUNK; e += 256) {   // past the end: "no pair"
      int ia = e < cnt ? pin[e] : -1, ib = e < cnt ? pout[e] : -1;
      if (dbg & 1) { ia = ia < 0 ? ia : (ia & 4095); ib = ib < 0 ? ib : (ib & 4095); }
      if (dbg & 2) { ia = -1; ib = -1; }
      s_in[e] = ia >= 0 ? (unsigned)ia * rowa : kOobOffset;
      s_out[e] = ib >= 0 ? (unsigned)ib * rowb : kOobOffset;
    }
  }
  __syncthreads();
  float* dst = partial + ((size_t)k * nchunks + chunk) * cin * cout;
  const int wa = wgrad_var_width(cin, sa), wb = wgrad_var_width(cout, sb);
#define MSMD_VAR_CASE(A, B)                                                                  \
  case A * 8 + B:                                                                            \
    wgrad_var_body<NP, A, B>(in, cin, dout, cout, s_in, s_out, (f32x4*)lds_raw, 64 * sa,     \
                             64 * sb, cnt, dst, dbg);                                        \
    break
  switch (wa * 8 + wb) {
    MSMD_VAR_CASE(4, 4);
    MSMD_VAR_CASE(4, 2);
    MSMD_VAR_CASE(4, 1);
    MSMD_VAR_CASE(2, 4);
    MSMD_VAR_CASE(2, 2);
    MSMD_VAR_CASE(2, 1);
    MSMD_VAR_CASE(1, 4);
    MSMD_VAR_CASE(1, 2);
    MSMD_VAR_CASE(1, 1);
  }
#undef MSMD_VAR_CASE
}


}  // namespace
}  // namespace msmd

using namespace msmd;

MSMD_EXPORT size_t msmd_spconv_packed_split_bytes(int kvol, int cin, int cout, int np) {
  return (size_t)kvol * ((cin + 31) / 32) * ((cout + 15) / 16) * np * 1024;
}

// `cin`/`cout` are the weight's own dims; with flags bit0 the packed image is of
// W[k]^T (contraction over c_out): its size is msmd_spconv_packed_split_bytes(K,
// cout, cin, np).
namespace {
int pack_split(const float* w, int kvol, int cin, int cout, int flags, int np, void* packed,
               void* packed_other, hipStream_t st) {
  if (kvol <= 0 || cin <= 0 || cout <= 0 || np < 1 || np > 3 || !w || !packed)
    return MSMD_ERR_INVALID_ARG;
  const int ci = (flags & 1) ? cout : cin, co = (flags & 1) ? cin : cout;
  long total = (long)kvol * ((ci + 31) / 32) * ((co + 15) / 16) * 64;
  if (packed_other) total += (long)kvol * ((co + 31) / 32) * ((ci + 15) / 16) * 64;
  int nblk = (int)((total + 255) / 256);
  if (nblk > 4096) nblk = 4096;
  if (np == 3)
    MSMD_LAUNCH(pack_weight_split_kernel<3>, dim3(nblk), dim3(256), 0, st, w, kvol, cin, cout,
                flags, (u32x4*)packed, (u32x4*)packed_other);
  else if (np == 2)
    MSMD_LAUNCH(pack_weight_split_kernel<2>, dim3(nblk), dim3(256), 0, st, w, kvol, cin, cout,
                flags, (u32x4*)packed, (u32x4*)packed_other);
  else
    MSMD_LAUNCH(pack_weight_split_kernel<1>, dim3(nblk), dim3(256), 0, st, w, kvol, cin, cout,
                flags, (u32x4*)packed, (u32x4*)packed_other);
  return launch_status();
}
}  // namespace

MSMD_EXPORT int msmd_spconv_pack_weight_split(const float* w, int kvol, int cin, int cout,
                                              int flags, int np, void* packed, msmd_stream_t stream) {
  return pack_split(w, kvol, cin, cout, flags, np, packed, nullptr, (hipStream_t)stream);
}

// Both images of one weight in one launch: `packed` as msmd_spconv_pack_weight_split
// with `flags`, `packed_transposed` the same with flags ^ 1 (what dgrad reads).
MSMD_EXPORT int msmd_spconv_pack_weight_split_pair(const float* w, int kvol, int cin, int cout,
                                                   int flags, int np, void* packed,
                                                   void* packed_transposed,
                                                   msmd_stream_t stream) {
  if (!packed_transposed) return MSMD_ERR_INVALID_ARG;
  return pack_split(w, kvol, cin, cout, flags, np, packed, packed_transposed, (hipStream_t)stream);
}

// `descs`: n_desc descriptors in DEVICE memory, 48 bytes each:
//   { const float* weight; void* packed; void* packed_transposed | NULL; int64 start;
//     int32 kernel_volume, c_in, c_out, flags; }
// start = running count of work units (one unit = the `np` 16-byte pieces of one (offset,
// k-block, tile, lane): msmd_spconv_packed_split_bytes / (16 np) per image written) in
// descriptor order, `total` their sum.  Same images as msmd_spconv_pack_weight_split[_pair].
MSMD_EXPORT int msmd_spconv_pack_weight_split_many(const void* descs, int n_desc, long total,
                                                   int np, msmd_stream_t stream) {
  static_assert(sizeof(PackDesc) == 48, "descriptor layout is part of the ABI");
  if (!descs || n_desc < 1 || total < 1 || np < 1 || np > 3) return MSMD_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  int nblk = (int)((total + 255) / 256);
  if (nblk > 8192) nblk = 8192;
  const PackDesc* d = (const PackDesc*)descs;
  if (np == 3)
    MSMD_LAUNCH(pack_weight_split_many_kernel<3>, dim3(nblk), dim3(256), 0, st, d, n_desc, total);
  else if (np == 2)
    MSMD_LAUNCH(pack_weight_split_many_kernel<2>, dim3(nblk), dim3(256), 0, st, d, n_desc, total);
  else
    MSMD_LAUNCH(pack_weight_split_many_kernel<1>, dim3(nblk), dim3(256), 0, st, d, n_desc, total);
  return launch_status();
}

MSMD_EXPORT int msmd_spconv_fwd_split_supported(int cin, int cout, int kvol) {
  // 8-channel gather pieces, 16-byte stores; below 32 channels on either side the
  // 32-wide k-blocks / paired output tiles would be mostly padding
  return cin >= 32 && (cin & 7) == 0 && cout >= 32 && (cout & 3) == 0 && kvol >= 1 &&
         kvol <= kMaxK;
}

MSMD_EXPORT int msmd_spconv_fwd_split_tile_rows(int cout) { return 32 * fwd_waves(cout); }

// The template arguments dispatch_fwd_split picks for a layer (its first pass; a short last
// pass of the 4-wave form may run a narrower instantiation): what rocprofv3 prints between
// the angle brackets of spconv_fwd_split_kernel, so that tools do not re-derive it.
MSMD_EXPORT int msmd_spconv_fwd_split_instantiation(int cout, int* params /* [7] */) {
  if (cout < 1 || !params) return MSMD_ERR_INVALID_ARG;
  const int nt_total = (cout + 15) / 16;
  const int n_pass = fwd_passes(cout);
  const int per = (nt_total + n_pass - 1) / n_pass;
  const int waves = fwd_waves(cout);
  int nt, ub = 1, nb = 2, pp = 0, tb = 2;
  if (waves == 8) {
    nt = per > 8 ? 12 : per > 6 ? 8 : 6;
    nb = 3;
    pp = 1;
    tb = nt == 12 ? 1 : 2;
  } else {
    nt = per > 6 ? 8 : per > 4 ? 6 : per > 2 ? 4 : 2;
    ub = nt == 4 ? 2 : nt == 2 ? 4 : 1;
  }
  params[0] = nt;
  params[1] = ub;
  params[2] = waves;
  params[3] = nb;
  params[4] = pp;
  params[5] = tb;
  params[6] = n_pass;
  return MSMD_OK;
}

MSMD_EXPORT size_t msmd_spconv_fwd_split_workspace_bytes(int n_out, int cout) {
  return fwd_sk_ws_bytes(n_out, kMaxK, cout);
}

// bn_partials (or NULL): [msmd_spconv_fwd_split_stats_blocks(n_out, c_out)][2][c_out] floats
// (one block per row tile of this width's kernel: 128 or 256 rows) -- per row tile the column sums
// and sums of squares of the rows written (what msmd_bn_act_fwd_from_partials_f32 takes)
MSMD_EXPORT int msmd_spconv_fwd_split_stats_blocks(int n_out, int cout) {
  return ceil_div(n_out > 0 ? n_out : 0, 32 * fwd_waves(cout));
}

MSMD_EXPORT int msmd_spconv_fwd_split_stats(const float* planes, int n_in, int cin,
                                            const void* packed, const int32_t* nbr, int ld,
                                            int n_out, int kvol, int weight_flip,
                                            const int32_t* row_order, int32_t* tile_counter,
                                            int sync_ints, float* out, int cout, int np,
                                            void* workspace, size_t workspace_bytes,
                                            const int32_t* tile_prefix, float* bn_partials,
                                            msmd_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!msmd_spconv_fwd_split_supported(cin, cout, kvol) || np < 1 || np > 3)
    return MSMD_ERR_UNSUPPORTED;
  if (!tile_counter || sync_ints < 1) return MSMD_ERR_INVALID_ARG;
  if (n_out <= 0) return MSMD_OK;
  // the gathers address the features through a 32-bit buffer offset (row * row bytes as a
  // 24-bit multiply)
  if ((size_t)n_in * cin * sizeof(float) >= (size_t)kOobOffset || n_in >= (1 << 24) ||
      cin * sizeof(float) >= (1u << 24))
    return MSMD_ERR_RANGE;
#define MSMD_ARGS planes, n_in, cin, packed, nbr, ld, n_out, kvol, weight_flip, row_order, \
                  tile_counter, sync_ints, out, cout, workspace, workspace_bytes, tile_prefix, \
                  bn_partials, st
  if (np == 3) return dispatch_fwd_split<3>(MSMD_ARGS);
  if (np == 2) return dispatch_fwd_split<2>(MSMD_ARGS);
  return dispatch_fwd_split<1>(MSMD_ARGS);
#undef MSMD_ARGS
}

MSMD_EXPORT int msmd_spconv_fwd_split(const float* planes, int n_in, int cin, const void* packed,
                                      const int32_t* nbr, int ld, int n_out, int kvol,
                                      int weight_flip, const int32_t* row_order,
                                      int32_t* tile_counter, int sync_ints, float* out, int cout,
                                      int np, void* workspace, size_t workspace_bytes,
                                      const int32_t* tile_prefix, msmd_stream_t stream) {
  return msmd_spconv_fwd_split_stats(planes, n_in, cin, packed, nbr, ld, n_out, kvol, weight_flip,
                                     row_order, tile_counter, sync_ints, out, cout, np, workspace,
                                     workspace_bytes, tile_prefix, nullptr, stream);
}

MSMD_EXPORT int msmd_rulebook_tile_prefix(const int32_t* nbr, int kvol, int ld, int n_rows,
                                          int rows_per_tile, int32_t* prefix,
                                          msmd_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (kvol < 1 || kvol > kMaxK || n_rows < 0 || ld < n_rows || !prefix ||
      (rows_per_tile != 128 && rows_per_tile != 256) ||
      (n_rows > 0 && !nbr))
    return MSMD_ERR_INVALID_ARG;
  const int n_tiles = ceil_div(n_rows, rows_per_tile);
  if (n_tiles == 0) {
    hipMemsetAsync(prefix, 0, sizeof(int32_t), st);
    return launch_status();
  }
  MSMD_LAUNCH(tile_weight_kernel, dim3(n_tiles), dim3(rows_per_tile), 0, st, nbr, kvol, ld, n_rows,
              rows_per_tile, sk_c1(), prefix);
  MSMD_LAUNCH(tile_prefix_kernel, dim3(1), dim3(1024), 0, st, prefix, n_tiles);
  return launch_status();
}

MSMD_EXPORT int msmd_rulebook_permute_cols(const int32_t* nbr, int kvol, int ld, int n,
                                           const int32_t* order, int32_t* out, msmd_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  if (kvol <= 0 || n < 0 || ld < n) return MSMD_ERR_INVALID_ARG;
  if (n == 0) return MSMD_OK;
  MSMD_LAUNCH(permute_cols_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, nbr, kvol, ld, n,
              order, out);
  return launch_status();
}

// wgrad at bf16 MFMA rate (64x64 slabs: c_in and c_out multiples of 64).  Writes
// the per-(k, chunk) partials in msmd_spconv_wgrad_f32's workspace layout; the
// caller finishes with that entry point's reduction (see msmd_spconv_wgrad_split
// in spconv.hip).
namespace msmd {
int stream_k_c1() { return sk_c1(); }     // plan_many.hip: the tile weights of all tables
int wgrad_split_partials(const float* in_feat, int c_in, const float* d_out, int c_out,
                         const int32_t* pairs, const int32_t* num, int ld, int kvol, int np,
                         int nchunks, float* ws, hipStream_t st) {
  // 32-bit row offsets (buffer loads): both operands must stay below 4 GiB
  if (!((double)ld * 4.0 * (c_in > c_out ? c_in : c_out) < (double)kOobOffset))
    return MSMD_ERR_RANGE;
  static const int wdbg = env_int2("MSMD_WGRAD_DBG", 0);
  const dim3 gv(wgrad_grid(nchunks, kvol, wgrad_var_slabs(c_in) * wgrad_var_slabs(c_out)));
  if (np == 3)
    MSMD_LAUNCH(spconv_wgrad_split_var_kernel<3>, gv, dim3(256), 0, st, in_feat, c_in, d_out,
                c_out, pairs, num, ld, nchunks, kvol, ws, wdbg);
  else if (np == 2)
    MSMD_LAUNCH(spconv_wgrad_split_var_kernel<2>, gv, dim3(256), 0, st, in_feat, c_in, d_out,
                c_out, pairs, num, ld, nchunks, kvol, ws, wdbg);
  else
    MSMD_LAUNCH(spconv_wgrad_split_var_kernel<1>, gv, dim3(256), 0, st, in_feat, c_in, d_out,
                c_out, pairs, num, ld, nchunks, kvol, ws, wdbg);
  return launch_status();
}
}  // namespace msmd

#ifdef MSMD_KERNEL_PROF
// out[16] <- phase cycle sums (then cleared): 0 top wait, 1 barrier, 2 weight issue,
// 3 gather issue + index fetch, 4 wait for rows, 5 convert + multiply, 6 loop glue,
// 7 items
MSMD_EXPORT int msmd_debug_kprof(unsigned long long* out) {
  hipDeviceSynchronize();
  unsigned long long z[16] = {0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_kprof), sizeof(z)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_kprof), z, sizeof(z)) != hipSuccess) return -1;
  return 0;
}
// out[cap][8] <- the per-tile trace since the last call (then cleared); returns the count
MSMD_EXPORT int msmd_debug_ktrace(unsigned long long* out, int cap) {
  hipDeviceSynchronize();
  unsigned n = 0, zero = 0;
  if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_ktrace_n), sizeof(n)) != hipSuccess) return -1;
  if (n > (unsigned)kTraceCap) n = kTraceCap;
  if ((int)n > cap) n = cap;
  if (n && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ktrace), sizeof(unsigned long long) * 8 * n) !=
               hipSuccess)
    return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_ktrace_n), &zero, sizeof(zero)) != hipSuccess) return -1;
  return (int)n;
}
#endif
