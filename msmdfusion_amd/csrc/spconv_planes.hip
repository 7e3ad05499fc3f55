// spconv_planes.hip -- bf16 PLANE tensors and the weight-gradient kernel that reads them.
//
// Why planes in HBM.  The split arithmetic (spconv_split.hip) writes every fp32 value
// as the exact sum of three bf16 values.  The forward kernel does that split in
// registers, amortised over all of c_out.  The weight gradient cannot: its contraction
// runs over PAIRS, every gathered row is re-split for each of the ~14 offsets it takes
// part in and for each channel slab -- 864 VALU operations beside 96 MFMAs per 32-pair
// step (r01: MFMA pipe 28 % busy, VALU-bound).  Here a tensor is split ONCE
// (msmd_split_planes_f32: HBM-bound, 4 B read + 6 B written per element) into
//     planes[row][plane][channel]  bf16,  rows 0 .. n, row n = zeros ("no pair")
// and the weight-gradient kernel gathers bf16 rows straight into LDS by LDS-DMA and
// feeds the matrix cores from there through the transposing LDS read: no VALU work on
// the operands at all.
//
// dW[k][ci][co] = sum_p in[i_p][ci] * dout[o_p][co]: the pair index p is the MFMA's
// contraction (32 pairs per v_mfma_f32_16x16x32_bf16), but memory is pair-major /
// channel-contiguous -- both operands need a transpose.  ds_read_b64_tr_b16 does it:
// each 16-lane group reads a [4 pairs][16 channels] block, lane i supplying the 8-byte
// address of (row i>>2, channel quad i&3) and receiving channel i of the 4 rows
// (row stride free; checked on the device by tools/scratch/tr_probe.hip).
//
// Work decomposition: workgroup = (chunk of kRowsPerChunk consecutive OUTPUT ROWS, offset
// k, slab of 128 c_in x 64 c_out channels), XCD-aware block order as the other wgrad
// kernels (common.hpp): block b runs on XCD b % 8, every XCD gets whole chunks and walks
// their (k, slab) workgroups consecutively.  The chunk is a range of output rows, not of
// pair positions (the pair lists are sorted by output row, so it is still one contiguous
// sub-range per offset: pair_ranges_kernel): all 27 offsets of a chunk then gather the
// SAME dout rows and the same few neighbourhoods of input rows, which stay in that XCD's
// 4 MiB L2 -- with chunks of pair positions (r01) chunk c of a corner offset and of the
// centre offset cover different rows and every gather went to HBM (L2 hit rate 22 %).  Its 4
// waves own disjoint output tiles (2 x 2 arrangement, up to 4 x 2 tiles of 16 x 16 each,
// partial slabs split evenly) -- no cross-wave reduction.  A STAGE = 32 pairs = one MFMA
// contraction step; its operands (36 KiB at 3 planes) are gathered by LDS-DMA into one
// of two buffers while the previous stage is multiplied: one barrier per stage.
// Partials per (k, chunk) + the fixed-order reduction of spconv.hip: deterministic.
//
// LDS image of one DMA op (1 KiB = 8 pairs x 64 channels of one plane):
//   [tile a = 16 channels][slot][32 B],  slot = (pair + 4 * (group & 1)) & 7
// so that the two 16-lane groups a transposing read services together (pairs 8g..8g+3
// of groups g, g+1) touch all 64 banks exactly once.
#include "common.hpp"

#include <stdlib.h>

#include <type_traits>

namespace msmd {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((address_space(1))) const void glb_void;

// products kept for NP planes, smallest terms first (as spconv_split.hip)
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

// ---------------------------------------------------------------- fp32 -> planes --
// planes[(row * NP + p) * c + ch] = plane p of x[row][ch]; row n is the zero row.
// One thread = 8 channels (2 x 16-byte loads, NP x 16-byte stores).
template <int NP>
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, int n,
                                                           int c, u32x4* __restrict__ planes) {
  const int c8 = c >> 3;
  const long total = (long)(n + 1) * c8;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int row = (int)(e / c8), q = (int)(e - (long)row * c8);
    u32x4 out[NP];
    if (row < n) {
      const f32x4* src = (const f32x4*)(x + (size_t)row * c + 8 * q);
      const f32x4 lo = src[0], hi = src[1];
      float r[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
      for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const f32x2 v = {r[2 * t], r[2 * t + 1]};
          const bf16x2 h = __builtin_convertvector(v, bf16x2);   // round to nearest even
          out[p][t] = __builtin_bit_cast(unsigned int, h);
          if (p + 1 < NP) {
            const f32x2 back = __builtin_convertvector(h, f32x2);
            r[2 * t] = v[0] - back[0];       // exact
            r[2 * t + 1] = v[1] - back[1];
          }
        }
      }
    } else {
#pragma unroll
      for (int p = 0; p < NP; ++p) out[p] = (u32x4){0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) planes[((size_t)row * NP + p) * c8 + q] = out[p];
  }
}

__device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// Phase timing of one wave (build with `make PROF=1`; never in the shipped library)
#ifdef MSMD_KERNEL_PROF
__device__ unsigned long long g_wprof[16];
#define WP_BEGIN() unsigned long long wp_t = __builtin_amdgcn_s_memtime()
#define WP_MARK(i)                                                                     \
  {                                                                                    \
    const unsigned long long wp_n = __builtin_amdgcn_s_memtime();                      \
    if (lane == 0 && (wave == 1 || wave == 5) && blockIdx.x < 64)                         \
      atomicAdd(&g_wprof[i + (wave == 5 ? 8 : 0)], wp_n - wp_t);                         \
    wp_t = wp_n;                                                                       \
  }
#else
#define WP_BEGIN()
#define WP_MARK(i)
#endif

// ------------------------------------------------------------------ wgrad --
// One workgroup per CU, 8 waves, slab of 128 c_in x 128 c_out channels, roles split:
//   waves 4-7  PRODUCERS: wave 4+g gathers the pairs [8g, 8g+8) of every stage, both sides
//              (up to 12 LDS-DMA ops of 1 KiB per stage), and does nothing else;
//   waves 0-3  CONSUMERS: 2 x 2 arrangement, up to 4 x 4 output tiles each (96 MFMAs per
//              stage), operands by transposing LDS reads; they issue no memory operations
//              besides the epilogue.
// Each SIMD hosts one wave of each role: the producers' DMA issue (~90 cycles per op) and
// index handling run beside the consumers' MFMAs instead of inside their instruction
// stream -- with all 8 waves doing both (r02's first versions) the phases ran in
// lockstep behind the stage barrier: 4100 cycles per stage around 770 cycles of MFMAs.
// Ring of 3 stage buffers (144 KiB): the gathers of stages s+1, s+2 are in flight while s
// is multiplied; ONE barrier per stage.
constexpr int kSlab = 128;        // channels per workgroup slab, both sides (8 tiles of 16)
constexpr int kWT = 4;            // output tiles per consumer wave and side, at most
constexpr int kBuffers = 3;

template <int NP>
__global__ __launch_bounds__(512) void spconv_wgrad_planes_kernel(
    const unsigned short* __restrict__ pa, int cin, int n_in,
    const unsigned short* __restrict__ pb, int cout, int n_out,
    const int32_t* __restrict__ pairs, const int32_t* __restrict__ ranges /* [K][nchunks+1] */,
    int ld, int nchunks, int kvol, float* __restrict__ partial /* [K][nchunks][cin][cout] */,
    int dbg /* experiments only: 2 no gathers, 4 no MFMAs, 8 rows folded onto 1024 */) {
  using P = Prod<NP>;
  // one stage buffer: per side [half 2][plane NP][group 4] KiB; A then B
  constexpr int kSide = 2 * NP * 4096, kBuf = 2 * kSide;
  extern __shared__ __attribute__((aligned(1024))) char smem[];

  const int slabs_b = (cout + kSlab - 1) / kSlab;
  int k, chunk, slab;
  if (!wgrad_work(nchunks, kvol, ((cin + kSlab - 1) / kSlab) * slabs_b, chunk, k, slab)) return;
  const int p_begin = ranges[(size_t)k * (nchunks + 1) + chunk];
  const int cnt = ranges[(size_t)k * (nchunks + 1) + chunk + 1] - p_begin;
  if (cnt <= 0) return;       // (the reduction skips empty (k, chunk) cells)
  const int n_stages = (cnt + 31) >> 5;
  const int a0 = (slab / slabs_b) * kSlab, b0 = (slab % slabs_b) * kSlab;
  const int remA = cin - a0, remB = cout - b0;
  const int nA = remA >= kSlab ? 8 : (remA + 15) >> 4;    // valid 16-channel tiles
  const int nB = remB >= kSlab ? 8 : (remB + 15) >> 4;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // 0..7
  const unsigned smem_base = (unsigned)(size_t)(lds_void*)smem;   // LDS byte address
  WP_BEGIN();

  if (wave >= 4) {
    // =============================== producer ===============================
    // One DMA op = 8 pairs x 64 channels of one plane = 1 KiB, lane-linear in LDS.  Lanes
    // 8p .. 8p+7 fetch the 128 contiguous bytes of pair p's row (one full line per 8
    // lanes), in a permuted piece order: position j of the LDS row holds piece
    // j ^ x(p, g), x = 2 * ((p >> 1) & 1) + 4 * (g & 1) -- the XOR keeps a tile's two
    // 16-byte pieces adjacent and spreads the 8 rows a transposing read touches together
    // (rows 4q..4q+3 of two 16-lane groups) over all 64 banks.
    const int g_grp = wave & 3;
    const int g_row = lane >> 3, g_pos = lane & 7;
    const int g_piece = g_pos ^ (2 * ((g_row >> 1) & 1) + 4 * (g_grp & 1));
    const int halvesA = nA > 4 ? 2 : 1, halvesB = nB > 4 ? 2 : 1;
    const int32_t* pin = pairs + ((size_t)k * 2 + 0) * ld + p_begin;
    const int32_t* pout = pairs + ((size_t)k * 2 + 1) * ld + p_begin;
    const size_t rowA = (size_t)NP * cin, rowB = (size_t)NP * cout;
    // The 8 row indices a wave needs per side and stage are read on the SCALAR unit (wave-
    // uniform addresses, lgkmcnt) a stage ahead of their use, so the vector memory queue
    // holds nothing but DMA ops and the counted waits below are exact.
    // (the loads are issued at the top of an iteration and consumed at its end: ~1 us of
    // scalar-cache miss latency otherwise sat in every stage)
    struct Rows8 { int v[8]; };
    auto load_rows = [&](const int32_t* list, int s) -> Rows8 {     // 8 uniform values
      Rows8 r;
      const int e0 = 32 * s + 8 * g_grp;
#pragma unroll
      for (int i = 0; i < 8; ++i) r.v[i] = list[e0 + i < cnt ? e0 + i : cnt - 1];   // s_load
      return r;
    };
    auto pick_row = [&](const Rows8& r, int s, int none) -> int {   // this lane's row
      const int e0 = 32 * s + 8 * g_grp;
      int mine = none;
#pragma unroll
      for (int i = 0; i < 8; ++i) mine = (g_row == i && e0 + i < cnt) ? r.v[i] : mine;
      return (dbg & 8) ? (mine & 1023) : mine;
    };
    auto rows_of = [&](const int32_t* list, int s, int none) -> int {
      return pick_row(load_rows(list, s), s, none);
    };
    auto issue = [&](int s, int ra, int rb) {
      if (dbg & 2) return;
      char* buf = smem + (s % kBuffers) * kBuf;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (h >= halvesA) break;
        const int ch = a0 + 64 * h + 8 * g_piece;
        const bool ok = ch < cin;     // channels past c_in: the zero row
        const unsigned short* src = pa + (ok ? (size_t)ra * rowA + ch : (size_t)n_in * rowA);
#pragma unroll
        for (int p = 0; p < NP; ++p)
          __builtin_amdgcn_global_load_lds((glb_void*)(src + (ok ? (size_t)p * cin : 0)),
                                           (lds_void*)(buf + ((h * NP + p) * 4 + g_grp) * 1024),
                                           16, 0, 0);
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (h >= halvesB) break;
        const int ch = b0 + 64 * h + 8 * g_piece;
        const bool ok = ch < cout;
        const unsigned short* src = pb + (ok ? (size_t)rb * rowB + ch : (size_t)n_out * rowB);
#pragma unroll
        for (int p = 0; p < NP; ++p)
          __builtin_amdgcn_global_load_lds(
              (glb_void*)(src + (ok ? (size_t)p * cout : 0)),
              (lds_void*)(buf + kSide + ((h * NP + p) * 4 + g_grp) * 1024), 16, 0, 0);
      }
    };
    const int ops = (halvesA + halvesB) * NP;     // DMA ops of this wave per stage
    auto wait_all_but = [&](int stages_in_flight) {   // newest stages allowed to be in flight
      const int n = (dbg & 2) ? 0 : stages_in_flight * ops;
      switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 2 * NP: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NP) : "memory"); break;
        case 3 * NP: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NP) : "memory"); break;
        case 4 * NP: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NP) : "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      }
    };
    int ra = rows_of(pin, 0, n_in), rb = rows_of(pout, 0, n_out);
    issue(0, ra, rb);
    if (n_stages > 1) {
      ra = rows_of(pin, 1, n_in);
      rb = rows_of(pout, 1, n_out);
      issue(1, ra, rb);
    }
    if (n_stages > 2) {
      ra = rows_of(pin, 2, n_in);
      rb = rows_of(pout, 2, n_out);
    }
    WP_MARK(0);
    for (int s = 0; s < n_stages; ++s) {
      Rows8 qa, qb;
      const bool more = s + 3 < n_stages;
      if (more) {                     // scalar loads for stage s+3: consumed at the end
        qa = load_rows(pin, s + 3);
        qb = load_rows(pout, s + 3);
      }
      // this wave's pieces of stage s have landed (stage s+1's may still be in flight)
      wait_all_but(s + 1 < n_stages ? 1 : 0);
      WP_MARK(1);
      __builtin_amdgcn_s_barrier();   // stage s visible to the consumers; they are past s-1
      WP_MARK(2);
      if (s + 2 < n_stages) {
        issue(s + 2, ra, rb);         // into the buffer stage s-1 was read from
        WP_MARK(3);
      }
      if (more) {
        ra = pick_row(qa, s + 3, n_in);
        rb = pick_row(qb, s + 3, n_out);
      }
      WP_MARK(4);
#ifdef MSMD_KERNEL_PROF
      if (lane == 0 && wave == 5 && blockIdx.x < 64) atomicAdd(&g_wprof[7], 1ull);
#endif
    }
    return;
  }

  // ================================= consumer =================================
  const int wa = wave >> 1, wb = wave & 1;
  const int hA = (nA + 1) >> 1, hB = (nB + 1) >> 1;
  const int ta0 = wa ? hA : 0, ta1 = wa ? nA : hA;
  const int tb0 = wb ? hB : 0, tb1 = wb ? nB : hB;
  const bool has_tiles = ta1 > ta0 && tb1 > tb0;
  // lane (i, g) reads rows 4q .. 4q+3 (q = 0, 1) of group g's op; byte offset, inside one
  // side of a stage buffer, of this lane's 8 bytes of tile t, plane 0:
  //   (t >> 2) * NP * 4096 + g * 1024 + row * 128 + ((32 (t & 3)) ^ (16 x(row, g))) + 8 (i & 3)
  const int m_i = lane & 15, m_g = lane >> 4;
  int offA[kWT][2], offB[kWT][2];   // loop-invariant, this wave's tiles
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int row = 4 * q + (m_i >> 2);
    const int x16 = 16 * (2 * ((row >> 1) & 1) + 4 * (m_g & 1));
    const int base = m_g * 1024 + row * 128 + 8 * (m_i & 3);
#pragma unroll
    for (int a = 0; a < kWT; ++a) {
      const int t = ta0 + a;
      offA[a][q] = (t >> 2) * (NP * 4096) + base + ((32 * (t & 3)) ^ x16);
    }
#pragma unroll
    for (int b = 0; b < kWT; ++b) {
      const int t = tb0 + b;
      offB[b][q] = kSide + (t >> 2) * (NP * 4096) + base + ((32 * (t & 3)) ^ x16);
    }
  }
  f32x4 acc[kWT][kWT];
#pragma unroll
  for (int a = 0; a < kWT; ++a)
#pragma unroll
    for (int b = 0; b < kWT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // The transposing reads are issued from inline asm: hipcc sees LDS-DMA in the kernel and
  // puts `s_waitcnt vmcnt(0)` in front of the first LDS read it knows about.  LDS returns
  // in order: counted lgkmcnt waits (this wave has no scalar loads in flight), and empty asm
  // statements tie the operand registers to the wait so no MFMA can move above it.
  auto read_half = [&](unsigned addr, int imm) -> u32x2 {   // imm: compile-time after unrolling
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(imm));
    return v;
  };
  // one stage: NA x NB tiles.  The A operands and the first B tile are waited for first;
  // the remaining B tiles land behind the MFMAs of the tiles before them.
  auto multiply = [&](unsigned buf, auto na_c, auto nb_c) {
    constexpr int NA = decltype(na_c)::value, NB = decltype(nb_c)::value;
    u32x2 ra[NA][NP][2], rb[NB][NP][2];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const unsigned addr = buf + (unsigned)offA[a][q];
        if constexpr (NP > 0) ra[a][0][q] = read_half(addr, 0);
        if constexpr (NP > 1) ra[a][1][q] = read_half(addr, 4096);
        if constexpr (NP > 2) ra[a][2][q] = read_half(addr, 8192);
      }
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const unsigned addr = buf + (unsigned)offB[b][q];
        if constexpr (NP > 0) rb[b][0][q] = read_half(addr, 0);
        if constexpr (NP > 1) rb[b][1][q] = read_half(addr, 4096);
        if constexpr (NP > 2) rb[b][2][q] = read_half(addr, 8192);
      }
    u32x4 opa[NA][NP];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      // reads still allowed in flight: those of the B tiles after b
      constexpr int per_tile = 2 * NP;
      switch ((NB - 1 - b) * per_tile) {
        case 0: asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); break;
        case 1 * per_tile: asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(1 * per_tile) : "memory"); break;
        case 2 * per_tile: asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * per_tile > 15 ? 15 : 2 * per_tile) : "memory"); break;
        default: asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(3 * per_tile > 15 ? 15 : 3 * per_tile) : "memory"); break;
      }
      if (b == 0) {
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
          for (int p = 0; p < NP; ++p) {
            asm volatile("" : "+v"(ra[a][p][0]), "+v"(ra[a][p][1]));
            opa[a][p] = (u32x4){ra[a][p][0][0], ra[a][p][0][1], ra[a][p][1][0], ra[a][p][1][1]};
          }
      }
      u32x4 opb[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        asm volatile("" : "+v"(rb[b][p][0]), "+v"(rb[b][p][1]));
        opb[p] = (u32x4){rb[b][p][0][0], rb[b][p][0][1], rb[b][p][1][0], rb[b][p][1][1]};
      }
#pragma unroll
      for (int t = 0; t < P::n; ++t)
#pragma unroll
        for (int a = 0; a < NA; ++a)
          acc[a][b] = mfma_bf16(opa[a][P::a[t]], opb[P::b[t]], acc[a][b]);
    }
  };
  const int my_na = ta1 - ta0, my_nb = tb1 - tb0;     // 0..4 each, wave-uniform
  const int shape = has_tiles && !(dbg & 4) ? my_na * 8 + my_nb : 0;
  WP_MARK(0);
  for (int s = 0; s < n_stages; ++s) {
    __builtin_amdgcn_s_barrier();     // stage s has landed
    WP_MARK(2);
    const unsigned buf = smem_base + (unsigned)((s % kBuffers) * kBuf);
#define MSMD_SHAPE(A_, B_)                                                              \
  case (A_) * 8 + (B_):                                                                  \
    multiply(buf, std::integral_constant<int, A_>{}, std::integral_constant<int, B_>{}); \
    break;
    switch (shape) {
      MSMD_SHAPE(4, 4) MSMD_SHAPE(4, 3) MSMD_SHAPE(4, 2) MSMD_SHAPE(4, 1)
      MSMD_SHAPE(3, 4) MSMD_SHAPE(3, 3) MSMD_SHAPE(3, 2) MSMD_SHAPE(3, 1)
      MSMD_SHAPE(2, 4) MSMD_SHAPE(2, 3) MSMD_SHAPE(2, 2) MSMD_SHAPE(2, 1)
      MSMD_SHAPE(1, 4) MSMD_SHAPE(1, 3) MSMD_SHAPE(1, 2) MSMD_SHAPE(1, 1)
      default: break;
    }
#undef MSMD_SHAPE
    WP_MARK(5);
  }
  // ---- epilogue: D of tile (a, b): lane (n = i, g), reg r -> ci = 16a + 4g + r, co = 16b + n
  float* dst = partial + ((size_t)k * nchunks + chunk) * cin * cout;
#pragma unroll
  for (int a = 0; a < kWT; ++a)
#pragma unroll
    for (int b = 0; b < kWT; ++b) {
      if (ta0 + a >= ta1 || tb0 + b >= tb1) continue;
      const int co = b0 + 16 * (tb0 + b) + m_i;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ci = a0 + 16 * (ta0 + a) + 4 * m_g + r;
        if (ci < cin && co < cout) dst[(size_t)ci * cout + co] = acc[a][b][r];
      }
    }
}

// output rows per chunk (MSMD_WGRAD_ROWS overrides, experiments)
inline int rows_per_chunk() {   // <= 2048: a workgroup keeps 64 stages of indices in registers
  static const int v = [] {
    const char* e = getenv("MSMD_WGRAD_ROWS");
    const int r = e ? atoi(e) : 2048;
    return r >= 64 && r <= 2048 ? r : 2048;
  }();
  return v;
}
#define kRowsPerChunk rows_per_chunk()

template <int NP>
int launch_wgrad_planes(const void* pa, int cin, int n_in, const void* pb, int cout, int n_out,
                        const int32_t* pairs, const int32_t* ranges, int ld, int nchunks, int kvol,
                        float* ws, hipStream_t st) {
  constexpr size_t smem = (size_t)kBuffers * 2 * (2 * NP * 4096);
  auto kern = spconv_wgrad_planes_kernel<NP>;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)smem);
    attr_done = true;
  }
  const int slabs = ceil_div(cin, kSlab) * ceil_div(cout, kSlab);
  static const int dbg = [] { const char* e = getenv("MSMD_DBG"); return e ? atoi(e) : 0; }();
  MSMD_LAUNCH(kern, dim3(wgrad_grid(nchunks, kvol, slabs)), dim3(512), smem, st,
              (const unsigned short*)pa, cin, n_in, (const unsigned short*)pb, cout, n_out, pairs,
              ranges, ld, nchunks, kvol, ws, dbg);
  return launch_status();
}

}  // namespace
}  // namespace msmd

using namespace msmd;

#ifdef MSMD_KERNEL_PROF
// out[16] <- phase cycle sums of the wgrad plane kernel (then cleared): 0 prologue, 1 DMA
// wait, 2 barrier, 3 DMA issue, 4 row indices, 5 LDS reads + MFMAs, 7 stages
MSMD_EXPORT int msmd_debug_wprof(unsigned long long* out) {
  (void)hipDeviceSynchronize();
  unsigned long long z[16] = {0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wprof), sizeof(z)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_wprof), z, sizeof(z)) != hipSuccess) return -1;
  return 0;
}
#endif

MSMD_EXPORT size_t msmd_planes_bytes(int n_rows, int channels, int planes) {
  return (size_t)(n_rows > 0 ? n_rows + 1 : 1) * (size_t)planes * (size_t)channels * 2;
}

MSMD_EXPORT int msmd_split_planes_f32(const float* x, int n_rows, int channels, int planes,
                                      void* out, msmd_stream_t stream) {
  if (n_rows < 0 || channels < 8 || (channels & 7) || planes < 1 || planes > 3 || !out ||
      (n_rows > 0 && !x))
    return MSMD_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  const long total = (long)(n_rows + 1) * (channels >> 3);
  int nblk = (int)((total + 255) / 256);
  if (nblk > 8192) nblk = 8192;
  if (planes == 3)
    MSMD_LAUNCH(split_planes_kernel<3>, dim3(nblk), dim3(256), 0, st, x, n_rows, channels,
                (u32x4*)out);
  else if (planes == 2)
    MSMD_LAUNCH(split_planes_kernel<2>, dim3(nblk), dim3(256), 0, st, x, n_rows, channels,
                (u32x4*)out);
  else
    MSMD_LAUNCH(split_planes_kernel<1>, dim3(nblk), dim3(256), 0, st, x, n_rows, channels,
                (u32x4*)out);
  return launch_status();
}

MSMD_EXPORT int msmd_spconv_wgrad_planes_supported(int c_in, int c_out) {
  // 16-byte pieces of 8 bf16 channels; below 64 channels the fp32 kernel's narrow slabs win
  return c_in >= 64 && c_out >= 64 && c_in % 8 == 0 && c_out % 8 == 0;
}

namespace {
struct PlanesWs {
  int32_t* ranges;
  float* partial;
};
template <typename A>
void carve_planes(A& a, PlanesWs* w, int kvol, int nchunks, int per_k) {
  int32_t* r = a.template take<int32_t>((size_t)kvol * (nchunks + 1));
  float* p = a.template take<float>((size_t)kvol * nchunks * per_k);
  if (w) *w = PlanesWs{r, p};
}
}  // namespace

MSMD_EXPORT size_t msmd_spconv_wgrad_planes_workspace_bytes(int kernel_volume, int n_out, int c_in,
                                                            int c_out) {
  if (kernel_volume < 1 || n_out < 0 || c_in < 1 || c_out < 1) return 0;
  ArenaSize a;
  carve_planes(a, (PlanesWs*)nullptr, kernel_volume, ceil_div(n_out > 0 ? n_out : 1, kRowsPerChunk),
               c_in * c_out);
  return a.off;
}

MSMD_EXPORT int msmd_spconv_wgrad_planes(const void* in_planes, int n_in, int c_in,
                                         const void* dout_planes, int n_out, int c_out,
                                         const int32_t* indice_pairs, const int32_t* indice_num,
                                         int ld, int kernel_volume, int planes, float* d_weight,
                                         int krsc_out, void* workspace, size_t workspace_bytes,
                                         msmd_stream_t stream) {
  if (!msmd_spconv_wgrad_planes_supported(c_in, c_out) || planes < 1 || planes > 3)
    return MSMD_ERR_UNSUPPORTED;
  if (kernel_volume < 1 || ld < 0 || n_in < 0 || n_out < 0 || !d_weight || !indice_num)
    return MSMD_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int per_k = c_in * c_out;
  if (ld == 0 || n_out == 0) {
    hipMemsetAsync(d_weight, 0, sizeof(float) * (size_t)kernel_volume * per_k, st);
    return launch_status();
  }
  if (!in_planes || !dout_planes || !indice_pairs) return MSMD_ERR_INVALID_ARG;
  const int nchunks = ceil_div(n_out, kRowsPerChunk);
  Arena a(workspace, workspace_bytes);
  PlanesWs w;
  carve_planes(a, &w, kernel_volume, nchunks, per_k);
  if (!a.ok()) return MSMD_ERR_WORKSPACE;
  MSMD_LAUNCH(pair_ranges_kernel, dim3(ceil_div(kernel_volume * (nchunks + 1), 256)), dim3(256), 0,
              st, indice_pairs, indice_num, ld, kernel_volume, kRowsPerChunk, nchunks, w.ranges);
  int rc;
  if (planes == 3)
    rc = launch_wgrad_planes<3>(in_planes, c_in, n_in, dout_planes, c_out, n_out, indice_pairs,
                                w.ranges, ld, nchunks, kernel_volume, w.partial, st);
  else if (planes == 2)
    rc = launch_wgrad_planes<2>(in_planes, c_in, n_in, dout_planes, c_out, n_out, indice_pairs,
                                w.ranges, ld, nchunks, kernel_volume, w.partial, st);
  else
    rc = launch_wgrad_planes<1>(in_planes, c_in, n_in, dout_planes, c_out, n_out, indice_pairs,
                                w.ranges, ld, nchunks, kernel_volume, w.partial, st);
  if (rc != MSMD_OK) return rc;
  int rb = ceil_div(per_k, 256);
  if (rb > 64) rb = 64;
  MSMD_LAUNCH(reduce_ranges_kernel, dim3(rb, kernel_volume), dim3(256), 0, st,
              (const float*)w.partial, (const int32_t*)w.ranges, nchunks, per_k, c_in, c_out,
              kernel_volume, krsc_out, d_weight);
  return launch_status();
}
