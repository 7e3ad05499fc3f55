// spconv.hip -- sparse convolution arithmetic on the gfx950 matrix cores.
//
//   out[o,:] = sum_k in[nbr[k,o],:] @ W[k]        (indiceConv, spconv_ops.h:260-361)
//
// The reference runs 27 x (gather kernel -> cuBLAS GEMM -> scatter-add kernel)
// per convolution.  Here one kernel does the whole convolution as an
// OUTPUT-STATIONARY implicit GEMM on v_mfma_f32_16x16x4_f32 (exact fp32, a
// k-ordered fmaf chain -- the 1e-4 parity bound holds with ~1e-6 to spare):
//
//   * a workgroup owns ROWS = 4 waves x R x 16 output rows and all of c_out;
//     accumulators stay in registers across all K offsets, each output row is
//     written once (no atomics, no scatter-add, deterministic);
//   * the MFMA is issued transposed: A = W fragment, B = gathered input rows,
//     so D[cout][row] leaves every lane with 4 consecutive output channels of
//     one row -> one 16-byte store per lane per 16-channel tile;
//   * the contraction index is permuted so that a lane's B operands for four
//     consecutive MFMAs are ONE float4 of its gathered input row
//     (channels 16t+4q .. +3): the gather is 16-byte loads straight from
//     HBM/L2 into MFMA operand registers, no LDS round trip, and the weights
//     are pre-packed (msmd_spconv_pack_weight) in exactly the order the A
//     operand wants them, so the LDS image is filled by straight 16-byte
//     copies and read with one conflict-free ds_read_b128 per 4 MFMAs;
//   * a wave skips the MFMAs of an offset none of its rows is connected by
//     (wave-uniform branch on a ballot).
//
// dgrad is the same kernel on the backward table with W[k]^T packed; wgrad
// (spconv_ops.h:399,438) contracts over the compact pair lists.
#include "common.hpp"
#include "tiling_key.hpp"

#include <stdlib.h>
#include <string.h>

namespace msmd {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kTC = 4;  // 16-channel steps of c_in staged per LDS fill (64 channels)

// ------------------------------------------------------------- packing ----
// packed[((k*T + t)*NT + n)*256 + l*4 + s] = W[k][16t + 4(l>>4) + s][16n + (l&15)]
// (zero outside c_in x c_out).  transpose: W[k] is read as W[k][col][row].
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, int kvol,
                                                          int cin, int cout, int flags,
                                                          float* __restrict__ packed) {
  const int transpose = flags & 1, krsc = flags & 2;  // krsc: w is [c_out][K][c_in]
  const int ci = transpose ? cout : cin, co = transpose ? cin : cout;  // effective dims
  const int T = (ci + 15) / 16, NT = (co + 15) / 16;
  long total = (long)kvol * T * NT * 256;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    int s = e & 3, l = (e >> 2) & 63;
    long tile = e >> 8;
    int n = tile % NT;
    int t = (tile / NT) % T;
    int k = tile / ((long)NT * T);
    int c = 16 * t + 4 * (l >> 4) + s, d = 16 * n + (l & 15);
    float v = 0.f;
    if (c < ci && d < co) {
      const int wi = transpose ? d : c, wo = transpose ? c : d;  // (c_in, c_out) index of W[k]
      v = krsc ? w[((size_t)wo * kvol + k) * cin + wi] : w[((size_t)k * cin + wi) * cout + wo];
    }
    packed[e] = v;
  }
}

// --------------------------------------------------------- forward/dgrad --
// NT: 16-wide output-channel tiles; R: 16-row groups per wave; VEC: c_in % 4 == 0.
// `order` (optional) is a permutation of the output rows: tile position p
// computes output row order[p].  The host passes rows sorted by their 27-bit
// neighbour mask, so the 32 rows of a wave (and the 128 rows of a workgroup)
// share their empty offsets and the wave-/block-uniform skips below remove
// most of the structural-zero MFMA work (measured on the synthetic cloud:
// issued/useful MFMA work 2.3x -> 1.3x at the 64-channel stage).
template <int NT, int R, bool VEC>
__global__ __launch_bounds__(256) void spconv_fwd_kernel(
    const float* __restrict__ in, int cin, const float* __restrict__ wp,
    const int32_t* __restrict__ nbr, int ld, int n_out, int kvol, int flip,
    const int32_t* __restrict__ order, int* __restrict__ tile_counter, float* __restrict__ out,
    int cout, int dbg) {
  __shared__ f32x4 wl[kTC * NT * 64];
  __shared__ int s_tile;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, q = lane >> 4;
  const int T = (cin + 15) / 16;
  constexpr int kTileRows = 4 * R * 16;
  const int n_tiles = (n_out + kTileRows - 1) / kTileRows;

  // Persistent workgroups pull tiles from a global counter.  With rows sorted
  // heaviest-mask-first the tiles arrive in decreasing cost, so this is
  // longest-processing-time-first list scheduling: CUs that drew cheap tiles
  // simply draw more (a static grid left the MFMA pipes idle ~45 % of the time
  // on the 128-channel layers because tile cost varies 3x with the mask).
  for (;;) {
    if (threadIdx.x == 0) {
      int t = (int)blockIdx.x;
      if (tile_counter) {
        t = atomicAdd(tile_counter, 1);
        // n_tiles + gridDim.x draws in total: the last one re-arms the counter
        if (t == n_tiles + (int)gridDim.x - 1) *tile_counter = 0;
      }
      s_tile = t;
    }
    __syncthreads();
    const int tile = s_tile;
    if (tile >= n_tiles) break;
    const int row0 = (tile * 4 + wave) * (R * 16);

    f32x4 acc[R][NT];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[r][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int rows[R];  // output row of this lane's column j in group r, or -1
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int p = row0 + r * 16 + j;
      rows[r] = p < n_out ? (order ? order[p] : p) : -1;
    }

    for (int k = 0; k < kvol; ++k) {
      const int kw = flip ? kvol - 1 - k : k;
      int src[R];
      bool any = false;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        src[r] = rows[r] >= 0 ? nbr[(size_t)k * ld + rows[r]] : -1;
        any |= src[r] >= 0;
      }
      const bool wave_any = __any(any);
      // block-uniform: nobody needs this offset -> no weight staging, no MFMAs
      // (the barrier also fences the previous offset's reads of wl / s_tile)
      if (!__syncthreads_or(wave_any)) continue;
      for (int t0 = 0; t0 < T; t0 += kTC) {
        const int tc = (T - t0) < kTC ? (T - t0) : kTC;
        // ---- stage W[kw][t0 .. t0+tc) : tc*NT*64 float4, straight copy ----
        if (t0 > 0) __syncthreads();
        if (!(dbg & 2)) {
          const f32x4* g = (const f32x4*)wp + ((size_t)kw * T + t0) * NT * 64;
          for (int e = threadIdx.x; e < tc * NT * 64; e += 256) wl[e] = g[e];
        }
        // ---- gather this wave's input rows (overlaps the fill) ----
        f32x4 b[R][kTC];
        if (wave_any) {
#pragma unroll
          for (int r = 0; r < R; ++r)
#pragma unroll
            for (int t = 0; t < kTC; ++t) {
              f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
              const int c0 = 16 * (t0 + t) + 4 * q;
              if (t < tc && src[r] >= 0 && !(dbg & 1)) {
                const float* p = in + (size_t)src[r] * cin + c0;
                if (VEC) {
                  if (c0 < cin) v = *(const f32x4*)p;
                } else {
#pragma unroll
                  for (int s = 0; s < 4; ++s)
                    if (c0 + s < cin) v[s] = p[s];
                }
              }
              b[r][t] = v;
            }
        }
        __syncthreads();
        if (wave_any) {
#pragma unroll
          for (int t = 0; t < kTC; ++t) {
            if (t < tc) {
#pragma unroll
              for (int n = 0; n < NT; ++n) {
                const f32x4 a = wl[(t * NT + n) * 64 + lane];
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                  for (int r = 0; r < R; ++r)
                    acc[r][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[r][t][s],
                                                                     acc[r][n], 0, 0, 0);
              }
            }
          }
        }
      }
    }
    // ---- epilogue: lane (j,q) holds out[row j][16n + 4q .. +3] ----
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (rows[r] < 0) continue;
      float* o = out + (size_t)rows[r] * cout;
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int c0 = 16 * n + 4 * q;
        if ((cout & 3) == 0) {
          if (c0 < cout) *(f32x4*)(o + c0) = acc[r][n];
        } else {
#pragma unroll
          for (int s = 0; s < 4; ++s)
            if (c0 + s < cout) o[c0 + s] = acc[r][n][s];
        }
      }
    }
    if (!tile_counter) break;
    __syncthreads();  // every thread has read s_tile / wl before the next draw
  }
}

// Tuning constants chosen from on-device sweeps (round-1 sweep script, since pruned: git history; the environment
// overrides of rounds 1-2 -- MSMD_FWD_SLOTS / _R / _PIPE / _KC, MSMD_PIPE_MIN_NT,
// MSMD_NARROW_ORDER, MSMD_WGRAD_MULTISLAB -- are gone: measured, decided).
inline int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}
inline int fwd_slots_per_cu() { return 3; }
inline int fwd_rows_variant() { return 0; }   // 0 = per-NT default

// ---------------------------------------------------- pipelined forward ----
// Same math and tiling as spconv_fwd_kernel, restructured so that no memory
// latency sits on the MFMA critical path (PMC on the unpipelined kernel: MFMA
// pipe 41 % busy, waves 36 % in s_waitcnt/barrier -- every offset paid one
// dependent nbr load, one weight copy and one row gather round trip):
//   * the tile's whole neighbour table slice nbr[0..K)[rows] is staged in LDS
//     once (one latency), which also yields the tile's list of active offsets;
//   * work items = (active offset, 64-channel chunk).  While item i's MFMAs
//     run out of LDS buffer i&1, item i+1's packed weights (global -> regs) and
//     gathered input rows (global -> MFMA operand regs) are already in flight;
//     one barrier per item (double-buffered weights);
//   * R == 1 interleaves two output tiles per weight step so consecutive
//     MFMAs never hit the same accumulator (40-cycle dependent latency vs
//     32-cycle issue on v_mfma_f32_16x16x4_f32).
constexpr int kMaxK = 32;

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// NT: c_out/16 tiles, R: 16-row groups per wave, KC: 16-channel steps per item
// (requires c_in % (16*KC) == 0 and c_out % 4 == 0: no tails, straight-line code).
template <int NT, int R, int KC>
__global__ __launch_bounds__(256) void spconv_fwd_pipe_kernel(
    const float* __restrict__ in, int cin, const float* __restrict__ wp,
    const int32_t* __restrict__ nbr, int ld, int n_out, int kvol, int flip,
    const int32_t* __restrict__ order, int* __restrict__ tile_counter, float* __restrict__ out,
    int cout) {
  constexpr int kRows = 4 * R * 16;
  constexpr int kPieces = KC * NT;      // 1-KiB (64 x float4) pieces per weight chunk
  constexpr int kWF4 = kPieces * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* wl = (f32x4*)smem;                      // [2][kWF4]
  int* nb = (int*)(wl + 2 * kWF4);               // [kMaxK][kRows]
  int* srow = nb + kMaxK * kRows;                // [kRows]
  int* act = srow + kRows;                       // [kMaxK]
  int* klist = act + kMaxK;                      // [kMaxK]
  int* sctl = klist + kMaxK;                     // [0] tile, [1] #active offsets
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, q = lane >> 4;
  const int nchunk = cin / (16 * KC);
  const int n_tiles = (n_out + kRows - 1) / kRows;
  const size_t chunk_f4 = (size_t)kWF4;          // packed weights: [k][chunk][kWF4]

  for (;;) {
    if (tid == 0) {
      int t = (int)blockIdx.x;
      if (tile_counter) {
        t = atomicAdd(tile_counter, 1);
        // n_tiles + gridDim.x draws in total: the last one re-arms the counter
        if (t == n_tiles + (int)gridDim.x - 1) *tile_counter = 0;
      }
      sctl[0] = t;
    }
    if (tid < kMaxK) act[tid] = 0;
    __syncthreads();
    const int tile = sctl[0];
    if (tile >= n_tiles) break;
    if (tid < kRows) {
      const int p = tile * kRows + tid;
      srow[tid] = p < n_out ? (order ? order[p] : p) : -1;
    }
    __syncthreads();
    for (int e = tid; e < kvol * kRows; e += 256) {
      const int k = e / kRows, rr = e - k * kRows;
      const int row = srow[rr];
      const int v = row >= 0 ? nbr[(size_t)k * ld + row] : -1;
      nb[k * kRows + rr] = v;
      if (v >= 0) act[k] = 1;
    }
    __syncthreads();
    if (tid == 0) {
      int n = 0;
      for (int k = 0; k < kvol; ++k)
        if (act[k]) klist[n++] = k;
      sctl[1] = n;
    }
    __syncthreads();
    const int n_items = sctl[1] * nchunk;

    f32x4 acc[R][NT];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int n = 0; n < NT; ++n) acc[r][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int lr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) lr[r] = (wave * R + r) * 16 + j;

    // Start item `it`: weights by LDS-DMA into buffer it&1 (no registers, no
    // ds_write pass), gathered rows into the operand registers `b`.
    auto issue = [&](int it, f32x4 (&b)[R][KC], int (&valid)[R]) {
      const int k = klist[it / nchunk];
      const int ch = it % nchunk;
      const int kw = flip ? kvol - 1 - k : k;
      const f32x4* g = (const f32x4*)wp + ((size_t)kw * nchunk + ch) * chunk_f4;
      f32x4* wb = wl + (it & 1) * kWF4;
#pragma unroll
      for (int p = 0; p < (kPieces + 3) / 4; ++p) {
        const int piece = wave + 4 * p;
        if (kPieces % 4 == 0 || piece < kPieces)
          __builtin_amdgcn_global_load_lds((glb_void*)(g + piece * 64 + lane),
                                           (lds_void*)(wb + piece * 64), 16, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int src = nb[k * kRows + lr[r]];
        valid[r] = src;
        const float* row = in + (size_t)(src < 0 ? 0 : src) * cin + ch * (16 * KC) + 4 * q;
#pragma unroll
        for (int t = 0; t < KC; ++t) b[r][t] = *(const f32x4*)(row + 16 * t);
      }
    };
    auto compute = [&](int it, f32x4 (&b)[R][KC], const int (&valid)[R]) {
      bool any = false;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        any |= valid[r] >= 0;
        if (valid[r] < 0) {
#pragma unroll
          for (int t = 0; t < KC; ++t) b[r][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
      if (!__any(any)) return;
      const f32x4* wb = wl + (it & 1) * kWF4;
#pragma unroll
      for (int t = 0; t < KC; ++t) {
        if (R >= 2) {
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            const f32x4 a = wb[(t * NT + n) * 64 + lane];
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
              for (int r = 0; r < R; ++r)
                acc[r][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[r][t][s], acc[r][n], 0,
                                                                 0, 0);
          }
        } else {
#pragma unroll
          for (int n = 0; n < NT; n += 2) {
            const f32x4 a0 = wb[(t * NT + n) * 64 + lane];
            const f32x4 a1 = wb[(t * NT + (n + 1 < NT ? n + 1 : n)) * 64 + lane];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              acc[0][n] =
                  __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], b[0][t][s], acc[0][n], 0, 0, 0);
              if (n + 1 < NT)
                acc[0][n + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], b[0][t][s],
                                                                     acc[0][n + 1], 0, 0, 0);
            }
          }
        }
      }
    };
    // one pipeline step: wait for item `it`'s data, start item it+1, compute it
    f32x4 b0[R][KC], b1[R][KC];
    int v0[R], v1[R];
#define MSMD_STEP(IT, BC, VC, BN, VN)                          \
  {                                                            \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           \
    __syncthreads();                                           \
    if ((IT) + 1 < n_items) issue((IT) + 1, BN, VN);           \
    compute((IT), BC, VC);                                     \
  }
    if (n_items > 0) issue(0, b0, v0);
    for (int it = 0; it < n_items; it += 2) {
      MSMD_STEP(it, b0, v0, b1, v1);
      if (it + 1 < n_items) MSMD_STEP(it + 1, b1, v1, b0, v0);
    }
#undef MSMD_STEP
    // ---- epilogue: lane (j,q) holds out[row j][16n + 4q .. +3] ----
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int row = srow[lr[r]];
      if (row < 0) continue;
      float* o = out + (size_t)row * cout + 4 * q;
#pragma unroll
      for (int n = 0; n < NT; ++n)
        if (16 * n + 4 * q < cout) *(f32x4*)(o + 16 * n) = acc[r][n];
    }
    if (!tile_counter) break;
    __syncthreads();  // LDS tables are reused by the next tile
  }
}

template <int NT, int R, int KC>
int launch_fwd_pipe(const float* in, int cin, const float* wp, const int32_t* nbr, int ld,
                    int n_out, int kvol, int flip, const int32_t* order, int* tile_counter,
                    float* out, int cout, hipStream_t st) {
  constexpr int kRows = 4 * R * 16;
  const size_t smem = sizeof(f32x4) * 2 * KC * NT * 64 +
                      sizeof(int) * ((size_t)kMaxK * kRows + kRows + 2 * kMaxK + 8);
  const int n_tiles = ceil_div(n_out, kRows);
  int nblk = n_tiles;
  if (tile_counter) {
    const int slots = 256 * fwd_slots_per_cu();
    if (nblk > slots) nblk = slots;
  }
  auto kern = spconv_fwd_pipe_kernel<NT, R, KC>;
  static LdsGrant granted;  // per instantiation
  const int lds_rc = optin_dynamic_lds((const void*)kern, smem, granted);
  if (lds_rc != MSMD_OK) return lds_rc;
  MSMD_LAUNCH(kern, dim3(nblk), dim3(256), smem, st, in, cin, wp, nbr, ld, n_out, kvol, flip,
              order, tile_counter, out, cout);
  return launch_status();
}

// Largest chunk (in 16-channel steps) that divides c_in and keeps two weight
// buffers within LDS: KC in {4, 2, 1}.
template <int NT, int R>
int dispatch_fwd_pipe(const float* in, int cin, const float* wp, const int32_t* nbr, int ld,
                      int n_out, int kvol, int flip, const int32_t* order, int* tile_counter,
                      float* out, int cout, hipStream_t st) {
  const int T = cin / 16;
  if (NT <= 8 && T % 4 == 0)
    return launch_fwd_pipe<NT, R, 4>(in, cin, wp, nbr, ld, n_out, kvol, flip, order,
                                     tile_counter, out, cout, st);
  if (T % 2 == 0)
    return launch_fwd_pipe<NT, R, 2>(in, cin, wp, nbr, ld, n_out, kvol, flip, order,
                                     tile_counter, out, cout, st);
  return launch_fwd_pipe<NT, R, 1>(in, cin, wp, nbr, ld, n_out, kvol, flip, order, tile_counter,
                                   out, cout, st);
}

template <int NT, int R>
int launch_fwd(const float* in, int cin, const float* wp, const int32_t* nbr, int ld, int n_out,
               int kvol, int flip, const int32_t* order, int* tile_counter, float* out, int cout,
               hipStream_t st) {
  // narrow layers too (c_in % 16 == 0): 16->16 34 -> 21 us, strided 16->32 65 -> 47, bit-identical
  if ((cin & 15) == 0 && (cout & 3) == 0 && kvol <= kMaxK)
    return dispatch_fwd_pipe<NT, R>(in, cin, wp, nbr, ld, n_out, kvol, flip, order, tile_counter,
                                    out, cout, st);
  const int rows_per_block = 4 * R * 16;
  const int n_tiles = ceil_div(n_out, rows_per_block);
  int nblk = n_tiles;
  if (tile_counter) {
    const int slots = 256 * fwd_slots_per_cu();
    if (nblk > slots) nblk = slots;
  }
  dim3 grid(nblk);
  static const int dbg = env_int("MSMD_DBG", 0);  // ablation bits, experiments only
  if ((cin & 3) == 0)
    MSMD_LAUNCH((spconv_fwd_kernel<NT, R, true>), grid, dim3(256), 0, st, in, cin, wp, nbr, ld,
                n_out, kvol, flip, order, tile_counter, out, cout, dbg);
  else
    MSMD_LAUNCH((spconv_fwd_kernel<NT, R, false>), grid, dim3(256), 0, st, in, cin, wp, nbr, ld,
                n_out, kvol, flip, order, tile_counter, out, cout, dbg);
  return launch_status();
}

// Neighbour mask of every output row: bit k set when nbr[k][row] >= 0 (K <= 64),
// and a sort key that makes rows with similar masks adjacent.
//   K == 27 (3x3x3): the mask with its bits re-ranked -- centre lowest, then the 6 face
//   neighbours, the 12 edge ones, the 8 corners highest (z before y before x inside a
//   class): rows that share their RARE offsets end up in the same wave / tile.  The
//   heaviest-first sequence the persistent scheduler wants is then imposed on whole
//   tiles (tile_cost_kernel), not on rows.  Simulated on the bench workload's
//   128-channel stage (tools/order_sim.py): items a workgroup walks / useful work
//   1.414 with the previous key ((27 - popcount) << 27 | mask) -> 1.299.
//   other K <= 31: (K - popcount) << K | mask; larger: the raw mask.
// (kRank27: tiling_key.hpp)
__global__ __launch_bounds__(256) void row_mask_kernel(const int32_t* __restrict__ nbr, int kvol,
                                                       int n, unsigned long long* __restrict__ m,
                                                       long long* __restrict__ key) {
  int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= n) return;
  unsigned long long v = 0, ranked = 0;
  for (int k = 0; k < kvol; ++k)
    if (nbr[(size_t)k * n + o] >= 0) {
      v |= 1ull << k;
      if (kvol == 27) ranked |= 1ull << kRank27[k];
    }
  if (m) m[o] = v;
  if (key)
    key[o] = kvol == 27   ? (long long)ranked
             : kvol <= 31 ? (long long)(((unsigned long long)(kvol - __popcll(v)) << kvol) | v)
                          : (long long)(v >> 1);
}

// The same key in 32 bits (K <= 31), for the in-library radix sort (tiling.hip).
__global__ __launch_bounds__(256) void row_key32_kernel(const int32_t* __restrict__ nbr, int kvol,
                                                        int n, uint32_t* __restrict__ key,
                                                        int32_t* __restrict__ row_ids) {
  int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= n) return;
  if (row_ids) row_ids[o] = o;          // the sort's payload (was a launch of its own)
  // (the Gray code of the ranked mask simulates 1.3 % better, 1.299 -> 1.282 items / useful
  // work; measured within noise, not used)
  key[o] = row_key32(nbr, kvol, (size_t)n, o);
}

// cost[t] = K - |union of the masks of tile t's rows| (ascending = heaviest first),
// tile t = positions [t * rows, (t+1) * rows) of `order`.  One block per tile.
__global__ __launch_bounds__(128) void tile_cost_kernel(const int32_t* __restrict__ nbr, int kvol,
                                                        int n, const int32_t* __restrict__ order,
                                                        int rows, int32_t* __restrict__ cost,
                                                        int32_t* __restrict__ tile_ids) {
  __shared__ unsigned long long u;
  if (threadIdx.x == 0) u = 0;
  __syncthreads();
  unsigned long long v = 0;
  for (int p = blockIdx.x * rows + threadIdx.x; p < (blockIdx.x + 1) * rows && p < n;
       p += blockDim.x) {
    const int o = order ? order[p] : p;
    for (int k = 0; k < kvol; ++k)
      if (nbr[(size_t)k * n + o] >= 0) v |= 1ull << k;
  }
  if (v) atomicOr(&u, v);
  __syncthreads();
  if (threadIdx.x == 0) {
    cost[blockIdx.x] = kvol - __popcll(u);
    if (tile_ids) tile_ids[blockIdx.x] = blockIdx.x;      // the tile sort's payload
  }
}

// ------------------------------------------------------------------ wgrad --
// dW[k] = sum_p in[i_p,:]^T (x) dout[o_p,:] over the compact pairs of offset k.
// MFMA 16x16x4 with the pair index as the contraction: A[ci][p], B[p][co], four
// pairs per instruction.  A workgroup = (CHUNK-pair chunk, offset k, slab of
// 16*SA x 16*SB channels); its 4 waves take interleaved groups of 4 pairs.
//   * the chunk's (in,out) row indices are staged in LDS once (coalesced);
//   * operands are ONE load of SA (SB) consecutive floats per lane per side:
//     lane (i,q) loads channels SA*i .. SA*i+SA-1 of pair q's row, and the slab's
//     16x16 tiles are defined over the permuted channel order
//         tile a, row i  <->  channel SA*i + a
//     so element a of that load IS the lane's A operand of tile a (same for B):
//     for 64-channel sides (SA = 4) 2 x 16-byte loads (256 B contiguous per row)
//     feed 16 MFMAs, no LDS for activations; narrow layers take SA/SB = 2 or 1
//     (8-/4-byte loads) so no matrix work is spent on channels that do not exist;
//   * the next groups' operands are fetched before the current MFMAs issue;
//   * the 4 wave partials are summed through LDS in fixed order and written to
//     a per-(k,chunk) partial; a second kernel reduces the partials in fixed
//     order (deterministic, no float atomics).
// CHUNK: 2048 pairs for the wide layers, 1024 / 512 for the narrow ones, whose
// full-resolution voxel sets have few pairs per offset (a 2048-pair chunking
// left half the CUs without a workgroup).
// Pairs per workgroup for an offset with `pairs` pairs.  (Doubling the chunk for
// pair-rich offsets halves the partials of the second pass but was measured
// slower overall: 558 us against 516 us on the 128x128 layers -- fewer, longer
// workgroups leave a longer tail.)
__host__ __device__ inline int wgrad_span(int pairs, int chunk) { return chunk; }

template <int S>
struct VecOf;
template <>
struct VecOf<1> { typedef float type; };
template <>
struct VecOf<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <>
struct VecOf<4> { typedef float type __attribute__((ext_vector_type(4))); };


template <int S, bool VEC>
__device__ __forceinline__ void load_side(const float* row, int c0, int c, float (&v)[S]) {
  if (VEC) {
    if (c0 < c) {
      const typename VecOf<S>::type t = *(const typename VecOf<S>::type*)(row + c0);
      if constexpr (S == 1) {
        v[0] = t;
      } else {
#pragma unroll
        for (int s = 0; s < S; ++s) v[s] = t[s];
      }
    }
  } else {
#pragma unroll
    for (int s = 0; s < S; ++s)
      if (c0 + s < c) v[s] = row[c0 + s];
  }
}

// WPS = waves per slab: 4 -- the workgroup owns one slab, its waves split the
// pairs; 2 / 1 -- it owns 2 / 4 slabs (same c_in rows first), every pair group is
// multiplied by 2 / 4 waves into different slabs: the second wave's gather of a
// row hits L1, so the L2 traffic of a 128x128 layer halves.
template <int SA, int SB, int CHUNK, bool VEC, int WPS>  // VEC: c_in % SA == 0 && c_out % SB == 0
__global__ __launch_bounds__(256) void spconv_wgrad_kernel(
    const float* __restrict__ in, int cin, const float* __restrict__ dout, int cout,
    const int32_t* __restrict__ pairs, const int32_t* __restrict__ num, int ld, int nchunks,
    int kvol, int slab_groups, float* __restrict__ partial /* [K][nchunks][cin][cout] */) {
  // LDS: the chunk's pair indices during the main loop, then the tree reduction
  // of the four wave partials (aliased)
  constexpr int kRedBytes = 2 * SA * SB * 64 * (int)sizeof(f32x4);
  constexpr int kIdxBytes = 2 * (2 * CHUNK) * (int)sizeof(int);   // chunks may be doubled
  __shared__ __attribute__((aligned(16))) char lds_raw[kRedBytes > kIdxBytes ? kRedBytes : kIdxBytes];
  int* s_in = (int*)lds_raw;
  int* s_out = s_in + 2 * CHUNK;
  f32x4* red = (f32x4*)lds_raw;
  int k, chunk, zz;
  if (!wgrad_work(nchunks, kvol, slab_groups, chunk, k, zz)) return;
  const int P = num[k];
  const int span = wgrad_span(P, CHUNK);   // pairs per workgroup for this offset
  const int p_begin = chunk * span;
  if (p_begin >= P) return;
  const int cnt = (P - p_begin) < span ? (P - p_begin) : span;
  const int NTs = (cout + 16 * SB - 1) / (16 * SB);  // slabs along c_out
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slab = zz * (4 / WPS) + wave / WPS, rank = wave % WPS;
  const int sa = slab / NTs, sb = slab % NTs;
  const int a0 = sa * SA * 16, b0 = sb * SB * 16;   // first channel of the slab
  const int i = lane & 15, q = lane >> 4;
  {
    const int32_t* pin = pairs + ((size_t)k * 2 + 0) * ld + p_begin;
    const int32_t* pout = pairs + ((size_t)k * 2 + 1) * ld + p_begin;
    for (int e = threadIdx.x; e < cnt; e += 256) {
      s_in[e] = pin[e];
      s_out[e] = pout[e];
    }
  }
  __syncthreads();

  f32x4 acc[SA][SB];
#pragma unroll
  for (int a = 0; a < SA; ++a)
#pragma unroll
    for (int b = 0; b < SB; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int ca = a0 + SA * i, cb = b0 + SB * i;  // this lane's channels on each side
  auto fetch = [&](int e, float (&av)[SA], float (&bv)[SB]) {
#pragma unroll
    for (int s = 0; s < SA; ++s) av[s] = 0.f;
#pragma unroll
    for (int s = 0; s < SB; ++s) bv[s] = 0.f;
    if (e < cnt) {
      load_side<SA, VEC>(in + (size_t)s_in[e] * cin, ca, cin, av);
      load_side<SB, VEC>(dout + (size_t)s_out[e] * cout, cb, cout, bv);
    }
  };
  // the WPS waves of a slab take pair groups rank, rank+WPS, ...; a group = 4
  // consecutive pairs (q).
  // Row gathers come from L2/MALL (~2 us under load) while a group is at most 16
  // MFMAs (512 cycles): keep kDepth groups in flight in a register ring.
  constexpr int kDepth = SA * SB >= 32 ? 4 : 8;   // (same MFMA time in flight either way)
  float ra[kDepth][SA], rb[kDepth][SB];
  constexpr int kStep = 4 * WPS;   // pairs between a wave's consecutive groups
  int e = 4 * rank + q;
#pragma unroll
  for (int d = 0; d < kDepth; ++d) fetch(e + kStep * d, ra[d], rb[d]);
  for (int g = 4 * rank; g < cnt; g += kStep * kDepth) {
#pragma unroll
    for (int d = 0; d < kDepth; ++d) {
      float av[SA], bv[SB];
#pragma unroll
      for (int s = 0; s < SA; ++s) av[s] = ra[d][s];
#pragma unroll
      for (int s = 0; s < SB; ++s) bv[s] = rb[d][s];
      fetch(e + kStep * (d + kDepth), ra[d], rb[d]);
#pragma unroll
      for (int a = 0; a < SA; ++a)
#pragma unroll
        for (int b = 0; b < SB; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[b], acc[a][b], 0, 0, 0);
    }
    e += kStep * kDepth;
  }
  // cross-wave sum within a slab, fixed order: deterministic
  __syncthreads();   // everyone is done with the index arrays (red aliases them)
  if (WPS == 4) {    // (w0+w2) + (w1+w3)
    if (wave >= 2) {
#pragma unroll
      for (int a = 0; a < SA; ++a)
#pragma unroll
        for (int b = 0; b < SB; ++b)
          red[((wave - 2) * SA * SB + a * SB + b) * 64 + lane] = acc[a][b];
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
      for (int a = 0; a < SA; ++a)
#pragma unroll
        for (int b = 0; b < SB; ++b)
          acc[a][b] += red[(wave * SA * SB + a * SB + b) * 64 + lane];
    }
    __syncthreads();
  }
  if (WPS >= 2) {    // odd wave of each slab -> its even wave
    if (rank == 1) {
#pragma unroll
      for (int a = 0; a < SA; ++a)
#pragma unroll
        for (int b = 0; b < SB; ++b)
          red[((wave / WPS) * SA * SB + a * SB + b) * 64 + lane] = acc[a][b];
    }
    __syncthreads();
  }
  if (rank == 0) {
    float* dst = partial + ((size_t)k * nchunks + chunk) * cin * cout;
#pragma unroll
    for (int a = 0; a < SA; ++a) {
      f32x4 v[SB];
#pragma unroll
      for (int b = 0; b < SB; ++b) {
        v[b] = acc[a][b];
        if (WPS >= 2) v[b] += red[((wave / WPS) * SA * SB + a * SB + b) * 64 + lane];
      }
      // D of tile (a,b): lane (col j = i, q) reg r  ->  ci = a0 + SA(4q+r) + a,
      // co = b0 + SB j + b : the SB b-tiles give SB consecutive co -> one store
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ci = a0 + SA * (4 * q + r) + a;
        if (ci >= cin) continue;
        float* o = dst + (size_t)ci * cout + cb;
        if (VEC) {
          if (cb < cout) {
            typename VecOf<SB>::type t;
            if constexpr (SB == 1) {
              t = v[0][r];
            } else {
#pragma unroll
              for (int b = 0; b < SB; ++b) t[b] = v[b][r];
            }
            *(typename VecOf<SB>::type*)o = t;
          }
        } else {
#pragma unroll
          for (int b = 0; b < SB; ++b)
            if (cb + b < cout) o[b] = v[b][r];
        }
      }
    }
  }
}

// Tiles per slab side for a channel count: 64-channel slabs for the wide layers,
// 32 / 16 for the narrow ones.
inline int wgrad_side(int c) { return c > 32 ? 4 : (c > 16 ? 2 : 1); }
// (128-channel slabs on the c_in side -- each dout row gathered half as often --
// need 288 registers per lane: one wave per SIMD, 714 us against 516 us.)
inline int wgrad_side_a(int c) { return wgrad_side(c); }
inline int wgrad_chunk(int cin, int cout) {
  return cin * cout <= 16 * 32 ? 512 : (cin * cout <= 32 * 64 ? 1024 : 2048);
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial,
                                                           const int32_t* __restrict__ num,
                                                           int nchunks, int per_k, int cin,
                                                           int cout, int kvol, int krsc,
                                                           int chunk_pairs,
                                                           float* __restrict__ dw) {
  const int k = blockIdx.y;
  const int P = num[k];
  const int span = wgrad_span(P, chunk_pairs);
  const int used = (P + span - 1) / span;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < per_k; e += gridDim.x * 256) {
    // fixed summation order (deterministic); 8 loads in flight per thread
    const float* src = partial + (size_t)k * nchunks * per_k + e;
    float s = 0.f;
    int c = 0;
    for (; c + 8 <= used; c += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(c + u) * per_k];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; c < used; ++c) s += src[(size_t)c * per_k];
    if (krsc) {  // d_weight is [c_out][K][c_in] (the module's parameter layout)
      const int ci = e / cout, co = e - ci * cout;
      dw[((size_t)co * kvol + k) * cin + ci] = s;
    } else {
      dw[(size_t)k * per_k + e] = s;
    }
  }
}

}  // namespace
}  // namespace msmd

namespace msmd {   // spconv_wgrad_block.hip
bool wgrad_block_supported(int c_in, int c_out, int kvol, int ld);
size_t wgrad_block_workspace_bytes(int kvol, int c_in, int c_out, int nchunk);
size_t wgrad_segment_table_ints(int kvol, int nchunk);
int wgrad_block_max_kvol();
int wgrad_pair_segments(const int32_t* pairs, const int32_t* num, int ld, int kvol,
                        int chunk_rows, int nchunk, int32_t* table, hipStream_t st);
int wgrad_block(const float* in_feat, int c_in, const float* d_out, int c_out,
                const int32_t* pairs, const int32_t* num, int ld, int kvol, int np,
                float* d_weight, int krsc_out, float* ws, const int32_t* segtab, int nchunk,
                hipStream_t st);
}
using namespace msmd;

MSMD_EXPORT size_t msmd_spconv_packed_weight_elems(int kernel_volume, int c_in, int c_out) {
  return (size_t)kernel_volume * ((c_in + 15) / 16) * ((c_out + 15) / 16) * 256;
}

MSMD_EXPORT int msmd_spconv_pack_weight(const float* weight, int kernel_volume, int c_in,
                                        int c_out, int flags, float* packed,
                                        msmd_stream_t stream) {
  if (!weight || !packed || kernel_volume < 1 || c_in < 1 || c_out < 1)
    return MSMD_ERR_INVALID_ARG;
  size_t total = msmd_spconv_packed_weight_elems(kernel_volume, c_in, c_out);
  int nb = ceil_div((long)total, 256);
  if (nb > 2048) nb = 2048;
  MSMD_LAUNCH(pack_weight_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, weight,
                     kernel_volume, c_in, c_out, flags, packed);
  return launch_status();
}

MSMD_EXPORT int msmd_rulebook_row_masks(const int32_t* nbr, int kernel_volume, int n_rows,
                                        uint64_t* masks, int64_t* sort_keys,
                                        msmd_stream_t stream) {
  if (kernel_volume < 1 || kernel_volume > 64) return MSMD_ERR_UNSUPPORTED;
  if (n_rows < 0 || (n_rows > 0 && (!nbr || (!masks && !sort_keys)))) return MSMD_ERR_INVALID_ARG;
  if (n_rows == 0) return MSMD_OK;
  MSMD_LAUNCH(row_mask_kernel, dim3(ceil_div(n_rows, 256)), dim3(256), 0, (hipStream_t)stream,
              nbr, kernel_volume, n_rows, (unsigned long long*)masks, (long long*)sort_keys);
  return launch_status();
}

namespace msmd {
// for tiling.hip: 32-bit sort keys (and how many of their bits matter) / tile costs
void launch_row_keys(const int32_t* nbr, int kvol, int n, uint32_t* keys, int32_t* row_ids,
                     int* key_bits, hipStream_t st) {
  MSMD_LAUNCH(row_key32_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st, nbr, kvol, n, keys,
              row_ids);
  *key_bits = row_key_bits(kvol);
}
void launch_tile_costs(const int32_t* nbr, int kvol, int n, const int32_t* order, int rows,
                       int32_t* cost, int32_t* tile_ids, hipStream_t st) {
  MSMD_LAUNCH(tile_cost_kernel, dim3(ceil_div(n, rows)), dim3(128), 0, st, nbr, kvol, n, order,
              rows, cost, tile_ids);
}
}  // namespace msmd

MSMD_EXPORT int msmd_rulebook_tile_costs(const int32_t* nbr, int kernel_volume, int n_rows,
                                         const int32_t* order, int rows_per_tile, int32_t* cost,
                                         msmd_stream_t stream) {
  if (kernel_volume < 1 || kernel_volume > 64) return MSMD_ERR_UNSUPPORTED;
  if (n_rows < 0 || rows_per_tile < 1 || (n_rows > 0 && (!nbr || !cost)))
    return MSMD_ERR_INVALID_ARG;
  if (n_rows == 0) return MSMD_OK;
  MSMD_LAUNCH(tile_cost_kernel, dim3(ceil_div(n_rows, rows_per_tile)), dim3(128), 0,
              (hipStream_t)stream, nbr, kernel_volume, n_rows, order, rows_per_tile, cost,
              (int32_t*)nullptr);
  return launch_status();
}

MSMD_EXPORT int msmd_spconv_fwd_f32(const float* in_feat, int n_in, int c_in,
                                    const float* packed_weight, const int32_t* nbr, int ld,
                                    int n_out, int kernel_volume, int weight_flip,
                                    const int32_t* row_order, int32_t* tile_counter,
                                    float* out_feat, int c_out, msmd_stream_t stream) {
  if (n_in < 0 || n_out < 0 || c_in < 1 || c_out < 1 || kernel_volume < 1 || ld < n_out)
    return MSMD_ERR_INVALID_ARG;
  if (n_out == 0) return MSMD_OK;
  if (!packed_weight || !nbr || !out_feat || (n_in > 0 && !in_feat)) return MSMD_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int NT = (c_out + 15) / 16;
  // (sorted order + persistent scheduler on the narrow layers: 16->16 21 -> 36 us --
  // scattered rows; they run in natural row order on a static grid)
  if (NT < 4) {
    row_order = nullptr;
    tile_counter = nullptr;
  }
#define FWD(NTv, Rv)                                                                       \
  return launch_fwd<NTv, Rv>(in_feat, c_in, packed_weight, nbr, ld, n_out, kernel_volume, \
                             weight_flip, row_order, tile_counter, out_feat, c_out, st)
  switch (NT) {
    case 1: FWD(1, 2);
    case 2: FWD(2, 2);
    case 3: FWD(3, 2);
    case 4: if (fwd_rows_variant() == 2) FWD(4, 2); else FWD(4, 1);
    case 5: if (fwd_rows_variant() == 2) FWD(5, 2); else FWD(5, 1);
    case 6: if (fwd_rows_variant() == 2) FWD(6, 2); else FWD(6, 1);
    case 8: if (fwd_rows_variant() == 2) FWD(8, 2); else FWD(8, 1);
    case 12: FWD(12, 1);
    default: break;
  }
#undef FWD
  return MSMD_ERR_UNSUPPORTED;
}

namespace {
size_t wgrad_ws_bytes(int kernel_volume, int ld, int c_in, int c_out, int n_chunks);
}
MSMD_EXPORT size_t msmd_spconv_wgrad_workspace_bytes(int kernel_volume, int ld, int c_in,
                                                     int c_out) {
  return wgrad_ws_bytes(kernel_volume, ld, c_in, c_out, 1);
}
// ... with a segment table of n_chunks row chunks (msmd_rulebook_pair_segments)
MSMD_EXPORT size_t msmd_spconv_wgrad_segments_workspace_bytes(int kernel_volume, int ld,
                                                              int c_in, int c_out,
                                                              int n_chunks) {
  return wgrad_ws_bytes(kernel_volume, ld, c_in, c_out, n_chunks < 1 ? 1 : n_chunks);
}
namespace {
size_t wgrad_ws_bytes(int kernel_volume, int ld, int c_in, int c_out, int n_chunks) {
  int chunk = wgrad_chunk(c_in, c_out);
  if (kWgradSplitChunk < chunk) chunk = kWgradSplitChunk;   // serves msmd_spconv_wgrad_split too
  size_t nchunks = (size_t)ceil_div(ld > 0 ? ld : 1, chunk);
  const size_t chunked = align_up(sizeof(float) * kernel_volume * nchunks * c_in * c_out);
  // the whole-block kernel's slots (spconv_wgrad_block.hip): one per workgroup + segment
  const size_t block = wgrad_block_workspace_bytes(kernel_volume, c_in, c_out, n_chunks);
  return chunked > block ? chunked : block;
}
}  // namespace

namespace {
template <int SA, int SB>
void launch_wgrad(int chunk, dim3 grid, hipStream_t st, const float* in_feat, int c_in,
                  const float* d_out, int c_out, const int32_t* pairs, const int32_t* num, int ld,
                  int nchunks, float* ws) {
  // vector loads need every row 4*S-byte aligned on both sides
  const bool vec = c_in % SA == 0 && c_out % SB == 0;
  // one slab per workgroup (measured on the 128x128 layers: 4 slabs per workgroup sharing
  // rows = 4x longer, 4x fewer workgroups -- 671 us against 516 us; removed)
  const int n_slabs = (int)grid.z;
  const int kvol = (int)grid.y, groups = n_slabs;
  const dim3 grid1(wgrad_grid(nchunks, kvol, groups));
#define MSMD_GOW(C_, V_, W_)                                                                  \
  MSMD_LAUNCH((spconv_wgrad_kernel<SA, SB, C_, V_, W_>), grid1, dim3(256), 0, st, in_feat,    \
              c_in, d_out, c_out, pairs, num, ld, nchunks, kvol, groups, ws)
#define MSMD_GOV(C_, W_)                                                                      \
  if (vec) MSMD_GOW(C_, true, W_); else MSMD_GOW(C_, false, W_)
  if (chunk == 512) {
    MSMD_GOV(512, 4);
  } else if (chunk == 1024) {
    MSMD_GOV(1024, 4);
  } else {
    MSMD_GOV(2048, 4);
  }
#undef MSMD_GOV
#undef MSMD_GOW
}
}  // namespace

MSMD_EXPORT int msmd_spconv_wgrad_f32(const float* in_feat, int c_in, const float* d_out,
                                      int c_out, const int32_t* indice_pairs,
                                      const int32_t* indice_num, int ld, int kernel_volume,
                                      float* d_weight, int krsc_out, void* workspace,
                                      size_t workspace_bytes, msmd_stream_t stream) {
  if (c_in < 1 || c_out < 1 || kernel_volume < 1 || ld < 0 || !d_weight || !indice_num)
    return MSMD_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int per_k = c_in * c_out;
  if (ld == 0) {
    hipMemsetAsync(d_weight, 0, sizeof(float) * (size_t)kernel_volume * per_k, st);
    return launch_status();
  }
  if (!in_feat || !d_out || !indice_pairs) return MSMD_ERR_INVALID_ARG;
  const int chunk = wgrad_chunk(c_in, c_out);
  const int nchunks = ceil_div(ld, chunk);
  if (workspace_bytes < sizeof(float) * (size_t)kernel_volume * nchunks * per_k ||
      ((uintptr_t)workspace & 255))
    return MSMD_ERR_WORKSPACE;
  const int SA = wgrad_side_a(c_in), SB = wgrad_side(c_out);
  const dim3 grid(nchunks, kernel_volume, ceil_div(c_in, 16 * SA) * ceil_div(c_out, 16 * SB));
#define MSMD_WG(A_, B_)                                                                        \
  if (SA == A_ && SB == B_)                                                                    \
    launch_wgrad<A_, B_>(chunk, grid, st, in_feat, c_in, d_out, c_out, indice_pairs,           \
                         indice_num, ld, nchunks, (float*)workspace);
  MSMD_WG(1, 1) MSMD_WG(1, 2) MSMD_WG(1, 4) MSMD_WG(2, 1) MSMD_WG(2, 2) MSMD_WG(2, 4)
  MSMD_WG(4, 1) MSMD_WG(4, 2) MSMD_WG(4, 4)
#undef MSMD_WG
  int rb = ceil_div(per_k, 256);
  if (rb > 64) rb = 64;
  MSMD_LAUNCH(wgrad_reduce_kernel, dim3(rb, kernel_volume), dim3(256), 0, st,
              (const float*)workspace, indice_num, nchunks, per_k, c_in, c_out, kernel_volume,
              krsc_out, chunk, d_weight);
  return launch_status();
}

namespace msmd {
int wgrad_split_partials(const float* in_feat, int c_in, const float* d_out, int c_out,
                         const int32_t* pairs, const int32_t* num, int ld, int kvol, int np,
                         int nchunks, float* ws, hipStream_t st);
}

MSMD_EXPORT int msmd_spconv_wgrad_split_supported(int c_in, int c_out) {
  // 64x64 channel slabs, the last one of a side possibly partial (16-byte pieces);
  // below 64 channels a slab is mostly padding: the fp32 kernel's narrow slabs win
  return c_in >= 64 && c_out >= 64 && c_in % 4 == 0 && c_out % 4 == 0;
}

// The segment table of a pair list for the whole-block wgrad kernel's row-chunk-major step
// sequence (spconv_wgrad_block.hip): int32 [msmd_rulebook_pair_segments_ints(K, n_chunks)],
// chunk c = output rows [c * chunk_rows, (c + 1) * chunk_rows).  Index data: computed once per
// rulebook, next to the pair lists.
// 0 for kernel volumes the whole-block kernel does not take (> 64, e.g. 5x5x5): no table -- the
// caller passes seg_table = NULL and msmd_spconv_wgrad_split_segments runs the slab kernel.
MSMD_EXPORT size_t msmd_rulebook_pair_segments_ints(int kernel_volume, int n_chunks) {
  return kernel_volume > 0 && kernel_volume <= wgrad_block_max_kvol() && n_chunks > 0
             ? wgrad_segment_table_ints(kernel_volume, n_chunks)
             : 0;
}
MSMD_EXPORT int msmd_rulebook_pair_segments(const int32_t* indice_pairs, const int32_t* indice_num,
                                            int ld, int kernel_volume, int chunk_rows,
                                            int n_chunks, int32_t* table, msmd_stream_t stream) {
  if (!indice_pairs || !indice_num || !table || ld < 1) return MSMD_ERR_INVALID_ARG;
  if ((long)chunk_rows * n_chunks < ld) return MSMD_ERR_INVALID_ARG;   // chunks must cover the rows
  return wgrad_pair_segments(indice_pairs, indice_num, ld, kernel_volume, chunk_rows, n_chunks,
                             table, (hipStream_t)stream);
}

MSMD_EXPORT int msmd_spconv_wgrad_split_segments(const float* in_feat, int c_in,
                                                 const float* d_out, int c_out,
                                                 const int32_t* indice_pairs,
                                                 const int32_t* indice_num, int ld,
                                                 int kernel_volume, int planes, float* d_weight,
                                                 int krsc_out, const int32_t* seg_table,
                                                 int n_chunks, void* workspace,
                                                 size_t workspace_bytes, msmd_stream_t stream) {
  if (!msmd_spconv_wgrad_split_supported(c_in, c_out) || planes < 1 || planes > 3)
    return MSMD_ERR_UNSUPPORTED;
  if (kernel_volume < 1 || ld < 0 || !d_weight || !indice_num) return MSMD_ERR_INVALID_ARG;
  if (seg_table && n_chunks < 1) return MSMD_ERR_INVALID_ARG;
  if (!seg_table) n_chunks = 1;
  hipStream_t st = (hipStream_t)stream;
  const int per_k = c_in * c_out;
  if (ld == 0) {
    hipMemsetAsync(d_weight, 0, sizeof(float) * (size_t)kernel_volume * per_k, st);
    return launch_status();
  }
  if (!in_feat || !d_out || !indice_pairs) return MSMD_ERR_INVALID_ARG;
  // the whole c_in x c_out block per workgroup (spconv_wgrad_block.hip) wherever both
  // widths are multiples of 16; MSMD_WGRAD=var keeps the 64 x 64 slab kernel (A/B runs)
  static const bool use_block = [] { const char* e = getenv("MSMD_WGRAD"); return !(e && !strcmp(e, "var")); }();
  if (use_block && wgrad_block_supported(c_in, c_out, kernel_volume, ld) &&
      (double)ld * 4.0 * (c_in > c_out ? c_in : c_out) < 4.0e9) {
    if (workspace_bytes < wgrad_block_workspace_bytes(kernel_volume, c_in, c_out, n_chunks) ||
        ((uintptr_t)workspace & 255))
      return MSMD_ERR_WORKSPACE;
    return wgrad_block(in_feat, c_in, d_out, c_out, indice_pairs, indice_num, ld, kernel_volume,
                       planes, d_weight, krsc_out, (float*)workspace, seg_table, n_chunks, st);
  }
  const int chunk = kWgradSplitChunk;   // pairs per workgroup (common.hpp)
  const int nchunks = ceil_div(ld, chunk);
  if (workspace_bytes < sizeof(float) * (size_t)kernel_volume * nchunks * per_k ||
      ((uintptr_t)workspace & 255))
    return MSMD_ERR_WORKSPACE;
  int rc = wgrad_split_partials(in_feat, c_in, d_out, c_out, indice_pairs, indice_num, ld,
                                kernel_volume, planes, nchunks, (float*)workspace, st);
  if (rc != MSMD_OK) return rc;
  int rb = ceil_div(per_k, 256);
  if (rb > 64) rb = 64;
  MSMD_LAUNCH(wgrad_reduce_kernel, dim3(rb, kernel_volume), dim3(256), 0, st,
              (const float*)workspace, indice_num, nchunks, per_k, c_in, c_out, kernel_volume,
              krsc_out, chunk, d_weight);
  return launch_status();
}

MSMD_EXPORT int msmd_spconv_wgrad_split(const float* in_feat, int c_in, const float* d_out,
                                        int c_out, const int32_t* indice_pairs,
                                        const int32_t* indice_num, int ld, int kernel_volume,
                                        int planes, float* d_weight, int krsc_out,
                                        void* workspace, size_t workspace_bytes,
                                        msmd_stream_t stream) {
  return msmd_spconv_wgrad_split_segments(in_feat, c_in, d_out, c_out, indice_pairs, indice_num,
                                          ld, kernel_volume, planes, d_weight, krsc_out, nullptr,
                                          1, workspace, workspace_bytes, stream);
}
