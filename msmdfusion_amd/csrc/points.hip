// points.hip -- GMA-Conv neighbour-search helpers: furthest point sampling,
// ball query, brute-force nearest key and the cluster -> member assignment.
//
// FPS: one 1024-thread workgroup per batch element (the m-1 rounds are serial
// by definition); every thread keeps its points AND their running distances
// in registers (no per-round HBM/LDS traffic at all), the per-round argmax is
// a 64-bit (distance, tie-order) max reduced with wave shuffles + one LDS hop.
// The tie order reproduces the reference kernel's block reduction exactly
// (furthest_point_sample_cuda.cu:17-23,55-137): among equal distances the
// point with the smallest (k mod block, k) wins, block = opt_n_threads(n).
#include "common.hpp"

#include <math.h>
#include <stdlib.h>

#include <type_traits>
#include <utility>

namespace msmd {
namespace {

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int o) {
  int lo = __shfl_xor((int)(uint32_t)v, o, 64), hi = __shfl_xor((int)(v >> 32), o, 64);
  return ((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo;
}

// ---- DPP reductions (no LDS crossbar: ~12 VALU per 32-bit wave reduce instead
// of 6 dependent ds_bpermute round trips).  Written as assembly: through
// __builtin_amdgcn_update_dpp the compiler emits v_mov_b32_dpp + a copy + the ALU
// op per step (4 instructions and 2 waits); the ALU op takes the DPP operand
// itself.  `s_nop 1` = the two wait states a DPP read needs after a VALU write of
// the same register. -------------------------------------------------------------
#define MSMD_DPP16(OP)                                                        \
  "s_nop 1\n" OP " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" \
  "s_nop 1\n" OP " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n" \
  "s_nop 1\n" OP " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n"     \
  "s_nop 1\n" OP " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n"
#define MSMD_DPP64(OP)                                                        \
  MSMD_DPP16(OP)                                                              \
  "s_nop 1\n" OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n"       \
  "s_nop 1\n" OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
__device__ __forceinline__ float fmax_dpp16(float v) {  // max over each 16-lane row, all lanes
  asm(MSMD_DPP16("v_max_f32_dpp") : "+v"(v));
  return v;
}
__device__ __forceinline__ uint32_t umin_dpp16(uint32_t v) {
  asm(MSMD_DPP16("v_min_u32_dpp") : "+v"(v));
  return v;
}
__device__ __forceinline__ float fmax_wave(float v) {  // wave-uniform result
  asm(MSMD_DPP64("v_max_f32_dpp") : "+v"(v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ uint32_t umin_wave(uint32_t v) {
  asm(MSMD_DPP64("v_min_u32_dpp") : "+v"(v));
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

constexpr int kFpsPrunedThreads = 512;
constexpr int kFpsSlotsPerBucket = 4;
constexpr int kFpsProbe0 = 32, kFpsProbe1 = 96;   // see fps_pruned_kernel

// Furthest point sampling, one 1024-thread workgroup per batch element; batch
// elements may have different sizes (offsets[b+1], ragged) or all n (offsets
// NULL).  PPT > 0: the element's points and running distances live in registers
// (n <= 1024*PPT; 4*PPT VGPRs, PPT = 24 is the most that fits 128 VGPRs);
// PPT == 0: any n, streamed from memory each round.
// Per round: pass 1 updates the running distances and takes the thread-local
// maximum; pass 2 finds, among the thread's points holding it, the smallest tie
// rank; the block reduces (max distance, then min rank among its holders) with
// DPP row operations and one LDS hop; the owner of the winner publishes its
// coordinates through LDS (no global read on the serial critical path).
template <int PPT>
__global__ __launch_bounds__(1024) void fps_kernel(const float* __restrict__ xyz_all,
                                                   const int* __restrict__ offsets, int n_fixed,
                                                   int m, float* __restrict__ temp_all,
                                                   int32_t* __restrict__ idx_all, int resume) {
  __shared__ float red_d[2][16];
  __shared__ uint32_t red_t[2][16];
  __shared__ float s_xyz[3];
  __shared__ float s_replay[kFpsProbe1][3];
  if (m <= 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long base = offsets ? offsets[blockIdx.x] : (long)blockIdx.x * n_fixed;
  const int n = offsets ? offsets[blockIdx.x + 1] - offsets[blockIdx.x] : n_fixed;
  const float* xyz = xyz_all + base * 3;
  float* temp = temp_all + base;
  int32_t* idx = idx_all + (size_t)blockIdx.x * m;
  if (n <= 0) {
    if (!resume)
      for (int j = tid; j < m; j += 1024) idx[j] = 0;
    return;
  }
  // resume: second half of a hand-over from fps_pruned_kernel (idx[m-1] = -(next
  // round j0)); an element it finished itself has idx[m-1] >= 0 and there is nothing
  // to do.  The running distances after samples 0..j0-2 are rebuilt by replaying them.
  int j0 = 1;
  if (resume) {
    const int mark = idx[m - 1];
    if (mark >= 0) return;
    j0 = -mark;
  }
  constexpr int P = PPT > 0 ? PPT : 1;
  // reference block size opt_n_threads(n): largest power of two <= n, max 1024
  // (furthest_point_sample_cuda.cu:11-15)
  const int bs_shift = min(31 - __clz(n), 10);
  const int bs_ref = 1 << bs_shift;
  // tie rank of point k: the reference's shared-memory tree keeps the LOWER slot
  // of each (t, t+s) pair, s = bs/2 .. 1, so between two thread ids the one
  // whose lowest differing bit is 0 wins: order by the bit-reversed thread id
  // (k mod bs), then by k div bs (the strided scan keeps its first maximum).
  auto tie_rank = [&](int k) -> uint32_t {
    uint32_t rev = bs_shift ? (__brev((uint32_t)(k & (bs_ref - 1))) >> (32 - bs_shift)) : 0u;
    return (rev << 21) | (uint32_t)(k >> bs_shift);
  };
  float px[P], py[P], pz[P], pd[P];
  if (PPT > 0) {
#pragma unroll
    for (int s = 0; s < P; ++s) {
      int k = tid + 1024 * s;
      bool ok = k < n;
      px[s] = ok ? xyz[k * 3 + 0] : 0.f;
      py[s] = ok ? xyz[k * 3 + 1] : 0.f;
      pz[s] = ok ? xyz[k * 3 + 2] : 0.f;
      pd[s] = ok ? 1e10f : -1.f;        // padding never wins (distances are >= 0)
    }
    if (resume) {
      for (int jj = tid; jj + 1 < j0 && jj < kFpsProbe1; jj += 1024) {
        const int si = idx[jj];
        s_replay[jj][0] = xyz[si * 3];
        s_replay[jj][1] = xyz[si * 3 + 1];
        s_replay[jj][2] = xyz[si * 3 + 2];
      }
      __syncthreads();
      for (int jj = 0; jj + 1 < j0; ++jj) {
        const float x1 = s_replay[jj][0], y1 = s_replay[jj][1], z1 = s_replay[jj][2];
#pragma unroll
        for (int s = 0; s < P; ++s) {
          float dx = px[s] - x1, dy = py[s] - y1, dz = pz[s] - z1;
          float d = dx * dx + dy * dy + dz * dz;
          pd[s] = fminf(d, pd[s]);
        }
      }
    }
  } else {
    for (int k = tid; k < n; k += 1024) temp[k] = 1e10f;
  }
  if (tid == 0) {
    const int first = resume ? idx[j0 - 1] : 0;
    if (!resume) idx[0] = 0;
    s_xyz[0] = xyz[first * 3];
    s_xyz[1] = xyz[first * 3 + 1];
    s_xyz[2] = xyz[first * 3 + 2];
  }
  __syncthreads();
  for (int j = j0; j < m; ++j) {
    const float x1 = s_xyz[0], y1 = s_xyz[1], z1 = s_xyz[2];
    float best_d = -1.f;
    uint32_t best_t = 0xFFFFFFFFu;
    if (PPT > 0) {
#pragma unroll
      for (int s = 0; s < P; ++s) {
        float dx = px[s] - x1, dy = py[s] - y1, dz = pz[s] - z1;
        float d = dx * dx + dy * dy + dz * dz;
        float d2 = fminf(d, pd[s]);      // padding: min(d, -1) = -1
        pd[s] = d2;
        best_d = fmaxf(best_d, d2);
      }
    } else {
      for (int k = tid; k < n; k += 1024) {
        float dx = xyz[k * 3] - x1, dy = xyz[k * 3 + 1] - y1, dz = xyz[k * 3 + 2] - z1;
        float d = dx * dx + dy * dy + dz * dz;
        float d2 = fminf(d, temp[k]);
        temp[k] = d2;
        const uint32_t t = tie_rank(k);
        bool better = d2 > best_d || (d2 == best_d && t < best_t);
        best_d = better ? d2 : best_d;
        best_t = better ? t : best_t;
      }
    }
    // wave: max distance, then min tie rank among the lanes/points holding it
    const float wd = fmax_wave(best_d);
    if (PPT > 0) {
      if (bs_shift == 10) {  // n >= 1024: k mod 1024 == tid, k div 1024 == s
        const uint32_t tbase = (__brev((uint32_t)tid) >> 22) << 21;
#pragma unroll
        for (int s = P - 1; s >= 0; --s)   // descending: the smallest s overwrites last
          best_t = pd[s] == wd ? (tbase | (uint32_t)s) : best_t;
      } else {
#pragma unroll
        for (int s = 0; s < P; ++s) {
          const uint32_t t = tie_rank(tid + 1024 * s);
          best_t = (pd[s] == wd && t < best_t) ? t : best_t;
        }
      }
    } else if (best_d != wd) {
      best_t = 0xFFFFFFFFu;
    }
    const uint32_t wt = umin_wave(best_t);
    const int buf = j & 1;   // double-buffered: one barrier separates write and read
    if (lane == 0) {
      red_d[buf][wave] = wd;
      red_t[buf][wave] = wt;
    }
    __syncthreads();
    const float cd = red_d[buf][lane & 15];
    const float bd = fmax_dpp16(cd);
    const uint32_t ct = cd == bd ? red_t[buf][lane & 15] : 0xFFFFFFFFu;
    const uint32_t tb = (uint32_t)__builtin_amdgcn_readfirstlane((int)umin_dpp16(ct));
    const uint32_t tid_ref = bs_shift ? (__brev(tb >> 21) >> (32 - bs_shift)) : 0u;
    const int old = (int)((tb & 0x1FFFFFu) << bs_shift) | (int)tid_ref;
    if (tid == 0) idx[j] = old;
    // the owner of `old` publishes its coordinates for the next round
    if (PPT > 0) {
      if ((old & 1023) == tid) {
        const int so = old >> 10;
#pragma unroll
        for (int s = 0; s < P; ++s)
          if (s == so) {
            s_xyz[0] = px[s];
            s_xyz[1] = py[s];
            s_xyz[2] = pz[s];
          }
      }
    } else if (tid == 0) {
      s_xyz[0] = xyz[old * 3];
      s_xyz[1] = xyz[old * 3 + 1];
      s_xyz[2] = xyz[old * 3 + 2];
    }
    __syncthreads();
  }
}

// ---- exact pruned FPS ---------------------------------------------------------
// The plain kernel above is VALU-bound on ONE CU: every round touches every point
// (24 points x ~10 VALU per lane, 16 waves on 4 SIMDs = ~4800 clk = 2.9 us), and
// the 2047 rounds are serial.  Most of that work changes nothing: after the first
// few hundred samples a new sample only lowers the running distance of points
// NEAR it.  Here the element's points are cut into buckets of 256 consecutive
// points (64 lanes x 4 register slots of one wave); per bucket the owning wave
// keeps, in the registers of lane g, the bucket's bounding box, its current
// (max running distance, tie rank) and the coordinates of that point.  A round
// updates bucket g only if  bound(sample, box_g) < maxdist_g , where bound is the
// box distance evaluated with the SAME fp32 operation sequence as the point
// distance: rounding is monotone, so bound <= d(sample, q) for every q in the
// box, hence min(d, running) == running for all of them and skipping is exact
// -- the selected indices are bit-identical to the unpruned kernel's (and the
// reference's: same distances, same tie order).  Skipped buckets contribute
// their cached (maxdist, rank).  One barrier per round: every wave publishes its
// best point's coordinates with its candidate, so the winner's coordinates are
// read back from LDS instead of being fetched by its owner in a second phase.
// Buckets are index ranges: the pruning is as good as the input order is
// spatially coherent (voxels in first-touch order of object-by-object virtual
// points are); on a shuffled cloud every bucket spans the scene and the kernel
// degrades to the plain one plus ~10 % bookkeeping.
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}


template <int P>
__global__ __launch_bounds__(kFpsPrunedThreads) void fps_pruned_kernel(
    const float* __restrict__ xyz_all, const int* __restrict__ offsets, int n_fixed, int m,
    int32_t* __restrict__ idx_all) {
  constexpr int SB = kFpsSlotsPerBucket, G = P / SB, NW = kFpsPrunedThreads / 64;
  static_assert(P % SB == 0 && G <= 16, "bucket state lives in lanes 0..15");
  __shared__ float red_d[2][NW];
  __shared__ uint32_t red_t[2][NW];
  __shared__ float red_p[2][NW][3];
  __shared__ int red_n[NW];
  if (m <= 0) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: ranks split below
  const long base = offsets ? offsets[blockIdx.x] : (long)blockIdx.x * n_fixed;
  const int n = offsets ? offsets[blockIdx.x + 1] - offsets[blockIdx.x] : n_fixed;
  const float* xyz = xyz_all + base * 3;
  int32_t* idx = idx_all + (size_t)blockIdx.x * m;
  if (n <= 0) {
    for (int j = tid; j < m; j += kFpsPrunedThreads) idx[j] = 0;
    return;
  }
  const int bs_shift = min(31 - __clz(n), 10);
  const int bs_ref = 1 << bs_shift;
  // Buckets are dealt round-robin to the waves (bucket g*NW + wave belongs to wave
  // `wave`): the few buckets a sample can reach are neighbours in index order, and this
  // way they are refreshed by different waves in parallel instead of one after another.
  // Slot s = g*SB + q of this lane holds point k = U_s + lane, U_s = ((g*NW + wave)*SB + q)*64.
  // tie rank of k (see fps_kernel) = (bitrev(k mod bs) << 21) | (k div bs); U_s is a
  // multiple of 64 and lane < 64, so the two never share a bit and the rank splits
  // into a per-lane constant OR a wave-uniform term (scalar ALU):
  const int rsh = 32 - bs_shift;
  const uint32_t lane_rank =
      ((bs_shift ? (__brev((uint32_t)lane) >> rsh) : 0u) << 21) |
      (bs_shift < 6 ? (uint32_t)lane >> bs_shift : 0u);
  auto rank_of = [&](uint32_t u) -> uint32_t {      // u: a multiple of 64
    const uint32_t urev = bs_shift ? (__brev(u) >> rsh) : 0u;
    return (urev << 21) | (u >> bs_shift);
  };
  // ... and U_s = U_g + 64*q (bucket g, slot q of SB=4; U_g a multiple of 256) splits
  // once more: 4 per-lane constants (lane + slot) and one scalar per bucket
  uint32_t lane_slot_rank[kFpsSlotsPerBucket];
#pragma unroll
  for (int q = 0; q < kFpsSlotsPerBucket; ++q) lane_slot_rank[q] = lane_rank | rank_of(64u * q);
  float px[P], py[P], pz[P], pd[P];
#pragma unroll
  for (int s = 0; s < P; ++s) {
    const int k = (((s / SB) * NW + wave) * SB + s % SB) * 64 + lane;
    const bool ok = k < n;
    px[s] = ok ? xyz[(size_t)k * 3 + 0] : 0.f;
    py[s] = ok ? xyz[(size_t)k * 3 + 1] : 0.f;
    pz[s] = ok ? xyz[(size_t)k * 3 + 2] : 0.f;
    pd[s] = ok ? 1e10f : -1.f;
  }
  // state of bucket g, held by lane g (lanes >= G: an empty bucket)
  float bx0 = 0.f, bx1 = 0.f, by0 = 0.f, by1 = 0.f, bz0 = 0.f, bz1 = 0.f;
  float mybd = -1.f, cx = 0.f, cy = 0.f, cz = 0.f;
  uint32_t mybt = 0xFFFFFFFFu;
  const float kInf = __int_as_float(0x7f800000);
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float lo[3] = {kInf, kInf, kInf}, hi[3] = {-kInf, -kInf, -kInf};
#pragma unroll
    for (int q = 0; q < SB; ++q) {
      const int s = g * SB + q;
      const bool ok = pd[s] >= 0.f;
      lo[0] = ok ? fminf(lo[0], px[s]) : lo[0];  hi[0] = ok ? fmaxf(hi[0], px[s]) : hi[0];
      lo[1] = ok ? fminf(lo[1], py[s]) : lo[1];  hi[1] = ok ? fmaxf(hi[1], py[s]) : hi[1];
      lo[2] = ok ? fminf(lo[2], pz[s]) : lo[2];  hi[2] = ok ? fmaxf(hi[2], pz[s]) : hi[2];
    }
    const float x0 = -fmax_wave(-lo[0]), x1 = fmax_wave(hi[0]);
    const float y0 = -fmax_wave(-lo[1]), y1 = fmax_wave(hi[1]);
    const float z0 = -fmax_wave(-lo[2]), z1 = fmax_wave(hi[2]);
    if (lane == g) {
      bx0 = x0; bx1 = x1; by0 = y0; by1 = y1; bz0 = z0; bz1 = z1;
      mybd = x0 <= x1 ? kInf : -1.f;     // not empty: round 1 updates it (bound < inf)
    }
  }
  // update of one bucket against the sample (sx, sy, sz) + refresh of its cached state
  auto update = [&](auto gc, float sx, float sy, float sz) {
    constexpr int g = decltype(gc)::value;
    float ld = -1.f;
#pragma unroll
    for (int q = 0; q < SB; ++q) {
      const int s = g * SB + q;
      float dx = px[s] - sx, dy = py[s] - sy, dz = pz[s] - sz;
      float d = dx * dx + dy * dy + dz * dz;
      float d2 = fminf(d, pd[s]);          // padding: min(d, -1) = -1
      pd[s] = d2;
      ld = fmaxf(ld, d2);
    }
    const float wd = fmax_wave(ld);
    uint32_t lt = 0xFFFFFFFFu;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (wd >= 0.f) {
      uint32_t bucket_rank = rank_of((uint32_t)((g * NW + wave) * SB) * 64u);
      // keep the OR below inside the loop: hoisted, its 48 results would live in VGPRs
      asm volatile("" : "+s"(bucket_rank));
#pragma unroll
      for (int q = 0; q < SB; ++q) {
        const int s = g * SB + q;
        const uint32_t r = lane_slot_rank[q] | bucket_rank;
        const bool better = pd[s] == wd && r < lt;
        lt = better ? r : lt;
        qx = better ? px[s] : qx;
        qy = better ? py[s] : qy;
        qz = better ? pz[s] : qz;
      }
    }
    const uint32_t wt = umin_wave(lt);
    const unsigned long long holders = __ballot(lt == wt);
    const int owner = holders ? __ffsll((long long)holders) - 1 : 0;
    const float ox = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qx), owner));
    const float oy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qy), owner));
    const float oz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(qz), owner));
    if (lane == g) {
      mybd = wd; mybt = wt; cx = ox; cy = oy; cz = oz;
    }
  };
  float sx = xyz[0], sy = xyz[1], sz = xyz[2];
  if (tid == 0) idx[0] = 0;
  // Is the pruning paying?  Refreshes are counted over rounds kFpsProbe0..kFpsProbe1; if
  // more than half of the (non-empty) buckets were refreshed per round -- an input
  // whose index order is not spatially coherent -- the element is handed to the plain
  // kernel, which is twice as fast when every point has to be touched anyway:
  // idx[m-1] = -(next round) marks the hand-over; the plain kernel rebuilds the running
  // distances from the samples chosen so far (a min over them: order-free, exact).
  const int nbuckets = (n + 64 * SB - 1) / (64 * SB);
  int refreshed = 0;
  auto one_round = [&](int j) {
    // which of this wave's buckets can the sample still reach?
    const float ex = fmaxf(fmaxf(bx0 - sx, sx - bx1), 0.f);
    const float ey = fmaxf(fmaxf(by0 - sy, sy - by1), 0.f);
    const float ez = fmaxf(fmaxf(bz0 - sz, sz - bz1), 0.f);
    const float bound = ex * ex + ey * ey + ez * ez;
    const unsigned long long need = __ballot(lane < G && bound < mybd);
    refreshed = j == kFpsProbe0 ? 0 : refreshed + __popcll(need);
    static_for<G>([&](auto gc) {
      if ((need >> decltype(gc)::value) & 1ull) update(gc, sx, sy, sz);
    });
    // wave candidate: max distance over its buckets, then the smallest rank holding it
    const float wd = __int_as_float(
        __builtin_amdgcn_readfirstlane(__float_as_int(fmax_dpp16(mybd))));
    const uint32_t ct = (lane < 16 && mybd == wd) ? mybt : 0xFFFFFFFFu;
    const uint32_t wt = (uint32_t)__builtin_amdgcn_readfirstlane((int)umin_dpp16(ct));
    const unsigned long long wh = __ballot(lane < 16 && ct == wt);
    const int gw = wh ? __ffsll((long long)wh) - 1 : 0;
    const float wx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cx), gw));
    const float wy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cy), gw));
    const float wz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cz), gw));
    const int buf = j & 1;
    if (lane == 0) {
      red_d[buf][wave] = wd;
      red_t[buf][wave] = wt;
      red_p[buf][wave][0] = wx;
      red_p[buf][wave][1] = wy;
      red_p[buf][wave][2] = wz;
    }
    __syncthreads();
    // every lane fetches candidate (lane mod NW) whole -- five independent LDS reads,
    // one wait -- and the winner's coordinates come out of a lane, not a second trip
    const int cw = lane & (NW - 1);
    const float cd = red_d[buf][cw];
    const uint32_t ctv = red_t[buf][cw];
    const float cpx = red_p[buf][cw][0], cpy = red_p[buf][cw][1], cpz = red_p[buf][cw][2];
    const float bd = fmax_dpp16(cd);
    const uint32_t tc = cd == bd ? ctv : 0xFFFFFFFFu;
    const uint32_t tb = (uint32_t)__builtin_amdgcn_readfirstlane((int)umin_dpp16(tc));
    const unsigned long long gh = __ballot(tc == tb);
    const int ww = gh ? __ffsll((long long)gh) - 1 : 0;
    sx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cpx), ww));
    sy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cpy), ww));
    sz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cpz), ww));
    if (tid == 0) {
      const uint32_t tid_ref = bs_shift ? (__brev(tb >> 21) >> (32 - bs_shift)) : 0u;
      idx[j] = (int)((tb & 0x1FFFFFu) << bs_shift) | (int)tid_ref;
    }
  };
  const bool probe = m > 2 * kFpsProbe1;
  const int m1 = probe ? kFpsProbe1 : m;
  for (int j = 1; j < m1; ++j) one_round(j);
  if (probe) {
    if (lane == 0) red_n[wave] = refreshed;
    __syncthreads();
    int total = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) total += red_n[w];
    if (2 * total > nbuckets * (kFpsProbe1 - kFpsProbe0)) {
      if (tid == 0) idx[m - 1] = -m1;
      return;
    }
    for (int j = m1; j < m; ++j) one_round(j);
  }
}

// One wave per centre; points visited 64 at a time in index order, hits
// compacted with a ballot so the first `nsample` hits keep their order
// (ball_query_cuda.cu:33-53).
__global__ __launch_bounds__(256) void ball_query_kernel(const float* __restrict__ centers,
                                                         const float* __restrict__ xyz, int n,
                                                         int m, float min_r2, float max_r2,
                                                         int nsample, int32_t* __restrict__ idx) {
  const int lane = threadIdx.x & 63;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.y;
  if (c >= m) return;
  centers += ((size_t)b * m + c) * 3;
  xyz += (size_t)b * n * 3;
  int32_t* out = idx + ((size_t)b * m + c) * nsample;
  const float cx = centers[0], cy = centers[1], cz = centers[2];
  int cnt = 0;
  for (int base = 0; base < n && cnt < nsample; base += 64) {
    int k = base + lane;
    bool hit = false;
    if (k < n) {
      float x = xyz[k * 3], y = xyz[k * 3 + 1], z = xyz[k * 3 + 2];
      float d2 = (cx - x) * (cx - x) + (cy - y) * (cy - y) + (cz - z) * (cz - z);
      hit = d2 == 0.f || (d2 >= min_r2 && d2 < max_r2);
    }
    unsigned long long mask = __ballot(hit);
    if (mask == 0) continue;
    if (cnt == 0) {  // first hit pre-fills every slot
      int first = base + __ffsll((long long)mask) - 1;
      for (int l = lane; l < nsample; l += 64) out[l] = first;
    }
    int pos = cnt + __popcll(mask & ((1ull << lane) - 1ull));
    if (hit && pos < nsample) out[pos] = k;
    cnt += __popcll(mask);
  }
  if (cnt == 0)  // no hit at all: reference leaves the zero-initialised row
    for (int l = lane; l < nsample; l += 64) out[l] = 0;
}

// nearest key: grid (query blocks, key chunks); packed (dist bits, key) min.
constexpr int kNnChunk = 1024;   // keys per block: one LDS tile; 4096 left most CUs idle (80 blocks)
__global__ __launch_bounds__(256) void nn_partial(const int32_t* __restrict__ q, int nq,
                                                  const int32_t* __restrict__ key, int nk,
                                                  unsigned long long* best) {
  __shared__ int ks[1024 * 3];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int k0 = blockIdx.y * kNnChunk;
  const int k1 = (k0 + kNnChunk) < nk ? (k0 + kNnChunk) : nk;
  float qz = 0, qy = 0, qx = 0;
  if (i < nq) {
    qz = (float)q[i * 3];
    qy = (float)q[i * 3 + 1];
    qx = (float)q[i * 3 + 2];
  }
  unsigned long long b = kEmptySlot;
  for (int base = k0; base < k1; base += 1024) {
    int cnt = (k1 - base) < 1024 ? (k1 - base) : 1024;
    __syncthreads();
    for (int e = threadIdx.x; e < cnt * 3; e += 256) ks[e] = key[(size_t)base * 3 + e];
    __syncthreads();
    for (int k = 0; k < cnt; ++k) {
      float dz = qz - (float)ks[k * 3], dy = qy - (float)ks[k * 3 + 1],
            dx = qx - (float)ks[k * 3 + 2];
      float d = sqrtf(dz * dz + dy * dy + dx * dx);
      unsigned long long v = ((unsigned long long)__float_as_uint(d) << 32) | (uint32_t)(base + k);
      b = v < b ? v : b;
    }
  }
  if (i < nq) atomicMin(&best[i], b);
}
__global__ __launch_bounds__(256) void nn_final(const unsigned long long* __restrict__ best,
                                                int nq, float thresh, int32_t* __restrict__ out) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nq) return;
  unsigned long long b = best[i];
  float d = __uint_as_float((uint32_t)(b >> 32));
  out[i] = (b != kEmptySlot && d < thresh) ? (int)(uint32_t)b : -1;
}

__global__ __launch_bounds__(256) void assign_max(const int32_t* __restrict__ group_idx,
                                                  const int32_t* __restrict__ rep_nn, int m,
                                                  int nsample, int nq, int32_t* winner) {
  long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)m * nsample) return;
  int r = (int)(t / nsample);
  if (rep_nn[r] < 0) return;
  int g = group_idx[t];
  if (g >= 0 && g < nq) atomicMax(&winner[g], r);
}
__global__ __launch_bounds__(256) void assign_final(const int32_t* __restrict__ winner,
                                                    const int32_t* __restrict__ rep_nn, int nq,
                                                    int32_t* __restrict__ out) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nq) return;
  int w = winner[i];
  out[i] = w >= 0 ? rep_nn[w] : -1;
}

int fps_block_size(int n) {  // opt_n_threads, furthest_point_sample_cuda.cu:11-15
  int pow_2 = (int)(log((double)n) / log(2.0));
  int t = 1 << pow_2;
  if (t > 1024) t = 1024;
  return t < 1 ? 1 : t;
}

}  // namespace
}  // namespace msmd

using namespace msmd;

namespace {
bool fps_prune_enabled() {
  static const bool on = [] {
    const char* e = getenv("MSMD_FPS_PRUNE");   // 0: the plain every-point kernel
    return !(e && e[0] == '0');
  }();
  return on;
}
void launch_fps_plain(const float* xyz, const int* offsets, int b, int n_max, int n_fixed, int m,
                      float* temp, int32_t* idx, int resume, hipStream_t st) {
  const int ppt = ceil_div(n_max, 1024);
#define FPS(P)                                                                               \
  MSMD_LAUNCH(fps_kernel<P>, dim3(b), dim3(1024), 0, st, xyz, offsets, n_fixed, m, temp, idx, \
              resume)
  if (ppt <= 2) FPS(2);
  else if (ppt <= 4) FPS(4);
  else if (ppt <= 8) FPS(8);
  else if (ppt <= 16) FPS(16);
  else if (ppt <= 20) FPS(20);
  else if (ppt <= 22) FPS(22);
  else if (ppt <= 24) FPS(24);
  else FPS(0);
#undef FPS
}
int launch_fps(const float* xyz, const int* offsets, int b, int n_max, int n_fixed, int m,
               float* temp, int32_t* idx, hipStream_t st) {
  // pruned kernel: elements of 6k..24k points sampled sparsely (below that a round is
  // mostly fixed cost either way; above, the points do not fit one workgroup's
  // registers).  It may hand an element over (see kFpsProbe1): the plain kernel runs
  // right behind it in resume mode and returns at once for finished elements.
  const int ppl = ceil_div(n_max, kFpsPrunedThreads);
  if (fps_prune_enabled() && n_max >= 6144 && ppl <= 48 && m > 2 * kFpsProbe1) {
#define FPSP(P)                                                                              \
  MSMD_LAUNCH(fps_pruned_kernel<P>, dim3(b), dim3(kFpsPrunedThreads), 0, st, xyz, offsets, \
              n_fixed, m, idx)
    if (ppl <= 16) FPSP(16);
    else if (ppl <= 32) FPSP(32);
    else FPSP(48);
#undef FPSP
    launch_fps_plain(xyz, offsets, b, n_max, n_fixed, m, temp, idx, 1, st);
    return launch_status();
  }
  launch_fps_plain(xyz, offsets, b, n_max, n_fixed, m, temp, idx, 0, st);
  return launch_status();
}
}  // namespace

MSMD_EXPORT int msmd_furthest_point_sample(const float* xyz, int b, int n, int m, float* temp,
                                           int32_t* idx, msmd_stream_t stream) {
  if (b < 1 || n < 1 || m < 0 || !xyz || !idx || !temp) return MSMD_ERR_INVALID_ARG;
  if (m == 0) return MSMD_OK;
  if (n >= (1 << 21)) return MSMD_ERR_RANGE;
  return launch_fps(xyz, nullptr, b, n, n, m, temp, idx, (hipStream_t)stream);
}

MSMD_EXPORT int msmd_furthest_point_sample_ragged(const float* xyz, const int32_t* offsets, int b,
                                                  int n_max, int m, float* temp, int32_t* idx,
                                                  msmd_stream_t stream) {
  if (b < 1 || n_max < 1 || m < 0 || !xyz || !offsets || !idx || !temp)
    return MSMD_ERR_INVALID_ARG;
  if (m == 0) return MSMD_OK;
  if (n_max >= (1 << 21)) return MSMD_ERR_RANGE;
  return launch_fps(xyz, offsets, b, n_max, 0, m, temp, idx, (hipStream_t)stream);
}

MSMD_EXPORT int msmd_ball_query(const float* center_xyz, const float* xyz, int b, int n, int m,
                                float min_radius, float max_radius, int nsample, int32_t* idx,
                                msmd_stream_t stream) {
  if (b < 1 || n < 1 || m < 0 || nsample < 1 || !center_xyz || !xyz || !idx)
    return MSMD_ERR_INVALID_ARG;
  if (m == 0) return MSMD_OK;
  MSMD_LAUNCH(ball_query_kernel, dim3(ceil_div(m, 4), b), dim3(256), 0,
                     (hipStream_t)stream, center_xyz, xyz, n, m, min_radius * min_radius,
                     max_radius * max_radius, nsample, idx);
  return launch_status();
}

MSMD_EXPORT int msmd_nn_search(const int32_t* query_zyx, int nq, const int32_t* key_zyx, int nk,
                               float dist_thresh, int32_t* out_idx, void* scratch,
                               msmd_stream_t stream) {
  if (nq < 0 || nk < 0 || (nq > 0 && (!query_zyx || !out_idx || !scratch)))
    return MSMD_ERR_INVALID_ARG;
  if (nq == 0) return MSMD_OK;
  hipStream_t st = (hipStream_t)stream;
  auto* best = (unsigned long long*)scratch;
  hipMemsetAsync(best, 0xFF, sizeof(unsigned long long) * nq, st);
  if (nk > 0)
    MSMD_LAUNCH(nn_partial, dim3(ceil_div(nq, 256), ceil_div(nk, kNnChunk)), dim3(256), 0,
                       st, query_zyx, nq, key_zyx, nk, best);
  MSMD_LAUNCH(nn_final, dim3(ceil_div(nq, 256)), dim3(256), 0, st, best, nq, dist_thresh,
                     out_idx);
  return launch_status();
}

MSMD_EXPORT int msmd_nn_assign(const int32_t* group_idx, const int32_t* rep_nn, int m,
                               int nsample, int nq, int32_t* query_nn, int32_t* scratch,
                               msmd_stream_t stream) {
  if (m < 0 || nsample < 1 || nq < 0 || (nq > 0 && (!query_nn || !scratch)))
    return MSMD_ERR_INVALID_ARG;
  if (nq == 0) return MSMD_OK;
  hipStream_t st = (hipStream_t)stream;
  hipMemsetAsync(scratch, 0xFF, sizeof(int32_t) * nq, st);
  if (m > 0)
    MSMD_LAUNCH(assign_max, dim3(ceil_div((long)m * nsample, 256)), dim3(256), 0, st,
                       group_idx, rep_nn, m, nsample, nq, scratch);
  MSMD_LAUNCH(assign_final, dim3(ceil_div(nq, 256)), dim3(256), 0, st, scratch, rep_nn, nq,
                     query_nn);
  return launch_status();
}

// ------------------------------------------------------------- misc API ----
MSMD_EXPORT const char* msmd_status_string(int status) {
  switch (status) {
    case MSMD_OK: return "ok";
    case MSMD_ERR_INVALID_ARG: return "invalid argument";
    case MSMD_ERR_WORKSPACE: return "workspace too small or misaligned";
    case MSMD_ERR_UNSUPPORTED: return "unsupported shape or parameter";
    case MSMD_ERR_LAUNCH: return "kernel launch failed";
    case MSMD_ERR_RANGE: return "linear voxel id does not fit 32 bits";
    default: return "unknown status";
  }
}
MSMD_EXPORT const char* msmd_last_launch_error(void) {
  return hipGetErrorString((hipError_t)g_last_detail);
}
MSMD_EXPORT int msmd_abi_version(void) { return 1; }
MSMD_EXPORT int msmd_device_ok(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return 0;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 0;
  const char* a = p.gcnArchName;
  return a[0] == 'g' && a[1] == 'f' && a[2] == 'x' && a[3] == '9' && a[4] == '5' && a[5] == '0';
}
