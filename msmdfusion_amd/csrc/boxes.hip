// boxes.hip -- the box / heat-map arithmetic under TransFusionHead.loss.
//
//   boxes_iou3d       BaseInstance3DBoxes.overlaps (core/bbox/structures/base_box3d.py:352-438):
//                     rotated BEV overlap (ops/iou3d/src/iou3d_kernel.cu:36-264) x height
//                     overlap / volumes, for every sample of the batch in ONE launch (the
//                     reference launches per sample and per decoder layer and copies the
//                     BEV boxes twice on the way).
//   heatmap_gaussian  the per-box python loop of get_targets_single
//                     (models/dense_heads/transfusion_head.py:1186-1210 ->
//                     core/utils/gaussian.py:5-53): all boxes of all samples in one launch.
//   gaussian_focal    clip_sigmoid + GaussianFocalLoss (transfusion_head.py:1247-1249):
//                     value, gradient and the positive count in one pass over the map
//                     (the reference: ~15 elementwise launches + .item()).
// All three are tiny (20k box pairs, a few hundred bumps, 650k map cells): what counts
// is the launch count and the absent host round trips, not a roofline.
#include "common.hpp"

namespace msmd {
namespace {

struct Pt {
  float x, y;
};

__device__ __forceinline__ float cross3(Pt a, Pt b, Pt o) {
  return (a.x - o.x) * (b.y - o.y) - (b.x - o.x) * (a.y - o.y);
}

// crossing of segments p0-p1 and q0-q1 (strict straddle test, reference order of operands)
__device__ __forceinline__ bool edge_hit(Pt p1, Pt p0, Pt q1, Pt q0, Pt& hit) {
  if (!(fminf(p0.x, p1.x) <= fmaxf(q0.x, q1.x) && fminf(q0.x, q1.x) <= fmaxf(p0.x, p1.x) &&
        fminf(p0.y, p1.y) <= fmaxf(q0.y, q1.y) && fminf(q0.y, q1.y) <= fmaxf(p0.y, p1.y)))
    return false;
  const float s1 = cross3(q0, p1, p0), s2 = cross3(p1, q1, p0);
  const float s3 = cross3(p0, q1, q0), s4 = cross3(q1, p1, q0);
  if (!(s1 * s2 > 0.f && s3 * s4 > 0.f)) return false;
  const float s5 = cross3(q1, p1, p0);
  if (fabsf(s5 - s1) > 1e-8f) {
    hit.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    hit.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    const float d = a0 * b1 - a1 * b0;
    hit.x = (b0 * c1 - b1 * c0) / d;
    hit.y = (a1 * c0 - a0 * c1) / d;
  }
  return true;
}

struct Box {          // x1, y1, x2, y2, angle
  float v[5];
};

__device__ __forceinline__ bool inside(const Box& box, Pt p) {
  const float margin = 1e-5f;
  const float cx = (box.v[0] + box.v[2]) / 2, cy = (box.v[1] + box.v[3]) / 2;
  const float c = cosf(-box.v[4]), s = sinf(-box.v[4]);
  const float rx = (p.x - cx) * c + (p.y - cy) * s + cx;
  const float ry = -(p.x - cx) * s + (p.y - cy) * c + cy;
  return rx > box.v[0] - margin && rx < box.v[2] + margin && ry > box.v[1] - margin &&
         ry < box.v[3] + margin;
}

__device__ __forceinline__ void corners_of(const Box& box, Pt* out /* 5, closed */) {
  const float cx = (box.v[0] + box.v[2]) / 2, cy = (box.v[1] + box.v[3]) / 2;
  const float c = cosf(box.v[4]), s = sinf(box.v[4]);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float dx = ((k == 1 || k == 2) ? box.v[2] : box.v[0]) - cx;
    const float dy = (k >= 2 ? box.v[3] : box.v[1]) - cy;
    out[k].x = dx * c + dy * s + cx;
    out[k].y = -dx * s + dy * c + cy;
  }
  out[4] = out[0];
}

// Area of the intersection polygon: vertices = edge crossings, then corners of b inside
// a / corners of a inside b alternating; ordered by atan2 about their mean (a bubble sort
// in the reference: stable, so an insertion sort on the precomputed angles gives the same
// order), summed as a fan from vertex 0.
__device__ float overlap_bev(const Box& a, const Box& b) {
  Pt ca[5], cb[5], v[24];
  float ang[24];
  corners_of(a, ca);
  corners_of(b, cb);
  int n = 0;
  float mx = 0.f, my = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      Pt h;
      if (edge_hit(ca[i + 1], ca[i], cb[j + 1], cb[j], h)) {
        mx += h.x;
        my += h.y;
        v[n++] = h;
      }
    }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (inside(a, cb[k])) {
      mx += cb[k].x;
      my += cb[k].y;
      v[n++] = cb[k];
    }
    if (inside(b, ca[k])) {
      mx += ca[k].x;
      my += ca[k].y;
      v[n++] = ca[k];
    }
  }
  if (n < 3) return 0.f;                       // fan over < 3 vertices is empty
  mx /= n;
  my /= n;
  for (int i = 0; i < n; ++i) ang[i] = atan2f(v[i].y - my, v[i].x - mx);
  for (int i = 1; i < n; ++i) {                // stable: moves left only past strictly larger
    const Pt p = v[i];
    const float t = ang[i];
    int j = i - 1;
    while (j >= 0 && ang[j] > t) {
      v[j + 1] = v[j];
      ang[j + 1] = ang[j];
      --j;
    }
    v[j + 1] = p;
    ang[j + 1] = t;
  }
  float area = 0.f;
  for (int k = 0; k < n - 1; ++k) {
    const float ux = v[k].x - v[0].x, uy = v[k].y - v[0].y;
    const float wx = v[k + 1].x - v[0].x, wy = v[k + 1].y - v[0].y;
    area += ux * wy - uy * wx;
  }
  return (float)(fabs((double)area) / 2.0);
}

// LiDARInstance3DBoxes.bev (columns 0,1,3,4,6) -> xywhr2xyxyr
__device__ __forceinline__ Box bev_of(const float* r) {
  Box b;
  const float hw = r[3] / 2.f, hh = r[4] / 2.f;
  b.v[0] = r[0] - hw;
  b.v[1] = r[1] - hh;
  b.v[2] = r[0] + hw;
  b.v[3] = r[1] + hh;
  b.v[4] = r[6];
  return b;
}

// mode 0: IoU, 1: IoF (over boxes a), 2: BEV overlap area of 5-column xyxyr boxes
__global__ __launch_bounds__(256) void boxes_iou3d_kernel(
    const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb,
    const int32_t* __restrict__ nb_valid, int batch, int na, int nb, int mode,
    float* __restrict__ out) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)batch * na * nb) return;
  const int j = (int)(t % nb);
  const long ij = t / nb;
  const int i = (int)(ij % na), s = (int)(ij / na);
  if (nb_valid && j >= nb_valid[s]) {
    out[t] = 0.f;
    return;
  }
  const float* ra = a + ((size_t)s * na + i) * lda;
  const float* rb = b + ((size_t)s * nb + j) * ldb;
  if (mode == 2) {
    Box ba, bb;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      ba.v[k] = ra[k];
      bb.v[k] = rb[k];
    }
    out[t] = overlap_bev(ba, bb);
    return;
  }
  const float top = fminf(ra[2] + ra[5], rb[2] + rb[5]), bot = fmaxf(ra[2], rb[2]);
  const float oh = fmaxf(top - bot, 0.f);
  const float inter = overlap_bev(bev_of(ra), bev_of(rb)) * oh;
  const float va = ra[3] * ra[4] * ra[5], vb = rb[3] * rb[4] * rb[5];
  out[t] = mode == 0 ? inter / fmaxf(va + vb - inter, 1e-8f) : inter / fmaxf(va, 1e-8f);
}

// one workgroup per box: heat[plane] = max(heat[plane], bump clipped to the map)
__global__ __launch_bounds__(256) void heatmap_gaussian_kernel(
    const int32_t* __restrict__ plane, const int32_t* __restrict__ cx,
    const int32_t* __restrict__ cy, const int32_t* __restrict__ radius, int planes, int h,
    int w, float* __restrict__ heat) {
  const int g = blockIdx.x;
  const int pl = plane[g], r = radius[g];
  if (pl < 0 || pl >= planes || r < 0) return;
  const int x0 = cx[g], y0 = cy[g];
  const int d = 2 * r + 1;
  const double sigma = (double)d / 6.0;
  const double denom = 2 * sigma * sigma;
  const double tiny = 2.220446049250313e-16;     // np.finfo(float64).eps * h.max(), max = 1
  for (int e = threadIdx.x; e < d * d; e += 256) {
    const int dy = e / d - r, dx = e % d - r;
    const int x = x0 + dx, y = y0 + dy;
    if (x < 0 || x >= w || y < 0 || y >= h) continue;
    double v = exp(-(double)(dx * dx + dy * dy) / denom);
    if (v < tiny) v = 0;
    // values are >= 0: their bit patterns order like the floats
    atomicMax((int*)(heat + ((size_t)pl * h + y) * w + x), __float_as_int((float)v));
  }
}

constexpr int kFocalBlock = 256;
constexpr int kFocalPerThread = 8;

// p = clamp(sigmoid(x), clip, 1 - clip);  loss = -log(p + e) (1 - p)^2 [t == 1]
//                                               - log(1 - p + e) p^2 (1 - t)^4
__global__ __launch_bounds__(kFocalBlock) void gaussian_focal_kernel(
    const float* __restrict__ x, const float* __restrict__ target, long n, float clip,
    float* __restrict__ grad /* d loss_i / d x_i, or null */, double* __restrict__ partial
    /* [blocks][2]: loss sum, positives */) {
  const float e = 1e-12f;
  double loss = 0, pos = 0;
  const long base = (long)blockIdx.x * kFocalBlock * kFocalPerThread;
#pragma unroll
  for (int u = 0; u < kFocalPerThread; ++u) {
    const long i = base + (long)u * kFocalBlock + threadIdx.x;
    if (i >= n) continue;
    const float xi = x[i], t = target[i];
    const float sg = 1.f / (1.f + expf(-xi));
    const float p = fminf(fmaxf(sg, clip), 1.f - clip);
    const bool live = sg >= clip && sg <= 1.f - clip;          // clamp passes the gradient
    const float q = 1.f - p;
    const float nw = (1.f - t) * (1.f - t) * (1.f - t) * (1.f - t);
    const float lp = logf(p + e), lq = logf(q + e);
    const bool is_pos = t == 1.f;
    const float li = (is_pos ? -lp * q * q : 0.f) + -lq * p * p * nw;
    loss += (double)li;
    pos += is_pos ? 1.0 : 0.0;
    if (grad) {
      float dp = nw * (p * p / (q + e) - 2.f * p * lq);
      if (is_pos) dp += -(q * q) / (p + e) + 2.f * q * lp;
      grad[i] = live ? dp * sg * (1.f - sg) : 0.f;
    }
  }
  __shared__ double sh[2][kFocalBlock / kWave];
  for (int off = kWave / 2; off > 0; off >>= 1) {
    loss += __shfl_down(loss, off, kWave);
    pos += __shfl_down(pos, off, kWave);
  }
  const int wave = threadIdx.x / kWave;
  if ((threadIdx.x & (kWave - 1)) == 0) {
    sh[0][wave] = loss;
    sh[1][wave] = pos;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double l = 0, c = 0;
    for (int k = 0; k < kFocalBlock / kWave; ++k) {
      l += sh[0][k];
      c += sh[1][k];
    }
    partial[(size_t)blockIdx.x * 2] = l;
    partial[(size_t)blockIdx.x * 2 + 1] = c;
  }
}

// fixed-order sum of the block partials (deterministic) -> out[0] = loss sum, out[1] = positives
__global__ __launch_bounds__(256) void focal_finish_kernel(const double* __restrict__ partial,
                                                           int blocks, float* __restrict__ out) {
  __shared__ double sh[2][256];
  double l = 0, c = 0;
  for (int b = threadIdx.x; b < blocks; b += 256) {
    l += partial[(size_t)b * 2];
    c += partial[(size_t)b * 2 + 1];
  }
  sh[0][threadIdx.x] = l;
  sh[1][threadIdx.x] = c;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      sh[0][threadIdx.x] += sh[0][threadIdx.x + s];
      sh[1][threadIdx.x] += sh[1][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = (float)sh[0][0];
    out[1] = (float)sh[1][0];
  }
}

}  // namespace
}  // namespace msmd

using namespace msmd;

MSMD_EXPORT int msmd_boxes_overlap_bev_f32(const float* boxes_a, int na, const float* boxes_b,
                                           int nb, float* out, msmd_stream_t stream) {
  if (na < 0 || nb < 0) return MSMD_ERR_INVALID_ARG;
  if ((long)na * nb == 0) return MSMD_OK;
  if (!boxes_a || !boxes_b || !out) return MSMD_ERR_INVALID_ARG;
  MSMD_LAUNCH(boxes_iou3d_kernel, dim3(ceil_div((long)na * nb, 256)), dim3(256), 0,
              (hipStream_t)stream, boxes_a, 5, boxes_b, 5, (const int32_t*)nullptr, 1, na, nb, 2,
              out);
  return launch_status();
}

MSMD_EXPORT int msmd_boxes_iou3d_f32(const float* boxes_a, int lda, const float* boxes_b, int ldb,
                                     const int32_t* nb_valid, int batch, int na, int nb, int mode,
                                     float* out, msmd_stream_t stream) {
  if (batch < 0 || na < 0 || nb < 0 || lda < 7 || ldb < 7 || (mode != 0 && mode != 1))
    return MSMD_ERR_INVALID_ARG;
  const long total = (long)batch * na * nb;
  if (total == 0) return MSMD_OK;
  if (total >= 2147483647L) return MSMD_ERR_RANGE;
  if (!boxes_a || !boxes_b || !out) return MSMD_ERR_INVALID_ARG;
  MSMD_LAUNCH(boxes_iou3d_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream,
              boxes_a, lda, boxes_b, ldb, nb_valid, batch, na, nb, mode, out);
  return launch_status();
}

MSMD_EXPORT int msmd_heatmap_gaussian_f32(const int32_t* plane, const int32_t* center_x,
                                          const int32_t* center_y, const int32_t* radius, int n,
                                          int planes, int h, int w, float* heatmap,
                                          msmd_stream_t stream) {
  if (n < 0 || planes < 1 || h < 1 || w < 1 || !heatmap) return MSMD_ERR_INVALID_ARG;
  if (n == 0) return MSMD_OK;
  if (!plane || !center_x || !center_y || !radius) return MSMD_ERR_INVALID_ARG;
  MSMD_LAUNCH(heatmap_gaussian_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, plane,
              center_x, center_y, radius, planes, h, w, heatmap);
  return launch_status();
}

MSMD_EXPORT size_t msmd_gaussian_focal_workspace_bytes(int64_t n) {
  if (n < 0) return 0;
  const long blocks = (n + kFocalBlock * kFocalPerThread - 1) / (kFocalBlock * kFocalPerThread);
  return align_up((size_t)(blocks > 0 ? blocks : 1) * 2 * sizeof(double));
}

MSMD_EXPORT int msmd_gaussian_focal_f32(const float* logits, const float* target, int64_t n,
                                        float clip, float* grad, float* sums, void* workspace,
                                        size_t workspace_bytes, msmd_stream_t stream) {
  if (n < 0 || !sums || !(clip >= 0.f && clip < 0.5f)) return MSMD_ERR_INVALID_ARG;
  if (n > 0 && (!logits || !target)) return MSMD_ERR_INVALID_ARG;
  if (!workspace || workspace_bytes < msmd_gaussian_focal_workspace_bytes(n))
    return MSMD_ERR_WORKSPACE;
  const int blocks = ceil_div(n, (long)kFocalBlock * kFocalPerThread);
  hipStream_t st = (hipStream_t)stream;
  if (blocks > 0)
    MSMD_LAUNCH(gaussian_focal_kernel, dim3(blocks), dim3(kFocalBlock), 0, st, logits, target,
                (long)n, clip, grad, (double*)workspace);
  MSMD_LAUNCH(focal_finish_kernel, dim3(1), dim3(256), 0, st, (const double*)workspace, blocks,
              sums);
  return launch_status();
}
