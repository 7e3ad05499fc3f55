// image_glue.hip -- the image-feature side of the virtual points
// and the channels-last BEV hand-over to the dense tail (a12 -> f1).
//
//   fg_gather      MSMDFusionDetector.get_foreground2D's per-(sample, camera)
//                  loop (MSMDFusion.py:195-224) for ALL cameras of ALL samples in
//                  one launch: pixel -> feature-map cell, 49-channel read, and
//                  the two row layouts the reference concatenates on the host
//                  ([pts15 | feat] and [feat | depth | lidar2img16]).
//   fg_scatter_add its backward (index's backward = index_put(accumulate)).
//   depth_canvas   depth_aware_channel_compression's B*6 index_put_ calls
//                  (MSMDFusion.py:336-356) as one launch.
//   bev_nhwc       dense() + view(N, C*D, H, W) + torch.cat (MSMDFusion.py:436-440)
//                  written straight into one channels-last [B,H,W,Ctot] buffer.
// All HBM-bound permutations of a few MB; the point is launch count (the
// reference issues ~40 small host->device copies and ~60 launches per scale).
#include "common.hpp"

namespace msmd {
namespace {

struct Strides4 {
  long p, c, h, w;   // element strides of the [P,C,H,W] feature map (any memory format)
};

// (fg_pxl * downscale_factor).long(): the product in the ARRAY's dtype (numpy keeps
// float32 for float32 * python float), truncation toward zero.
template <typename T>
__device__ __forceinline__ long cell_of_pixel(T v, double scale) {
  T prod = v * (T)scale;
  return (long)prod;
}

// python-style negative wrap of an index tensor entry; -1 = out of range
__device__ __forceinline__ long wrap_index(long i, int size) {
  if (i < 0) i += size;
  return (i < 0 || i >= size) ? -1 : i;
}

// one 16-lane group per point; lanes stride the channels
template <typename T>
__global__ __launch_bounds__(256) void fg_gather_kernel(
    const float* __restrict__ img, Strides4 st, int P, int C, int H, int W,
    const T* __restrict__ pix /* [n,3] x,y,depth */, const int32_t* __restrict__ plane,
    double scale, const float* __restrict__ pts, int pts_dim,
    const float* __restrict__ lidar2img /* [P,16] */, int n, float* __restrict__ fg_pcd,
    float* __restrict__ score_in, int32_t* __restrict__ cells /* [n] or null */,
    int* __restrict__ bad) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int i = (int)(t >> 4), sub = (int)(t & 15);
  if (i >= n) return;
  const int pl = plane[i];
  long w = wrap_index(cell_of_pixel(pix[(size_t)i * 3 + 0], scale), W);
  long h = wrap_index(cell_of_pixel(pix[(size_t)i * 3 + 1], scale), H);
  const bool ok = pl >= 0 && pl < P && w >= 0 && h >= 0;
  if (!ok && sub == 0) atomicAdd(bad, 1);
  if (cells && sub == 0) cells[i] = ok ? (int32_t)((pl * (long)H + h) * W + w) : -1;
  const int row = pts_dim + C;
  float* o0 = fg_pcd + (size_t)i * row;
  float* o1 = score_in + (size_t)i * (C + 17);
  const float* src = img + (ok ? pl * st.p + h * st.h + w * st.w : 0);
  for (int c = sub; c < C; c += 16) {
    float v = ok ? src[c * st.c] : 0.f;
    o0[pts_dim + c] = v;
    o1[c] = v;
  }
  for (int c = sub; c < pts_dim; c += 16) o0[c] = pts[(size_t)i * pts_dim + c];
  if (sub == 0) o1[C] = (float)pix[(size_t)i * 3 + 2];
  {
    const int c = sub;   // 16 lanes, 16 matrix entries
    o1[C + 1 + c] = (pl >= 0 && pl < P) ? lidar2img[(size_t)pl * 16 + c] : 0.f;
  }
}

// The inference form of get_foreground2D's whole tail (MSMDFusion.py:213-228): gather, the
// score_net row [feat | depth | lidar2img] . w + b, ReLU, and fg_pcd = [pts | feat * score]
// in ONE pass -- score_in (264 B per point written, read back by the Linear, and the cat /
// multiply passes over fg_pcd after it) never exists.  Points [0, n_scaled) are scaled, the
// rest keep the plain gather (reference_quirks: :229-234 writes the scaled channels back for
// sample 0 -- and 1 when B == 2 -- only).  16 lanes per point; the 66-term dot product is
// summed per lane over its channels (c = sub, sub + 16, ...), then across the 16 lanes by
// xor-shuffles: a fixed order, fp32.
template <typename T>
__global__ __launch_bounds__(256) void fg_gather_scored_kernel(
    const float* __restrict__ img, Strides4 st, int P, int C, int H, int W,
    const T* __restrict__ pix, const int32_t* __restrict__ plane, double scale,
    const float* __restrict__ pts, int pts_dim, const float* __restrict__ lidar2img,
    const float* __restrict__ sw /* [C + 17] */, const float* __restrict__ sb /* [1] */,
    int n, int n_scaled, float* __restrict__ fg_pcd, int* __restrict__ bad) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int i = (int)(t >> 4), sub = (int)(t & 15);
  const bool live = i < n;
  const int ii = live ? i : 0;
  const int pl = live ? plane[ii] : 0;
  long w = wrap_index(cell_of_pixel(pix[(size_t)ii * 3 + 0], scale), W);
  long h = wrap_index(cell_of_pixel(pix[(size_t)ii * 3 + 1], scale), H);
  const bool ok = live && pl >= 0 && pl < P && w >= 0 && h >= 0;
  if (live && !ok && sub == 0) atomicAdd(bad, 1);
  const float* src = img + (ok ? pl * st.p + h * st.h + w * st.w : 0);
  float v[4];           // C <= 64: this lane's channels sub, sub + 16, sub + 32, sub + 48
  float acc = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = sub + 16 * q;
    v[q] = (ok && c < C) ? src[c * st.c] : 0.f;
    acc = (c < C) ? fmaf(v[q], sw[c], acc) : acc;
  }
  if (sub == 0) acc = fmaf((float)pix[(size_t)ii * 3 + 2], sw[C], acc);
  const float m = (pl >= 0 && pl < P) ? lidar2img[(size_t)pl * 16 + sub] : 0.f;
  acc = fmaf(m, sw[C + 1 + sub], acc);
#pragma unroll
  for (int o = 8; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 16);
  float score = fmaxf(acc + sb[0], 0.f);
  if (i >= n_scaled) score = 1.f;
  if (!live) return;
  float* o0 = fg_pcd + (size_t)i * (pts_dim + C);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = sub + 16 * q;
    if (c < C) o0[pts_dim + c] = i < n_scaled ? v[q] * score : v[q];
  }
  for (int c = sub; c < pts_dim; c += 16) o0[c] = pts[(size_t)i * pts_dim + c];
}

// grad_img[plane,c,h,w] += grad[i, col0 + c]   (cells from the forward)
__global__ __launch_bounds__(256) void fg_scatter_add_kernel(
    const float* __restrict__ grad, int grad_stride, int col0, const int32_t* __restrict__ cells,
    int n, int C, int H, int W, Strides4 st, float* __restrict__ grad_img) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int i = (int)(t >> 4), sub = (int)(t & 15);
  if (i >= n) return;
  const int cell = cells[i];
  if (cell < 0) return;
  const int w = cell % W, h = (cell / W) % H, pl = cell / (W * H);
  float* dst = grad_img + pl * st.p + h * st.h + w * st.w;
  const float* g = grad + (size_t)i * grad_stride + col0;
  for (int c = sub; c < C; c += 16) unsafeAtomicAdd(dst + c * st.c, g[c]);
}

// canvas[plane, y, x] = depth of the LAST row that lands on the pixel (what a
// sequential index_put_ leaves behind; the reference's CUDA index_put_ with
// duplicate indices is unordered -- this picks the deterministic answer).
template <typename T>
__global__ __launch_bounds__(256) void canvas_claim_kernel(const T* __restrict__ pix,
                                                           const int32_t* __restrict__ plane,
                                                           int n, int P, int H, int W,
                                                           int32_t* __restrict__ winner,
                                                           int* __restrict__ bad) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int pl = plane[i];
  long x = wrap_index((long)pix[(size_t)i * 3 + 0], W);
  long y = wrap_index((long)pix[(size_t)i * 3 + 1], H);
  if (pl < 0 || pl >= P || x < 0 || y < 0) {
    atomicAdd(bad, 1);
    return;
  }
  atomicMax(&winner[((long)pl * H + y) * W + x], i);
}
template <typename T>
__global__ __launch_bounds__(256) void canvas_fill_kernel(const T* __restrict__ pix,
                                                          const int32_t* __restrict__ winner,
                                                          long cells, float* __restrict__ canvas) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= cells) return;
  const int r = winner[e];
  canvas[e] = r < 0 ? 0.f : (float)pix[(size_t)r * 3 + 2];
}

// ---------------------------------------------------------------- bev nhwc ----
// out[b, y, x, coff + ch*D + z] = feat[row, ch]; one 64-lane wave per 2 rows would
// waste lanes on narrow c, so: thread = (row, ch) with ch fastest (coalesced feature
// side; the BEV side has stride D floats, and the z-neighbour voxel fills the gaps).
template <bool SCATTER>
__global__ __launch_bounds__(256) void bev_nhwc_kernel(float* __restrict__ feat,
                                                       const int32_t* __restrict__ idx, int n,
                                                       int c, int D, int H, int W,
                                                       float* __restrict__ bev, int ctot,
                                                       int coff) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)n * c) return;
  const int row = (int)(e / c), ch = (int)(e - (long)row * c);
  const int4 r = ((const int4*)idx)[row];   // b, z, y, x
  float* p = bev + (((long)r.x * H + r.z) * W + r.w) * ctot + coff + ch * D + r.y;
  if (SCATTER) *p = feat[e]; else feat[e] = *p;
}

}  // namespace
}  // namespace msmd

using namespace msmd;

static bool strides_ok(const int64_t* s) { return s && s[0] >= 0 && s[1] >= 0 && s[2] >= 0 && s[3] >= 0; }

MSMD_EXPORT int msmd_fg_gather_f32(const float* img_feat, const int64_t* strides, int planes,
                                   int c, int h, int w, const void* pixels, int pixel_is_f64,
                                   const int32_t* plane, double downscale, const float* pts,
                                   int pts_dim, const float* lidar2img, int n, float* fg_pcd,
                                   float* score_in, int32_t* cells, int32_t* n_bad,
                                   msmd_stream_t stream) {
  if (n < 0 || planes < 1 || c < 1 || h < 1 || w < 1 || pts_dim < 0 || !strides_ok(strides) ||
      !n_bad)
    return MSMD_ERR_INVALID_ARG;
  if ((double)planes * h * w >= 2147483647.0) return MSMD_ERR_RANGE;
  hipStream_t st = (hipStream_t)stream;
  hipMemsetAsync(n_bad, 0, sizeof(int32_t), st);
  if (n == 0) return MSMD_OK;
  if (!img_feat || !pixels || !plane || !lidar2img || !fg_pcd || !score_in || (pts_dim && !pts))
    return MSMD_ERR_INVALID_ARG;
  Strides4 s{strides[0], strides[1], strides[2], strides[3]};
  const int grid = ceil_div((long)n * 16, 256);
  if (pixel_is_f64)
    MSMD_LAUNCH(fg_gather_kernel<double>, dim3(grid), dim3(256), 0, st, img_feat, s, planes, c, h,
                w, (const double*)pixels, plane, downscale, pts, pts_dim, lidar2img, n, fg_pcd,
                score_in, cells, n_bad);
  else
    MSMD_LAUNCH(fg_gather_kernel<float>, dim3(grid), dim3(256), 0, st, img_feat, s, planes, c, h,
                w, (const float*)pixels, plane, downscale, pts, pts_dim, lidar2img, n, fg_pcd,
                score_in, cells, n_bad);
  return launch_status();
}

MSMD_EXPORT int msmd_fg_gather_scored_f32(const float* img_feat, const int64_t* strides, int planes,
                                          int c, int h, int w, const void* pixels,
                                          int pixel_is_f64, const int32_t* plane,
                                          double downscale, const float* pts, int pts_dim,
                                          const float* lidar2img, const float* score_weight,
                                          const float* score_bias, int n, int n_scaled,
                                          float* fg_pcd, int32_t* n_bad, msmd_stream_t stream) {
  if (n < 0 || planes < 1 || c < 1 || c > 64 || h < 1 || w < 1 || pts_dim < 0 ||
      !strides_ok(strides) || !n_bad || n_scaled < 0)
    return MSMD_ERR_INVALID_ARG;
  if ((double)planes * h * w >= 2147483647.0) return MSMD_ERR_RANGE;
  hipStream_t st = (hipStream_t)stream;
  hipMemsetAsync(n_bad, 0, sizeof(int32_t), st);
  if (n == 0) return MSMD_OK;
  if (!img_feat || !pixels || !plane || !lidar2img || !fg_pcd || !score_weight || !score_bias ||
      (pts_dim && !pts))
    return MSMD_ERR_INVALID_ARG;
  Strides4 s{strides[0], strides[1], strides[2], strides[3]};
  const int grid = ceil_div((long)n * 16, 256);
  if (pixel_is_f64)
    MSMD_LAUNCH(fg_gather_scored_kernel<double>, dim3(grid), dim3(256), 0, st, img_feat, s, planes,
                c, h, w, (const double*)pixels, plane, downscale, pts, pts_dim, lidar2img,
                score_weight, score_bias, n, n_scaled, fg_pcd, n_bad);
  else
    MSMD_LAUNCH(fg_gather_scored_kernel<float>, dim3(grid), dim3(256), 0, st, img_feat, s, planes,
                c, h, w, (const float*)pixels, plane, downscale, pts, pts_dim, lidar2img,
                score_weight, score_bias, n, n_scaled, fg_pcd, n_bad);
  return launch_status();
}

MSMD_EXPORT int msmd_fg_scatter_add_f32(const float* grad, int grad_stride, int col0,
                                        const int32_t* cells, int n, int planes, int c, int h,
                                        int w, const int64_t* strides, float* grad_img,
                                        msmd_stream_t stream) {
  if (n < 0 || planes < 1 || c < 1 || h < 1 || w < 1 || grad_stride < col0 + c || col0 < 0 ||
      !strides_ok(strides))
    return MSMD_ERR_INVALID_ARG;
  if (n == 0) return MSMD_OK;
  if (!grad || !cells || !grad_img) return MSMD_ERR_INVALID_ARG;
  Strides4 s{strides[0], strides[1], strides[2], strides[3]};
  MSMD_LAUNCH(fg_scatter_add_kernel, dim3(ceil_div((long)n * 16, 256)), dim3(256), 0,
              (hipStream_t)stream, grad, grad_stride, col0, cells, n, c, h, w, s, grad_img);
  return launch_status();
}

MSMD_EXPORT size_t msmd_depth_canvas_workspace_bytes(int planes, int h, int w) {
  if (planes < 1 || h < 1 || w < 1) return 0;
  return align_up((size_t)planes * h * w * sizeof(int32_t));
}

MSMD_EXPORT int msmd_depth_canvas_f32(const void* pixels, int pixel_is_f64, const int32_t* plane,
                                      int n, int planes, int h, int w, float* canvas,
                                      int32_t* n_bad, void* workspace, size_t workspace_bytes,
                                      msmd_stream_t stream) {
  if (n < 0 || planes < 1 || h < 1 || w < 1 || !canvas || !n_bad) return MSMD_ERR_INVALID_ARG;
  if (n > 0 && (!pixels || !plane)) return MSMD_ERR_INVALID_ARG;
  const size_t cells = (size_t)planes * h * w;
  if (cells >= 2147483647u) return MSMD_ERR_RANGE;
  if (!workspace || workspace_bytes < cells * sizeof(int32_t)) return MSMD_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  int32_t* winner = (int32_t*)workspace;
  hipMemsetAsync(winner, 0xff, cells * sizeof(int32_t), st);
  hipMemsetAsync(n_bad, 0, sizeof(int32_t), st);
  const int grid_c = ceil_div((long)cells, 256);
  if (pixel_is_f64) {
    if (n > 0)
      MSMD_LAUNCH(canvas_claim_kernel<double>, dim3(ceil_div(n, 256)), dim3(256), 0, st,
                  (const double*)pixels, plane, n, planes, h, w, winner, n_bad);
    MSMD_LAUNCH(canvas_fill_kernel<double>, dim3(grid_c), dim3(256), 0, st,
                (const double*)pixels, winner, (long)cells, canvas);
  } else {
    if (n > 0)
      MSMD_LAUNCH(canvas_claim_kernel<float>, dim3(ceil_div(n, 256)), dim3(256), 0, st,
                  (const float*)pixels, plane, n, planes, h, w, winner, n_bad);
    MSMD_LAUNCH(canvas_fill_kernel<float>, dim3(grid_c), dim3(256), 0, st, (const float*)pixels,
                winner, (long)cells, canvas);
  }
  return launch_status();
}

static int bev_args(int n, int c, int batch, const int* shape, int ctot, int coff) {
  if (n < 0 || c < 1 || batch < 1 || !shape || shape[0] < 1 || shape[1] < 1 || shape[2] < 1 ||
      coff < 0 || ctot < coff + c * shape[0])
    return MSMD_ERR_INVALID_ARG;
  if ((double)batch * shape[1] * shape[2] * ctot >= 9.0e18) return MSMD_ERR_RANGE;
  return MSMD_OK;
}

MSMD_EXPORT int msmd_bev_scatter_nhwc_f32(const float* feat, const int32_t* indices, int n, int c,
                                          int batch_size, const int* spatial_shape, float* bev,
                                          int total_channels, int channel_offset,
                                          msmd_stream_t stream) {
  int rc = bev_args(n, c, batch_size, spatial_shape, total_channels, channel_offset);
  if (rc) return rc;
  if (n == 0) return MSMD_OK;
  if (!feat || !indices || !bev) return MSMD_ERR_INVALID_ARG;
  MSMD_LAUNCH(bev_nhwc_kernel<true>, dim3(ceil_div((long)n * c, 256)), dim3(256), 0,
              (hipStream_t)stream, (float*)feat, indices, n, c, spatial_shape[0], spatial_shape[1],
              spatial_shape[2], bev, total_channels, channel_offset);
  return launch_status();
}

MSMD_EXPORT int msmd_bev_gather_nhwc_f32(const float* bev, const int32_t* indices, int n, int c,
                                         int batch_size, const int* spatial_shape, float* feat,
                                         int total_channels, int channel_offset,
                                         msmd_stream_t stream) {
  int rc = bev_args(n, c, batch_size, spatial_shape, total_channels, channel_offset);
  if (rc) return rc;
  if (n == 0) return MSMD_OK;
  if (!feat || !indices || !bev) return MSMD_ERR_INVALID_ARG;
  MSMD_LAUNCH(bev_nhwc_kernel<false>, dim3(ceil_div((long)n * c, 256)), dim3(256), 0,
              (hipStream_t)stream, feat, indices, n, c, spatial_shape[0], spatial_shape[1],
              spatial_shape[2], (float*)bev, total_channels, channel_offset);
  return launch_status();
}
