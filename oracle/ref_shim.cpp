// ref_shim.cpp -- C-callable entry points onto the REFERENCE's own CPU code.
//
// TEST INFRASTRUCTURE ONLY (see oracle/msmd_oracle.c).  This file contains no
// reference source: it only calls functions that are compiled from where they
// lie under $(REF) (oracle/Makefile target _ref/libmsmd_ref.so):
//   mmdet3d/ops/voxel/src/voxelization_cpu.cpp        voxelization::hard_voxelize_cpu
//   mmdet3d/ops/spconv/include/spconv/geometry.h      getIndicePairsSubM / getIndicePairsConv
//   mmdet3d/ops/spconv/src/reordering.cc              SparseGatherFunctor / SparseScatterAddFunctor (CPU)
// The per-offset loops around gather -> torch::mm -> scatter-add are the only
// thing re-stated here (spconv_ops.h:260-361 forward, :363-456 backward: the
// header cannot be built, it drags in ATen/cuda and the GPU functor
// specialisations); they are marked below.
// tensorview.h needs <cuda_runtime_api.h>; the image ships one with triton
// (python3.10/dist-packages/triton/backends/nvidia/include) and the Makefile
// points -I there -- no stand-in header is written.
#include <torch/torch.h>

#include <spconv/geometry.h>
#include <spconv/reordering.h>
#include <tensorview/tensorview.h>

#include <cstdint>
#include <vector>

namespace voxelization {
int hard_voxelize_cpu(const at::Tensor& points, at::Tensor& voxels,
                      at::Tensor& coors, at::Tensor& num_points_per_voxel,
                      const std::vector<float> voxel_size,
                      const std::vector<float> coors_range,
                      const int max_points, const int max_voxels,
                      const int NDim);
}

namespace {
template <typename T>
tv::TensorView<T> view(T* p, std::initializer_list<int> dims) {
  tv::Shape s;
  for (int d : dims) s.push_back(d);
  return tv::TensorView<T>(p, s);
}
}  // namespace

extern "C" {

int ref_hard_voxelize(const float* points, int n, int nfeat,
                      const float* voxel_size, const float* range,
                      int max_points, int max_voxels, float* voxels,
                      int32_t* coors, int32_t* num_points) {
  auto f = torch::TensorOptions().dtype(torch::kFloat32);
  auto i = torch::TensorOptions().dtype(torch::kInt32);
  auto tp = torch::from_blob(const_cast<float*>(points), {n, nfeat}, f);
  auto tv_ = torch::from_blob(voxels, {max_voxels, max_points, nfeat}, f);
  auto tc = torch::from_blob(coors, {max_voxels, 3}, i);
  auto tn = torch::from_blob(num_points, {max_voxels}, i);
  std::vector<float> vs(voxel_size, voxel_size + 3), rg(range, range + 6);
  return voxelization::hard_voxelize_cpu(tp, tv_, tc, tn, vs, rg, max_points,
                                         max_voxels, 3);
}

// indice_pairs [kvol,2,n] pre-filled -1, indice_num [kvol] zero, grid
// [batch*out_volume] pre-filled -1, out_indices [n*kvol,4].
int ref_get_indice_pairs(const int32_t* indices, int n, int batch,
                         const int* out_shape, const int* ksize,
                         const int* stride, const int* pad, const int* dil,
                         int subm, int32_t* out_indices, int32_t* indice_pairs,
                         int32_t* indice_num, int32_t* grid) {
  int kvol = ksize[0] * ksize[1] * ksize[2];
  long vol = (long)out_shape[0] * out_shape[1] * out_shape[2];
  auto in_v = view<const int>(indices, {n, 4});
  auto grid_v = view<int>(grid, {(int)(vol * batch)});
  auto pair_v = view<int>(indice_pairs, {kvol, 2, n});
  auto num_v = view<int>(indice_num, {kvol});
  int st[3], pd[3];
  for (int i = 0; i < 3; ++i) {  // spconv_ops.h:73-85
    st[i] = subm ? 1 : stride[i];
    pd[i] = subm ? ksize[i] / 2 : pad[i];
  }
  if (subm)
    return spconv::getIndicePairsSubM<int, int, 3>(in_v, grid_v, pair_v, num_v,
                                                   ksize, st, pd, dil,
                                                   out_shape);
  auto out_v = view<int>(out_indices, {n * kvol, 4});
  return spconv::getIndicePairsConv<int, int, 3>(in_v, out_v, grid_v, pair_v,
                                                 num_v, ksize, st, pd, dil,
                                                 out_shape);
}

// RESTATED LOOP (spconv_ops.h:260-361) around the reference's CPU gather /
// scatter-add functors and torch::mm.
void ref_indice_conv_fwd(const float* feat, int n_in, int cin,
                         const float* filters, int kvol, int cout,
                         const int32_t* pairs, const int32_t* num, int ld,
                         int n_out, int subm, float* out) {
  auto f = torch::TensorOptions().dtype(torch::kFloat32);
  auto features = torch::from_blob(const_cast<float*>(feat), {n_in, cin}, f);
  auto w = torch::from_blob(const_cast<float*>(filters), {kvol, cin, cout}, f);
  auto output = torch::from_blob(out, {n_out, cout}, f);
  output.zero_();
  int centre = 0, hot_max = 0;
  for (int k = 0; k < kvol; ++k)
    if (num[k] > hot_max) { hot_max = num[k]; centre = k; }
  auto ibuf = torch::zeros({hot_max + 1, cin}, f);
  auto obuf = torch::zeros({hot_max + 1, cout}, f);
  if (subm) torch::mm_out(output, features, w[centre]);
  spconv::functor::SparseGatherFunctor<tv::CPU, float, int> gather;
  spconv::functor::SparseScatterAddFunctor<tv::CPU, float, int> scatter;
  for (int k = 0; k < kvol; ++k) {
    int hot = num[k];
    if (hot <= 0 || (subm && k == centre)) continue;
    auto ib = torch::from_blob(ibuf.data_ptr<float>(), {hot, cin}, f);
    auto ob = torch::from_blob(obuf.data_ptr<float>(), {hot, cout}, f);
    gather(tv::CPU(), view<float>(ibuf.data_ptr<float>(), {hot_max + 1, cin}),
           view<const float>(feat, {n_in, cin}),
           view<const int>(pairs + ((long)k * 2 + 0) * ld, {ld}), hot);
    torch::mm_out(ob, ib, w[k]);
    scatter(tv::CPU(), view<float>(out, {n_out, cout}),
            view<const float>(obuf.data_ptr<float>(), {hot_max + 1, cout}),
            view<const int>(pairs + ((long)k * 2 + 1) * ld, {ld}), hot, true);
  }
}

// RESTATED LOOP (indiceConvBackward, spconv_ops.h:363-456, _inverse = 0) around the
// reference's CPU gather / scatter-add functors and torch::mm: per offset
//   dW[k] = gather(features)^T . gather(outGrad),
//   dIn  += scatter(gather(outGrad) . W[k]^T);
// the SubM centre offset is taken as two dense mms (:398-402).
void ref_indice_conv_bwd(const float* feat, int n_in, int cin,
                         const float* filters, int kvol, int cout,
                         const float* out_grad, int n_out,
                         const int32_t* pairs, const int32_t* num, int ld,
                         int subm, float* din, float* dw) {
  auto f = torch::TensorOptions().dtype(torch::kFloat32);
  auto features = torch::from_blob(const_cast<float*>(feat), {n_in, cin}, f);
  auto w = torch::from_blob(const_cast<float*>(filters), {kvol, cin, cout}, f);
  auto g = torch::from_blob(const_cast<float*>(out_grad), {n_out, cout}, f);
  auto in_grad = torch::from_blob(din, {n_in, cin}, f);
  auto w_grad = torch::from_blob(dw, {kvol, cin, cout}, f);
  in_grad.zero_();
  w_grad.zero_();
  int centre = 0, hot_max = 0;
  for (int k = 0; k < kvol; ++k)
    if (num[k] > hot_max) { hot_max = num[k]; centre = k; }
  auto ibuf = torch::zeros({hot_max + 1, cin}, f);
  auto obuf = torch::zeros({hot_max + 1, cout}, f);
  if (subm) {
    auto sub = w_grad[centre];
    torch::mm_out(sub, features.t(), g);
    torch::mm_out(in_grad, g, w[centre].t());
  }
  spconv::functor::SparseGatherFunctor<tv::CPU, float, int> gather_in, gather_out;
  spconv::functor::SparseScatterAddFunctor<tv::CPU, float, int> scatter;
  for (int k = 0; k < kvol; ++k) {
    int hot = num[k];
    if (hot <= 0 || (subm && k == centre)) continue;
    gather_in(tv::CPU(), view<float>(ibuf.data_ptr<float>(), {hot_max + 1, cin}),
              view<const float>(feat, {n_in, cin}),
              view<const int>(pairs + ((long)k * 2 + 0) * ld, {ld}), hot);
    gather_out(tv::CPU(), view<float>(obuf.data_ptr<float>(), {hot_max + 1, cout}),
               view<const float>(out_grad, {n_out, cout}),
               view<const int>(pairs + ((long)k * 2 + 1) * ld, {ld}), hot);
    auto sub = w_grad[k];
    auto ob = torch::from_blob(obuf.data_ptr<float>(), {hot, cout}, f);
    auto ib = torch::from_blob(ibuf.data_ptr<float>(), {hot, cin}, f);
    torch::mm_out(sub, ib.t(), ob);
    torch::mm_out(ib, ob, w[k].t());
    scatter(tv::CPU(), view<float>(din, {n_in, cin}),
            view<const float>(ibuf.data_ptr<float>(), {hot_max + 1, cin}),
            view<const int>(pairs + ((long)k * 2 + 0) * ld, {ld}), hot, true);
  }
}

}  // extern "C"
