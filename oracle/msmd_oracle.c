/*
 * msmd_oracle.c -- CPU oracle for the MSMDFusion sparse-voxel hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file restates, in plain C, the algorithms of
 * the reference's CPU code so that tests/ (and bench.py's cpu_baseline leg and
 * __graft_entry__.smoke()) can check the HIP library against them.  Nothing in
 * msmdfusion_amd/ may import, link or call it: the product path is HIP only.
 *
 * Pinning: tests/test_oracle_cpu.py compares every function here that has a
 * compilable reference counterpart (voxelization, rulebooks, gather/scatter
 * conv forward and backward) against the reference's own C++ built into
 * oracle/_ref/ (live in the build container; through the committed outputs
 * tests/golden/reference_vectors.npz everywhere), and against the literal
 * vectors of the reference's tests.  sparse_add,
 * dense() and the modality split have no runnable reference here (spconv 2.x
 * and numba are absent); they are pinned by hand-derived cases and by
 * cross-checks against dense torch ops -- see DESIGN.md "parity pins".
 *
 * Each function cites the reference lines it follows (paths relative to the
 * reference checkout, /root/reference in the build container).
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off: float results must not
 * depend on FMA contraction, the HIP kernels are built the same way).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

#ifdef _OPENMP
#include <omp.h>
#endif
/* Threads used by the OpenMP loops of the conv restatement (cpu_baseline). */
ORC_API int orc_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
  return omp_get_max_threads();
#else
  (void)n;
  return 1;
#endif
}

/* ------------------------------------------------------------------------- *
 * Hard voxelization.
 * Follows mmdet3d/ops/voxel/src/voxelization_cpu.cpp:8-40 (coordinate of a
 * point: float subtract, float divide, floor, reject c<0 || c>=grid, stored
 * reversed as z,y,x) and :44-99 (first-appearance voxel ids, the break at
 * max_voxels, slots < max_points), grid from :119-122 (round of the float
 * quotient).  voxels/coors/num_points must be zeroed by the caller exactly as
 * mmdet3d/ops/voxel/voxelize.py:46-50 does.
 * ------------------------------------------------------------------------- */
ORC_API void orc_grid_size(const float* voxel_size, const float* range,
                           int* grid /* x,y,z */) {
  for (int i = 0; i < 3; ++i)
    grid[i] = (int)roundf((range[3 + i] - range[i]) / voxel_size[i]);
}

static int point_coor(const float* p, const float* voxel_size,
                      const float* range, const int* grid, int* zyx) {
  for (int j = 0; j < 3; ++j) {
    int c = (int)floorf((p[j] - range[j]) / voxel_size[j]);
    if (c < 0 || c >= grid[j]) return 0;
    zyx[2 - j] = c;
  }
  return 1;
}

ORC_API int orc_hard_voxelize(const float* points, int n, int nfeat,
                              const float* voxel_size, const float* range,
                              int max_points, int max_voxels, float* voxels,
                              int32_t* coors, int32_t* num_points) {
  int grid[3];
  orc_grid_size(voxel_size, range, grid);
  size_t cells = (size_t)grid[0] * grid[1] * grid[2];
  /* coor_to_voxelidx, stored +1 so calloc's zero means "no voxel yet" */
  int32_t* cell2vox = (int32_t*)calloc(cells, sizeof(int32_t));
  int voxel_num = 0;
  for (int i = 0; i < n; ++i) {
    int c[3];
    if (!point_coor(points + (size_t)i * nfeat, voxel_size, range, grid, c))
      continue;
    size_t cell = ((size_t)c[0] * grid[1] + c[1]) * grid[0] + c[2];
    int v = cell2vox[cell] - 1;
    if (v == -1) {
      v = voxel_num;
      if (max_voxels != -1 && voxel_num >= max_voxels) break;
      voxel_num += 1;
      cell2vox[cell] = v + 1;
      for (int k = 0; k < 3; ++k) coors[(size_t)v * 3 + k] = c[k];
    }
    int num = num_points[v];
    if (max_points == -1 || num < max_points) {
      memcpy(voxels + ((size_t)v * max_points + num) * nfeat,
             points + (size_t)i * nfeat, sizeof(float) * nfeat);
      num_points[v] = num + 1;
    }
  }
  free(cell2vox);
  return voxel_num;
}

/* HardSimpleVFE: mmdet3d/models/voxel_encoders/voxel_encoder.py:44-46
 * (sum over the point slots, padding included, divided by the count). */
ORC_API void orc_voxel_mean(const float* voxels, const int32_t* num_points,
                            int m, int max_points, int nfeat, int out_feat,
                            float* out) {
  for (int v = 0; v < m; ++v)
    for (int f = 0; f < out_feat; ++f) {
      float s = 0.f;
      for (int p = 0; p < max_points; ++p)
        s += voxels[((size_t)v * max_points + p) * nfeat + f];
      out[(size_t)v * out_feat + f] = s / (float)num_points[v];
    }
}

/* ------------------------------------------------------------------------- *
 * Rulebooks.
 * valid_out_pos restates getValidOutPos, spconv/geometry.h:24-85: for one
 * input position enumerate the output positions it reaches and the kernel
 * offset (row-major over the kernel, :62-73) through which it does.
 * C integer division (truncation) is kept as in the reference.
 * ------------------------------------------------------------------------- */
static int valid_out_pos(const int* in_pos, const int* ksize, const int* stride,
                         const int* pad, const int* dil, const int* out_shape,
                         int* out /* [kvol][4]: z,y,x,offset */) {
  int lo[3], up[3], cnt[3], csize[3], npts = 1, found = 0;
  for (int i = 0; i < 3; ++i) {
    lo[i] = (in_pos[i] - (ksize[i] - 1) * dil[i] - 1 + stride[i] + pad[i]) /
            stride[i];
    up[i] = (in_pos[i] + pad[i]) / stride[i];
  }
  for (int i = 0; i < 3; ++i) {
    csize[i] = (up[i] - lo[i]) / dil[i] + 1;
    npts *= csize[i];
    cnt[i] = 0;
  }
  for (int p = 0; p < npts; ++p) {
    int ok = 1, m = 1, offset = 0;
    for (int j = 2; j >= 0; --j) {
      int val = up[j] - cnt[j] * dil[j];
      out[found * 4 + j] = val;
      if (val < 0 || val > out_shape[j] - 1) ok = 0;
      offset += m * (in_pos[j] - val * stride[j] + pad[j]) / dil[j];
      m *= ksize[j];
    }
    out[found * 4 + 3] = offset;
    if (ok) ++found;
    cnt[2] += 1;
    for (int c = 2; c >= 0; --c)
      if (cnt[c] == csize[c] && c > 0) {
        cnt[c - 1] += 1;
        cnt[c] = 0;
      }
  }
  return found;
}

static size_t lin_id(const int* zyx, const int* shape) {
  return ((size_t)zyx[0] * shape[1] + zyx[1]) * shape[2] + zyx[2];
}

/* getIndicePairsSubM (geometry.h:247-297) when subm != 0, else
 * getIndicePairsConv (geometry.h:144-194); parameter handling of
 * spconv_ops.h:73-85 (SubM forces stride 1, padding ksize/2).
 * indice_pairs is [kvol,2,n] pre-filled with -1, indice_num [kvol] zeroed,
 * out_indices [n*kvol,4] (strided only).  Returns numActOut.
 * Output row order is the CPU reference's first-touch order; tests
 * canonicalise before comparing with the GPU path. */
ORC_API int orc_get_indice_pairs(const int32_t* indices, int n, int batch,
                                 const int* out_shape, const int* ksize,
                                 const int* stride_in, const int* pad_in,
                                 const int* dil, int subm,
                                 int32_t* out_indices, int32_t* indice_pairs,
                                 int32_t* indice_num) {
  int stride[3], pad[3];
  int kvol = ksize[0] * ksize[1] * ksize[2];
  for (int i = 0; i < 3; ++i) {
    stride[i] = subm ? 1 : stride_in[i];
    pad[i] = subm ? ksize[i] / 2 : pad_in[i];
  }
  size_t vol = (size_t)out_shape[0] * out_shape[1] * out_shape[2];
  /* gridsOut, stored +1 (0 == the reference's -1) */
  int32_t* grid = (int32_t*)calloc(vol * (size_t)batch, sizeof(int32_t));
  int* vp = (int*)malloc(sizeof(int) * 4 * (size_t)kvol);
  int num_act = 0;
  if (subm) {
    for (int j = 0; j < n; ++j) {
      const int32_t* r = indices + (size_t)j * 4;
      int pos[3] = {r[1], r[2], r[3]};
      grid[lin_id(pos, out_shape) + vol * (size_t)r[0]] = j + 1;
    }
    for (int j = 0; j < n; ++j) {
      const int32_t* r = indices + (size_t)j * 4;
      int pos[3] = {r[1], r[2], r[3]};
      int nv = valid_out_pos(pos, ksize, stride, pad, dil, out_shape, vp);
      for (int i = 0; i < nv; ++i) {
        int off = vp[i * 4 + 3];
        int32_t g = grid[lin_id(vp + i * 4, out_shape) + vol * (size_t)r[0]];
        if (g > 0) {
          int slot = indice_num[off]++;
          indice_pairs[((size_t)off * 2 + 0) * n + slot] = j;
          indice_pairs[((size_t)off * 2 + 1) * n + slot] = g - 1;
        }
      }
    }
    num_act = n;
  } else {
    for (int j = 0; j < n; ++j) {
      const int32_t* r = indices + (size_t)j * 4;
      int pos[3] = {r[1], r[2], r[3]};
      int nv = valid_out_pos(pos, ksize, stride, pad, dil, out_shape, vp);
      for (int i = 0; i < nv; ++i) {
        int off = vp[i * 4 + 3];
        size_t cell = lin_id(vp + i * 4, out_shape) + vol * (size_t)r[0];
        if (grid[cell] == 0) {
          int32_t* o = out_indices + (size_t)num_act * 4;
          o[0] = r[0];
          o[1] = vp[i * 4 + 0];
          o[2] = vp[i * 4 + 1];
          o[3] = vp[i * 4 + 2];
          grid[cell] = ++num_act;
        }
        int slot = indice_num[off]++;
        indice_pairs[((size_t)off * 2 + 0) * n + slot] = j;
        indice_pairs[((size_t)off * 2 + 1) * n + slot] = grid[cell] - 1;
      }
    }
  }
  free(vp);
  free(grid);
  return num_act;
}

/* ------------------------------------------------------------------------- *
 * Native sparse convolution: gather -> dense [nHot,Cin]x[Cin,Cout] -> scatter
 * add, one kernel offset at a time.
 * Follows indiceConv, spconv/spconv_ops.h:260-361: SubM runs the offset with
 * the most pairs (the centre) as one dense product over all rows (:300-303)
 * and skips it in the loop (:310); gather/scatter-add as in
 * spconv/src/reordering.cc:20-50.  filters is [kvol,Cin,Cout]
 * (mmdet3d/ops/spconv/conv.py:98-99 layout flattened, spconv_ops.h:299).
 * The dense product is a k-ordered float accumulation (what a plain SGEMM
 * reference loop does); tests compare with a 1e-4 tolerance.
 * ------------------------------------------------------------------------- */
static void mm_acc(const float* a, const float* b, float* c, int m, int k,
                   int n) { /* c[m,n] += a[m,k] b[k,n] */
#pragma omp parallel for schedule(static)
  for (int i = 0; i < m; ++i)
    for (int p = 0; p < k; ++p) {
      float av = a[(size_t)i * k + p];
      const float* br = b + (size_t)p * n;
      float* cr = c + (size_t)i * n;
      for (int j = 0; j < n; ++j) cr[j] += av * br[j];
    }
}

static int max_offset(const int32_t* num, int kvol) {
  int best = 0;
  for (int i = 1; i < kvol; ++i)
    if (num[i] > num[best]) best = i; /* std::max_element: first maximum */
  return best;
}

ORC_API void orc_indice_conv_fwd(const float* feat, int n_in, int cin,
                                 const float* filters, int kvol, int cout,
                                 const int32_t* pairs, const int32_t* num,
                                 int ld, int n_out, int inverse, int subm,
                                 float* out /* [n_out,cout], zeroed here */) {
  (void)n_in;
  memset(out, 0, sizeof(float) * (size_t)n_out * cout);
  int centre = max_offset(num, kvol);
  int hot_max = num[centre];
  float* ibuf = (float*)malloc(sizeof(float) * (size_t)(hot_max + 1) * cin);
  float* obuf = (float*)malloc(sizeof(float) * (size_t)(hot_max + 1) * cout);
  if (subm)
    mm_acc(feat, filters + (size_t)centre * cin * cout, out, n_out, cin, cout);
  for (int k = 0; k < kvol; ++k) {
    int hot = num[k];
    if (hot <= 0 || (subm && k == centre)) continue;
    const int32_t* gi = pairs + ((size_t)k * 2 + (inverse ? 1 : 0)) * ld;
    const int32_t* si = pairs + ((size_t)k * 2 + (inverse ? 0 : 1)) * ld;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < hot; ++i)
      memcpy(ibuf + (size_t)i * cin, feat + (size_t)gi[i] * cin,
             sizeof(float) * cin);
    memset(obuf, 0, sizeof(float) * (size_t)hot * cout);
    mm_acc(ibuf, filters + (size_t)k * cin * cout, obuf, hot, cin, cout);
    /* within one offset every output row appears once (reordering.cu.h:99) */
#pragma omp parallel for schedule(static)
    for (int i = 0; i < hot; ++i) {
      float* o = out + (size_t)si[i] * cout;
      const float* b = obuf + (size_t)i * cout;
      for (int j = 0; j < cout; ++j) o[j] += b[j];
    }
  }
  free(ibuf);
  free(obuf);
}

/* indiceConvBackward, spconv/spconv_ops.h:363-456:
 *   dW[k] = gather(in)^T . gather(dout)        (:438)
 *   din  += scatter(gather(dout) . W[k]^T)     (:439-452)
 * with the SubM centre offset done densely (:396-401). */
ORC_API void orc_indice_conv_bwd(const float* feat, int n_in, int cin,
                                 const float* filters, int kvol, int cout,
                                 const float* dout, const int32_t* pairs,
                                 const int32_t* num, int ld, int n_out,
                                 int inverse, int subm, float* din,
                                 float* dfilters) {
  memset(din, 0, sizeof(float) * (size_t)n_in * cin);
  memset(dfilters, 0, sizeof(float) * (size_t)kvol * cin * cout);
  int centre = max_offset(num, kvol);
  for (int k = 0; k < kvol; ++k) {
    int dense = subm && k == centre;
    int hot = dense ? n_out : num[k];
    if (hot <= 0) continue;
    const int32_t* gi = pairs + ((size_t)k * 2 + (inverse ? 1 : 0)) * ld;
    const int32_t* go = pairs + ((size_t)k * 2 + (inverse ? 0 : 1)) * ld;
    const float* w = filters + (size_t)k * cin * cout;
    float* dw = dfilters + (size_t)k * cin * cout;
#pragma omp parallel
    {
      /* per-thread dW partial, summed in thread order below */
      float* part = (float*)calloc((size_t)cin * cout, sizeof(float));
#pragma omp for schedule(static)
      for (int p = 0; p < hot; ++p) {
        size_t ri = dense ? (size_t)p : (size_t)gi[p];
        size_t ro = dense ? (size_t)p : (size_t)go[p];
        const float* x = feat + ri * cin;
        const float* g = dout + ro * cout;
        float* dx = din + ri * cin; /* an input row appears once per offset */
        for (int a = 0; a < cin; ++a) {
          float xa = x[a], acc = 0.f;
          const float* wr = w + (size_t)a * cout;
          float* dwr = part + (size_t)a * cout;
          for (int b = 0; b < cout; ++b) {
            dwr[b] += xa * g[b];
            acc += g[b] * wr[b];
          }
          dx[a] += acc;
        }
      }
#pragma omp critical
      for (size_t e = 0; e < (size_t)cin * cout; ++e) dw[e] += part[e];
      free(part);
    }
  }
}

/* ------------------------------------------------------------------------- *
 * SparseConvTensor.dense(): mmdet3d/ops/spconv/structure.py:5-18,55-64 --
 * zero [B,D,H,W,C], scatter rows, permute to [B,C,D,H,W] contiguous.
 * ------------------------------------------------------------------------- */
ORC_API void orc_dense(const float* feat, const int32_t* indices, int n, int c,
                       int batch, const int* shape, float* out) {
  size_t vol = (size_t)shape[0] * shape[1] * shape[2];
  memset(out, 0, sizeof(float) * vol * (size_t)batch * c);
  for (int i = 0; i < n; ++i) {
    const int32_t* r = indices + (size_t)i * 4;
    int pos[3] = {r[1], r[2], r[3]};
    size_t cell = lin_id(pos, shape);
    for (int ch = 0; ch < c; ++ch)
      out[((size_t)r[0] * c + ch) * vol + cell] = feat[(size_t)i * c + ch];
  }
}

/* ------------------------------------------------------------------------- *
 * sparse_add(a, b): call site
 * mmdet3d/models/middle_encoders/sparse_multimodal_encoder_painting.py:455.
 * spconv 2.x is not in the reference tree (README.md:19-20 pins v2.1.21):
 * its published behaviour -- sparse COO add + coalesce -- is restated: the
 * union of the coordinate sets in ascending linear (b,z,y,x) id, features
 * summed where both operands hold the voxel.  PARITY UNPINNED against
 * spconv itself; pinned by hand-derived cases and a dense-add cross-check.
 * Returns the number of output rows; out_* sized n_a + n_b.
 * ------------------------------------------------------------------------- */
typedef struct {
  uint64_t key;
  int32_t row, src;
} orc_kv;
static int kv_cmp(const void* x, const void* y) {
  const orc_kv *a = (const orc_kv*)x, *b = (const orc_kv*)y;
  if (a->key != b->key) return a->key < b->key ? -1 : 1;
  if (a->src != b->src) return a->src - b->src;
  return a->row - b->row;
}
static uint64_t row_key(const int32_t* r, const int* shape) {
  int pos[3] = {r[1], r[2], r[3]};
  return (uint64_t)r[0] * shape[0] * shape[1] * shape[2] + lin_id(pos, shape);
}

ORC_API int orc_sparse_add(const float* fa, const int32_t* ia, int na,
                           const float* fb, const int32_t* ib, int nb, int c,
                           const int* shape, int32_t* out_idx, float* out_feat,
                           int32_t* map_a, int32_t* map_b) {
  int n = na + nb, m = 0;
  orc_kv* kv = (orc_kv*)malloc(sizeof(orc_kv) * (size_t)(n + 1));
  for (int i = 0; i < na; ++i)
    kv[i] = (orc_kv){row_key(ia + (size_t)i * 4, shape), i, 0};
  for (int i = 0; i < nb; ++i)
    kv[na + i] = (orc_kv){row_key(ib + (size_t)i * 4, shape), i, 1};
  qsort(kv, (size_t)n, sizeof(orc_kv), kv_cmp);
  for (int i = 0; i < n; ++i) {
    if (i == 0 || kv[i].key != kv[i - 1].key) {
      const int32_t* r = kv[i].src ? ib + (size_t)kv[i].row * 4
                                   : ia + (size_t)kv[i].row * 4;
      memcpy(out_idx + (size_t)m * 4, r, sizeof(int32_t) * 4);
      memset(out_feat + (size_t)m * c, 0, sizeof(float) * c);
      ++m;
    }
    const float* f = kv[i].src ? fb + (size_t)kv[i].row * c
                               : fa + (size_t)kv[i].row * c;
    float* o = out_feat + (size_t)(m - 1) * c;
    for (int j = 0; j < c; ++j) o[j] += f[j];
    if (kv[i].src) map_b[kv[i].row] = m - 1; else map_a[kv[i].row] = m - 1;
  }
  free(kv);
  return m;
}

/* ------------------------------------------------------------------------- *
 * voxel_modality_split + type_assign:
 * mmdet3d/models/detectors/MSMDFusion.py:27-45 (two-pointer walk over the two
 * sorted key lists; equal keys mark both and advance both) and :251-325.
 * float_keys != 0 reproduces the reference's float32 key z*1e6 + y*1e3 + x
 * (:271-272; int32 * python float promotes to float32, each step rounded)
 * with a stable sort; float_keys == 0 uses the exact linear id (the fixed
 * behaviour the HIP path implements).  One sample (the
 * caller loops over the batch as :262-314 does).  pair_* receive the matched
 * rows in key order; returns their count.
 * ------------------------------------------------------------------------- */
typedef struct {
  double key;
  int32_t row;
} orc_fk;
static int fk_cmp(const void* x, const void* y) {
  const orc_fk *a = (const orc_fk*)x, *b = (const orc_fk*)y;
  if (a->key != b->key) return a->key < b->key ? -1 : 1;
  return a->row - b->row; /* stable */
}
static double split_key(const int32_t* zyx, const int* shape, int float_keys) {
  if (float_keys) {
    float k = (float)zyx[0] * 1e6f;
    k = k + (float)zyx[1] * 1e3f;
    k = k + (float)zyx[2];
    return (double)k;
  }
  int pos[3] = {zyx[0], zyx[1], zyx[2]};
  return (double)lin_id(pos, shape);
}

ORC_API int orc_modality_split(const int32_t* zyx3, int n3, const int32_t* zyx2,
                               int n2, const int* shape, int float_keys,
                               int32_t* mix3, int32_t* mix2, int32_t* pair3,
                               int32_t* pair2) {
  orc_fk* a = (orc_fk*)malloc(sizeof(orc_fk) * (size_t)(n3 + 1));
  orc_fk* b = (orc_fk*)malloc(sizeof(orc_fk) * (size_t)(n2 + 1));
  for (int i = 0; i < n3; ++i)
    a[i] = (orc_fk){split_key(zyx3 + (size_t)i * 3, shape, float_keys), i};
  for (int i = 0; i < n2; ++i)
    b[i] = (orc_fk){split_key(zyx2 + (size_t)i * 3, shape, float_keys), i};
  qsort(a, (size_t)n3, sizeof(orc_fk), fk_cmp);
  qsort(b, (size_t)n2, sizeof(orc_fk), fk_cmp);
  memset(mix3, 0, sizeof(int32_t) * (size_t)n3);
  memset(mix2, 0, sizeof(int32_t) * (size_t)n2);
  int ii = 0, jj = 0, m = 0;
  while (ii < n3 && jj < n2) {
    if (a[ii].key < b[jj].key) {
      ++ii;
    } else if (a[ii].key == b[jj].key) {
      mix3[a[ii].row] = 1;
      mix2[b[jj].row] = 1;
      pair3[m] = a[ii].row;
      pair2[m] = b[jj].row;
      ++m; ++ii; ++jj;
    } else {
      ++jj;
    }
  }
  free(a);
  free(b);
  return m;
}

/* ------------------------------------------------------------------------- *
 * Furthest point sampling, one batch element.  Emulates, thread by thread,
 * furthest_point_sampling_kernel
 * (mmdet3d/ops/furthest_point_sample/src/furthest_point_sample_cuda.cu:25-141):
 * block size from opt_n_threads (:11-15), each thread's strided scan with the
 * strictly-greater update (:55-71), the shared-memory tree whose __update
 * keeps the lower slot on ties (:17-23,76-137).  First sample is index 0.
 * ------------------------------------------------------------------------- */
ORC_API int orc_fps_block_size(int n) {
  int pow_2 = (int)(log((double)n) / log(2.0));
  int t = 1 << pow_2;
  if (t > 1024) t = 1024;
  return t < 1 ? 1 : t;
}

ORC_API void orc_fps(const float* xyz, int n, int m, float* temp /* [n] */,
                     int32_t* idx /* [m] */) {
  if (m <= 0) return;
  int bs = orc_fps_block_size(n);
  float* dv = (float*)malloc(sizeof(float) * (size_t)bs);
  int* di = (int*)malloc(sizeof(int) * (size_t)bs);
  for (int k = 0; k < n; ++k) temp[k] = 1e10f;
  int old = 0;
  idx[0] = 0;
  for (int j = 1; j < m; ++j) {
    float x1 = xyz[old * 3 + 0], y1 = xyz[old * 3 + 1], z1 = xyz[old * 3 + 2];
    for (int tid = 0; tid < bs; ++tid) {
      int besti = 0;
      float best = -1.f;
      for (int k = tid; k < n; k += bs) {
        float x2 = xyz[k * 3 + 0], y2 = xyz[k * 3 + 1], z2 = xyz[k * 3 + 2];
        float d = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) +
                  (z2 - z1) * (z2 - z1);
        float d2 = d < temp[k] ? d : temp[k];
        temp[k] = d2;
        besti = d2 > best ? k : besti;
        best = d2 > best ? d2 : best;
      }
      dv[tid] = best;
      di[tid] = besti;
    }
    for (int s = bs / 2; s >= 1; s /= 2)
      for (int tid = 0; tid < s; ++tid) {
        float v1 = dv[tid], v2 = dv[tid + s];
        int i1 = di[tid], i2 = di[tid + s];
        dv[tid] = v1 > v2 ? v1 : v2;
        di[tid] = v2 > v1 ? i2 : i1;
      }
    old = di[0];
    idx[j] = old;
  }
  free(dv);
  free(di);
}

/* ball_query_kernel, mmdet3d/ops/ball_query/src/ball_query_cuda.cu:11-54:
 * per centre scan all points in order; keep those with d2 == 0 or
 * min_r^2 <= d2 < max_r^2; the first hit pre-fills every slot. idx zeroed by
 * the caller (ball_query.py:36). */
ORC_API void orc_ball_query(const float* centers, const float* xyz, int n,
                            int m, float min_r, float max_r, int nsample,
                            int32_t* idx /* [m,nsample] */) {
  float max2 = max_r * max_r, min2 = min_r * min_r;
  for (int c = 0; c < m; ++c) {
    float cx = centers[c * 3], cy = centers[c * 3 + 1], cz = centers[c * 3 + 2];
    int32_t* o = idx + (size_t)c * nsample;
    int cnt = 0;
    for (int k = 0; k < n && cnt < nsample; ++k) {
      float x = xyz[k * 3], y = xyz[k * 3 + 1], z = xyz[k * 3 + 2];
      float d2 = (cx - x) * (cx - x) + (cy - y) * (cy - y) + (cz - z) * (cz - z);
      if (d2 == 0 || (d2 >= min2 && d2 < max2)) {
        if (cnt == 0)
          for (int l = 0; l < nsample; ++l) o[l] = k;
        o[cnt++] = k;
      }
    }
  }
}

/* Brute-force nearest key of fps_NN_fast,
 * sparse_multimodal_encoder_painting.py:289-293,303-306: dist = ||q - k||_2 in
 * float32, min over keys (first minimum), valid when dist < thresh. */
ORC_API void orc_nn_search(const int32_t* q, int nq, const int32_t* key, int nk,
                           float thresh, int32_t* out) {
  for (int i = 0; i < nq; ++i) {
    float best = INFINITY;
    int bi = -1;
    for (int k = 0; k < nk; ++k) {
      float dz = (float)q[i * 3] - (float)key[k * 3];
      float dy = (float)q[i * 3 + 1] - (float)key[k * 3 + 1];
      float dx = (float)q[i * 3 + 2] - (float)key[k * 3 + 2];
      float d = sqrtf(dz * dz + dy * dy + dx * dx);
      if (d < best) {
        best = d;
        bi = k;
      }
    }
    out[i] = (bi >= 0 && best < thresh) ? bi : -1;
  }
}

/* The scatter that ends fps_NN_fast (:311-321): for representatives in
 * order, for their ball members in order, query_nn[member] = rep's key when
 * the representative is valid; later writes overwrite earlier ones (the
 * order a sequential index_put_ leaves). */
ORC_API void orc_nn_assign(const int32_t* group_idx, const int32_t* rep_nn,
                           int m, int nsample, int nq, int32_t* query_nn) {
  for (int i = 0; i < nq; ++i) query_nn[i] = -1;
  for (int r = 0; r < m; ++r) {
    if (rep_nn[r] < 0) continue;
    for (int s = 0; s < nsample; ++s)
      query_nn[group_idx[(size_t)r * nsample + s]] = rep_nn[r];
  }
}

/* ------------------------------------------------------------------------
 * Rotated BEV overlap of two box sets -- the algorithm of
 * mmdet3d/ops/iou3d/src/iou3d_kernel.cu:36-239 (box_overlap and its helpers),
 * entry boxes_overlap_kernel :250-264 / boxes_overlap_bev_gpu iou3d.cpp:70-98,
 * used by BaseInstance3DBoxes.overlaps (core/bbox/structures/base_box3d.py:384-438).
 * The reference has NO CPU twin of this kernel (CUDA only), so this is a
 * restatement in float arithmetic, pinned by the known answers of the
 * reference's tests/test_utils/test_box3d.py:897-936 and by an independent fp64
 * polygon clip (tests/test_head_loss_cpu.py).  Not bit-comparable with the CUDA
 * build (nvcc contracts a*b+c, device cosf/atan2f differ in the last ulp).
 *
 * Boxes are (x1, y1, x2, y2, angle).  Quirks kept: corners are turned by
 * R(-angle) about the box centre (rot = (dx*c + dy*s, -dx*s + dy*c)); a corner
 * counts as inside the other box with a 1e-5 margin; edge crossings need both
 * strict "straddle" products > 0; the vertex list (up to 24 entries, duplicates
 * allowed) is bubble-sorted by atan2 about its mean and summed as a fan from
 * vertex 0.
 */
typedef struct { float x, y; } orc_pt;

static float orc_cross3(orc_pt a, orc_pt b, orc_pt o) {
  return (a.x - o.x) * (b.y - o.y) - (b.x - o.x) * (a.y - o.y);
}

static int orc_edge_hit(orc_pt p1, orc_pt p0, orc_pt q1, orc_pt q0, orc_pt* hit) {
  /* bounding rectangles of the two segments must touch */
  if (!(fminf(p0.x, p1.x) <= fmaxf(q0.x, q1.x) && fminf(q0.x, q1.x) <= fmaxf(p0.x, p1.x) &&
        fminf(p0.y, p1.y) <= fmaxf(q0.y, q1.y) && fminf(q0.y, q1.y) <= fmaxf(p0.y, p1.y)))
    return 0;
  float s1 = orc_cross3(q0, p1, p0), s2 = orc_cross3(p1, q1, p0);
  float s3 = orc_cross3(p0, q1, q0), s4 = orc_cross3(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
  float s5 = orc_cross3(q1, p1, p0);
  if (fabsf(s5 - s1) > 1e-8f) {
    hit->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    hit->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {                                  /* general line-line solve */
    float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    float d = a0 * b1 - a1 * b0;
    hit->x = (b0 * c1 - b1 * c0) / d;
    hit->y = (a1 * c0 - a0 * c1) / d;
  }
  return 1;
}

static int orc_inside(const float* box, orc_pt p) {
  const float margin = 1e-5f;
  float cx = (box[0] + box[2]) / 2, cy = (box[1] + box[3]) / 2;
  float c = cosf(-box[4]), s = sinf(-box[4]);
  float rx = (p.x - cx) * c + (p.y - cy) * s + cx;
  float ry = -(p.x - cx) * s + (p.y - cy) * c + cy;
  return rx > box[0] - margin && rx < box[2] + margin && ry > box[1] - margin &&
         ry < box[3] + margin;
}

static void orc_corners(const float* box, orc_pt* out /* 5, closed */) {
  float cx = (box[0] + box[2]) / 2, cy = (box[1] + box[3]) / 2;
  float c = cosf(box[4]), s = sinf(box[4]);
  const float xs[4] = {box[0], box[2], box[2], box[0]};
  const float ys[4] = {box[1], box[1], box[3], box[3]};
  for (int k = 0; k < 4; ++k) {
    float dx = xs[k] - cx, dy = ys[k] - cy;
    out[k].x = dx * c + dy * s + cx;
    out[k].y = -dx * s + dy * c + cy;
  }
  out[4] = out[0];
}

static float orc_box_overlap(const float* a, const float* b) {
  orc_pt ca[5], cb[5], v[24], mean = {0.f, 0.f};
  orc_corners(a, ca);
  orc_corners(b, cb);
  int n = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      if (orc_edge_hit(ca[i + 1], ca[i], cb[j + 1], cb[j], &v[n])) {
        mean.x += v[n].x;
        mean.y += v[n].y;
        ++n;
      }
  for (int k = 0; k < 4; ++k) {
    if (orc_inside(a, cb[k])) {
      mean.x += cb[k].x;
      mean.y += cb[k].y;
      v[n++] = cb[k];
    }
    if (orc_inside(b, ca[k])) {
      mean.x += ca[k].x;
      mean.y += ca[k].y;
      v[n++] = ca[k];
    }
  }
  mean.x /= n;                               /* n == 0: NaN, never read */
  mean.y /= n;
  for (int j = 0; j < n - 1; ++j)
    for (int i = 0; i < n - j - 1; ++i)
      if (atan2f(v[i].y - mean.y, v[i].x - mean.x) >
          atan2f(v[i + 1].y - mean.y, v[i + 1].x - mean.x)) {
        orc_pt t = v[i];
        v[i] = v[i + 1];
        v[i + 1] = t;
      }
  float area = 0.f;
  for (int k = 0; k < n - 1; ++k) {
    float ux = v[k].x - v[0].x, uy = v[k].y - v[0].y;
    float wx = v[k + 1].x - v[0].x, wy = v[k + 1].y - v[0].y;
    area += ux * wy - uy * wx;
  }
  return (float)(fabs((double)area) / 2.0);
}

ORC_API void orc_boxes_overlap_bev(const float* boxes_a, int na, const float* boxes_b, int nb,
                                   float* out /* [na, nb] */) {
  for (int i = 0; i < na; ++i)
    for (int j = 0; j < nb; ++j)
      out[(size_t)i * nb + j] = orc_box_overlap(boxes_a + 5 * i, boxes_b + 5 * j);
}
