"""numpy/ctypes binding of oracle/libmsmd_oracle.so (and, when present,
oracle/_ref/libmsmd_ref.so, the reference's own CPU code).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.
Each wrapper mirrors the reference call it stands for (cited in
oracle/msmd_oracle.c) and returns numpy arrays.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libmsmd_oracle.so")
_REF = os.path.join(_HERE, "_ref", "libmsmd_ref.so")


def build(with_ref=None):
    """Compile the C restatement (gcc) and, when the reference checkout is
    present (build container only), the reference's own sources."""
    subprocess.check_call(["make", "-s", "-C", _HERE])
    if with_ref is None:
        with_ref = os.path.isdir("/root/reference")
    if with_ref and not os.path.exists(_REF):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


def _load():
    if not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(
            os.path.join(_HERE, "msmd_oracle.c")):
        build(with_ref=False)
    return C.CDLL(_LIB)


_lib = _load()
_ref = None


def have_ref():
    return os.path.exists(_REF)


def ref_lib():
    global _ref
    if _ref is None:
        import torch  # noqa: F401  (libmsmd_ref.so links libtorch)
        _ref = C.CDLL(_REF)
    return _ref


def set_threads(n):
    """OpenMP threads of the conv loops; returns the count in effect."""
    fn = _lib.orc_set_threads
    fn.restype = C.c_int
    return fn(int(n))


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _f(a):
    return _p(a, C.c_float)


def _i(a):
    return _p(a, C.c_int32)


def _i3(v):
    return (C.c_int * 3)(*[int(x) for x in v])


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def expand3(v):
    return [int(v)] * 3 if np.isscalar(v) else [int(x) for x in v]


# --------------------------------------------------------------------------- voxelization
def grid_size(voxel_size, pc_range):
    g = (C.c_int * 3)()
    _lib.orc_grid_size(_f(_c(voxel_size, np.float32)), _f(_c(pc_range, np.float32)), g)
    return list(g)


def hard_voxelize(points, voxel_size, pc_range, max_points, max_voxels, use_ref=False):
    """-> (voxels[M,max_points,C], coors[M,3] zyx, num_points[M])"""
    pts = _c(points, np.float32)
    n, c = pts.shape
    voxels = np.zeros((max_voxels, max_points, c), np.float32)
    coors = np.zeros((max_voxels, 3), np.int32)
    npv = np.zeros((max_voxels,), np.int32)
    vs = _c(voxel_size, np.float32)
    rg = _c(pc_range, np.float32)
    fn = ref_lib().ref_hard_voxelize if use_ref else _lib.orc_hard_voxelize
    fn.restype = C.c_int
    m = fn(_f(pts), n, c, _f(vs), _f(rg), int(max_points), int(max_voxels),
           _f(voxels), _i(coors), _i(npv))
    return voxels[:m].copy(), coors[:m].copy(), npv[:m].copy()


def voxel_mean(voxels, num_points, out_features=None):
    v = _c(voxels, np.float32)
    m, mp, c = v.shape
    of = c if out_features is None else out_features
    out = np.zeros((m, of), np.float32)
    _lib.orc_voxel_mean(_f(v), _i(_c(num_points, np.int32)), m, mp, c, of, _f(out))
    return out


# --------------------------------------------------------------------------- rulebooks
def conv_output_size(in_shape, ksize, stride, padding, dilation):
    """mmdet3d/ops/spconv/ops.py:20-30."""
    return [(in_shape[i] + 2 * padding[i] - dilation[i] * (ksize[i] - 1) - 1) // stride[i] + 1
            for i in range(3)]


def get_indice_pairs(indices, batch_size, spatial_shape, ksize, stride=1, padding=0,
                     dilation=1, subm=False, use_ref=False):
    """-> (out_indices[Nout,4], indice_pairs[K,2,N], indice_num[K], out_shape)
    in the CPU reference's raw (first-touch) order."""
    idx = _c(indices, np.int32)
    n = idx.shape[0]
    ks, st, pd, dl = expand3(ksize), expand3(stride), expand3(padding), expand3(dilation)
    out_shape = list(spatial_shape) if subm else conv_output_size(spatial_shape, ks, st, pd, dl)
    kvol = ks[0] * ks[1] * ks[2]
    pairs = np.full((kvol, 2, max(n, 1)), -1, np.int32)
    num = np.zeros((kvol,), np.int32)
    outs = np.zeros((max(n * kvol, 1), 4), np.int32)
    if use_ref:
        vol = int(np.prod(out_shape))
        grid = np.full((vol * batch_size,), -1, np.int32)
        fn = ref_lib().ref_get_indice_pairs
        fn.restype = C.c_int
        m = fn(_i(idx), n, batch_size, _i3(out_shape), _i3(ks), _i3(st), _i3(pd), _i3(dl),
               int(subm), _i(outs), _i(pairs), _i(num), _i(grid))
    else:
        fn = _lib.orc_get_indice_pairs
        fn.restype = C.c_int
        m = fn(_i(idx), n, batch_size, _i3(out_shape), _i3(ks), _i3(st), _i3(pd), _i3(dl),
               int(subm), _i(outs), _i(pairs), _i(num))
    out_indices = idx.copy() if subm else outs[:m].copy()
    return out_indices, pairs[:, :, :n] if n else pairs[:, :, :0], num, out_shape


def linear_ids(indices, spatial_shape):
    idx = np.asarray(indices, np.int64)
    d, h, w = [int(x) for x in spatial_shape]
    return ((idx[:, 0] * d + idx[:, 1]) * h + idx[:, 2]) * w + idx[:, 3]


def canonical_rulebook(out_indices, pairs, num, out_shape, keep_rows=False):
    """SURVEY Appendix B.2 canonical form: output rows sorted by linear id
    (keep_rows=True for SubM, whose output rows ARE the input rows, in input
    order); pairs of each offset sorted by (out, in).  Returns
    (out_indices_sorted, [array[P_k,2] (in,out) per offset], perm) where
    perm[new_row] = old_row."""
    lid = linear_ids(out_indices, out_shape)
    perm = np.arange(lid.size) if keep_rows else np.argsort(lid, kind="stable")
    inv = np.empty_like(perm)
    inv[perm] = np.arange(perm.size)
    per_offset = []
    for k in range(pairs.shape[0]):
        p = int(num[k])
        i = pairs[k, 0, :p].astype(np.int64)
        o = inv[pairs[k, 1, :p].astype(np.int64)] if p else np.zeros((0,), np.int64)
        order = np.lexsort((i, o))
        per_offset.append(np.stack([i[order], o[order]], 1).astype(np.int32))
    return np.asarray(out_indices)[perm], per_offset, perm


def nbr_table_from_pairs(per_offset, n_out):
    """canonical pair lists -> output-stationary table nbr[K,n_out] (the HIP
    library's native format)."""
    k = len(per_offset)
    nbr = np.full((k, n_out), -1, np.int32)
    for kk, po in enumerate(per_offset):
        nbr[kk, po[:, 1]] = po[:, 0]
    return nbr


# --------------------------------------------------------------------------- conv arithmetic
def indice_conv_fwd(feat, filters, pairs, num, n_out, inverse=False, subm=False, use_ref=False):
    """filters [K,Cin,Cout]; pairs [K,2,ld] -> out [n_out,Cout]"""
    f = _c(feat, np.float32)
    w = _c(filters, np.float32)
    pr = _c(pairs, np.int32)
    nm = _c(num, np.int32)
    kvol, cin, cout = w.shape
    out = np.zeros((n_out, cout), np.float32)
    if use_ref:
        assert not inverse
        ref_lib().ref_indice_conv_fwd(_f(f), f.shape[0], cin, _f(w), kvol, cout, _i(pr), _i(nm),
                                      pr.shape[2], n_out, int(subm), _f(out))
    else:
        _lib.orc_indice_conv_fwd(_f(f), f.shape[0], cin, _f(w), kvol, cout, _i(pr), _i(nm),
                                 pr.shape[2], n_out, int(inverse), int(subm), _f(out))
    return out


def indice_conv_bwd(feat, filters, dout, pairs, num, inverse=False, subm=False, use_ref=False):
    """-> (din [n_in,Cin], dfilters [K,Cin,Cout])"""
    f = _c(feat, np.float32)
    w = _c(filters, np.float32)
    g = _c(dout, np.float32)
    pr = _c(pairs, np.int32)
    nm = _c(num, np.int32)
    kvol, cin, cout = w.shape
    din = np.zeros_like(f)
    dw = np.zeros_like(w)
    if use_ref:
        assert not inverse
        ref_lib().ref_indice_conv_bwd(_f(f), f.shape[0], cin, _f(w), kvol, cout, _f(g),
                                      g.shape[0], _i(pr), _i(nm), pr.shape[2], int(subm),
                                      _f(din), _f(dw))
        return din, dw
    _lib.orc_indice_conv_bwd(_f(f), f.shape[0], cin, _f(w), kvol, cout, _f(g), _i(pr), _i(nm),
                             pr.shape[2], g.shape[0], int(inverse), int(subm), _f(din), _f(dw))
    return din, dw


def dense(feat, indices, batch_size, spatial_shape):
    f = _c(feat, np.float32)
    idx = _c(indices, np.int32)
    n, c = f.shape
    d, h, w = [int(x) for x in spatial_shape]
    out = np.zeros((batch_size, c, d, h, w), np.float32)
    _lib.orc_dense(_f(f), _i(idx), n, c, batch_size, _i3(spatial_shape), _f(out))
    return out


def sparse_add(fa, ia, fb, ib, spatial_shape):
    """-> (out_indices, out_feat, map_a, map_b)"""
    fa, fb = _c(fa, np.float32), _c(fb, np.float32)
    ia, ib = _c(ia, np.int32), _c(ib, np.int32)
    na, nb, c = fa.shape[0], fb.shape[0], fa.shape[1]
    oi = np.zeros((na + nb, 4), np.int32)
    of = np.zeros((na + nb, c), np.float32)
    ma = np.zeros((na,), np.int32)
    mb = np.zeros((nb,), np.int32)
    fn = _lib.orc_sparse_add
    fn.restype = C.c_int
    m = fn(_f(fa), _i(ia), na, _f(fb), _i(ib), nb, c, _i3(spatial_shape), _i(oi), _f(of),
           _i(ma), _i(mb))
    return oi[:m].copy(), of[:m].copy(), ma, mb


def modality_split(zyx3, zyx2, spatial_shape, float_keys=False):
    """one sample -> (mix3[n3], mix2[n2], pair3[m], pair2[m])"""
    a, b = _c(zyx3, np.int32), _c(zyx2, np.int32)
    n3, n2 = a.shape[0], b.shape[0]
    m3, m2 = np.zeros((n3,), np.int32), np.zeros((n2,), np.int32)
    cap = max(min(n3, n2), 1)
    p3, p2 = np.zeros((cap,), np.int32), np.zeros((cap,), np.int32)
    fn = _lib.orc_modality_split
    fn.restype = C.c_int
    m = fn(_i(a), n3, _i(b), n2, _i3(spatial_shape), int(float_keys), _i(m3), _i(m2), _i(p3),
           _i(p2))
    return m3, m2, p3[:m].copy(), p2[:m].copy()


# --------------------------------------------------------------------------- GMA-Conv helpers
def fps_block_size(n):
    fn = _lib.orc_fps_block_size
    fn.restype = C.c_int
    return fn(int(n))


def furthest_point_sample(xyz, m):
    """xyz [B,N,3] -> idx [B,m] int32"""
    x = _c(xyz, np.float32)
    b, n, _ = x.shape
    out = np.zeros((b, m), np.int32)
    tmp = np.zeros((n,), np.float32)
    for i in range(b):
        _lib.orc_fps(_f(x[i]), n, m, _f(tmp), _i(out[i]))
    return out


def ball_query(min_radius, max_radius, nsample, xyz, center_xyz):
    """argument order of mmdet3d/ops/ball_query/ball_query.py:14 -> [B,M,nsample]"""
    x, cx = _c(xyz, np.float32), _c(center_xyz, np.float32)
    b, n, _ = x.shape
    m = cx.shape[1]
    out = np.zeros((b, m, nsample), np.int32)
    for i in range(b):
        _lib.orc_ball_query(_f(cx[i]), _f(x[i]), n, m, C.c_float(min_radius),
                            C.c_float(max_radius), int(nsample), _i(out[i]))
    return out


def nn_search(query_zyx, key_zyx, dist_thresh):
    q, k = _c(query_zyx, np.int32), _c(key_zyx, np.int32)
    out = np.zeros((q.shape[0],), np.int32)
    _lib.orc_nn_search(_i(q), q.shape[0], _i(k), k.shape[0], C.c_float(dist_thresh), _i(out))
    return out


def nn_assign(group_idx, rep_nn, nq):
    g, r = _c(group_idx, np.int32), _c(rep_nn, np.int32)
    out = np.zeros((nq,), np.int32)
    _lib.orc_nn_assign(_i(g), _i(r), g.shape[0], g.shape[1], nq, _i(out))
    return out
