"""Tensor-level wrappers over the C ABI (include/msmd_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every
computation is a call into libmsmd_hip.so on torch's current stream.  All
functions require CUDA (ROCm) tensors and raise otherwise -- no CPU path.
"""
import ctypes as C
import os

import torch

from ._lib import check, float_arr, int3, lib


def _raw_stream(device_index=None):
    """Handle of torch's current stream (an int).  torch.cuda.current_stream() builds a
    Python Stream object per call -- 8 us each, 85 calls per training step."""
    if device_index is None:
        device_index = torch._C._cuda_getDevice()     # (torch.cuda.current_device() minus its
    return torch._C._cuda_getCurrentRawStream(device_index)   # lazy-init check: 1 us a call)


def _stream():
    return C.c_void_p(_raw_stream())


def _p(t):
    # (the address as a Python int: ctypes converts it for a c_void_p parameter itself, and
    # building a c_void_p object per argument was a quarter of a 17-argument call's 3.5 us)
    return None if t is None else t.data_ptr()


def _need_cuda(*ts):
    """Every operand on the GPU, and on the CURRENT one: the C ABI launches on the
    current HIP device and on torch's current stream of that device, so a tensor
    living elsewhere (single-process multi-GPU, a forgotten set_device) would be
    touched from the wrong device and stream -- refuse instead."""
    cur = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("msmdfusion_amd kernels need CUDA/ROCm tensors "
                               "(there is no CPU fallback)")
        if cur is None:
            cur = torch._C._cuda_getDevice()
        if t.device.index != cur:
            raise RuntimeError("tensor on cuda:%d but the current device is cuda:%d: wrap the "
                               "call in torch.cuda.device(tensor.device)" % (t.device.index, cur))


def _need_bzyx(*index_tensors):
    """The index kernels read rows as one int4 (b,z,y,x): anything else (the
    5-column tensors voxel_modality_split leaves, MSMDFusion.py:322-323) would be
    read misaligned and past the end."""
    for t in index_tensors:
        if t.dim() != 2 or t.shape[1] != 4:
            raise ValueError("expected [N,4] (b,z,y,x) voxel indices, got %s" % (tuple(t.shape),))


_WS_CACHE = {}
_WS_CACHE_MAX = 1 << 30


def _ws(nbytes, device):
    """Scratch for one library call (or a count -> host read -> fill pair).  Grow-only
    buffer per (device, stream): every user is stream-ordered behind the previous one, a
    stream is driven by one thread at a time, and nothing returned to callers aliases the
    scratch -- so ~250 allocator calls per LC step (two threads contending for the caching
    allocator's lock) become dictionary look-ups."""
    nbytes = max(int(nbytes), 256)
    if nbytes > _WS_CACHE_MAX:
        return torch.empty(nbytes, dtype=torch.uint8, device=device)
    key = (device.index, _raw_stream(device.index))
    buf = _WS_CACHE.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = _WS_CACHE[key] = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8,
                                           device=device)
    return buf


class _Count:
    """Where a counting kernel puts its total and how the host gets it: a device int read
    with .item().  (Round 2 also tried pinned, device-mapped host slots the host spins on:
    134.2-134.3 against 134.4 samples/s -- what a read costs next to the feature pass is the
    counting kernels' turn on a busy GPU, not the copy; removed.)"""

    def __init__(self, device):
        self.tensor = torch.empty((1,), dtype=torch.int32, device=device)
        self.ptr = _p(self.tensor)

    def read(self):
        return int(self.tensor.item())


def _expand3(v):
    return [int(v)] * 3 if isinstance(v, int) else [int(x) for x in v]


def kernel_volume(ksize):
    k = _expand3(ksize)
    return k[0] * k[1] * k[2]


# ------------------------------------------------------------------ voxelization
VOXELIZE_MANY = os.environ.get("MSMD_VOXELIZE_MANY", "1") == "1"


def hard_voxelize(points, voxel_size, coors_range, max_points, max_voxels,
                  want_voxels=True, want_mean=False):
    """voxel_layer.hard_voxelize (mmdet3d/ops/voxel/src/voxelization.h:51-69).

    Returns (voxels[M,max_points,C] | None, coors[M,3], num_points[M],
    mean[M,C] | None).  One host read (the voxel count), like the reference.
    """
    _need_cuda(points)
    pts = points.contiguous().float()
    n, c = pts.shape
    dev = pts.device
    voxels = torch.empty((max_voxels, max_points, c), dtype=torch.float32, device=dev) \
        if want_voxels else None
    mean = torch.empty((max_voxels, c), dtype=torch.float32, device=dev) if want_mean else None
    coors = torch.empty((max_voxels, 3), dtype=torch.int32, device=dev)
    npv = torch.empty((max_voxels,), dtype=torch.int32, device=dev)
    count = torch.empty((1,), dtype=torch.int32, device=dev)
    nbytes = lib.msmd_voxelize_workspace_bytes(n, max_voxels, max_points)
    ws = _ws(nbytes, dev)
    check(lib.msmd_hard_voxelize(_p(pts), n, c, float_arr(voxel_size), float_arr(coors_range),
                                 int(max_points), int(max_voxels), _p(voxels), _p(coors), _p(npv),
                                 _p(mean), _p(count), _p(ws), nbytes, _stream()),
          "msmd_hard_voxelize")
    m = int(count.item())
    return (voxels[:m] if want_voxels else None, coors[:m], npv[:m],
            mean[:m] if want_mean else None)


class _VoxDesc(C.Structure):        # include/msmd_hip.h: msmd_voxelize_desc
    _fields_ = [("points", C.c_void_p), ("num_points", C.c_int32), ("num_features", C.c_int32),
                ("voxel_size", C.c_float * 3), ("coors_range", C.c_float * 6),
                ("max_points", C.c_int32), ("max_voxels", C.c_int32), ("voxels", C.c_void_p),
                ("coors", C.c_void_p), ("num_points_per_voxel", C.c_void_p),
                ("voxel_mean", C.c_void_p), ("voxel_num", C.c_void_p)]


def hard_voxelize_batch(points_list, voxel_size, coors_range, max_points, max_voxels,
                        want_voxels=True, want_mean=False):
    """hard_voxelize for every sample of a batch in ONE launch set and with ONE host read
    (msmd_hard_voxelize_many: the reference syncs 4 times per sample,
    voxelization_cuda.cu:231-323, and is called once per sample and scale).
    voxel_size: one size for all clouds, or one size per cloud (the LiDAR sweep
    and the four scales of virtual points of a step go out in a single call).
    MSMD_VOXELIZE_MANY=0: one msmd_hard_voxelize call per cloud (rounds 1-5)."""
    per_cloud = len(voxel_size) > 0 and isinstance(voxel_size[0], (list, tuple))
    if per_cloud and len(voxel_size) != len(points_list):
        raise ValueError("need one voxel size per cloud")
    if not points_list:
        return []
    many = VOXELIZE_MANY and len(points_list) > 1
    pending = []
    descs = (_VoxDesc * len(points_list))() if many else None
    counts_all = None
    for ci, points in enumerate(points_list):
        _need_cuda(points)
        pts = points.contiguous().float()
        n, c = pts.shape
        dev = pts.device
        if counts_all is None:
            counts_all = torch.empty((len(points_list),), dtype=torch.int32, device=dev)
        voxels = torch.empty((max_voxels, max_points, c), dtype=torch.float32, device=dev) \
            if want_voxels else None
        mean = torch.empty((max_voxels, c), dtype=torch.float32, device=dev) if want_mean else None
        coors = torch.empty((max_voxels, 3), dtype=torch.int32, device=dev)
        npv = torch.empty((max_voxels,), dtype=torch.int32, device=dev)
        count = counts_all[ci:ci + 1]
        vs = voxel_size[ci] if per_cloud else voxel_size
        if many:
            d = descs[ci]
            d.points, d.num_points, d.num_features = _p(pts), n, c
            d.voxel_size = (C.c_float * 3)(*[float(v) for v in vs])
            d.coors_range = (C.c_float * 6)(*[float(v) for v in coors_range])
            d.max_points, d.max_voxels = int(max_points), int(max_voxels)
            d.voxels, d.coors, d.num_points_per_voxel = _p(voxels), _p(coors), _p(npv)
            d.voxel_mean, d.voxel_num = _p(mean), _p(count)
        else:
            nbytes = lib.msmd_voxelize_workspace_bytes(n, max_voxels, max_points)
            ws = _ws(nbytes, dev)
            check(lib.msmd_hard_voxelize(_p(pts), n, c, float_arr(vs), float_arr(coors_range),
                                         int(max_points), int(max_voxels), _p(voxels), _p(coors),
                                         _p(npv), _p(mean), _p(count), _p(ws), nbytes, _stream()),
                  "msmd_hard_voxelize")
        pending.append((voxels, coors, npv, mean, pts))
    if many:
        nbytes = lib.msmd_hard_voxelize_many_workspace_bytes(descs, len(points_list))
        ws = _ws(nbytes, counts_all.device)
        check(lib.msmd_hard_voxelize_many(descs, len(points_list), _p(ws), nbytes, _stream()),
              "msmd_hard_voxelize_many")
    counts = counts_all.tolist()
    return [(v[:m] if want_voxels else None, c[:m], n[:m], mu[:m] if want_mean else None)
            for (v, c, n, mu, _), m in zip(pending, counts)]


def voxel_mean(voxels, num_points, out_features=None):
    _need_cuda(voxels, num_points)
    v = voxels.contiguous().float()
    m, mp, c = v.shape
    of = c if out_features is None else int(out_features)
    out = torch.empty((m, of), dtype=torch.float32, device=v.device)
    check(lib.msmd_voxel_mean(_p(v), _p(num_points.contiguous().int()), m, mp, c, of, _p(out),
                              _stream()), "msmd_voxel_mean")
    return out


# ------------------------------------------------------------------ rulebooks
def conv_output_size(in_shape, ksize, stride, padding, dilation=(1, 1, 1)):
    """mmdet3d/ops/spconv/ops.py:20-30."""
    return [(in_shape[i] + 2 * padding[i] - dilation[i] * (ksize[i] - 1) - 1) // stride[i] + 1
            for i in range(3)]


# SubM index: "hash" (2N-slot table, cost ~ voxels), "bitmap" (occupancy words of the grid,
# cost ~ grid cells + a third of the hash's per-voxel cost) or "auto" by size
_SUBM_INDEX = os.environ.get("MSMD_SUBM_INDEX", "auto")
_SUBM_BITMAP_WORDS_PER_VOXEL = float(os.environ.get("MSMD_SUBM_BITMAP_WORDS_PER_VOXEL", "100"))
_SUBM_BITMAP_MIN_VOXELS = int(os.environ.get("MSMD_SUBM_BITMAP_MIN_VOXELS", "60000"))


def subm_index_method(n, batch_size, spatial_shape, method=None):
    method = method or _SUBM_INDEX
    if method != "auto":
        return method
    # the bitmap is laid out in 4 x 8 x 8 bricks (csrc/rulebook.hip): the grid padded to
    # whole bricks is what it clears and counts, and the entry point takes fewer than 2^24
    # bricks -- a thin grid (D = 1..3 pads 4x along z) or a huge one takes the hash index
    d, h, w = (int(v) for v in spatial_shape)
    bricks = int(batch_size) * ((d + 3) // 4) * ((h + 7) // 8) * ((w + 7) // 8)
    if bricks >= 1 << 24:
        return "hash"
    words = bricks * 8          # 256 cells per brick
    # the bitmap's clear + count (~3 ps per 32-cell word) against what it saves per voxel
    # (~0.3 ns), and not below the size where either is launch-bound anyway
    return "bitmap" if n >= _SUBM_BITMAP_MIN_VOXELS and words <= _SUBM_BITMAP_WORDS_PER_VOXEL * n \
        else "hash"


def rulebook_subm(indices, batch_size, spatial_shape, ksize, method=None):
    """-> nbr[K,N] int32 (output-stationary neighbour table)."""
    _need_bzyx(indices)
    _need_cuda(indices)
    idx = indices.contiguous().int()
    n = idx.shape[0]
    ks = _expand3(ksize)
    nbr = torch.empty((kernel_volume(ks), n), dtype=torch.int32, device=idx.device)
    if subm_index_method(n, batch_size, spatial_shape, method) == "bitmap":
        nbytes = lib.msmd_rulebook_subm_bitmap_workspace_bytes(n, int(batch_size),
                                                               int3(spatial_shape))
        ws = _ws(nbytes, idx.device)
        check(lib.msmd_rulebook_subm3d_bitmap(_p(idx), n, int(batch_size), int3(spatial_shape),
                                              int3(ks), _p(nbr), _p(ws), nbytes, _stream()),
              "msmd_rulebook_subm3d_bitmap")
        return nbr
    nbytes = lib.msmd_rulebook_subm_workspace_bytes(n)
    ws = _ws(nbytes, idx.device)
    check(lib.msmd_rulebook_subm3d(_p(idx), n, int(batch_size), int3(spatial_shape), int3(ks),
                                   _p(nbr), _p(ws), nbytes, _stream()), "msmd_rulebook_subm3d")
    return nbr


def add_conv_chain(extras, batch_size, in_shape, geoms, need_bwd=True):
    """The fusion stack's stage chain with ONE host read: level l's strided conv (geoms[l] =
    (ksize, stride, padding)) runs over total_l = extras[0] for l = 0 and
    sparse_add(extras[l], out_{l-1}) after that (a = the extra set, b = the previous level's
    output set, as sparse_add_index(extras[l], out_prev)).  All union and output sets are
    counted on the device back to back (msmd_rulebook_add_conv_count_chain), read together,
    then filled level by level.  -> per level dict(total_indices, map_a, map_b (None at level
    0), out_indices, nbr_fwd, nbr_bwd, in_shape, out_shape): tensor for tensor what
    sparse_add_index + rulebook_conv give stage by stage."""
    levels = len(geoms)
    if len(extras) != levels or levels < 1:
        raise ValueError("add_conv_chain: one extra voxel set per level")
    idx = []
    for e in extras:
        _need_bzyx(e)
        _need_cuda(e)
        idx.append(e if (e.dtype == torch.int32 and e.is_contiguous()) else e.contiguous().int())
    dev = idx[0].device
    ks = [_expand3(g[0]) for g in geoms]
    st = [_expand3(g[1]) for g in geoms]
    pd = [_expand3(g[2]) for g in geoms]
    in_shapes, out_shapes, shape = [], [], list(in_shape)
    for l in range(levels):
        in_shapes.append(list(shape))
        shape = conv_output_size(shape, ks[l], st[l], pd[l])
        out_shapes.append(list(shape))
    flat = lambda rows: (C.c_int * (3 * levels))(*[int(v) for r in rows for v in r])  # noqa: E731
    f_in, f_out, f_ks, f_st, f_pd = flat(in_shapes), flat(out_shapes), flat(ks), flat(st), flat(pd)
    nbytes = lib.msmd_rulebook_add_conv_chain_workspace_bytes(int(batch_size), levels, f_in, f_out)
    # (not the per-stream scratch: the fills below run after other library calls may have
    # used that one)
    ws = torch.empty((max(int(nbytes), 256),), dtype=torch.uint8, device=dev)
    ptrs = (C.c_void_p * levels)(*[t.data_ptr() for t in idx])
    ns = (C.c_int * levels)(*[int(t.shape[0]) for t in idx])
    counts = torch.empty((2 * levels,), dtype=torch.int32, device=dev)
    check(lib.msmd_rulebook_add_conv_count_chain(ptrs, ns, int(batch_size), levels, f_in, f_out,
                                                 f_ks, f_st, f_pd, _p(counts), _p(ws), nbytes,
                                                 _stream()), "msmd_rulebook_add_conv_count_chain")
    ms = counts.tolist()                                    # the chain's one host read
    out, off, base, prev = [], 0, ws.data_ptr(), None
    for l in range(levels):
        ub = 0 if l == 0 else \
            (lib.msmd_rulebook_conv_workspace_bytes(int(batch_size), int3(in_shapes[l])) + 255) \
            // 256 * 256
        cb = (lib.msmd_rulebook_conv_workspace_bytes(int(batch_size), int3(out_shapes[l])) + 255) \
            // 256 * 256
        kvol = kernel_volume(ks[l])
        ma = mb = None
        if l == 0:
            total = idx[0]
        else:
            na, nb, m_u = idx[l].shape[0], prev.shape[0], int(ms[2 * l])
            total = torch.empty((m_u, 4), dtype=torch.int32, device=dev)
            ma = torch.empty((na,), dtype=torch.int32, device=dev)
            mb = torch.empty((nb,), dtype=torch.int32, device=dev)
            check(lib.msmd_sparse_add_fill(None, _p(idx[l]), na, None, _p(prev), nb, 0,
                                           int(batch_size), int3(in_shapes[l]), m_u, _p(total),
                                           None, _p(ma), _p(mb), base + off, ub, _stream()),
                  "msmd_sparse_add_fill")
        n, m = total.shape[0], int(ms[2 * l + 1])
        out_idx = torch.empty((m, 4), dtype=torch.int32, device=dev)
        nbr_fwd = torch.empty((kvol, m), dtype=torch.int32, device=dev)
        nbr_bwd = torch.empty((kvol, n), dtype=torch.int32, device=dev) if need_bwd else None
        check(lib.msmd_rulebook_conv3d_fill(_p(total), n, int(batch_size), int3(out_shapes[l]),
                                            int3(ks[l]), int3(st[l]), int3(pd[l]), m, _p(out_idx),
                                            _p(nbr_fwd), _p(nbr_bwd), base + off + ub, cb,
                                            _stream()), "msmd_rulebook_conv3d_fill")
        out.append(dict(total_indices=total, map_a=ma, map_b=mb, out_indices=out_idx,
                        nbr_fwd=nbr_fwd, nbr_bwd=nbr_bwd, in_shape=in_shapes[l],
                        out_shape=out_shapes[l]))
        prev = out_idx
        off += ub + cb
    return out


class _SubmDesc(C.Structure):      # include/msmd_hip.h: msmd_subm_desc
    _fields_ = [("indices", C.c_void_p), ("n", C.c_int32), ("batch_size", C.c_int32),
                ("spatial_shape", C.c_int32 * 3), ("ksize", C.c_int32 * 3),
                ("method", C.c_int32), ("reserved", C.c_int32), ("nbr", C.c_void_p)]


def subm_table(indices, ksize):
    """The empty [K, N] table rulebook_subm_many fills for a voxel set."""
    return torch.empty((kernel_volume(_expand3(ksize)), indices.shape[0]), dtype=torch.int32,
                       device=indices.device)


def rulebook_subm_many(jobs):
    """rulebook_subm for MANY voxel sets in one library call and one launch set
    (csrc/rulebook.hip: msmd_rulebook_subm3d_many).  jobs: dicts with indices [N,4] int32
    (b,z,y,x) contiguous, batch_size, spatial_shape, ksize, nbr (= subm_table(...), filled
    here) and optionally method.  The index method of each set is chosen as rulebook_subm
    chooses it; the tables are those of the single calls."""
    if not jobs:
        return
    descs = (_SubmDesc * len(jobs))()
    keep = []
    for d, j in zip(descs, jobs):
        idx = j["indices"]
        _need_bzyx(idx)
        _need_cuda(idx, j["nbr"])
        if idx.dtype != torch.int32 or not idx.is_contiguous():
            idx = idx.contiguous().int()
        keep.append(idx)
        n = idx.shape[0]
        ks = _expand3(j["ksize"])
        if tuple(j["nbr"].shape) != (kernel_volume(ks), n) or not j["nbr"].is_contiguous():
            raise ValueError("rulebook_subm_many: nbr must be the contiguous [K, N] table")
        d.indices, d.n, d.batch_size = idx.data_ptr(), n, int(j["batch_size"])
        d.spatial_shape[:] = [int(v) for v in j["spatial_shape"]]
        d.ksize[:] = ks
        d.method = int(subm_index_method(n, j["batch_size"], j["spatial_shape"],
                                         j.get("method")) == "bitmap")
        d.nbr = j["nbr"].data_ptr()
    dev = jobs[0]["nbr"].device
    nbytes = lib.msmd_rulebook_subm3d_many_workspace_bytes(descs, len(jobs))
    ws = _ws(nbytes, dev)
    check(lib.msmd_rulebook_subm3d_many(descs, len(jobs), _p(ws), nbytes, _stream()),
          "msmd_rulebook_subm3d_many")


def rulebook_conv(indices, batch_size, spatial_shape, ksize, stride, padding, need_bwd=True):
    """-> (out_indices[M,4], nbr_fwd[K,M], nbr_bwd[K,N] | None, out_shape)"""
    _need_bzyx(indices)
    _need_cuda(indices)
    idx = indices.contiguous().int()
    n = idx.shape[0]
    dev = idx.device
    ks, st, pd = _expand3(ksize), _expand3(stride), _expand3(padding)
    out_shape = conv_output_size(list(spatial_shape), ks, st, pd)
    kvol = kernel_volume(ks)
    nbytes = lib.msmd_rulebook_conv_workspace_bytes(int(batch_size), int3(out_shape))
    ws = _ws(nbytes, dev)
    count = _Count(dev)
    check(lib.msmd_rulebook_conv3d_count(_p(idx), n, int(batch_size), int3(out_shape), int3(ks),
                                         int3(st), int3(pd), count.ptr, _p(ws), nbytes, _stream()),
          "msmd_rulebook_conv3d_count")
    m = count.read()
    out_idx = torch.empty((m, 4), dtype=torch.int32, device=dev)
    nbr_fwd = torch.empty((kvol, m), dtype=torch.int32, device=dev)
    nbr_bwd = torch.empty((kvol, n), dtype=torch.int32, device=dev) if need_bwd else None
    check(lib.msmd_rulebook_conv3d_fill(_p(idx), n, int(batch_size), int3(out_shape), int3(ks),
                                        int3(st), int3(pd), m, _p(out_idx), _p(nbr_fwd),
                                        _p(nbr_bwd), _p(ws), nbytes, _stream()),
          "msmd_rulebook_conv3d_fill")
    return out_idx, nbr_fwd, nbr_bwd, out_shape


def rulebook_conv_chain(indices, batch_size, spatial_shape, geoms, need_bwd=True):
    """A chain of strided convs, each over the previous one's output set: geoms = [(ksize,
    stride, padding), ...].  All output sets are counted on the device back to back and the
    counts read ONCE.  -> [(out_indices, nbr_fwd, nbr_bwd | None, out_shape), ...], tensor for
    tensor what rulebook_conv gives level by level."""
    _need_bzyx(indices)
    _need_cuda(indices)
    idx = indices.contiguous().int()
    dev = idx.device
    levels = len(geoms)
    ks = [_expand3(g[0]) for g in geoms]
    st = [_expand3(g[1]) for g in geoms]
    pd = [_expand3(g[2]) for g in geoms]
    shapes, shape = [], list(spatial_shape)
    for l in range(levels):
        shape = conv_output_size(shape, ks[l], st[l], pd[l])
        shapes.append(list(shape))
    flat = lambda rows: (C.c_int * (3 * levels))(*[int(v) for r in rows for v in r])  # noqa: E731
    f_shapes, f_ks, f_st, f_pd = flat(shapes), flat(ks), flat(st), flat(pd)
    nbytes = lib.msmd_rulebook_conv_chain_workspace_bytes(int(batch_size), levels, f_shapes)
    ws = _ws(nbytes, dev)
    counts = torch.empty((levels,), dtype=torch.int32, device=dev)
    check(lib.msmd_rulebook_conv3d_count_chain(_p(idx), idx.shape[0], int(batch_size), levels,
                                               f_shapes, f_ks, f_st, f_pd, _p(counts), _p(ws),
                                               nbytes, _stream()),
          "msmd_rulebook_conv3d_count_chain")
    ms = counts.tolist()                                    # the chain's one host read
    out, off, base = [], 0, ws.data_ptr()
    for l in range(levels):
        n, m, kvol = idx.shape[0], int(ms[l]), kernel_volume(ks[l])
        lvl_bytes = lib.msmd_rulebook_conv_workspace_bytes(int(batch_size), int3(shapes[l]))
        out_idx = torch.empty((m, 4), dtype=torch.int32, device=dev)
        nbr_fwd = torch.empty((kvol, m), dtype=torch.int32, device=dev)
        nbr_bwd = torch.empty((kvol, n), dtype=torch.int32, device=dev) if need_bwd else None
        check(lib.msmd_rulebook_conv3d_fill(_p(idx), n, int(batch_size), int3(shapes[l]),
                                            int3(ks[l]), int3(st[l]), int3(pd[l]), m, _p(out_idx),
                                            _p(nbr_fwd), _p(nbr_bwd), C.c_void_p(base + off),
                                            lvl_bytes, _stream()), "msmd_rulebook_conv3d_fill")
        out.append((out_idx, nbr_fwd, nbr_bwd, shapes[l]))
        off += (lvl_bytes + 255) // 256 * 256
        idx = out_idx
    return out


def rulebook_pairs(nbr, ld=None):
    """nbr[K,M] -> (indice_pairs[K,2,ld], indice_num[K]): the reference's
    rulebook format (spconv_ops.h:55-59), pairs sorted by output row."""
    _need_cuda(nbr)
    kvol, m = nbr.shape
    ld = m if ld is None else int(ld)
    dev = nbr.device
    pairs = torch.empty((kvol, 2, ld), dtype=torch.int32, device=dev)
    num = torch.empty((kvol,), dtype=torch.int32, device=dev)
    nbytes = lib.msmd_rulebook_pairs_workspace_bytes(kvol, m)
    ws = _ws(nbytes, dev)
    check(lib.msmd_rulebook_pairs(_p(nbr.contiguous()), kvol, m, _p(pairs), ld, _p(num), _p(ws),
                                  nbytes, _stream()), "msmd_rulebook_pairs")
    return pairs, num


# ------------------------------------------------------------------ convolution
def pack_weight(weight, transpose=False, krsc=False):
    """weight -> MFMA-fragment order (see spconv.hip).  weight is [K,Cin,Cout]
    or, with krsc=True, the modules' KRSC parameter [Cout,kd,kh,kw,Cin]."""
    _need_cuda(weight)
    w = weight.contiguous().float()
    if krsc:
        cout, cin = w.shape[0], w.shape[-1]
        kvol = w.numel() // (cout * cin)
    else:
        kvol, cin, cout = w.shape
    packed = torch.empty((lib.msmd_spconv_packed_weight_elems(kvol, cin, cout),),
                         dtype=torch.float32, device=w.device)
    check(lib.msmd_spconv_pack_weight(_p(w), kvol, cin, cout, int(bool(transpose)) | (2 if krsc else 0),
                                      _p(packed), _stream()), "msmd_spconv_pack_weight")
    return packed


# bench.py sets this to a list to time individual launches with events recorded
# on the launch stream: entries are (kind, start_event, end_event, meta).
PROFILE = None


def _prof_begin():
    if PROFILE is None:
        return None
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    return ev


def _prof_end(kind, start, **meta):
    if start is not None:
        end = torch.cuda.Event(enable_timing=True)
        end.record()
        PROFILE.append((kind, start, end, meta))


# tools/split_bench.py --lc-b2 sets this to a list: every conv launch of a step is kept as
# (kind, meta, relaunch) -- relaunch() issues the same library call on the same operands
# again, alone on the chip -- to separate a layer's own time from what the step costs it.
CAPTURE = None


def _launch(kind, call, **meta):
    """One conv library call, bracketed by PROFILE's events and kept for CAPTURE."""
    ev = _prof_begin()
    call()
    _prof_end(kind, ev, **meta)
    if CAPTURE is not None:
        CAPTURE.append((kind, meta, call))


TILE_ROWS = 128     # rows per workgroup tile of the conv kernels


def row_mask_order(nbr, tile_lpt=True):
    """Tiling order of the rows of nbr[K,n]: rows sorted by (rank-permuted) neighbour
    mask so that the rows of a wave / a 128-row tile share their empty offsets, then
    -- tile_lpt -- whole tiles re-sequenced heaviest first for the persistent
    scheduler.  Masks and tile costs come from the C ABI; the sorts are torch's
    device radix sort.  It only chooses a tiling: results do not depend on it."""
    _need_cuda(nbr)
    kvol, n = nbr.shape
    if kvol > 64 or n == 0:
        return None
    keys = torch.empty((n,), dtype=torch.int64, device=nbr.device)
    check(lib.msmd_rulebook_row_masks(_p(nbr), kvol, n, None, _p(keys), _stream()),
          "msmd_rulebook_row_masks")
    if 2 * kvol <= 32 or (kvol <= 27):   # key fits 32 bits: half the radix passes
        keys = (keys - 2147483648).int()   # unsigned order under a signed compare
    order = torch.sort(keys, stable=False)[1].int()
    full = n // TILE_ROWS
    if tile_lpt and full > 1:
        cost = torch.empty(((n + TILE_ROWS - 1) // TILE_ROWS,), dtype=torch.int32,
                           device=nbr.device)
        check(lib.msmd_rulebook_tile_costs(_p(nbr), kvol, n, _p(order), TILE_ROWS, _p(cost),
                                           _stream()), "msmd_rulebook_tile_costs")
        seq = torch.sort(cost[:full], stable=True)[1]      # the partial last tile stays last
        head = order[:full * TILE_ROWS].view(full, TILE_ROWS).index_select(0, seq).reshape(-1)
        order = torch.cat([head, order[full * TILE_ROWS:]]) if n % TILE_ROWS else head
    return order


def rulebook_tiling(nbr, want_table=True):
    """(order[n], table in tile order | None) of nbr[K,n] in one library call: what
    row_mask_order + permute_cols compute through torch, with the sorts done in the
    library (K <= 31; falls back to those two otherwise)."""
    _need_cuda(nbr)
    kvol, n = nbr.shape
    if n == 0:
        return None, nbr
    if kvol > 31:
        order = row_mask_order(nbr)
        return order, (permute_cols(nbr, order) if want_table and order is not None else nbr)
    t = nbr.contiguous()
    order = torch.empty((n,), dtype=torch.int32, device=t.device)
    tiled = torch.empty_like(t) if want_table else None
    nbytes = lib.msmd_rulebook_tiling_workspace_bytes(n, TILE_ROWS)
    ws = _ws(nbytes, t.device)
    check(lib.msmd_rulebook_tiling(_p(t), kvol, n, TILE_ROWS, _p(order), _p(tiled), _p(ws), nbytes,
                                   _stream()), "msmd_rulebook_tiling")
    return order, tiled


def rulebook_plan(nbr, tile_rows=(), want_pairs=False, ld=None):
    """rulebook_tiling + tile_prefix (for each height in tile_rows, 128 / 256) +
    rulebook_pairs of one table in ONE library call.
    -> dict(order, tiled, prefix={rows: tensor}, pairs=(pairs, num) | None)"""
    _need_cuda(nbr)
    kvol, n = nbr.shape
    t = nbr.contiguous()
    dev = t.device
    order = torch.empty((n,), dtype=torch.int32, device=dev)
    tiled = torch.empty_like(t)
    pre = {r: torch.empty(((n + r - 1) // r + 1,), dtype=torch.int32, device=dev)
           for r in set(tile_rows)}
    if any(r not in (128, 256) for r in pre):
        raise ValueError("tile heights are 128 or 256 rows")
    pairs = num = None
    ld = n if ld is None else int(ld)
    if want_pairs:
        pairs = torch.empty((kvol, 2, ld), dtype=torch.int32, device=dev)
        num = torch.empty((kvol,), dtype=torch.int32, device=dev)
    nbytes = lib.msmd_rulebook_plan_workspace_bytes(kvol, n, TILE_ROWS)
    ws = _ws(nbytes, dev)
    check(lib.msmd_rulebook_plan(_p(t), kvol, n, TILE_ROWS, _p(order), _p(tiled), _p(pre.get(128)),
                                 _p(pre.get(256)), _p(pairs), ld, _p(num), _p(ws), nbytes,
                                 _stream()), "msmd_rulebook_plan")
    return dict(order=order, tiled=tiled, prefix=pre,
                pairs=(pairs, num) if want_pairs else None)


class _PlanDesc(C.Structure):      # include/msmd_hip.h: msmd_plan_desc (80 bytes)
    _fields_ = [("nbr", C.c_void_p), ("kvol", C.c_int32), ("n_rows", C.c_int32),
                ("order", C.c_void_p), ("tiled", C.c_void_p), ("prefix128", C.c_void_p),
                ("prefix256", C.c_void_p), ("indice_pairs", C.c_void_p),
                ("indice_num", C.c_void_p), ("segtab", C.c_void_p), ("ld", C.c_int32),
                ("reserved", C.c_int32)]


def rulebook_plan_many(jobs):
    """rulebook_plan (+ the one-chunk pair_segments table) of MANY tables in one library call
    and one launch set (csrc/plan_many.hip).  jobs: dicts with nbr [K,n] and optionally
    tile_rows (heights, 128 / 256; implies the tile-ordered table), want_table, want_pairs,
    want_segments (implies pairs), ld.  -> one dict per job:
    order, tiled | None, prefix {rows: tensor}, pairs (pairs, num) | None,
    segments (table, 1) | None -- the values the single calls give.
    All outputs of a call live in ONE int32 allocation (views; 256-byte aligned pieces)."""
    if not jobs:
        return []
    dev = jobs[0]["nbr"].device
    _need_cuda(*[j["nbr"] for j in jobs])
    descs = (_PlanDesc * len(jobs))()
    lay, total = [], 0

    def take(n):            # offset (int32 units) of an n-int piece
        nonlocal total
        o = total
        total += (n + 63) & ~63
        return o
    keep = []
    for d, j in zip(descs, jobs):
        t = j["nbr"]
        if not t.is_contiguous():
            t = t.contiguous()
        keep.append(t)
        kvol, n = t.shape
        rows = set(j.get("tile_rows") or ())
        if any(r not in (128, 256) for r in rows):
            raise ValueError("tile heights are 128 or 256 rows")
        want_seg = bool(j.get("want_segments"))
        want_pairs = bool(j.get("want_pairs")) or want_seg
        want_table = bool(j.get("want_table")) or bool(rows)
        ld = n if j.get("ld") is None else int(j["ld"])
        if want_seg and (WGRAD_CHUNK_ROWS > 0 or ld <= 0):
            want_seg = False            # chunked tables: pair_segments() after the call
        o = dict(kvol=kvol, n=n, ld=ld, order=take(n), tiled=take(kvol * n) if want_table else None,
                 prefix={r: take((n + r - 1) // r + 1) for r in sorted(rows)},
                 pairs=take(kvol * 2 * ld) if want_pairs else None,
                 num=take(kvol) if want_pairs else None,
                 seg=take(int(lib.msmd_rulebook_pair_segments_ints(kvol, 1))) if want_seg else None)
        lay.append(o)
        d.nbr, d.kvol, d.n_rows, d.ld = t.data_ptr(), kvol, n, ld
    slab = torch.empty((max(total, 1),), dtype=torch.int32, device=dev)
    base = slab.data_ptr()
    at = lambda off: None if off is None else base + 4 * off
    for d, o in zip(descs, lay):
        d.order, d.tiled = at(o["order"]), at(o["tiled"])
        d.prefix128, d.prefix256 = at(o["prefix"].get(128)), at(o["prefix"].get(256))
        d.indice_pairs, d.indice_num, d.segtab = at(o["pairs"]), at(o["num"]), at(o["seg"])
    nbytes = lib.msmd_rulebook_plan_many_workspace_bytes(descs, len(jobs))
    ws = _ws(nbytes, dev)
    check(lib.msmd_rulebook_plan_many(descs, len(jobs), _p(ws), nbytes, _stream()),
          "msmd_rulebook_plan_many")
    def view(off, *shape):
        size = 1
        for v in shape:
            size *= v
        return slab[off:off + size].view(*shape)
    out = []
    for o in lay:
        kvol, n, ld = o["kvol"], o["n"], o["ld"]
        pairs = None
        if o["pairs"] is not None:
            pairs = (view(o["pairs"], kvol, 2, ld), view(o["num"], kvol))
        seg = None
        if o["seg"] is not None:
            seg = (slab[o["seg"]:o["seg"] + 3 * kvol + 1], 1)
        out.append(dict(order=view(o["order"], n),
                        tiled=None if o["tiled"] is None else view(o["tiled"], kvol, n),
                        prefix={r: slab[off:off + (n + r - 1) // r + 1]
                                for r, off in o["prefix"].items()},
                        pairs=pairs, segments=seg))
    return out


_TILE_COUNTERS = {}


_SYNC_INTS = 1 + 8192       # tile counter + one stream-K exchange flag per workgroup ticket


def _tile_counter(device):
    """Zeroed int32s per (device, stream): [0] is the counter the persistent conv
    kernels draw tiles from, the rest are the split kernels' exchange flags; every
    launch leaves all of them at 0 again (see msmd_spconv_fwd_f32 / _fwd_split)."""
    key = (device, _raw_stream(device.index))
    c = _TILE_COUNTERS.get(key)
    if c is None:
        c = _TILE_COUNTERS[key] = torch.zeros((_SYNC_INTS,), dtype=torch.int32, device=device)
    return c


_EXCHANGE = {}


def _exchange_buffer(device, nbytes):
    """Grow-only scratch per (device, stream) for the split kernels' tile halves:
    launches on one stream are ordered, so one buffer serves them all."""
    key = (device, _raw_stream(device.index))
    b = _EXCHANGE.get(key)
    if b is None or b.numel() < nbytes:
        b = _EXCHANGE[key] = torch.empty((max(int(nbytes), 256),), dtype=torch.uint8,
                                         device=device)
    return b


def conv_forward(feat, packed_weight, nbr, n_out, c_out, weight_flip=False, row_order=None):
    """out[o] = sum_k feat[nbr[k,o]] @ W[k]  (implicit GEMM on MFMA)."""
    _need_cuda(feat, packed_weight, nbr)
    f = feat.contiguous().float()
    n_in, c_in = f.shape
    kvol, ld = nbr.shape
    out = torch.empty((n_out, c_out), dtype=torch.float32, device=f.device)
    # persistent tile scheduler whenever a (heaviest-first) order is supplied
    counter = _tile_counter(f.device) if row_order is not None else None
    _launch("spconv_fwd", lambda: check(
        lib.msmd_spconv_fwd_f32(_p(f), n_in, c_in, _p(packed_weight), _p(nbr), ld, int(n_out),
                                kvol, int(bool(weight_flip)), _p(row_order), _p(counter),
                                _p(out), int(c_out), _stream()), "msmd_spconv_fwd_f32"),
            nbr=nbr, c_in=c_in, c_out=int(c_out), n_in=n_in, n_out=int(n_out))
    return out


# ---------------------------------------------- split-bf16 ("fp32-equivalent") conv path
_SPLIT_SUPPORTED = {}


def split_supported(c_in, c_out, kvol=27):
    """Does the bf16-split implicit GEMM cover (contraction c_in, outputs c_out)?"""
    key = (c_in, c_out, kvol)
    v = _SPLIT_SUPPORTED.get(key)
    if v is None:
        v = _SPLIT_SUPPORTED[key] = bool(
            lib.msmd_spconv_fwd_split_supported(int(c_in), int(c_out), int(kvol)))
    return v


def pack_weight_split(weight, planes=3, transpose=False, krsc=False):
    """weight -> split bf16 planes in MFMA 16x16x32 fragment order."""
    _need_cuda(weight)
    w = weight.contiguous().float()
    if krsc:
        cout, cin = w.shape[0], w.shape[-1]
        kvol = w.numel() // (cout * cin)
    else:
        kvol, cin, cout = w.shape
    ci, co = (cout, cin) if transpose else (cin, cout)
    packed = torch.empty((lib.msmd_spconv_packed_split_bytes(kvol, ci, co, planes),),
                         dtype=torch.uint8, device=w.device)
    check(lib.msmd_spconv_pack_weight_split(_p(w), kvol, cin, cout,
                                            int(bool(transpose)) | (2 if krsc else 0), planes,
                                            _p(packed), _stream()),
          "msmd_spconv_pack_weight_split")
    return packed


def pack_weight_split_pair(weight, planes=3, krsc=False):
    """(pack_weight_split(w), pack_weight_split(w, transpose=True)) in one launch."""
    _need_cuda(weight)
    w = weight.contiguous().float()
    if krsc:
        cout, cin = w.shape[0], w.shape[-1]
        kvol = w.numel() // (cout * cin)
    else:
        kvol, cin, cout = w.shape
    a = torch.empty((lib.msmd_spconv_packed_split_bytes(kvol, cin, cout, planes),),
                    dtype=torch.uint8, device=w.device)
    b = torch.empty((lib.msmd_spconv_packed_split_bytes(kvol, cout, cin, planes),),
                    dtype=torch.uint8, device=w.device)
    check(lib.msmd_spconv_pack_weight_split_pair(_p(w), kvol, cin, cout, 2 if krsc else 0, planes,
                                                 _p(a), _p(b), _stream()),
          "msmd_spconv_pack_weight_split_pair")
    return a, b


class _PackDesc(C.Structure):      # csrc/spconv_split.hip: PackDesc (48 bytes)
    _fields_ = [("w", C.c_void_p), ("packed_a", C.c_void_p), ("packed_b", C.c_void_p),
                ("start", C.c_int64), ("kvol", C.c_int32), ("cin", C.c_int32),
                ("cout", C.c_int32), ("flags", C.c_int32)]


def pack_weight_split_many(jobs, planes=3):
    """pack_weight_split / pack_weight_split_pair of several weights in ONE launch.
    jobs: [(weight, krsc, packed, packed_transposed | None)] with `packed*` preallocated uint8
    tensors of msmd_spconv_packed_split_bytes each (same device).  The descriptor table goes to
    the device through a pinned staging copy (no host wait)."""
    if not jobs:
        return
    descs = (_PackDesc * len(jobs))()
    start = 0
    keep = []
    for i, (weight, krsc, packed, packed_t) in enumerate(jobs):
        _need_cuda(weight, packed, packed_t)
        w = weight.detach()
        if not (w.is_contiguous() and w.dtype == torch.float32):
            w = w.contiguous().float()
        keep.append(w)
        if krsc:
            cout, cin = w.shape[0], w.shape[-1]
            kvol = w.numel() // (cout * cin)
        else:
            kvol, cin, cout = w.shape
        d = descs[i]
        d.w, d.packed_a = w.data_ptr(), packed.data_ptr()
        d.packed_b = packed_t.data_ptr() if packed_t is not None else None
        d.start, d.kvol, d.cin, d.cout, d.flags = start, kvol, cin, cout, 2 if krsc else 0
        # one work unit = the `planes` 16-byte pieces of one (offset, k-block, tile, lane)
        start += lib.msmd_spconv_packed_split_bytes(kvol, cin, cout, planes) // (16 * planes)
        if packed_t is not None:
            start += lib.msmd_spconv_packed_split_bytes(kvol, cout, cin, planes) // (16 * planes)
    dev = jobs[0][0].device
    host = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).pin_memory()
    table = host.to(dev, non_blocking=True)
    check(lib.msmd_spconv_pack_weight_split_many(_p(table), len(jobs), start, int(planes),
                                                 _stream()), "msmd_spconv_pack_weight_split_many")
    # (the pinned buffer and the descriptor table must outlive the copy / the launch: they are
    # referenced by the stream until it has passed them)
    table.record_stream(torch.cuda.current_stream(dev))
    return host, table, keep


def permute_cols(nbr, order):
    """nbr[K,n] -> the table in tile order: out[k][p] = nbr[k][order[p]]."""
    _need_cuda(nbr, order)
    kvol, n = nbr.shape
    out = torch.empty_like(nbr)
    check(lib.msmd_rulebook_permute_cols(_p(nbr), kvol, n, n, _p(order), _p(out), _stream()),
          "msmd_rulebook_permute_cols")
    return out


_TILE_ROWS = {}


def split_tile_rows(c_out):
    """Rows per tile of the split kernel for a layer with c_out output channels."""
    v = _TILE_ROWS.get(c_out)
    if v is None:
        v = _TILE_ROWS[c_out] = int(lib.msmd_spconv_fwd_split_tile_rows(int(c_out)))
    return v


def split_instantiation(c_out):
    """dict(nt, ub, waves, buffers, pingpong, tables, passes): the template arguments
    msmd_spconv_fwd_split runs a layer of c_out output channels with (asked of the library:
    msmd_spconv_fwd_split_instantiation)."""
    p = (C.c_int * 7)()
    check(lib.msmd_spconv_fwd_split_instantiation(int(c_out), p),
          "msmd_spconv_fwd_split_instantiation")
    return dict(nt=p[0], ub=p[1], waves=p[2], buffers=p[3], pingpong=bool(p[4]), tables=p[5],
                passes=p[6])


def tile_prefix(nbr, rows=TILE_ROWS):
    """Stream-K work table of a neighbour table given in the order the conv kernel tiles
    it (tile order when a row order is used), for tiles of `rows` rows
    (split_tile_rows(c_out) of the layer): int32 [n_tiles + 1], see
    msmd_rulebook_tile_prefix."""
    _need_cuda(nbr)
    kvol, n = nbr.shape
    t = nbr.contiguous()
    out = torch.empty(((n + rows - 1) // rows + 1,), dtype=torch.int32, device=t.device)
    check(lib.msmd_rulebook_tile_prefix(_p(t), kvol, n, n, int(rows), _p(out), _stream()),
          "msmd_rulebook_tile_prefix")
    return out


def conv_forward_split(feat, packed_weight, nbr, n_out, c_out, planes=3, weight_flip=False,
                       row_order=None, split_tiles=True, tile_prefix=None, bn_stats=False):
    """conv_forward at bf16 MFMA rate with fp32-equivalent results: fp32 features
    are split into `planes` bf16 planes in registers, weights are pre-split
    (pack_weight_split).  With row_order, `nbr` must be in tile order (permute_cols).
    split_tiles=False withholds the exchange buffer: every 128-row tile is then one
    scheduling unit and the result does not depend on the tiling order at all (with
    it, tiles that a scheduling boundary cuts are summed in pieces: same value to the
    last bit or two).  tile_prefix (tile_prefix(nbr), with the exchange buffer): stream-K
    scheduling -- every workgroup gets the same share of the launch's work."""
    _need_cuda(feat, packed_weight, nbr)
    f = feat.contiguous().float()
    n_in, c_in = f.shape
    kvol, ld = nbr.shape
    out = torch.empty((n_out, c_out), dtype=torch.float32, device=f.device)
    counter = _tile_counter(f.device)
    nbytes = lib.msmd_spconv_fwd_split_workspace_bytes(int(n_out), int(c_out))
    ws = _exchange_buffer(f.device, nbytes) if split_tiles else None
    if tile_prefix is not None and split_tiles:
        rows = split_tile_rows(c_out)
        if tile_prefix.numel() != (int(n_out) + rows - 1) // rows + 1:
            raise ValueError("tile_prefix was not computed for %d-row tiles (K.tile_prefix(nbr, "
                             "K.split_tile_rows(c_out)))" % rows)
    # bn_stats: also the per-tile column sums / sums of squares of the output, for the
    # BatchNorm that follows (bn_act_forward(partials=...)) -> (out, partials)
    part = None
    if bn_stats and int(n_out) > 0:
        part = torch.empty((int(lib.msmd_spconv_fwd_split_stats_blocks(int(n_out), int(c_out))), 2,
                            int(c_out)), dtype=torch.float32, device=f.device)
    _launch("spconv_fwd_split", lambda: check(
        lib.msmd_spconv_fwd_split_stats(_p(f), n_in, c_in, _p(packed_weight), _p(nbr), ld,
                                        int(n_out), kvol, int(bool(weight_flip)),
                                        _p(row_order), _p(counter), counter.numel(), _p(out),
                                        int(c_out), int(planes), _p(ws),
                                        0 if ws is None else ws.numel(),
                                        _p(tile_prefix) if split_tiles else None, _p(part),
                                        _stream()), "msmd_spconv_fwd_split_stats"),
            nbr=nbr, c_in=c_in, c_out=int(c_out), n_in=n_in, n_out=int(n_out))
    return (out, part) if bn_stats else out


def conv_wgrad(feat, d_out, pairs, num, krsc_shape=None):
    """dW from the compact pair lists: [K,Cin,Cout], or laid out as the KRSC
    parameter when krsc_shape (= weight.shape) is given."""
    _need_cuda(feat, d_out, pairs, num)
    f, g = feat.contiguous().float(), d_out.contiguous().float()
    kvol, _, ld = pairs.shape
    c_in, c_out = f.shape[1], g.shape[1]
    dw = torch.empty((kvol, c_in, c_out) if krsc_shape is None else tuple(krsc_shape),
                     dtype=torch.float32, device=f.device)
    nbytes = lib.msmd_spconv_wgrad_workspace_bytes(kvol, ld, c_in, c_out)
    ws = _ws(nbytes, f.device)
    _launch("spconv_wgrad", lambda: check(
        lib.msmd_spconv_wgrad_f32(_p(f), c_in, _p(g), c_out, _p(pairs), _p(num), ld, kvol,
                                  _p(dw), int(krsc_shape is not None), _p(ws), nbytes,
                                  _stream()), "msmd_spconv_wgrad_f32"),
            num=num, c_in=c_in, c_out=c_out, n_in=f.shape[0], n_out=g.shape[0])
    return dw


def wgrad_split_supported(c_in, c_out):
    return bool(lib.msmd_spconv_wgrad_split_supported(int(c_in), int(c_out)))


# Rows per chunk of the wgrad kernel's row-chunk-major sequence; 0 (default) = offset-major.
# Measured (MI355X, tools/wgrad_ablate.py, 128 x 128 at 89.7 k rows): offset-major 263 us;
# 8192 / 4096 / 2048 / 1024-row chunks 270 / 278 / 308 / 367 us -- every (chunk, offset)
# boundary is another 64 KB partial slot for the reduction pass to read, and the main kernel
# does not get faster with its rows in L2: it is not the bandwidth-bound kernel the PMC
# traffic figure (651 MB per launch) suggested.
WGRAD_CHUNK_ROWS = int(os.environ.get("MSMD_WGRAD_CHUNK_ROWS", "0"))
WGRAD_MAX_CHUNKS = 256


def pair_segments(pairs, num, chunk_rows=None):
    """Row-chunk segment table of pair lists (pairs[K,2,ld], num[K]) for conv_wgrad_split:
    -> (table int32, n_chunks); MSMD_WGRAD_CHUNK_ROWS=0 (default) = one chunk.
    Index data: computed once per rulebook (IndiceData.pair_segments)."""
    _need_cuda(pairs, num)
    chunk_rows = WGRAD_CHUNK_ROWS if chunk_rows is None else int(chunk_rows)
    kvol, _, ld = pairs.shape
    if ld <= 0:
        return None
    if chunk_rows <= 0:          # one chunk: the offset-major sequence, its table built HERE
        chunk_rows = ld          # (index pass) instead of by every wgrad launch
    n_chunks = (ld + chunk_rows - 1) // chunk_rows
    if n_chunks > WGRAD_MAX_CHUNKS:          # very large sets: larger chunks
        chunk_rows = (ld + WGRAD_MAX_CHUNKS - 1) // WGRAD_MAX_CHUNKS
        n_chunks = (ld + chunk_rows - 1) // chunk_rows
    n_ints = int(lib.msmd_rulebook_pair_segments_ints(kvol, n_chunks))
    if n_ints == 0:      # kernel volume > 64 (5x5x5 ...): the slab wgrad kernel, which takes no table
        return None
    table = torch.empty((n_ints,),
                        dtype=torch.int32, device=pairs.device)
    check(lib.msmd_rulebook_pair_segments(_p(pairs), _p(num), ld, kvol, chunk_rows, n_chunks,
                                          _p(table), _stream()), "msmd_rulebook_pair_segments")
    return table, n_chunks


def conv_wgrad_split(feat, d_out, pairs, num, planes=3, krsc_shape=None, segments=None):
    """conv_wgrad at bf16 MFMA rate (operands split into `planes` bf16 planes in
    registers; planes=3 is fp32-equivalent).  c_in, c_out >= 64, multiples of 4.
    segments = pair_segments(pairs, num): the whole-block kernel walks the pairs row chunk by
    row chunk (all offsets of a chunk together: its rows stay in L2) instead of offset by
    offset; same sums in another fixed order."""
    _need_cuda(feat, d_out, pairs, num)
    f, g = feat.contiguous().float(), d_out.contiguous().float()
    kvol, _, ld = pairs.shape
    c_in, c_out = f.shape[1], g.shape[1]
    dw = torch.empty((kvol, c_in, c_out) if krsc_shape is None else tuple(krsc_shape),
                     dtype=torch.float32, device=f.device)
    table, n_chunks = segments if segments is not None else (None, 1)
    nbytes = lib.msmd_spconv_wgrad_segments_workspace_bytes(kvol, ld, c_in, c_out, n_chunks)
    ws = _ws(nbytes, f.device)
    _launch("spconv_wgrad_split", lambda: check(
        lib.msmd_spconv_wgrad_split_segments(_p(f), c_in, _p(g), c_out, _p(pairs), _p(num), ld,
                                             kvol, int(planes), _p(dw),
                                             int(krsc_shape is not None), _p(table),
                                             int(n_chunks), _p(ws), nbytes, _stream()),
        "msmd_spconv_wgrad_split_segments"),
            num=num, c_in=c_in, c_out=c_out, n_in=f.shape[0], n_out=g.shape[0])
    return dw


# ------------------------------------------------------------------ BN (+residual)(+ReLU)
def bn_act_forward(x, residual, gamma, beta, running_mean, running_var, training, momentum, eps,
                   relu, partials=None):
    """-> (y, save_mean, save_invstd).  partials (training only): [blocks, 2, c] column sums /
    sums of squares of disjoint row blocks covering x, from the kernel that produced x
    (conv_forward_split(bn_stats=True)): the statistics pass is skipped."""
    _need_cuda(x, gamma, beta)
    xx = x.contiguous().float()
    n, c = xx.shape
    dev = xx.device
    y = torch.empty_like(xx)
    stats = torch.empty((2, c), dtype=torch.float32, device=dev)
    mean, invstd = stats[0], stats[1]
    nbytes = lib.msmd_bn_workspace_bytes(n, c)
    ws = _ws(nbytes, dev)
    res = None if residual is None else residual.contiguous().float()
    if partials is not None and training:
        assert partials.shape[1:] == (2, c) and partials.is_contiguous()
    if partials is not None and training and n > 0:
        check(lib.msmd_bn_act_fwd_from_partials_f32(
            _p(xx), _p(res), n, c, _p(gamma), _p(beta), _p(running_mean), _p(running_var),
            float(momentum), float(eps), int(bool(relu)), _p(y), _p(mean), _p(invstd),
            _p(partials), int(partials.shape[0]), _stream()), "msmd_bn_act_fwd_from_partials_f32")
        return y, mean, invstd
    check(lib.msmd_bn_act_fwd_f32(_p(xx), _p(res), n, c, _p(gamma), _p(beta), _p(running_mean),
                                  _p(running_var), int(bool(training)), float(momentum),
                                  float(eps), int(bool(relu)), _p(y), _p(mean), _p(invstd), _p(ws),
                                  nbytes, _stream()), "msmd_bn_act_fwd_f32")
    return y, mean, invstd


def bn_act_backward(x, y, dy, gamma, save_mean, save_invstd, training, relu, want_residual):
    """-> (dx, dresidual | None, dgamma, dbeta)"""
    _need_cuda(x, dy)
    xx, g = x.contiguous().float(), dy.contiguous().float()
    n, c = xx.shape
    dev = xx.device
    dx = torch.empty_like(xx)
    dres = torch.empty_like(xx) if want_residual else None
    dgb = torch.empty((2, c), dtype=torch.float32, device=dev)
    dgamma, dbeta = dgb[0], dgb[1]
    nbytes = lib.msmd_bn_workspace_bytes(n, c)
    ws = _ws(nbytes, dev)
    check(lib.msmd_bn_act_bwd_f32(_p(xx), _p(y), _p(g), n, c, _p(gamma), _p(save_mean),
                                  _p(save_invstd), int(bool(training)), int(bool(relu)), _p(dx),
                                  _p(dres), _p(dgamma), _p(dbeta), _p(ws), nbytes, _stream()),
          "msmd_bn_act_bwd_f32")
    return dx, dres, dgamma, dbeta


def bn_relu_backward(x, dy, gamma, beta, save_mean, save_invstd, training):
    """bn_act_backward for BatchNorm + ReLU without a residual, without reading y (the mask is
    recomputed from x: csrc/bn.hip BwdStatRecompute).  -> (dx, dgamma, dbeta)"""
    _need_cuda(x, dy)
    xx, g = x.contiguous().float(), dy.contiguous().float()
    n, c = xx.shape
    dev = xx.device
    dx = torch.empty_like(xx)
    dgb = torch.empty((2, c), dtype=torch.float32, device=dev)
    nbytes = lib.msmd_bn_workspace_bytes(n, c)
    ws = _ws(nbytes, dev)
    check(lib.msmd_bn_relu_bwd_f32(_p(xx), _p(g), n, c, _p(gamma), _p(beta), _p(save_mean),
                                   _p(save_invstd), int(bool(training)), _p(dx), _p(dgb[0]),
                                   _p(dgb[1]), _p(ws), nbytes, _stream()), "msmd_bn_relu_bwd_f32")
    return dx, dgb[0], dgb[1]


# ------------------------------------------------------------------ dense / sets
def dense_scatter(feat, indices, batch_size, spatial_shape):
    _need_bzyx(indices)
    _need_cuda(feat, indices)
    f = feat.contiguous().float()
    n, c = f.shape
    d, h, w = [int(x) for x in spatial_shape]
    out = torch.empty((int(batch_size), c, d, h, w), dtype=torch.float32, device=f.device)
    check(lib.msmd_dense_scatter_f32(_p(f), _p(indices.contiguous().int()), n, c, int(batch_size),
                                     int3(spatial_shape), _p(out), _stream()),
          "msmd_dense_scatter_f32")
    return out


def dense_gather(dense, indices, spatial_shape):
    _need_bzyx(indices)
    _need_cuda(dense, indices)
    dn = dense.contiguous().float()
    b, c = dn.shape[0], dn.shape[1]
    n = indices.shape[0]
    feat = torch.empty((n, c), dtype=torch.float32, device=dn.device)
    check(lib.msmd_dense_gather_f32(_p(dn), _p(indices.contiguous().int()), n, c, b,
                                    int3(spatial_shape), _p(feat), _stream()),
          "msmd_dense_gather_f32")
    return feat


def bev_scatter_nhwc(feat, indices, batch_size, spatial_shape, bev, channel_offset):
    """feat rows -> channels [channel_offset, +c*D) of the zero-filled NHWC buffer
    bev [B,H,W,Ctot] (in place)."""
    _need_bzyx(indices)
    _need_cuda(feat, indices, bev)
    f = feat.contiguous().float()
    n, c = f.shape
    d, h, w = [int(x) for x in spatial_shape]
    if bev.dtype != torch.float32 or not bev.is_contiguous() or \
            tuple(bev.shape[:3]) != (int(batch_size), h, w):
        raise ValueError("bev must be a contiguous float32 [B,H,W,Ctot] buffer, got %s"
                         % (tuple(bev.shape),))
    check(lib.msmd_bev_scatter_nhwc_f32(_p(f), _p(indices.contiguous().int()), n, c,
                                        int(batch_size), int3(spatial_shape), _p(bev),
                                        int(bev.shape[3]), int(channel_offset), _stream()),
          "msmd_bev_scatter_nhwc_f32")
    return bev


def bev_gather_nhwc(bev, indices, c, batch_size, spatial_shape, channel_offset):
    _need_bzyx(indices)
    _need_cuda(bev, indices)
    if bev.dtype != torch.float32 or not bev.is_contiguous() or bev.dim() != 4:
        raise ValueError("bev must be a contiguous float32 [B,H,W,Ctot] buffer")
    n = indices.shape[0]
    feat = torch.empty((n, int(c)), dtype=torch.float32, device=bev.device)
    check(lib.msmd_bev_gather_nhwc_f32(_p(bev), _p(indices.contiguous().int()), n, int(c),
                                       int(batch_size), int3(spatial_shape), _p(feat),
                                       int(bev.shape[3]), int(channel_offset), _stream()),
          "msmd_bev_gather_nhwc_f32")
    return feat


# ------------------------------------------------------------ image-side glue
def _strides4(t):
    return (C.c_int64 * 4)(*[int(s) for s in t.stride()])


def _pixels(pixels):
    if pixels.dim() != 2 or pixels.shape[1] != 3 or \
            pixels.dtype not in (torch.float32, torch.float64):
        raise ValueError("pixels must be [n,3] float32/float64 (x, y, depth)")
    return pixels.contiguous(), int(pixels.dtype == torch.float64)


def fg_gather(img_feat, pixels, plane, downscale, pts, lidar2img, want_cells=True):
    """-> (fg_pcd [n, pts_dim+C], score_in [n, C+17], cells [n] | None, n_bad [1])"""
    _need_cuda(img_feat, pixels, plane, pts, lidar2img)
    if img_feat.dim() != 4 or img_feat.dtype != torch.float32:
        raise ValueError("img_feat must be float32 [planes, C, H, W]")
    px, is64 = _pixels(pixels)
    n = px.shape[0]
    planes, c, h, w = img_feat.shape
    pts = pts.contiguous().float()
    if pts.shape[0] != n or plane.shape[0] != n:
        raise ValueError("pixels / points / plane ids disagree on the number of points")
    l2i = lidar2img.contiguous().float().view(-1, 16)
    if l2i.shape[0] != planes:
        raise ValueError("need one 4x4 lidar2img per (sample, camera) plane")
    dev = img_feat.device
    fg = torch.empty((n, pts.shape[1] + c), dtype=torch.float32, device=dev)
    sc = torch.empty((n, c + 17), dtype=torch.float32, device=dev)
    cells = torch.empty((n,), dtype=torch.int32, device=dev) if want_cells else None
    bad = torch.empty((1,), dtype=torch.int32, device=dev)
    check(lib.msmd_fg_gather_f32(_p(img_feat), _strides4(img_feat), planes, c, h, w, _p(px), is64,
                                 _p(plane.contiguous().int()), float(downscale), _p(pts),
                                 int(pts.shape[1]), _p(l2i), n, _p(fg), _p(sc), _p(cells), _p(bad),
                                 _stream()), "msmd_fg_gather_f32")
    return fg, sc, cells, bad


def fg_gather_scored(img_feat, pixels, plane, downscale, pts, lidar2img, score_weight,
                     score_bias, n_scaled=None):
    """get_foreground2D without gradients, one launch: -> (fg_pcd [n, pts_dim+C] = [pts |
    feat * ReLU(score_net row)] for the first n_scaled points (default all), [pts | feat]
    for the rest; n_bad [1]).  score_weight [C+17] (or [1, C+17]), score_bias [1]."""
    _need_cuda(img_feat, pixels, plane, pts, lidar2img, score_weight, score_bias)
    if img_feat.dim() != 4 or img_feat.dtype != torch.float32:
        raise ValueError("img_feat must be float32 [planes, C, H, W]")
    px, is64 = _pixels(pixels)
    n = px.shape[0]
    planes, c, h, w = img_feat.shape
    pts = pts.contiguous().float()
    if pts.shape[0] != n or plane.shape[0] != n:
        raise ValueError("pixels / points / plane ids disagree on the number of points")
    l2i = lidar2img.contiguous().float().view(-1, 16)
    if l2i.shape[0] != planes:
        raise ValueError("need one 4x4 lidar2img per (sample, camera) plane")
    sw = score_weight.detach().contiguous().float().view(-1)
    sb = score_bias.detach().contiguous().float().view(-1)
    if sw.numel() != c + 17 or sb.numel() != 1:
        raise ValueError("score_net is Linear(%d, 1): got weight %s, bias %s"
                         % (c + 17, tuple(score_weight.shape), tuple(score_bias.shape)))
    dev = img_feat.device
    fg = torch.empty((n, pts.shape[1] + c), dtype=torch.float32, device=dev)
    bad = torch.empty((1,), dtype=torch.int32, device=dev)
    check(lib.msmd_fg_gather_scored_f32(_p(img_feat), _strides4(img_feat), planes, c, h, w, _p(px),
                                        is64, _p(plane.contiguous().int()), float(downscale),
                                        _p(pts), int(pts.shape[1]), _p(l2i), _p(sw), _p(sb), n,
                                        n if n_scaled is None else int(n_scaled), _p(fg), _p(bad),
                                        _stream()), "msmd_fg_gather_scored_f32")
    return fg, bad


def fg_scatter_add(grad, col0, c, cells, like):
    """Backward of fg_gather w.r.t. the feature map: zeros_like(like) + scatter."""
    _need_cuda(grad, cells, like)
    g = grad.contiguous().float()
    out = torch.zeros_like(like)
    planes, cc, h, w = like.shape
    check(lib.msmd_fg_scatter_add_f32(_p(g), int(g.shape[1]), int(col0), _p(cells),
                                      int(g.shape[0]), planes, int(c), h, w, _strides4(out),
                                      _p(out), _stream()), "msmd_fg_scatter_add_f32")
    return out


def depth_canvas(pixels, plane, planes, h, w):
    """-> (canvas [planes,h,w] float32, n_bad [1])"""
    _need_cuda(pixels, plane)
    px, is64 = _pixels(pixels)
    dev = px.device
    canvas = torch.empty((int(planes), int(h), int(w)), dtype=torch.float32, device=dev)
    bad = torch.empty((1,), dtype=torch.int32, device=dev)
    nbytes = lib.msmd_depth_canvas_workspace_bytes(int(planes), int(h), int(w))
    ws = _ws(nbytes, dev)
    check(lib.msmd_depth_canvas_f32(_p(px), is64, _p(plane.contiguous().int()), int(px.shape[0]),
                                    int(planes), int(h), int(w), _p(canvas), _p(bad), _p(ws),
                                    nbytes, _stream()), "msmd_depth_canvas_f32")
    return canvas, bad


# ------------------------------------------------ head targets / loss (row f3)
def boxes_overlap_bev(boxes_a, boxes_b):
    """iou3d_cuda.boxes_overlap_bev_gpu: rotated rectangles (x1, y1, x2, y2, angle) ->
    overlap AREAS [na, nb]."""
    _need_cuda(boxes_a, boxes_b)
    a, b = boxes_a.contiguous().float(), boxes_b.contiguous().float()
    if a.dim() != 2 or b.dim() != 2 or a.shape[1] != 5 or b.shape[1] != 5:
        raise ValueError("boxes must be [n,5] (x1, y1, x2, y2, angle)")
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    check(lib.msmd_boxes_overlap_bev_f32(_p(a), a.shape[0], _p(b), b.shape[0], _p(out), _stream()),
          "msmd_boxes_overlap_bev_f32")
    return out


def boxes_iou3d(boxes_a, boxes_b, nb_valid=None, mode="iou"):
    """BaseInstance3DBoxes.overlaps for LiDAR boxes (x, y, z_bottom, dx, dy, dz, yaw, ...).
    [na, ca] x [nb, cb] -> [na, nb], or batched [B, na, ca] x [B, nb, cb] (+ nb_valid [B]:
    rows of boxes_b in use per sample) -> [B, na, nb]."""
    _need_cuda(boxes_a, boxes_b)
    if mode not in ("iou", "iof"):
        raise ValueError("mode must be 'iou' or 'iof'")
    a, b = boxes_a.contiguous().float(), boxes_b.contiguous().float()
    single = a.dim() == 2
    if single:
        a, b = a[None], b[None]
    if a.dim() != 3 or b.dim() != 3 or a.shape[0] != b.shape[0] or a.shape[2] < 7 or b.shape[2] < 7:
        raise ValueError("boxes must be [B,n,>=7] with equal B (or [n,>=7])")
    batch, na, nb = a.shape[0], a.shape[1], b.shape[1]
    if nb_valid is not None:
        _need_cuda(nb_valid)
        nb_valid = nb_valid.contiguous().int()
        if nb_valid.numel() != batch:
            raise ValueError("nb_valid must hold one count per sample")
    out = torch.empty((batch, na, nb), dtype=torch.float32, device=a.device)
    check(lib.msmd_boxes_iou3d_f32(_p(a), a.shape[2], _p(b), b.shape[2],
                                   _p(nb_valid) if nb_valid is not None else None, batch, na, nb,
                                   0 if mode == "iou" else 1, _p(out), _stream()),
          "msmd_boxes_iou3d_f32")
    return out[0] if single else out


def heatmap_gaussian(heatmap, plane, center_x, center_y, radius):
    """heatmap [planes, H, W] (or [B, C, H, W]) float32, updated in place: maximum with one
    Gaussian bump per box (plane = sample * C + class; plane < 0 or radius < 0 skips)."""
    _need_cuda(heatmap, plane, center_x, center_y, radius)
    if heatmap.dtype != torch.float32 or not heatmap.is_contiguous() or heatmap.dim() < 3:
        raise ValueError("heatmap must be a contiguous float32 [..., H, W] buffer")
    h, w = heatmap.shape[-2:]
    planes = heatmap.numel() // (h * w)
    n = plane.numel()
    args = [t.contiguous().int() for t in (plane, center_x, center_y, radius)]
    if any(t.numel() != n for t in args):
        raise ValueError("plane / center / radius must have one entry per box")
    check(lib.msmd_heatmap_gaussian_f32(_p(args[0]), _p(args[1]), _p(args[2]), _p(args[3]), n,
                                        planes, h, w, _p(heatmap), _stream()),
          "msmd_heatmap_gaussian_f32")
    return heatmap


def gaussian_focal(logits, target, clip=1e-4, want_grad=True):
    """clip_sigmoid + GaussianFocalLoss(alpha=2, gamma=4), unreduced sums.
    -> (sums [2] = (sum of losses, cells with target == 1), grad like logits | None)"""
    _need_cuda(logits, target)
    x, t = logits.contiguous().float(), target.contiguous().float()
    if x.shape != t.shape:
        raise ValueError("logits and target must have the same shape")
    n = x.numel()
    sums = torch.empty((2,), dtype=torch.float32, device=x.device)
    grad = torch.empty_like(x) if want_grad else None
    nbytes = lib.msmd_gaussian_focal_workspace_bytes(n)
    ws = _ws(nbytes, x.device)
    check(lib.msmd_gaussian_focal_f32(_p(x), _p(t), n, float(clip),
                                      _p(grad) if grad is not None else None, _p(sums), _p(ws),
                                      nbytes, _stream()), "msmd_gaussian_focal_f32")
    return sums, grad


def sparse_add(feat_a, idx_a, feat_b, idx_b, batch_size, spatial_shape):
    """-> (out_indices, out_feat, map_a, map_b)"""
    _need_bzyx(idx_a, idx_b)
    _need_cuda(feat_a, idx_a, feat_b, idx_b)
    fa, fb = feat_a.contiguous().float(), feat_b.contiguous().float()
    ia, ib = idx_a.contiguous().int(), idx_b.contiguous().int()
    na, nb, c = fa.shape[0], fb.shape[0], fa.shape[1]
    dev = fa.device
    nbytes = lib.msmd_sparse_add_workspace_bytes(int(batch_size), int3(spatial_shape))
    ws = _ws(nbytes, dev)
    count = _Count(dev)
    check(lib.msmd_sparse_add_count(_p(ia), na, _p(ib), nb, int(batch_size), int3(spatial_shape),
                                    count.ptr, _p(ws), nbytes, _stream()), "msmd_sparse_add_count")
    m = count.read()
    oi = torch.empty((m, 4), dtype=torch.int32, device=dev)
    of = torch.empty((m, c), dtype=torch.float32, device=dev)
    ma = torch.empty((na,), dtype=torch.int32, device=dev)
    mb = torch.empty((nb,), dtype=torch.int32, device=dev)
    check(lib.msmd_sparse_add_fill(_p(fa), _p(ia), na, _p(fb), _p(ib), nb, c, int(batch_size),
                                   int3(spatial_shape), m, _p(oi), _p(of), _p(ma), _p(mb), _p(ws),
                                   nbytes, _stream()), "msmd_sparse_add_fill")
    return oi, of, ma, mb


def sparse_add_index(idx_a, idx_b, batch_size, spatial_shape):
    """Index half of sparse_add -> (out_indices[m,4], map_a[na], map_b[nb])."""
    _need_bzyx(idx_a, idx_b)
    _need_cuda(idx_a, idx_b)
    ia, ib = idx_a.contiguous().int(), idx_b.contiguous().int()
    na, nb = ia.shape[0], ib.shape[0]
    dev = ia.device
    nbytes = lib.msmd_sparse_add_workspace_bytes(int(batch_size), int3(spatial_shape))
    ws = _ws(nbytes, dev)
    count = _Count(dev)
    check(lib.msmd_sparse_add_count(_p(ia), na, _p(ib), nb, int(batch_size), int3(spatial_shape),
                                    count.ptr, _p(ws), nbytes, _stream()), "msmd_sparse_add_count")
    m = count.read()
    oi = torch.empty((m, 4), dtype=torch.int32, device=dev)
    ma = torch.empty((na,), dtype=torch.int32, device=dev)
    mb = torch.empty((nb,), dtype=torch.int32, device=dev)
    check(lib.msmd_sparse_add_fill(None, _p(ia), na, None, _p(ib), nb, 0, int(batch_size),
                                   int3(spatial_shape), m, _p(oi), None, _p(ma), _p(mb), _p(ws),
                                   nbytes, _stream()), "msmd_sparse_add_fill")
    return oi, ma, mb


def sparse_add_rows(feat_a, map_a, feat_b, map_b, n_out):
    """Feature half of sparse_add with the maps of sparse_add_index (no host read)."""
    _need_cuda(feat_a, map_a, feat_b, map_b)
    fa, fb = feat_a.contiguous().float(), feat_b.contiguous().float()
    assert fa.shape[1] == fb.shape[1] and map_a.dtype == map_b.dtype == torch.int32
    assert map_a.shape[0] == fa.shape[0] and map_b.shape[0] == fb.shape[0]
    of = torch.empty((int(n_out), fa.shape[1]), dtype=torch.float32, device=fa.device)
    check(lib.msmd_sparse_add_rows(_p(fa), _p(map_a.contiguous()), fa.shape[0], _p(fb),
                                   _p(map_b.contiguous()), fb.shape[0], fa.shape[1], int(n_out),
                                   _p(of), _stream()), "msmd_sparse_add_rows")
    return of


def rows_inverse(row_map, n_out):
    """inv[j] = the last row i with row_map[i] == j, -1 where none (int32 [n_out])."""
    _need_cuda(row_map)
    m = row_map if (row_map.dtype == torch.int32 and row_map.is_contiguous()) \
        else row_map.contiguous().int()
    inv = torch.empty((int(n_out),), dtype=torch.int32, device=m.device)
    check(lib.msmd_rows_inverse(_p(m), m.shape[0], int(n_out), _p(inv), _stream()),
          "msmd_rows_inverse")
    return inv


def sparse_add_rows_gather(feat_a, map_a, inv_a, feat_b, map_b, inv_b, n_out):
    """sparse_add_rows as a gather through the inverse maps (rows_inverse of map_a / map_b):
    every output row written once, no zero fill, no float atomics unless a tensor repeats a
    coordinate."""
    _need_cuda(feat_a, feat_b, inv_a, inv_b)
    fa, fb = feat_a.contiguous().float(), feat_b.contiguous().float()
    assert fa.shape[1] == fb.shape[1] and fa.shape[1] % 4 == 0
    assert map_a.shape[0] == fa.shape[0] and map_b.shape[0] == fb.shape[0]
    assert inv_a.shape[0] == inv_b.shape[0] == int(n_out)
    of = torch.empty((int(n_out), fa.shape[1]), dtype=torch.float32, device=fa.device)
    check(lib.msmd_sparse_add_rows_gather(_p(fa), _p(map_a), _p(inv_a), fa.shape[0], _p(fb),
                                          _p(map_b), _p(inv_b), fb.shape[0], fa.shape[1],
                                          int(n_out), _p(of), _stream()),
          "msmd_sparse_add_rows_gather")
    return of


class _GmaAssemble(torch.autograd.Function):
    """Feature assembly of a GMA-Conv stage, one launch each way (csrc/gma.hip)."""

    @staticmethod
    def forward(ctx, conv3, cross_gate, gate, feat3, feat2, nn3, rows_o2, rows_m3, rows_m2,
                n_o2_pad, n_mix_pad, order, starts):
        n_o3, c3 = conv3.shape
        n3, c2 = cross_gate.shape[0] - 1, cross_gate.shape[1]
        n_o2, n_mix = rows_o2.shape[0], rows_m3.shape[0]
        rows = n_o3 + n_o2 + n_o2_pad + n_mix + n_mix_pad
        out = torch.empty((rows, c3 + c2), dtype=torch.float32, device=conv3.device)
        check(lib.msmd_gma_assemble_fwd_f32(
            _p(conv3), n_o3, c3, _p(cross_gate), n3, c2, _p(nn3), _p(feat2), _p(rows_o2), n_o2,
            int(n_o2_pad), _p(feat3), _p(rows_m3), _p(gate), _p(rows_m2), n_mix, int(n_mix_pad),
            _p(out), _stream()), "msmd_gma_assemble_fwd_f32")
        ctx.save_for_backward(feat2, nn3, rows_o2, rows_m2)
        ctx.csr = (order, starts)
        ctx.dims = (n_o3, c3, n3, c2, n_o2, int(n_o2_pad), n_mix, int(n_mix_pad))
        return out

    @staticmethod
    def backward(ctx, d_out):
        feat2, nn3, rows_o2, rows_m2 = ctx.saved_tensors
        n_o3, c3, n3, c2, n_o2, n_o2_pad, n_mix, n_mix_pad = ctx.dims
        d = d_out.contiguous().float()
        need = ctx.needs_input_grad
        d_conv3 = torch.empty((n_o3, c3), dtype=torch.float32, device=d.device)
        d_cross = torch.empty((n3 + 1, c2), dtype=torch.float32, device=d.device) if need[1] else None
        d_gate = torch.empty((n_mix, c2), dtype=torch.float32, device=d.device) if need[2] else None
        ws = None
        if ctx.csr[0] is not None and d_cross is not None:
            ws = _ws(4 * lib.msmd_gma_assemble_bwd_workspace_floats(c2), d.device)
        check(lib.msmd_gma_assemble_bwd_f32(
            _p(d), n_o3, c3, n3, c2, _p(nn3), _p(feat2), _p(rows_o2), n_o2, n_o2_pad,
            _p(rows_m2), n_mix, n_mix_pad, _p(d_conv3), _p(d_cross), _p(d_gate),
            _p(ctx.csr[0]), _p(ctx.csr[1]), _p(ws), _stream()), "msmd_gma_assemble_bwd_f32")
        return (d_conv3 if need[0] else None, d_cross, d_gate) + (None,) * 10


def gma_nn_segments(nn3, n3):
    """(order, starts) for gma_assemble's backward: the only-2D rows sorted by their nearest
    3D voxel (-1 -> n3, stable) and, per target t in 0..n3, its slice starts[t]:starts[t+1] of
    `order`.  Index-only (no host read): the index pass builds it ahead of the feature pass."""
    key = torch.where(nn3 >= 0, nn3, torch.full_like(nn3, n3))
    skey, order = torch.sort(key, stable=True)
    starts = torch.searchsorted(skey, torch.arange(n3 + 2, device=nn3.device, dtype=skey.dtype))
    return order.contiguous(), starts.contiguous()


def gma_assemble(conv3, cross_gate, gate, feat3, feat2, nn3, rows_o2, rows_m3, rows_m2,
                 n_o2_pad=0, n_mix_pad=0, segments=None):
    """Unified features of a GMA-Conv stage (mmdet3d/models/middle_encoders/
    sparse_encoder_multimodal_encoderpaint_double_aware.py:349-421): rows
    [conv3 | 0], [0 | cross_gate[nn3] * feat2[rows_o2]], zero pad rows,
    [feat3[rows_m3] | gate * feat2[rows_m2]], zero pad rows.  Gradients flow to conv3,
    cross_gate and gate; feat3 / feat2 must not require one (the caller falls back to the
    unfused ops otherwise).  segments = gma_nn_segments(nn3, n3): d cross_gate is then summed
    in a fixed order (deterministic); without it by float atomics."""
    _need_cuda(conv3, cross_gate, gate, feat3, feat2, nn3, rows_o2, rows_m3, rows_m2)
    assert not feat3.requires_grad and not feat2.requires_grad
    assert nn3.dtype == rows_o2.dtype == rows_m3.dtype == rows_m2.dtype == torch.int64
    assert nn3.shape[0] == rows_o2.shape[0] and rows_m3.shape[0] == rows_m2.shape[0] == gate.shape[0]
    assert conv3.shape[1] == feat3.shape[1] and cross_gate.shape[1] == gate.shape[1] == feat2.shape[1]
    f = lambda t: t.contiguous().float()
    order, starts = segments if segments is not None else (None, None)
    if order is not None:
        assert order.dtype == starts.dtype == torch.int64 and order.shape[0] == nn3.shape[0]
        assert starts.shape[0] == cross_gate.shape[0] + 1
    return _GmaAssemble.apply(f(conv3), f(cross_gate), f(gate), f(feat3), f(feat2),
                              nn3.contiguous(), rows_o2.contiguous(), rows_m3.contiguous(),
                              rows_m2.contiguous(), int(n_o2_pad), int(n_mix_pad), order, starts)


def _modality_split_launch(idx_3d, idx_2d, batch_size, spatial_shape, float_keys,
                           reference_offsets, want_stats):
    """Enqueue one split -> (mix3d, mix2d, pair_3d, pair_2d, out) with out = int32
    [count | 4 * batch stats] on the device (capacity-sized pair lists: the caller trims)."""
    _need_bzyx(idx_3d, idx_2d)
    _need_cuda(idx_3d, idx_2d)
    a, b = idx_3d.contiguous().int(), idx_2d.contiguous().int()
    n3, n2 = a.shape[0], b.shape[0]
    dev = a.device
    B = int(batch_size)
    cap = max(min(n3, n2), 1)
    mix3 = torch.empty((n3,), dtype=torch.int32, device=dev)
    mix2 = torch.empty((n2,), dtype=torch.int32, device=dev)
    p3 = torch.empty((cap,), dtype=torch.int32, device=dev)
    p2 = torch.empty((cap,), dtype=torch.int32, device=dev)
    out = torch.empty((1 + 4 * B,), dtype=torch.int32, device=dev)   # count | stats
    stats_p = C.c_void_p(out.data_ptr() + 4) if want_stats else None
    if float_keys:
        nbytes = lib.msmd_modality_split_float_keys_workspace_bytes(n3, n2, B)
        ws = _ws(nbytes, dev)
        check(lib.msmd_modality_split_float_keys(_p(a), n3, _p(b), n2, B, int3(spatial_shape),
                                                 _p(mix3), _p(mix2), _p(p3), _p(p2), _p(out),
                                                 stats_p, int(bool(reference_offsets)), _p(ws),
                                                 nbytes, _stream()),
              "msmd_modality_split_float_keys")
    else:
        nbytes = lib.msmd_modality_split_workspace_bytes(B, int3(spatial_shape))
        ws = _ws(nbytes, dev)
        if want_stats:
            check(lib.msmd_modality_split_stats(_p(a), n3, _p(b), n2, B, int3(spatial_shape),
                                                _p(mix3), _p(mix2), _p(p3), _p(p2), _p(out),
                                                stats_p, _p(ws), nbytes, _stream()),
                  "msmd_modality_split_stats")
        else:
            check(lib.msmd_modality_split(_p(a), n3, _p(b), n2, B, int3(spatial_shape), _p(mix3),
                                          _p(mix2), _p(p3), _p(p2), _p(out), _p(ws), nbytes,
                                          _stream()), "msmd_modality_split")
    return mix3, mix2, p3, p2, out, (a, b, ws)


def _check_split_count(m):
    """msmd_modality_split_float_keys reports bad input through a count of -1."""
    if m < 0:
        raise ValueError("modality_split(float_keys=True): a voxel row lies outside the grid / "
                         "the batch, or (reference_offsets) the rows are not grouped by sample "
                         "in ascending order (include/msmd_hip.h)")


def modality_split(idx_3d, idx_2d, batch_size, spatial_shape, float_keys=False,
                   reference_offsets=False):
    """-> (mix3d[n3], mix2d[n2], pair_3d[m], pair_2d[m]).
    float_keys: the reference's float32 keys and two-pointer merge (csrc/modality_float.hip;
    MSMDFusion.py:271-272, 27-45) instead of exact integer keys; reference_offsets: pair rows
    numbered with the reference's non-cumulative batch offsets (:288-289,313-314) -- this
    needs each set's rows grouped by sample in ascending order (what voxelize() returns; the
    reference selects by mask and has no such requirement, nor has the exact-key path).
    Rows outside the grid and ungrouped rows raise ValueError (checked on the device)."""
    mix3, mix2, p3, p2, out, _keep = _modality_split_launch(
        idx_3d, idx_2d, batch_size, spatial_shape, float_keys, reference_offsets, False)
    m = int(out[0].item())
    _check_split_count(m)
    return mix3, mix2, p3[:m], p2[:m]


def modality_split_many(jobs, batch_size, float_keys=False, reference_offsets=False):
    """Several independent modality splits (the four image scales of a step) with ONE
    host read.  jobs: [(idx_3d, idx_2d, spatial_shape), ...] -> per job
    (mix3d, mix2d, pair_3d, pair_2d, stats) with stats = dict of per-sample row
    counts as python lists: c3_plain, c3_mixed, c2_plain, c2_mixed."""
    B = int(batch_size)
    pending = [_modality_split_launch(i3, i2, B, shape, float_keys, reference_offsets, True)
               for i3, i2, shape in jobs]
    if not pending:
        return []
    host = torch.stack([p[4] for p in pending]).tolist()
    res = []
    for (mix3, mix2, p3, p2, _, _), h in zip(pending, host):
        m = h[0]
        _check_split_count(m)
        stats = dict(c3_plain=h[1:1 + B], c3_mixed=h[1 + B:1 + 2 * B],
                     c2_plain=h[1 + 2 * B:1 + 3 * B], c2_mixed=h[1 + 3 * B:1 + 4 * B])
        res.append((mix3, mix2, p3[:m], p2[:m], stats))
    return res


CHECK_COUNTS = os.environ.get("MSMD_CHECK_COUNTS", "0") == "1"


def rows_where_eq(flags, value, count):
    """Row numbers i (int64, ascending) with flags[i] == value for a 1-D int32 tensor (any
    stride: a column of an index tensor) whose number of such rows the host already knows:
    one scan (two launches, csrc/dense.hip) instead of a compare + nonzero_static."""
    count = int(count)
    if count == 0 or flags.shape[0] == 0:
        return torch.empty((0,), dtype=torch.long, device=flags.device)
    if not flags.is_cuda or flags.dtype != torch.int32 or flags.dim() != 1:
        return rows_where(flags == value, count)
    n = flags.shape[0]
    # `count` is the host's number (msmd_modality_split_stats).  The kernel never leaves a
    # row of the output unwritten: rows past the device's own total are set to -1
    # (rows_tail_fill, csrc/dense.hip), so a stale host count cannot hand uninitialised row
    # ids to index_select -- it hands it -1, which torch's bounds check rejects.
    # MSMD_CHECK_COUNTS=1 (tests, debugging) reads the device total back and compares.
    rows = torch.empty((count,), dtype=torch.long, device=flags.device)
    nbytes = lib.msmd_rows_where_workspace_bytes(n)
    ws = _ws(nbytes, flags.device)
    total = torch.empty((1,), dtype=torch.int32, device=flags.device) if CHECK_COUNTS else None
    check(lib.msmd_rows_where_eq(_p(flags), int(flags.stride(0)), n, int(value), _p(rows), count,
                                 _p(total), _p(ws), nbytes, _stream()), "msmd_rows_where_eq")
    if total is not None and int(total.item()) != count:
        raise RuntimeError("rows_where_eq: the host expects %d rows with flag %d, the device "
                           "found %d (stale per-sample statistics?)"
                           % (count, int(value), int(total.item())))
    return rows


def rows_where_eq_many(jobs):
    """rows_where_eq for several (flags, value, count) triples in ONE scan (two launches):
    -> list of int64 row tensors.  flags: 1-D int32 CUDA tensors (any stride)."""
    if not jobs:
        return []
    dev = jobs[0][0].device
    n = len(jobs)
    outs, ptr_f, ptr_r = [], (C.c_void_p * n)(), (C.c_void_p * n)()
    strides, lens, values, caps = ((C.c_int * n)() for _ in range(4))
    for i, (flags, value, count) in enumerate(jobs):
        _need_cuda(flags)
        if flags.dtype != torch.int32 or flags.dim() != 1:
            raise ValueError("rows_where_eq_many: 1-D int32 flag vectors")
        count = int(count)
        out = torch.empty((count,), dtype=torch.long, device=dev)
        outs.append(out)
        ptr_f[i], ptr_r[i] = _p(flags), (_p(out) if count else None)
        strides[i] = int(flags.stride(0)) if flags.shape[0] else 1
        lens[i], values[i], caps[i] = flags.shape[0], int(value), count
    nbytes = lib.msmd_rows_where_eq_many_workspace_bytes(lens, n)
    ws = _ws(nbytes, dev)
    check(lib.msmd_rows_where_eq_many(ptr_f, strides, lens, values, ptr_r, caps, n, _p(ws), nbytes,
                                      _stream()), "msmd_rows_where_eq_many")
    return outs


def rows_where(mask, count):
    """Row numbers where a 1-D bool tensor is set, ascending, when their number is
    already known on the host: mask.nonzero() would wait for the device to size its
    output."""
    count = int(count)
    if count == 0:
        return torch.empty((0,), dtype=torch.long, device=mask.device)
    if _nonzero_static_ok(mask.device):
        return torch.nonzero_static(mask, size=count).flatten()
    # stable sort of the inverted mask: the selected rows come first, in order
    return torch.sort((~mask).to(torch.uint8), stable=True)[1][:count]


_NONZERO_STATIC = {}      # device type -> whether torch.nonzero_static works there


def _nonzero_static_ok(device):
    """Probed ONCE per device type on a tiny tensor: the hot path never swallows a
    RuntimeError (an asynchronous HIP error surfacing at that call would be hidden and the
    slow fallback taken for the rest of the process)."""
    ok = _NONZERO_STATIC.get(device.type)
    if ok is None:
        ok = False
        if hasattr(torch, "nonzero_static"):
            try:
                probe = torch.tensor([True, False, True], device=device)
                ok = torch.nonzero_static(probe, size=2).flatten().tolist() == [0, 2]
            except (NotImplementedError, RuntimeError):
                ok = False
        _NONZERO_STATIC[device.type] = ok
    return ok


# ------------------------------------------------------------------ GMA-Conv helpers
def furthest_point_sample(points_xyz, num_points):
    """mmdet3d/ops/furthest_point_sample/furthest_point_sample.py:14-35."""
    _need_cuda(points_xyz)
    x = points_xyz.contiguous().float()
    b, n, _ = x.shape
    out = torch.empty((b, num_points), dtype=torch.int32, device=x.device)
    tmp = torch.empty((b, n), dtype=torch.float32, device=x.device)
    check(lib.msmd_furthest_point_sample(_p(x), b, n, int(num_points), _p(tmp), _p(out),
                                         _stream()), "msmd_furthest_point_sample")
    return out


def furthest_point_sample_ragged(xyz, offsets, n_max, num_points):
    """FPS over a ragged batch: xyz [total,3], offsets int32 [b+1] (device),
    n_max = largest element (host int).  -> int32 [b, num_points], indices local
    to each element.  All elements run concurrently, one workgroup each."""
    _need_cuda(xyz, offsets)
    x = xyz.contiguous().float()
    b = offsets.shape[0] - 1
    out = torch.empty((b, num_points), dtype=torch.int32, device=x.device)
    tmp = torch.empty((max(x.shape[0], 1),), dtype=torch.float32, device=x.device)
    check(lib.msmd_furthest_point_sample_ragged(_p(x), _p(offsets.contiguous().int()), b,
                                                int(n_max), int(num_points), _p(tmp), _p(out),
                                                _stream()), "msmd_furthest_point_sample_ragged")
    return out


def ball_query(min_radius, max_radius, sample_num, xyz, center_xyz):
    """mmdet3d/ops/ball_query/ball_query.py:14-40 (same argument order)."""
    _need_cuda(xyz, center_xyz)
    assert min_radius < max_radius
    x, cx = xyz.contiguous().float(), center_xyz.contiguous().float()
    b, n, _ = x.shape
    m = cx.shape[1]
    idx = torch.empty((b, m, int(sample_num)), dtype=torch.int32, device=x.device)
    check(lib.msmd_ball_query(_p(cx), _p(x), b, n, m, float(min_radius), float(max_radius),
                              int(sample_num), _p(idx), _stream()), "msmd_ball_query")
    return idx


def nn_search(query_zyx, key_zyx, dist_thresh):
    _need_cuda(query_zyx, key_zyx)
    q, k = query_zyx.contiguous().int(), key_zyx.contiguous().int()
    nq, nk = q.shape[0], k.shape[0]
    out = torch.empty((nq,), dtype=torch.int32, device=q.device)
    scratch = torch.empty((max(nq, 1),), dtype=torch.int64, device=q.device)
    check(lib.msmd_nn_search(_p(q), nq, _p(k), nk, float(dist_thresh), _p(out), _p(scratch),
                             _stream()), "msmd_nn_search")
    return out


def nn_assign(group_idx, rep_nn, nq):
    _need_cuda(group_idx, rep_nn)
    g, r = group_idx.contiguous().int(), rep_nn.contiguous().int()
    m, ns = g.shape
    out = torch.empty((nq,), dtype=torch.int32, device=g.device)
    scratch = torch.empty((max(nq, 1),), dtype=torch.int32, device=g.device)
    check(lib.msmd_nn_assign(_p(g), _p(r), m, ns, int(nq), _p(out), _p(scratch), _stream()),
          "msmd_nn_assign")
    return out


# ------------------------------------------------------------------ gate tables (a16)
def rows_linear_supported(c_in, c_out):
    return bool(lib.msmd_rows_linear_supported(int(c_in), int(c_out)))


class _RowsLinear(torch.autograd.Function):
    """relu?(cat(x, x_tail) @ w^T + b) over feature rows (csrc/gma.hip); the weight and bias
    gradients come from per-block partial sums added in block order (deterministic); the
    input gradient, when something upstream wants it, is one matmul."""

    @staticmethod
    def forward(ctx, x, x_tail, w, b, relu):
        xx = x.contiguous().float()
        n, c_in = xx.shape
        tail = None if x_tail is None else x_tail.contiguous().float()
        n_tail = 0 if tail is None else tail.shape[0]
        ww = w.contiguous().float()
        bb = None if b is None else b.contiguous().float()
        c_out = ww.shape[0]
        y = torch.empty((n + n_tail, c_out), dtype=torch.float32, device=xx.device)
        check(lib.msmd_rows_linear_fwd_f32(_p(xx), n, _p(tail), n_tail, c_in, _p(ww), _p(bb), c_out,
                                           int(bool(relu)), _p(y), _stream()),
              "msmd_rows_linear_fwd_f32")
        ctx.save_for_backward(xx, tail, ww, y)
        ctx.relu, ctx.has_bias = bool(relu), b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        xx, tail, ww, y = ctx.saved_tensors
        g = dy.contiguous().float()
        n, c_in = xx.shape
        n_tail = 0 if tail is None else tail.shape[0]
        c_out = ww.shape[0]
        dx = dtail = dw = db = None
        if ctx.needs_input_grad[2] or (ctx.has_bias and ctx.needs_input_grad[3]):
            dw = torch.empty_like(ww)
            db = torch.empty((c_out,), dtype=torch.float32, device=ww.device) if ctx.has_bias else None
            nbytes = lib.msmd_rows_linear_bwd_workspace_bytes(n + n_tail, c_in, c_out)
            ws = _ws(nbytes, ww.device)
            check(lib.msmd_rows_linear_bwd_f32(_p(xx), n, _p(tail), n_tail, c_in, _p(y), _p(g),
                                               c_out, int(ctx.relu), _p(dw), _p(db), _p(ws), nbytes,
                                               _stream()), "msmd_rows_linear_bwd_f32")
        if ctx.needs_input_grad[0] or (tail is not None and ctx.needs_input_grad[1]):
            gm = g * (y > 0) if ctx.relu else g
            if ctx.needs_input_grad[0]:
                dx = gm[:n] @ ww
            if tail is not None and ctx.needs_input_grad[1]:
                dtail = gm[n:] @ ww
        return dx, dtail, dw, db, None


def rows_linear(x, weight, bias=None, relu=False, x_tail=None):
    """nn.Linear (+ ReLU) over feature rows, optionally with extra rows appended (x_tail)
    without materialising the concatenation."""
    _need_cuda(x, weight)
    return _RowsLinear.apply(x, x_tail, weight, bias, relu)
