"""Autograd glue over the C ABI: the role of spconv.pytorch.functional
(legacy twin: mmdet3d/ops/spconv/functional.py:20-75).

Weights arrive as [K, Cin, Cout] (the layout indiceConv contracts with,
spconv_ops.h:299); modules keep their parameter in spconv-2.x's KRSC layout
and hand a permuted view in, so autograd maps the gradient back.
"""
import os

import torch
from torch.autograd import Function

from .. import kernels as K


def conv_planes():
    """Arithmetic of the sparse convolutions (forward, dgrad and, for channel
    counts that are multiples of 64, wgrad):
    3 (default) -- operands split into three bf16 planes, six products on the
                   bf16 matrix cores, fp32 accumulate: fp32-equivalent results
                   (csrc/spconv_split.hip) at 2.7x fewer MFMA cycles;
    2           -- two planes, three products (relative error ~2^-17);
    0           -- the fp32 matrix instruction (csrc/spconv.hip).
    Layers the split kernels do not cover (fewer than 32 channels on a side,
    c_in % 8 != 0; for wgrad: not multiples of 64) always use the fp32 kernels."""
    return int(os.environ.get("MSMD_CONV_PLANES", "3"))


# The split kernel gathers through a 32-bit byte offset into the feature tensor
# (buffer loads, csrc/spconv_split.hip): tensors of 4 GiB and more take the fp32 kernels.
SPLIT_MAX_FEATURE_BYTES = 0xFFFFFF00


def _use_split(c_in, c_out, kvol, n_in=0):
    # c_in % 8 == 0 with a partial last k-block (the fusion stack's 80-channel layers)
    # included: 80->80 104 vs 114 us, 80->96 102 vs 135 since the split tiles (before
    # them the fp32 kernel won there, 118 vs 160).
    # Its gathers use 32-bit byte offsets: features beyond 4 GiB go the fp32 way.
    return (conv_planes() in (1, 2, 3) and n_in * c_in * 4 < SPLIT_MAX_FEATURE_BYTES
            and K.split_supported(c_in, c_out, kvol))


def _wants_order(c_in, c_out):
    """The pipelined kernel (c_out >= 64, c_in % 16 == 0) profits from the
    mask-sorted tiling order; the narrow layers ignore it."""
    return c_out >= 64 and c_in % 16 == 0


_F32_TILE_COUNTS = (12, 8, 6, 5, 4, 3, 2, 1)    # instantiations of msmd_spconv_fwd_f32


def _conv_f32(x, weight, krsc, transpose, nbr, n_out, weight_flip=False, row_order=None):
    """The fp32-MFMA kernel over any output width: its instantiations cover
    1-6, 8 and 12 tiles of 16 output channels; other widths (97..112, 129..176,
    > 192 channels) run as column passes over slices of the weight."""
    if krsc:
        c_out = weight.shape[-1] if transpose else weight.shape[0]
    else:
        c_out = weight.shape[1] if transpose else weight.shape[2]
    tiles = (c_out + 15) // 16
    if tiles in _F32_TILE_COUNTS:
        return K.conv_forward(x, _pack_f32(weight, transpose, krsc), nbr,
                              n_out, c_out, weight_flip=weight_flip, row_order=row_order)
    outs, c0 = [], 0
    while c0 < c_out:
        left = (c_out - c0 + 15) // 16
        width = min(16 * next(t for t in _F32_TILE_COUNTS if t <= left), c_out - c0)
        if krsc:
            w = weight[..., c0:c0 + width] if transpose else weight[c0:c0 + width]
        else:
            w = weight[:, c0:c0 + width, :] if transpose else weight[:, :, c0:c0 + width]
        outs.append(K.conv_forward(x, K.pack_weight(w, transpose=transpose, krsc=krsc), nbr,
                                   n_out, width, weight_flip=weight_flip, row_order=row_order))
        c0 += width
    return torch.cat(outs, 1)


# ---- packed weight images of module parameters -------------------------------------------
# The conv kernels read weights in MFMA fragment order (split into bf16 planes): a copy made
# from the KRSC parameter.  For a module PARAMETER the copy is kept and refreshed only when
# the parameter's `_version` has moved (the optimizer's in-place update; never, for the frozen
# LiDAR encoder of the LC recipe, tools/train.py:185-219) -- and all stale copies of a device
# are refreshed TOGETHER by one launch (kernels.pack_weight_split_many) at the first conv that
# finds its own stale: 16 trained convs were 15 pack launches per LC step on the feature
# queue.  An entry belongs to ONE live tensor object (weak reference: a new Parameter that
# inherits a dead one's id and storage address misses, a dead one's entry is evicted).
# Writes through `.data` do not bump the version: SparseConvolution drops the cache when a
# state dict is loaded into it, and code that rewrites a weight through `.data` after a
# forward calls invalidate_packed_weights() itself.  Temporaries (grid_conv's KRSC copies of
# Conv2d weights) are never cached: their ids and addresses are reused.
_PACKS = {}        # id(weight) -> entry dict
_F32_PACKS = {}    # (id(weight), transpose) -> (ref, key, packed)


def _pack_entry(weight, np_, krsc, want_t):
    import weakref
    wid = id(weight)
    key = (weight.data_ptr(), tuple(weight.shape), np_, krsc, weight.device)
    e = _PACKS.get(wid)
    if e is None or e["ref"]() is not weight or e["key"] != key:
        if krsc:
            cout, cin = weight.shape[0], weight.shape[-1]
            kvol = weight.numel() // (cout * cin)
        else:
            kvol, cin, cout = weight.shape
        e = dict(ref=weakref.ref(weight, lambda _r, wid=wid: _PACKS.pop(wid, None)), key=key,
                 dims=(kvol, cin, cout), version=None, packed_t=None,
                 packed=torch.empty((K.lib.msmd_spconv_packed_split_bytes(kvol, cin, cout, np_),),
                                    dtype=torch.uint8, device=weight.device))
        _PACKS[wid] = e
    if want_t and e["packed_t"] is None:
        kvol, cin, cout = e["dims"]
        e["packed_t"] = torch.empty((K.lib.msmd_spconv_packed_split_bytes(kvol, cout, cin, np_),),
                                    dtype=torch.uint8, device=weight.device)
        e["version"] = None
    return e


def _packed_parameter(weight, np_, krsc, want_t):
    """(packed, packed W^T | None) of a module parameter, refreshed if stale -- together with
    every other stale parameter of its device and plane count."""
    e = _pack_entry(weight, np_, krsc, want_t)
    if e["version"] != weight._version:
        jobs, entries = [], []
        for o in list(_PACKS.values()):
            w = o["ref"]()
            if w is None or o["key"][2] != np_ or o["key"][4] != weight.device:
                continue
            if o["version"] != w._version:
                jobs.append((w, o["key"][3], o["packed"], o["packed_t"]))
                entries.append((o, w._version))
        K.pack_weight_split_many(jobs, np_)
        for o, v in entries:
            o["version"] = v
    return e["packed"], e["packed_t"]


def _pack_f32(weight, transpose, krsc):
    """kernels.pack_weight (fp32 fragment order, the narrow layers) -- kept for a module
    parameter until its version moves."""
    if not isinstance(weight, torch.nn.Parameter):
        return K.pack_weight(weight, transpose=transpose, krsc=krsc)
    import weakref
    ck = (id(weight), bool(transpose))
    key = (weight.data_ptr(), weight._version, krsc, tuple(weight.shape))
    hit = _F32_PACKS.get(ck)
    if hit is not None and hit[0]() is weight and hit[1] == key:
        return hit[2]
    packed = K.pack_weight(weight, transpose=transpose, krsc=krsc)
    _F32_PACKS[ck] = (weakref.ref(weight, lambda _r, ck=ck: _F32_PACKS.pop(ck, None)), key, packed)
    return packed


def invalidate_packed_weights(weight=None):
    """Mark the cached packed images stale -- all of them, or those of one parameter: the
    next conv that uses a weight repacks it (together with every other stale one) INTO THE
    BUFFERS IT ALREADY HAS.  Writes through `.data` (mmcv's Fp16OptimizerHook.
    copy_params_to_fp16, which runs every iteration; EMAHook's parameter swap; `param.data.
    copy_` in custom loops) do not move `_version`: hooks that write that way must call this
    after the write.  SparseConvolution calls it from its state-dict load hook and from
    train() / eval() (EMA swaps sit at those boundaries -- a per-epoch or per-iteration
    toggle costs one pack launch, no allocation); distributed.TrainStep calls it when its
    optimizer is not one of torch's in-place ones."""
    if weight is None:
        for e in _PACKS.values():
            e["version"] = None
        _F32_PACKS.clear()          # (27 KB images of the 5- / 16-channel layers)
        return
    e = _PACKS.get(id(weight))
    if e is not None:
        e["version"] = None
    for t in (False, True):
        _F32_PACKS.pop((id(weight), t), None)


def _conv_forward(features, weight, rb, krsc, want_dgrad, bn_stats=False):
    """-> (out, packed W^T for dgrad | None[, BN partials | None when bn_stats])."""
    c_in, c_out = (weight.shape[-1], weight.shape[0]) if krsc else weight.shape[1:]
    rb.check_ready()
    if _use_split(c_in, c_out, rb.nbr_fwd.shape[0], features.shape[0]):
        np_ = conv_planes()
        packed_t = None
        # dgrad will want W^T packed too
        want_t = want_dgrad and _use_split(c_out, c_in, rb.nbr_fwd.shape[0], rb.n_out)
        if isinstance(weight, torch.nn.Parameter):
            packed, packed_t = _packed_parameter(weight, np_, krsc, want_t)
            packed_t = packed_t if want_t else None
        elif want_t:      # a temporary (not cached): both images in one launch
            packed, packed_t = K.pack_weight_split_pair(weight, np_, krsc=krsc)
        else:
            packed = K.pack_weight_split(weight, np_, krsc=krsc)
        table, order = rb.tiling_fwd()
        res = K.conv_forward_split(features, packed, table, rb.n_out, c_out, np_,
                                   row_order=order, tile_prefix=rb.prefix_fwd(c_out),
                                   bn_stats=bn_stats)
        return (res[0], packed_t, res[1]) if bn_stats else (res, packed_t)
    out = _conv_f32(features, weight, krsc, False, rb.nbr_fwd, rb.n_out,
                    row_order=rb.order_fwd() if _wants_order(c_in, c_out) else None)
    return (out, None, None) if bn_stats else (out, None)


class _SparseConvFunction(Function):
    """indice_conv / indice_subm_conv / implicit_gemm in one: forward and dgrad
    are the same implicit-GEMM kernel, wgrad contracts over the pair lists."""

    @staticmethod
    def forward(ctx, features, weight, rb, krsc, stats=None):
        ctx.rb, ctx.krsc = rb, krsc
        ctx.save_for_backward(features, weight)
        if stats is not None:     # a one-slot list: receives the BN partials of the output
            out, ctx.packed_t, stats[0] = _conv_forward(features, weight, rb, krsc,
                                                        ctx.needs_input_grad[0], bn_stats=True)
        else:
            out, ctx.packed_t = _conv_forward(features, weight, rb, krsc, ctx.needs_input_grad[0])
        return out

    @staticmethod
    def backward(ctx, grad_out):
        features, weight = ctx.saved_tensors
        rb, krsc = ctx.rb, ctx.krsc
        c_in, c_out = (weight.shape[-1], weight.shape[0]) if krsc else weight.shape[1:]
        grad_out = grad_out.contiguous()
        d_feat = d_w = None
        if ctx.needs_input_grad[1]:
            # (rounds 1-2 could put wgrad on a side stream, MSMD_WGRAD_STREAM=1: its thousands
            # of short workgroups filled the CUs the persistent dgrad kernel left idle in its
            # tail, +2-3 %.  The whole-block kernel is one 144 KB workgroup per CU: next to
            # dgrad it measured -1 % on the LC path and -3 % on configs[1]; removed.)
            pairs, num = rb.pairs()     # (cached; built on this stream if not yet)
            if conv_planes() in (1, 2, 3) and K.wgrad_split_supported(c_in, c_out):
                d_w = K.conv_wgrad_split(features, grad_out, pairs, num, conv_planes(),
                                         krsc_shape=weight.shape if krsc else None,
                                         segments=rb.pair_segments())
            else:
                d_w = K.conv_wgrad(features, grad_out, pairs, num,
                                   krsc_shape=weight.shape if krsc else None)
        if ctx.needs_input_grad[0] and _use_split(c_out, c_in, rb.nbr_fwd.shape[0],
                                                  grad_out.shape[0]):
            np_ = conv_planes()
            packed_t = ctx.packed_t if ctx.packed_t is not None else \
                K.pack_weight_split(weight, np_, transpose=True, krsc=krsc)
            table, order = rb.tiling_bwd()
            d_feat = K.conv_forward_split(grad_out, packed_t, table, rb.n_in, c_in, np_,
                                          weight_flip=rb.is_subm, row_order=order,
                                          tile_prefix=rb.prefix_bwd(c_in))
        elif ctx.needs_input_grad[0]:
            order = rb.order_bwd() if _wants_order(c_out, c_in) else None
            # SubM: forward table + flipped weights == backward table
            d_feat = _conv_f32(grad_out, weight, krsc, True,
                               rb.nbr_fwd if rb.is_subm else rb.nbr_bwd, rb.n_in,
                               weight_flip=rb.is_subm, row_order=order)
        return d_feat, d_w, None, None, None


def sparse_conv(features, weight, rb, krsc=False, bn_stats=None):
    """weight: [K,Cin,Cout], or the KRSC module parameter with krsc=True (read
    and differentiated in place -- no permute/contiguous copies per step).
    bn_stats: a one-slot list that receives the output's BatchNorm partials (per-tile column
    sums / sums of squares, or None where the kernel in use does not produce them) for
    bn_act(..., stats=...)."""
    if not (torch.is_grad_enabled() and (features.requires_grad or weight.requires_grad)):
        if bn_stats is not None:
            out, _, bn_stats[0] = _conv_forward(features, weight, rb, krsc, False, bn_stats=True)
            return out
        return _conv_forward(features, weight, rb, krsc, False)[0]   # nothing to record
    return _SparseConvFunction.apply(features, weight, rb, krsc, bn_stats)


class _BNActFunction(Function):
    """BatchNorm1d (+ residual) (+ ReLU) in two fused passes each way
    (csrc/bn.hip) instead of torch's 3-5 launches per conv."""

    @staticmethod
    def forward(ctx, x, residual, gamma, beta, running_mean, running_var, training, momentum,
                eps, relu, stats=None):
        y, mean, invstd = K.bn_act_forward(x, residual, gamma, beta, running_mean, running_var,
                                           training, momentum, eps, relu, partials=stats)
        ctx.save_for_backward(x, y, gamma, beta, mean, invstd)
        ctx.cfg = (bool(training), bool(relu), residual is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, beta, mean, invstd = ctx.saved_tensors
        training, relu, has_res = ctx.cfg
        if relu and not has_res and gamma.dtype == torch.float32 and beta.dtype == torch.float32:
            # BN + ReLU: the mask comes from x (bit-identical decision), y is not read
            dx, dgamma, dbeta = K.bn_relu_backward(x, dy, gamma, beta, mean, invstd, training)
            return dx, None, dgamma, dbeta, None, None, None, None, None, None, None
        dx, dres, dgamma, dbeta = K.bn_act_backward(x, y, dy, gamma, mean, invstd, training, relu,
                                                    has_res and ctx.needs_input_grad[1])
        return dx, dres, dgamma, dbeta, None, None, None, None, None, None, None


# num_batches_tracked: every training BatchNorm adds 1 per forward -- one 5 us launch each,
# 16 per LC step, for a counter nothing reads while `momentum` is set (all reference configs:
# momentum=0.01).  Inside `deferred_batch_counters()` (distributed.TrainStep wraps the step in
# it) the increments are collected on the host and applied by ONE multi-tensor launch when the
# context closes; outside it -- plain module calls, tests, evaluation -- each forward updates
# its counter at once, as torch does.
_DEFERRED_COUNTS = None      # None: immediate; dict id -> [tensor, pending] while deferring


def count_batch(counter):
    if _DEFERRED_COUNTS is None:
        counter.add_(1)
        return
    e = _DEFERRED_COUNTS.get(id(counter))
    if e is None:
        _DEFERRED_COUNTS[id(counter)] = [counter, 1]
    else:
        e[1] += 1


def flush_batch_counters():
    """Apply the increments collected so far (one launch per device and dtype)."""
    if not _DEFERRED_COUNTS:
        return
    entries = list(_DEFERRED_COUNTS.values())
    _DEFERRED_COUNTS.clear()
    groups = {}
    for t, n in entries:
        groups.setdefault((t.device, t.dtype), []).append((t, n))
    for items in groups.values():
        tensors = [t for t, _ in items]
        if len({n for _, n in items}) == 1:
            torch._foreach_add_(tensors, items[0][1])
        else:
            for t, n in items:
                t.add_(n)


class deferred_batch_counters:
    """Context: BatchNorm batch counters are applied together at exit (see above)."""

    def __enter__(self):
        global _DEFERRED_COUNTS
        self.outer = _DEFERRED_COUNTS
        if _DEFERRED_COUNTS is None:
            _DEFERRED_COUNTS = {}
        return self

    def __exit__(self, *exc):
        global _DEFERRED_COUNTS
        if self.outer is None:
            try:
                flush_batch_counters()
            finally:
                _DEFERRED_COUNTS = None
        return False


def bn_act(x, bn, relu=False, residual=None, stats=None):
    """Apply an nn.BatchNorm1d module (its parameters / buffers / mode) fused
    with an optional residual add and ReLU.  Same semantics as
    relu(bn(x) + residual), including the running-stat update.
    stats: x's per-block column sums / sums of squares when the kernel that produced x left
    them (sparse_conv(bn_stats=[None]) -> the list's entry): the statistics pass over x is
    skipped in training mode."""
    if x.shape[0] == 0:     # SparseSequential skips dense modules on empty tensors
        return x
    fusable = (x.is_cuda and x.dtype == torch.float32 and x.shape[1] % 4 == 0 and bn.affine
               and bn.momentum is not None)
    if not fusable:
        y = bn(x)
        if residual is not None:
            y = y + residual
        return torch.relu(y) if relu else y
    # nn.modules.batchnorm._BatchNorm.forward: batch statistics when training or
    # when there are no buffers; the buffers are handed to the kernel (and, in
    # training, updated) only when `not training or track_running_stats`
    # (tools/train.py:205-211 flips track_running_stats on frozen layers).
    use_batch_stats = bn.training or bn.running_mean is None
    pass_running = (not bn.training) or bn.track_running_stats
    rm = bn.running_mean if pass_running else None
    rv = bn.running_var if pass_running else None
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        count_batch(bn.num_batches_tracked)
    # (any of the four: a BatchNorm with frozen weight but trainable bias still needs its
    # bias gradient, as torch's own BatchNorm produces)
    if not (torch.is_grad_enabled() and (x.requires_grad or bn.weight.requires_grad
                                         or bn.bias.requires_grad
                                         or (residual is not None and residual.requires_grad))):
        return K.bn_act_forward(x, residual, bn.weight, bn.bias, rm, rv, use_batch_stats,
                                bn.momentum, bn.eps, relu, partials=stats)[0]
    return _BNActFunction.apply(x, residual, bn.weight, bn.bias, rm, rv, use_batch_stats,
                                bn.momentum, bn.eps, relu, stats)


class _DenseFunction(Function):
    @staticmethod
    def forward(ctx, features, indices, batch_size, spatial_shape):
        ctx.save_for_backward(indices)
        ctx.shape = list(spatial_shape)
        return K.dense_scatter(features, indices, batch_size, spatial_shape)

    @staticmethod
    def backward(ctx, grad):
        (indices,) = ctx.saved_tensors
        return K.dense_gather(grad.contiguous(), indices, ctx.shape), None, None, None


def dense(features, indices, batch_size, spatial_shape):
    return _DenseFunction.apply(features, indices, batch_size, spatial_shape)


class _BevConcatFunction(Function):
    """N sparse tensors -> one channels-last [B,H,W,sum(C_i*D_i)] buffer."""

    @staticmethod
    def forward(ctx, batch_size, metas, *features):
        # metas: per tensor (indices, spatial_shape)
        h, w = metas[0][1][1], metas[0][1][2]
        widths = [f.shape[1] * m[1][0] for f, m in zip(features, metas)]
        bev = torch.zeros((int(batch_size), h, w, sum(widths)), dtype=torch.float32,
                          device=features[0].device)
        off = 0
        for f, (idx, shape), width in zip(features, metas, widths):
            if shape[1] != h or shape[2] != w:
                raise ValueError("bev_concat: BEV sizes differ: %s vs %s" % (shape, metas[0][1]))
            K.bev_scatter_nhwc(f, idx, batch_size, shape, bev, off)
            off += width
        ctx.metas, ctx.batch_size = metas, int(batch_size)
        ctx.channels = [f.shape[1] for f in features]
        return bev.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, grad):
        g = grad.permute(0, 2, 3, 1).contiguous().float()
        out, off = [], 0
        for i, ((idx, shape), c) in enumerate(zip(ctx.metas, ctx.channels)):
            out.append(K.bev_gather_nhwc(g, idx, c, ctx.batch_size, shape, off)
                       if ctx.needs_input_grad[2 + i] else None)
            off += c * shape[0]
        return (None, None, *out)


def bev_concat(tensors):
    """torch.cat([t.dense().view(N, C*D, H, W) for t in tensors], 1)
    (MSMDFusion.py:436-440 with sparse_encoder.py:187-190) without the dense
    intermediates: returns the [B, sum(C*D), H, W] map as a channels-last view."""
    metas = []
    for t in tensors:
        assert t.indices.shape[1] == 4, "bev_concat needs (b,z,y,x) indices"
        metas.append((t.indices, [int(v) for v in t.spatial_shape]))
    return _BevConcatFunction.apply(tensors[0].batch_size, metas, *[t.features for t in tensors])


class _SparseAddFunction(Function):
    @staticmethod
    def forward(ctx, fa, ia, fb, ib, batch_size, spatial_shape):
        oi, of, ma, mb = K.sparse_add(fa, ia, fb, ib, batch_size, spatial_shape)
        ctx.save_for_backward(ma, mb)
        ctx.mark_non_differentiable(oi)
        return of, oi

    @staticmethod
    def backward(ctx, g_feat, _g_idx):
        ma, mb = ctx.saved_tensors
        g = g_feat.contiguous()
        return g.index_select(0, ma.long()), None, g.index_select(0, mb.long()), None, None, None


def sparse_add(*tensors):
    """spconv.pytorch.functional.sparse_add (call site
    sparse_multimodal_encoder_painting.py:455): union of the operands' voxel
    sets in ascending linear id, features summed where they coincide."""
    from .core import SparseConvTensor
    assert len(tensors) >= 2
    acc = tensors[0]
    for other in tensors[1:]:
        assert acc.spatial_shape == other.spatial_shape, "sparse_add needs equal spatial_shape"
        assert acc.batch_size == other.batch_size
        feat, idx = _SparseAddFunction.apply(acc.features, acc.indices, other.features,
                                             other.indices, acc.batch_size, acc.spatial_shape)
        acc = SparseConvTensor(feat, idx, acc.spatial_shape, acc.batch_size)
    return acc


class _SparseAddRowsFunction(Function):
    """Feature half of sparse_add over the maps of an index pass done earlier."""

    @staticmethod
    def forward(ctx, fa, fb, plan):
        ctx.plan = plan
        n_out = plan["sum"].indices.shape[0]
        if "inv_a" in plan and fa.shape[1] % 4 == 0:       # gather: no fill, no float atomics
            return K.sparse_add_rows_gather(fa, plan["ma"], plan["inv_a"], fb, plan["mb"],
                                            plan["inv_b"], n_out)
        return K.sparse_add_rows(fa, plan["ma"], fb, plan["mb"], n_out)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        return g.index_select(0, ctx.plan["ma_l"]), g.index_select(0, ctx.plan["mb_l"]), None


def plan_sparse_add(a_indices, b_indices, batch_size, spatial_shape):
    """Index half of sparse_add(a, b) from the two coordinate sets alone: the
    union (an index-only SparseConvTensor whose rulebooks can be planned) and the
    row maps the feature half and its backward need."""
    from .core import SparseConvTensor
    oi, ma, mb = K.sparse_add_index(a_indices, b_indices, batch_size, spatial_shape)
    total = SparseConvTensor(torch.empty((oi.shape[0], 0), dtype=torch.float32,
                                         device=oi.device), oi, spatial_shape, batch_size)
    return add_plan(total, ma, mb)


def add_plan(total, ma, mb):
    """What sparse_add_planned needs besides the union tensor: the row maps (the kernels'
    and the backward's index_select) and their inverses (the gather)."""
    n_out = total.indices.shape[0]
    # (index_select takes int32 indices as they are: no int64 copies of the maps)
    plan = dict(sum=total, ma=ma, mb=mb, ma_l=ma, mb_l=mb)
    if ma.is_cuda and os.environ.get("MSMD_ADD_GATHER", "1") == "1":
        plan["inv_a"], plan["inv_b"] = K.rows_inverse(ma, n_out), K.rows_inverse(mb, n_out)
    return plan


def sparse_add_planned(a, b, plan):
    """sparse_add(a, b) with plan = plan_sparse_add(a.indices, b.indices, ...):
    same result, no host read; the output shares the planned tensor's rulebooks."""
    assert a.features.shape[0] == plan["ma"].shape[0] and b.features.shape[0] == plan["mb"].shape[0]
    return plan["sum"].replace_feature(_SparseAddRowsFunction.apply(a.features, b.features, plan))
