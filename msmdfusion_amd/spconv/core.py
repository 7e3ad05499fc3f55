"""SparseConvTensor and rulebook cache.

Mirrors the spconv-2.x type the reference call sites construct and mutate
(legacy twin mmdet3d/ops/spconv/structure.py:21-69):
SparseConvTensor(features, indices, spatial_shape, batch_size) with
.features/.indices/.spatial_shape/.batch_size/.indice_dict, .replace_feature(),
.dense(), .find_indice_pair(), assignable .indices (MSMDFusion.py:322-323).
"""
from typing import List, Optional

import os
import threading

import numpy as np
import torch

from .. import kernels as K


_PLAN = threading.local()
# MSMD_PLAN_BATCH=0: every table planned at once by its own prepare() (A/B, tests)
PLAN_BATCHING = os.environ.get("MSMD_PLAN_BATCH", "1") != "0"
# MSMD_SUBM_BATCH=0: SubM tables built at once even inside a plan_batch (plans stay batched)
SUBM_BATCHING = os.environ.get("MSMD_SUBM_BATCH", "1") != "0"
PLAN_SCOPE = os.environ.get("MSMD_PLAN_SCOPE", "call")


class plan_batch:
    """`with plan_batch():` -- IndiceData.prepare() calls inside (SparseConvTensor.plan, the
    fusion stack's stage planning) record what each table's kernels will need; the exit of
    the OUTERMOST context computes all of it in one K.rulebook_plan_many call on the current
    stream.  Nothing in an index pass reads a plan (the feature pass does), so the ~21 tables
    of an LC step are planned together at the end of fusion.prepare(): 8 launches instead of
    ~280, 3 ms less on the index queue.  Per thread (the prefetch worker and
    the step thread plan independently); results are those of the table-by-table calls."""

    LEVELS = {"call": 0, "stage": 1, "all": 2}

    def __init__(self, level="call"):
        """level: how much of an index pass this context spans -- "call" (one
        SparseConvTensor.plan), "stage" (an encoder / one fusion stage), "all" (a whole
        prepare()).  MSMD_PLAN_SCOPE names the widest level that batches (default call: on a
        GPU-bound step a launch set spanning more displaces the conv kernels)."""
        self.jobs = {}
        self.subm_jobs = []
        self.outer = None
        self.level = self.LEVELS[level]

    def __enter__(self):
        self.outer = getattr(_PLAN, "batch", None)
        self.active = PLAN_BATCHING and self.level <= self.LEVELS.get(PLAN_SCOPE, 0)
        if self.outer is None and self.active:
            _PLAN.batch = self
        return self

    def __exit__(self, exc_type, exc, tb):
        if self.outer is None and self.active:
            _PLAN.batch = None
            if exc_type is None:
                self.flush()
            else:       # tables never filled: a rulebook that survives the error must not be used
                for j in self.subm_jobs:        # (cached_rulebook rebuilds a poisoned entry)
                    j["rb"].nbr_fwd = None
                    j["rb"].pending = False
                self.subm_jobs, self.jobs = [], {}
        return False

    def job(self, rb, side):
        if self.outer is not None:
            return self.outer.job(rb, side)
        key = (id(rb), side)
        j = self.jobs.get(key)
        if j is None:
            j = self.jobs[key] = dict(rb=rb, side=side, tile_rows=set(), want_order=False,
                                      want_pairs=False, want_segments=False)
        return j

    def subm(self, rb, batch_size):
        """A SubM rulebook whose table is to be filled at the exit (build_rulebook)."""
        if self.outer is not None:
            return self.outer.subm(rb, batch_size)
        self.subm_jobs.append(dict(rb=rb, indices=rb.indices, batch_size=batch_size,
                                   spatial_shape=rb.spatial_shape, ksize=rb.ksize,
                                   nbr=rb.nbr_fwd))

    def flush(self):
        subm, self.subm_jobs = self.subm_jobs, []
        try:
            K.rulebook_subm_many(subm)          # the tables first: the plans read them
        except Exception:
            for j in subm:                      # never filled: must not be used
                j["rb"].nbr_fwd = None
                j["rb"].pending = False
            self.jobs = {}
            raise
        for j in subm:
            j["rb"].pending = False
        jobs, todo = list(self.jobs.values()), []
        self.jobs = {}
        for j in jobs:
            rb, fwd = j["rb"], j["side"] == "fwd"
            have = rb._prefix_fwd if fwd else rb._prefix_bwd
            tiled = rb._tiled_fwd if fwd else rb._tiled_bwd
            order = rb._order_fwd if fwd else rb._order_bwd
            rows = {r for r in j["tile_rows"] if r not in have}
            want_table = bool(j["tile_rows"]) and tiled is None
            if rows and tiled is not None:          # a further height of a table already tiled
                for r in rows:
                    have[r] = K.tile_prefix(tiled[0], r)
                rows = set()
            want_pairs = fwd and j["want_pairs"] and rb._pairs is None
            want_seg = fwd and j["want_segments"] and rb._pair_segments is False
            if want_seg and rb._pairs is not None:
                rb.pair_segments()
                want_seg = False
            want_order = j["want_order"] and order is None
            if not (rows or want_table or want_pairs or want_seg or want_order):
                continue
            todo.append((rb, j["side"], dict(nbr=rb.nbr_fwd if fwd else rb.nbr_bwd, tile_rows=rows,
                                             want_order=want_order,
                                             want_table=want_table, want_pairs=want_pairs,
                                             want_segments=want_seg,
                                             ld=max(rb.n_in, rb.n_out, 1))))
        for (rb, side, job), res in zip(todo, K.rulebook_plan_many([t[2] for t in todo])):
            rb._planned(side, res, job["want_order"])
            if job["want_segments"] and res["segments"] is None:
                rb.pair_segments()              # chunked segment tables (MSMD_WGRAD_CHUNK_ROWS)


class IndiceData:
    """One rulebook (spconv-2.x ImplicitGemmIndiceData's role,
    bug_fix/conv.py:416-436): the output-stationary neighbour tables plus,
    built lazily for the weight gradient, the reference-format pair lists."""

    def __init__(self, out_indices, indices, nbr_fwd, nbr_bwd, is_subm, spatial_shape,
                 out_spatial_shape, ksize, stride, padding, dilation, algo=None):
        self.out_indices = out_indices
        self.indices = indices
        self.nbr_fwd = nbr_fwd          # [K, n_out]
        self.nbr_bwd = nbr_bwd          # [K, n_in]; None for SubM (fwd table, flipped)
        self.is_subm = is_subm
        self.spatial_shape = spatial_shape
        self.out_spatial_shape = out_spatial_shape
        self.ksize, self.stride, self.padding, self.dilation = ksize, stride, padding, dilation
        self.algo = algo
        # True between build_rulebook() inside a plan_batch() and that context's exit: the
        # SubM table is allocated but not filled yet -- nothing may read it (table() checks)
        self.pending = False
        self._pairs = None
        self._pair_segments = False     # not computed yet (None = chunking off)
        self._order_fwd = None
        self._order_bwd = None
        self._tiled_fwd = None
        self._tiled_bwd = None
        self._prefix_fwd = {}
        self._prefix_bwd = {}

    def check_ready(self):
        """Raise if the table cannot be read yet / any more (deferred fill still pending, or
        the launch set that should have filled it failed)."""
        if self.pending:
            raise RuntimeError("this SubM rulebook's table is filled when the enclosing "
                               "spconv.plan_batch() closes; it cannot be used inside it")
        if self.nbr_fwd is None:
            raise RuntimeError("this rulebook's table was never filled (its plan_batch() "
                               "failed); build it again")

    @property
    def n_in(self):
        return self.indices.shape[0]

    @property
    def n_out(self):
        return self.out_indices.shape[0]

    def pairs(self):
        """(indice_pairs[K,2,ld], indice_num[K]) -- spconv_ops.h:55-59 format."""
        if self._pairs is None:
            self._pairs = K.rulebook_pairs(self.nbr_fwd, ld=max(self.n_in, self.n_out, 1))
        return self._pairs

    def pair_segments(self):
        """Row-chunk segment table of the pair lists for the whole-block wgrad kernel
        (kernels.pair_segments), or None when chunking is off."""
        if self._pair_segments is False:
            self._pair_segments = K.pair_segments(*self.pairs())
        return self._pair_segments

    def prepare(self, need_grad, c_in=None, c_out=None):
        """Compute everything derived from the table now -- the pair lists when a
        weight gradient will be needed and, for a conv of c_in -> c_out channels,
        the tiling order / tile-ordered table its kernels will ask for -- so that
        the feature pass enqueues no index work and never waits on the host.
        Inside `plan_batch()` the work is only recorded; the batch's exit runs it for
        all tables together (K.rulebook_plan_many: one launch set)."""
        batch = getattr(_PLAN, "batch", None)
        if c_in is not None:
            from .functional import _use_split, _wants_order
            kvol = self.nbr_fwd.shape[0]
            # one library call for what the split kernels want from each table (tiling
            # order, table in tile order, stream-K prefix, pair lists) instead of four:
            # the index pass is bound by host time
            fwd_split = _use_split(c_in, c_out, kvol, self.n_in)
            bwd_split = need_grad and _use_split(c_out, c_in, kvol, self.n_out)
            if batch is not None and kvol <= 31 and self.n_out > 0 and self.n_in > 0:
                wgs = need_grad and K.wgrad_split_supported(c_in, c_out)
                fwd = batch.job(self, "fwd")
                if fwd_split:
                    fwd["tile_rows"].add(K.split_tile_rows(c_out))
                elif _wants_order(c_in, c_out):
                    fwd["want_order"] = True
                fwd["want_pairs"] |= need_grad
                fwd["want_segments"] |= wgs
                if need_grad:
                    side = fwd if self.is_subm else batch.job(self, "bwd")
                    if bwd_split:
                        side["tile_rows"].add(K.split_tile_rows(c_in))
                    elif _wants_order(c_out, c_in):
                        side["want_order"] = True
                return self
            if kvol <= 31 and self.n_out > 0 and self._tiled_fwd is None and \
                    (fwd_split or (bwd_split and self.is_subm)):
                rows = {K.split_tile_rows(c_out)} if fwd_split else set()
                if bwd_split and self.is_subm:
                    rows.add(K.split_tile_rows(c_in))
                plan = K.rulebook_plan(self.nbr_fwd, rows, need_grad and self._pairs is None,
                                       ld=max(self.n_in, self.n_out, 1))
                self._order_fwd, self._tiled_fwd = (plan["order"],), (plan["tiled"],)
                self._prefix_fwd.update(plan["prefix"])
                if plan["pairs"] is not None:
                    self._pairs = plan["pairs"]
            if bwd_split and not self.is_subm and kvol <= 31 and self.n_in > 0 and \
                    self._tiled_bwd is None:
                plan = K.rulebook_plan(self.nbr_bwd, {K.split_tile_rows(c_in)})
                self._order_bwd, self._tiled_bwd = (plan["order"],), (plan["tiled"],)
                self._prefix_bwd.update(plan["prefix"])
        elif batch is not None and need_grad and self.nbr_fwd.shape[0] <= 31 and \
                self.n_out > 0 and self.n_in > 0:
            batch.job(self, "fwd")["want_pairs"] = True
            return self
        if need_grad:
            self.pairs()
            if c_in is not None and K.wgrad_split_supported(c_in, c_out):
                self.pair_segments()
        if c_in is not None:
            if _use_split(c_in, c_out, kvol, self.n_in):
                self.prefix_fwd(c_out)
            elif _wants_order(c_in, c_out):
                self.order_fwd()
            if need_grad:       # dgrad: the same kernel over the mirrored problem
                if _use_split(c_out, c_in, kvol, self.n_out):
                    self.prefix_bwd(c_in)
                elif _wants_order(c_out, c_in):
                    self.order_bwd()
        return self

    def _planned(self, side, res, keep_order):
        """Take one K.rulebook_plan_many result (plan_batch.flush)."""
        if side == "fwd":
            if res["tiled"] is not None:
                self._order_fwd, self._tiled_fwd = (res["order"],), (res["tiled"],)
            elif keep_order and self._order_fwd is None:
                self._order_fwd = (res["order"],)
            self._prefix_fwd.update(res["prefix"])
            if res["pairs"] is not None:
                self._pairs = res["pairs"]
            if res["segments"] is not None:
                self._pair_segments = res["segments"]
        else:
            if res["tiled"] is not None:
                self._order_bwd, self._tiled_bwd = (res["order"],), (res["tiled"],)
            elif keep_order and self._order_bwd is None:
                self._order_bwd = (res["order"],)
            self._prefix_bwd.update(res["prefix"])

    def order_fwd(self):
        """Tiling order of the output rows (similar neighbour masks adjacent, tiles
        heaviest first); for SubM the same order serves dgrad (its table is the
        forward one mirrored)."""
        if self._order_fwd is None:
            self._order_fwd = (K.rulebook_tiling(self.nbr_fwd, want_table=False)[0],)
        return self._order_fwd[0]

    def order_bwd(self):
        if self.is_subm:
            return self.order_fwd()
        if self._order_bwd is None:
            self._order_bwd = (K.rulebook_tiling(self.nbr_bwd, want_table=False)[0],)
        return self._order_bwd[0]

    def tiling_fwd(self):
        """(table, row_order) the split-bf16 kernel tiles the forward pass by: the
        table in mask-sorted tile order (column p belongs to output row order[p]).
        SubM rulebooks serve 8 launches per step (4 convs x forward/dgrad), a strided
        conv's tables one launch each -- but those need it most: a stride-2 output
        row has ~5 of the 27 offsets and its neighbours in linear order all have
        different ones, so a 128-row tile in natural order walks every offset
        (issued / useful work 5.4 forward, 8.1 backward on the bench workload;
        1.8 / 1.0 sorted -- tools/order_sim.py).  The sort + permute (~100 us) runs
        in the index pass, off the feature pass."""
        if self._tiled_fwd is None:
            order, table = K.rulebook_tiling(self.nbr_fwd)
            self._order_fwd, self._tiled_fwd = (order,), (table,)
        return self._tiled_fwd[0], self._order_fwd[0]

    def prefix_fwd(self, c_out):
        """Stream-K work table of the forward tiling (K.tile_prefix) for a conv with
        c_out output channels (the kernel's tile size depends on the width): with it
        every workgroup of the split kernel takes the same share of the launch."""
        rows = K.split_tile_rows(c_out)
        if rows not in self._prefix_fwd:
            self._prefix_fwd[rows] = K.tile_prefix(self.tiling_fwd()[0], rows)
        return self._prefix_fwd[rows]

    def prefix_bwd(self, c_in):
        if self.is_subm:
            return self.prefix_fwd(c_in)
        rows = K.split_tile_rows(c_in)
        if rows not in self._prefix_bwd:
            self._prefix_bwd[rows] = K.tile_prefix(self.tiling_bwd()[0], rows)
        return self._prefix_bwd[rows]

    def tiling_bwd(self):
        if self.is_subm:      # forward table + flipped weights == backward table
            return self.tiling_fwd()
        if self._tiled_bwd is None:
            order, table = K.rulebook_tiling(self.nbr_bwd)
            self._order_bwd, self._tiled_bwd = (order,), (table,)
        return self._tiled_bwd[0], self._order_bwd[0]


def build_rulebook(indices, batch_size, spatial_shape, ksize, stride, padding, dilation, subm,
                   algo=None):
    if any(d != 1 for d in dilation):
        raise NotImplementedError("only dilation 1 is built (all reference configs use it)")
    if subm:
        if any(k % 2 == 0 for k in ksize):
            raise NotImplementedError("SubM needs odd kernel sizes")
        batch = getattr(_PLAN, "batch", None)
        # inside plan_batch(): the table is filled when the context closes, together with
        # every other SubM table of the index pass (K.rulebook_subm_many).  Only tables whose
        # prepare() is deferred too (K <= 31, not empty): nothing may read one before the exit.
        defer = batch is not None and SUBM_BATCHING and indices.shape[0] > 0 and \
            K.kernel_volume(ksize) <= 31 and indices.dtype == torch.int32 and \
            indices.is_contiguous()
        nbr = K.subm_table(indices, ksize) if defer else \
            K.rulebook_subm(indices, batch_size, spatial_shape, ksize)
        rb = IndiceData(indices, indices, nbr, None, True, list(spatial_shape),
                        list(spatial_shape), ksize, [1, 1, 1], [k // 2 for k in ksize],
                        dilation, algo)
        if defer:
            rb.pending = True
            batch.subm(rb, batch_size)
        return rb
    out_idx, nbr_fwd, nbr_bwd, out_shape = K.rulebook_conv(indices, batch_size, spatial_shape,
                                                           ksize, stride, padding)
    return IndiceData(out_idx, indices, nbr_fwd, nbr_bwd, False, list(spatial_shape),
                      list(out_shape), ksize, stride, padding, dilation, algo)


class SparseConvTensor:

    def __init__(self, features: torch.Tensor, indices: torch.Tensor,
                 spatial_shape: List[int], batch_size: int, grid=None, voxel_num=None,
                 indice_dict: Optional[dict] = None, benchmark: bool = False):
        assert features.dim() == 2 and indices.dim() == 2
        assert indices.dtype == torch.int32, "indices must be int32 (b,z,y,x)"
        assert features.shape[0] == indices.shape[0], \
            f"{features.shape[0]} feature rows for {indices.shape[0]} voxels"
        self._features = features
        self.indices = indices
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)
        self.indice_dict = {} if indice_dict is None else indice_dict
        self.grid = grid
        self.voxel_num = voxel_num
        self.benchmark = benchmark
        self.benchmark_record = {}
        self._timer = None
        self.thrust_allocator = None
        # rulebooks keyed by geometry + the identity of the indices tensor:
        # a rulebook depends on nothing else, so every SubM conv over the same
        # voxel set shares one (the reference rebuilds it for each of the 16
        # indice_key=None convs of SparseEncoder).  Results are identical.
        self._rb_cache = {}

    # spconv 2.x forbids `x.features = ...` (hence replace_feature); the legacy
    # type allowed it.  Both spellings work here.
    @property
    def features(self):
        return self._features

    @features.setter
    def features(self, val):
        self._features = val
        self.__dict__.pop("bn_stats", None)

    def replace_feature(self, feature: torch.Tensor):
        assert feature.shape[0] == self.indices.shape[0], \
            f"{feature.shape[0]} feature rows for {self.indices.shape[0]} voxels"
        new = self.shadow_copy()
        new._features = feature
        return new

    def shadow_copy(self):
        """A second handle on the same data (callers go on to re-point .indices /
        .features one after the other, so no row-count check here)."""
        new = SparseConvTensor.__new__(SparseConvTensor)
        new.__dict__.update(self.__dict__)
        # (the BatchNorm partials a conv leaves on ITS output describe those features only)
        new.__dict__.pop("bn_stats", None)
        return new

    @property
    def spatial_size(self):
        return int(np.prod(self.spatial_shape))

    @property
    def sparity(self):
        return self.indices.shape[0] / np.prod(self.spatial_shape) / self.batch_size

    def find_indice_pair(self, key) -> Optional[IndiceData]:
        if key is None:
            return None
        return self.indice_dict.get(key)

    def cached_rulebook(self, ksize, stride, padding, dilation, subm):
        # after voxel_modality_split the tensors carry 5-column indices
        # (b,mix,z,y,x: MSMDFusion.py:322-323); spconv asserts on ndim there too
        assert self.indices.shape[1] == 4, \
            f"sparse conv needs (b,z,y,x) indices, got {self.indices.shape[1]} columns"
        ident = (self.indices.data_ptr(), self.indices.shape[0], tuple(self.spatial_shape),
                 tuple(ksize), tuple(stride), tuple(padding), tuple(dilation), bool(subm))
        hit = self._rb_cache.get(ident)
        # (an entry whose deferred fill failed is rebuilt, not handed out again)
        if hit is not None and hit.indices is self.indices and hit.nbr_fwd is not None:
            return hit
        rb = build_rulebook(self.indices, self.batch_size, self.spatial_shape, list(ksize),
                            list(stride), list(padding), list(dilation), subm)
        self._rb_cache[ident] = rb
        return rb

    def seed_strided_chain(self, convs):
        """The strided convs of `convs` (execution order) as ONE chain: when every conv
        between two of them keeps the voxel set (SubM), each strided conv's input set is the
        previous one's output set, and all their output sets can be counted on the device
        back to back with a single host read (kernels.rulebook_conv_chain) instead of one
        read per conv.  The rulebooks land in the shared cache under the keys plan() /
        forward() will look up; results are those of the level-by-level path."""
        strided = [c for c in convs if not c.subm and not getattr(c, "conv1x1", False)]
        if len(strided) < 2 or os.environ.get("MSMD_CONV_CHAIN", "1") != "1" or \
                any(d != 1 for c in strided for d in c.dilation):
            return
        geoms = [(list(c.kernel_size), list(c.stride), list(c.padding)) for c in strided]
        idx, shape = self.indices, list(self.spatial_shape)
        for c, (out_idx, nbr_fwd, nbr_bwd, out_shape) in zip(
                strided, K.rulebook_conv_chain(idx, self.batch_size, shape, geoms)):
            ident = (idx.data_ptr(), idx.shape[0], tuple(shape), tuple(c.kernel_size),
                     tuple(c.stride), tuple(c.padding), tuple(c.dilation), False)
            self._rb_cache[ident] = IndiceData(out_idx, idx, nbr_fwd, nbr_bwd, False, list(shape),
                                               list(out_shape), list(c.kernel_size),
                                               list(c.stride), list(c.padding), list(c.dilation),
                                               None)
            idx, shape = out_idx, list(out_shape)

    def plan(self, convs, need_grad, strided_outputs=None):
        """Index-only pre-pass: build (or fetch) the rulebook of every sparse
        conv in `convs` (execution order), following the voxel set through the
        strided ones.  Rulebooks depend on indices only, never on features, so
        the whole chain -- including its host reads of output-voxel counts --
        runs before the first feature kernel; the feature pass then finds every
        rulebook in the shared cache.  `strided_outputs` (a list) receives the
        (indices, spatial_shape) after every strided conv, in order."""
        t = self
        with plan_batch():
            for conv in convs:
                if getattr(conv, "conv1x1", False):
                    continue
                rb = t.find_indice_pair(conv.indice_key) if conv.subm else None
                if rb is None:
                    rb = t.cached_rulebook(conv.kernel_size, conv.stride, conv.padding,
                                           conv.dilation, conv.subm)
                rb.prepare(need_grad, conv.in_channels, conv.out_channels)
                if not conv.subm:
                    t = t.shadow_copy()
                    t.indices = rb.out_indices
                    t._features = t._features.new_empty((rb.out_indices.shape[0], 0))
                    t.spatial_shape = rb.out_spatial_shape
                    if strided_outputs is not None:
                        strided_outputs.append((rb.out_indices, list(rb.out_spatial_shape)))
        return t

    def dense(self, channels_first: bool = True):
        """[B,C,D,H,W] (structure.py:55-64); channels_last returns the permuted
        view of the same buffer."""
        from .functional import dense as _dense
        assert self.indices.shape[1] == 4, \
            f"dense() needs (b,z,y,x) indices, got {self.indices.shape[1]} columns"
        out = _dense(self._features, self.indices, self.batch_size, self.spatial_shape)
        if channels_first:
            return out
        nd = len(self.spatial_shape)
        return out.permute(0, *range(2, nd + 2), 1).contiguous()
