"""Host logic of spconv.plan_batch (no GPU): which context collects the requests at each
MSMD_PLAN_SCOPE level, that nested contexts hand their requests to the outermost active one,
that one request per (rulebook, side) accumulates what several convs ask for, and that a
context left by an exception poisons the SubM tables it never filled."""
import pytest
import torch

from msmdfusion_amd.spconv import core


class _FakeRb:
    def __init__(self, n=10, kvol=27, subm=True):
        self.nbr_fwd = torch.zeros((kvol, n), dtype=torch.int32)
        self.nbr_bwd = None if subm else torch.zeros((kvol, n), dtype=torch.int32)
        self.indices = torch.zeros((n, 4), dtype=torch.int32)
        self.spatial_shape, self.ksize = [4, 4, 4], [3, 3, 3]
        self.is_subm = subm
        self.n_in = self.n_out = n
        self._pairs = None
        self._pair_segments = False
        self._order_fwd = self._order_bwd = self._tiled_fwd = self._tiled_bwd = None
        self._prefix_fwd, self._prefix_bwd = {}, {}
        self.planned = []

    def _planned(self, side, res, keep_order):
        self.planned.append((side, res, keep_order))

    def pair_segments(self):      # (the chunked-table fall-back after a launch set)
        self._pair_segments = ("table", 1)
        return self._pair_segments


@pytest.fixture
def calls(monkeypatch):
    log = dict(plan=[], subm=[])

    def plan_many(jobs):
        log["plan"].append(jobs)
        return [dict(order="o", tiled=None, prefix={}, pairs=None, segments=None) for _ in jobs]

    def subm_many(jobs):
        log["subm"].append(jobs)
    monkeypatch.setattr(core.K, "rulebook_plan_many", plan_many)
    monkeypatch.setattr(core.K, "rulebook_subm_many", subm_many)
    monkeypatch.setattr(core, "PLAN_BATCHING", True)
    return log


@pytest.mark.parametrize("scope,flushes", [("call", 3), ("stage", 2), ("all", 1)])
def test_scope_levels_decide_who_flushes(calls, monkeypatch, scope, flushes):
    """all > stage > call: a context is active when its level is within MSMD_PLAN_SCOPE, and
    an active context nested in another active one hands everything to the outer one."""
    monkeypatch.setattr(core, "PLAN_SCOPE", scope)
    rbs = [_FakeRb() for _ in range(3)]
    with core.plan_batch("all"):
        with core.plan_batch("stage"):
            with core.plan_batch("call") as b:
                (getattr(core._PLAN, "batch", None) or b).job(rbs[0], "fwd")["want_pairs"] = True
            with core.plan_batch("call") as b:
                (getattr(core._PLAN, "batch", None) or b).job(rbs[1], "fwd")["want_order"] = True
        with core.plan_batch("stage"):
            with core.plan_batch("call") as b:
                (getattr(core._PLAN, "batch", None) or b).job(rbs[2], "fwd")["want_pairs"] = True
    assert getattr(core._PLAN, "batch", None) is None
    assert len(calls["plan"]) == flushes
    assert sum(len(j) for j in calls["plan"]) == 3
    assert all(len(rb.planned) == 1 for rb in rbs)


def test_requests_of_one_table_accumulate(calls, monkeypatch):
    monkeypatch.setattr(core, "PLAN_SCOPE", "all")
    rb = _FakeRb()
    with core.plan_batch("all") as b:
        j = b.job(rb, "fwd")
        j["tile_rows"].add(128)
        j["want_pairs"] = True
        j2 = b.job(rb, "fwd")            # a second conv on the same table
        assert j2 is j
        j2["tile_rows"].add(256)
        j2["want_segments"] = True
    (jobs,) = calls["plan"]
    assert len(jobs) == 1
    assert jobs[0]["tile_rows"] == {128, 256} and jobs[0]["want_table"]
    assert jobs[0]["want_pairs"] and jobs[0]["want_segments"]
    assert rb.planned[0][0] == "fwd" and rb.planned[0][2] is False


def test_inactive_batching_is_inert(calls, monkeypatch):
    monkeypatch.setattr(core, "PLAN_BATCHING", False)
    with core.plan_batch("all"):
        assert getattr(core._PLAN, "batch", None) is None
    assert calls["plan"] == [] and calls["subm"] == []


def test_exception_poisons_unfilled_subm_tables(calls, monkeypatch):
    monkeypatch.setattr(core, "PLAN_SCOPE", "all")
    rb = _FakeRb()
    with pytest.raises(RuntimeError, match="boom"):
        with core.plan_batch("all") as b:
            b.subm(rb, 1)
            raise RuntimeError("boom")
    assert rb.nbr_fwd is None and calls["subm"] == []
    assert getattr(core._PLAN, "batch", None) is None
    # ... and a launch set that fails does the same
    rb2 = _FakeRb()

    def failing(jobs):
        raise RuntimeError("launch failed")
    monkeypatch.setattr(core.K, "rulebook_subm_many", failing)
    with pytest.raises(RuntimeError, match="launch failed"):
        with core.plan_batch("all") as b:
            b.subm(rb2, 1)
    assert rb2.nbr_fwd is None


def test_pending_flag_guards_a_deferred_table(calls, monkeypatch):
    """A SubM rulebook built inside plan_batch() has an allocated but unfilled table until the
    context closes: check_ready() refuses it meanwhile and accepts it afterwards; a rulebook
    whose fill failed is refused for good and rebuilt by cached_rulebook."""
    monkeypatch.setattr(core, "PLAN_SCOPE", "all")
    idx = torch.zeros((10, 4), dtype=torch.int32)
    rb = core.IndiceData(idx, idx, torch.zeros((27, 10), dtype=torch.int32), None, True,
                         [4, 4, 4], [4, 4, 4], [3, 3, 3], [1, 1, 1], [1, 1, 1], [1, 1, 1])
    with core.plan_batch("all") as b:
        rb.pending = True
        b.subm(rb, 1)
        with pytest.raises(RuntimeError, match="plan_batch"):
            rb.check_ready()
    assert rb.pending is False
    rb.check_ready()
    # a failed context: poisoned, and the shared cache does not hand it out again
    rb2 = core.IndiceData(idx, idx, torch.zeros((27, 10), dtype=torch.int32), None, True,
                          [4, 4, 4], [4, 4, 4], [3, 3, 3], [1, 1, 1], [1, 1, 1], [1, 1, 1])
    with pytest.raises(RuntimeError, match="boom"):
        with core.plan_batch("all") as b:
            rb2.pending = True
            b.subm(rb2, 1)
            raise RuntimeError("boom")
    with pytest.raises(RuntimeError, match="never filled"):
        rb2.check_ready()
    t = core.SparseConvTensor(torch.zeros((10, 4)), idx, [4, 4, 4], 1)
    ident = (idx.data_ptr(), 10, (4, 4, 4), (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), True)
    t._rb_cache[ident] = rb2
    fresh = object()
    monkeypatch.setattr(core, "build_rulebook", lambda *a, **k: fresh)
    assert t.cached_rulebook([3, 3, 3], [1, 1, 1], [1, 1, 1], [1, 1, 1], True) is fresh


def test_subm_index_choice_counts_the_brick_padding():
    """The bitmap SubM index works on the grid padded to 4 x 8 x 8 bricks: the choice between
    it and the hash index uses the padded size, and grids of 2^24 bricks and more (which the
    entry point refuses) always take the hash index."""
    from msmdfusion_amd import kernels as K
    n = 200000
    assert K.subm_index_method(n, 2, [41, 1440, 1440], "auto") == "bitmap"
    # a thin grid pads 4x along z: 1 x 8192 x 8192 = 2^26 cells = 2^20 bricks of 256 cells
    thin = K.subm_index_method(n, 1, [1, 8192, 8192], "auto")
    words_padded = 1 * 1 * 1024 * 1024 * 8
    assert thin == ("bitmap" if words_padded <= K._SUBM_BITMAP_WORDS_PER_VOXEL * n else "hash")
    # 2^24 bricks: never the bitmap, however many voxels
    assert K.subm_index_method(10 ** 9, 16, [4, 8192, 8192], "auto") == "hash"
    assert K.subm_index_method(n, 2, [41, 1440, 1440], "hash") == "hash"
