"""Index work one step ahead, on a high-priority side stream.

Everything a step derives from voxel COORDINATES alone -- hard voxelization, the
encoder's 8 rulebooks, tiling orders, pair lists -- is independent of the
weights and of the previous step, exactly like a data loader's work.  It is
latency-bound (dozens of small launches, five host reads of voxel counts).
IndexPrefetcher runs it for batch i+1 on a side stream while the main stream is
busy with batch i's feature pass; the main stream picks the result up through an
event.

Lifetime rule: a prepared batch lives in the side stream's allocator pool and is
read by main-stream kernels, so its tensors must not be freed (= handed to the
next submit) before the main stream is past its last reader.  retire() records
that point; the object is dropped only once the event has completed.
"""
import collections
from concurrent.futures import ThreadPoolExecutor

import torch


_SIDE_STREAMS = {}


def side_stream(device, priority=-1, slot=0):
    """Process-wide side streams, created once per (device, priority, slot).  torch hands
    out streams from a pool of 32 per priority, round robin: objects that each create their
    own (one prefetcher + four search streams per model) start to ALIAS after a few models
    have lived in the process, and two chains that were meant to overlap serialise on one
    queue (bench.py's later workloads ran up to 40 % slower than the same workload first)."""
    device = torch.device(device)
    key = (device.index if device.index is not None else torch.cuda.current_device(),
           int(priority), int(slot))
    s = _SIDE_STREAMS.get(key)
    if s is None:
        cus = cu_partition()
        if cus is not None and cus[0 if slot < 8 else 1] > 0:
            # experiment (MSMD_CU_PARTITION): the index / search queues
            # on CUs of their own -- [0, index CUs) for the prefetcher's slots, the next
            # `search CUs` for the neighbour-search slots (8..11)
            lo = 0 if slot < 8 else cus[0]
            s = masked_stream(device, range(lo, lo + cus[0 if slot < 8 else 1]))
        else:
            s = torch.cuda.Stream(device=device, priority=priority)
        _SIDE_STREAMS[key] = s
    return s


def cu_partition():
    """MSMD_CU_PARTITION="i,s": CUs given to the index stream(s) and to the neighbour-search
    streams (either may be 0 = unmasked, high priority as usual); None when unset."""
    import os
    v = os.environ.get("MSMD_CU_PARTITION")
    if not v:
        return None
    a = [int(x) for x in v.split(",")]
    return (a[0], a[1] if len(a) > 1 else 0)


_HIP = None
_MASKED = []      # (handle, ExternalStream): kept alive for the life of the process


def masked_stream(device, cu_bits, total_cus=256):
    """A stream whose kernels run on the given CUs only (hipExtStreamCreateWithCUMask).  Bit i
    of the mask is CU i / 8 of XCD i % 8 on MI300-class parts (the driver deals the mask's bits
    round robin over the XCDs), so a contiguous range of bits is spread evenly over the eight
    XCDs.  Default priority (the extension has no priority argument)."""
    import ctypes as C
    global _HIP
    device = torch.device(device)
    if _HIP is None:
        # the HIP runtime THIS process already runs on (torch's bundled copy): opening
        # "libamdhip64.so" by name could map /opt/rocm's as a second runtime instance whose
        # streams torch's would not know
        path = None
        with open("/proc/self/maps") as maps:
            for line in maps:
                if "libamdhip64" in line:
                    path = line.split()[-1]
                    break
        if path is None:
            raise RuntimeError("no HIP runtime is mapped into this process")
        _HIP = C.CDLL(path)
    words = (total_cus + 31) // 32
    mask = (C.c_uint32 * words)()
    for b in cu_bits:
        mask[b // 32] |= 1 << (b % 32)
    handle = C.c_void_p()
    with torch.cuda.device(device):
        rc = _HIP.hipExtStreamCreateWithCUMask(C.byref(handle), C.c_uint32(words), mask)
    if rc != 0:
        raise RuntimeError("hipExtStreamCreateWithCUMask failed: %d" % rc)
    s = torch.cuda.ExternalStream(handle.value, device=device)
    _MASKED.append((handle, s))
    return s


class IndexPrefetcher:

    def __init__(self, prepare_fn, device, priority=-1, threaded=True, depth=1, workers=None):
        """threaded: run prepare_fn on a worker thread.  Its host reads (voxel and
        pair counts) wait for the side stream with the GIL released, so the calling
        thread keeps the main stream fed meanwhile -- without it those waits come
        straight out of the time the host has to enqueue the feature pass.

        On a CPU device (the gloo tests of the multi-rank step structure) there
        are no streams: prepare_fn runs inline or on the worker thread, and the
        hand-over / retire events are no-ops.

        depth: batches in flight.  prepare() is a chain of ~150 small dependent
        launches and ~35 host reads; next to the feature pass (whose persistent
        kernels fill every CU) its latency grows past the feature pass's own
        duration and becomes the step time.  With depth 2 two batches are prepared
        concurrently, each on its own worker thread and side stream, and the step
        is bound by throughput again.
        workers: threads (and side streams) the `depth` batches share, default one each.
        workers=1 with depth=2 keeps ONE prepare() running at any time but lets the next one
        start the moment the worker is free instead of when the step thread gets round to
        submitting it."""
        self.prepare_fn = prepare_fn
        self.device = torch.device(device)
        self.on_gpu = self.device.type == "cuda"
        self.depth = max(int(depth), 1) if threaded else 1
        self.workers = min(self.depth, max(int(workers), 1)) if workers else self.depth
        self._sides = [side_stream(self.device, priority, slot=i)
                       for i in range(self.workers)] if self.on_gpu else [None]
        self.side = self._sides[0]
        self._next = 0
        self._retired = collections.deque()
        self.max_behind = 2      # steps the host may run ahead of the main stream
        self._pool = ThreadPoolExecutor(self.workers, thread_name_prefix="msmd-index") \
            if threaded else None

    def _run(self, grad, side, args, kw):
        if not self.on_gpu:
            with torch.set_grad_enabled(grad):
                return {"value": self.prepare_fn(*args, **kw), "ready": None}
        torch.cuda.set_device(self.device)          # current device / stream / grad mode
        with torch.set_grad_enabled(grad), torch.cuda.stream(side):    # are per thread
            value = self.prepare_fn(*args, **kw)
            ready = torch.cuda.Event()
            ready.record(side)
        return {"value": value, "ready": ready}

    def submit(self, *args, **kw):
        self._collect()
        grad = torch.is_grad_enabled()
        side = self._sides[self._next % len(self._sides)]
        self._next += 1
        if self._pool is None:
            return self._run(grad, side, args, kw)
        return {"future": self._pool.submit(self._run, grad, side, args, kw)}

    def take(self, ticket):
        if "future" in ticket:
            ticket.update(ticket.pop("future").result())
        if ticket["ready"] is not None:
            torch.cuda.current_stream(self.device).wait_event(ticket["ready"])
        return ticket["value"]

    def retire(self, ticket):
        if not self.on_gpu:
            return
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(self.device))
        self._retired.append((done, ticket))

    def close(self):
        """Stop the worker thread(s) (idle threads of discarded prefetchers otherwise live as
        long as the process) and drop the retired batches.  The side streams are process-wide
        and stay."""
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None
        self._retired.clear()

    def _collect(self):
        # the side stream's host reads do not throttle the host to the MAIN stream's
        # pace: without a bound it runs many steps ahead and every retired batch
        # (hundreds of MB of tables) stays alive until the GPU catches up
        while len(self._retired) > self.max_behind:
            self._retired[0][0].synchronize()
            self._retired.popleft()
        while self._retired and self._retired[0][0].query():
            self._retired.popleft()
