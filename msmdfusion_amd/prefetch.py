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

import torch


class IndexPrefetcher:

    def __init__(self, prepare_fn, device, priority=-1):
        self.prepare_fn = prepare_fn
        self.device = torch.device(device)
        self.side = torch.cuda.Stream(device=self.device, priority=priority)
        self._retired = collections.deque()
        self.max_behind = 2      # steps the host may run ahead of the main stream

    def submit(self, *args, **kw):
        self._collect()
        with torch.cuda.stream(self.side):
            value = self.prepare_fn(*args, **kw)
            ready = torch.cuda.Event()
            ready.record(self.side)
        return {"value": value, "ready": ready}

    def take(self, ticket):
        torch.cuda.current_stream(self.device).wait_event(ticket["ready"])
        return ticket["value"]

    def retire(self, ticket):
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(self.device))
        self._retired.append((done, ticket))

    def _collect(self):
        # the side stream's host reads do not throttle the host to the MAIN stream's
        # pace: without a bound it runs many steps ahead and every retired batch
        # (hundreds of MB of tables) stays alive until the GPU catches up
        while len(self._retired) > self.max_behind:
            self._retired[0][0].synchronize()
            self._retired.popleft()
        while self._retired and self._retired[0][0].query():
            self._retired.popleft()
