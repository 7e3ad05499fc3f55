"""Index work in a WORKER PROCESS (the data-loader arrangement).

`IndexPrefetcher` runs prepare() on a thread; on the LC path that thread and the
training thread spend the step taking turns on the interpreter lock (DESIGN.md 8.5:
6.7 + 6.9 ms of Python per 15.8 ms step, the device needs 13.3).  Here prepare() --
voxelization, rulebooks, tilings, modality split, neighbour search: everything that
depends on the inputs alone -- runs in its own process, next to the data it would be
loading anyway, and a prepared batch crosses the process boundary as

  * a few flat device buffers (one per dtype) holding every tensor of the batch,
    shared through torch's CUDA IPC (one handle per buffer, not per tensor), and
  * a pickled skeleton of the object graph in which each tensor is (buffer, offset,
    shape): rebuilding ~300 views costs the training process < 1 ms.

The worker synchronises its device before it hands a batch over, so the consumer needs
no cross-process event; a consumed batch's buffers are kept referenced until the
consumer's stream has passed its last reader (retire(), as in IndexPrefetcher).
"""
import collections
import queue
import io
import pickle

import torch

_ALIGN = 64      # elements: every tensor starts 256-byte aligned inside its buffer


class _Packer(pickle.Pickler):
    def __init__(self, file):
        super().__init__(file, protocol=pickle.HIGHEST_PROTOCOL)
        self.tensors = []

    def persistent_id(self, obj):
        if isinstance(obj, torch.Tensor):
            self.tensors.append(obj)
            return len(self.tensors) - 1
        if isinstance(obj, torch.cuda.Event):      # same-process hand-over marker: the worker
            return "none"                          # synchronises instead
        return None


def pack(obj):
    """-> (skeleton bytes, [(dtype, shape, buffer key, offset)], {key: flat buffer})"""
    f = io.BytesIO()
    p = _Packer(f)
    p.dump(obj)
    groups, table = {}, []
    for t in p.tensors:
        key = "%s@%s" % (t.dtype, t.device)
        n = t.numel()
        off = groups.setdefault(key, [0, [], [], t.dtype, t.device])
        table.append((key, tuple(t.shape), off[0]))
        if n:
            off[1].append((off[0], n))
            off[2].append(t.detach().reshape(-1) if t.is_contiguous()
                          else t.detach().contiguous().reshape(-1))
        off[0] += (n + _ALIGN - 1) // _ALIGN * _ALIGN
    buffers = {}
    for key, (total, spans, srcs, dtype, device) in groups.items():
        buf = torch.empty((max(total, 1),), dtype=dtype, device=device)
        if srcs:
            torch._foreach_copy_([buf[o:o + n] for o, n in spans], srcs)
        buffers[key] = buf
    return f.getvalue(), table, buffers


class _Unpacker(pickle.Unpickler):
    def __init__(self, file, table, buffers):
        super().__init__(file)
        self.table, self.buffers = table, buffers

    def persistent_load(self, pid):
        if pid == "none":
            return None
        key, shape, off = self.table[pid]
        n = 1
        for s in shape:
            n *= s
        return self.buffers[key][off:off + n].view(shape)


def unpack(skeleton, table, buffers):
    return _Unpacker(io.BytesIO(skeleton), table, buffers).load()


def _worker_main(init, init_args, device_index, queue, stop):
    torch.cuda.set_device(device_index)
    produce = init(*init_args)            # zero-argument callable -> one prepared batch
    try:
        while not stop.is_set():
            value = produce()
            torch.cuda.synchronize()      # side streams included: no events cross the boundary
            item = pack(value)
            torch.cuda.synchronize()
            del value
            queue.put(item)
    except (KeyboardInterrupt, BrokenPipeError, EOFError):
        pass


class ProcessPrefetcher:
    """Same submit / take / retire surface as IndexPrefetcher; the worker owns the
    stream of batches (like a DataLoader worker), so submit() carries nothing.

    init(*init_args) runs in the worker and returns a zero-argument callable that
    yields prepared batches; both must be picklable (module-level function, plain
    arguments)."""
    depth = 1         # TrainStep keeps one ticket pending; the queue holds the look-ahead

    def __init__(self, init, init_args, device, queue_depth=2):
        import torch.multiprocessing as mp
        self.device = torch.device(device)
        ctx = mp.get_context("spawn")
        self.queue = ctx.Queue(maxsize=queue_depth)
        self.stop = ctx.Event()
        self.proc = ctx.Process(target=_worker_main,
                                args=(init, tuple(init_args), self.device.index, self.queue,
                                      self.stop), daemon=True)
        self.proc.start()
        self._retired = collections.deque()
        self.on_gpu = True

    def submit(self, *args, **kw):
        return {"proc": True}

    def take(self, ticket):
        while True:
            try:
                skeleton, table, buffers = self.queue.get(timeout=5.0)
                break
            except queue.Empty:          # nothing yet: is the worker still there?
                # (only Empty: an IPC-rebuild or unpickling error raised by get() has
                # consumed the item and must surface, not be retried forever)
                if not self.proc.is_alive():
                    raise RuntimeError("index worker process died (exit code %s)"
                                       % self.proc.exitcode)
        ticket["buffers"] = buffers
        return unpack(skeleton, table, buffers)

    def retire(self, ticket):
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(self.device))
        self._retired.append((done, ticket))
        while self._retired and self._retired[0][0].query():
            self._retired.popleft()
        while len(self._retired) > 3:
            self._retired[0][0].synchronize()
            self._retired.popleft()

    def close(self):
        self.stop.set()
        try:
            while True:                  # unblock a worker waiting on a full queue
                self.queue.get_nowait()
        except Exception:
            pass
        self.proc.join(timeout=5.0)
        if self.proc.is_alive():
            self.proc.terminate()
        self._retired.clear()
