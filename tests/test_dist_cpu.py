"""N>1 path on CPU: two gloo processes exercise the same helpers bench.py uses
on RCCL (sample sharding, DDP gradient averaging, max-over-ranks timing)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from msmdfusion_amd import distributed as D
    r, lr, w = D.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    ids = D.shard_sample_ids(r, w, 4, step=3)
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))
    ddp = D.wrap_data_parallel(model)
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(4, 8, generator=g)
    ddp(x).pow(2).sum().backward()
    grads = torch.cat([p.grad.flatten() for p in model.parameters()])
    # reference: average of the per-rank gradients computed without DDP
    torch.manual_seed(0)
    ref = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))
    acc = None
    for rr in range(world):
        ref.zero_grad()
        gg = torch.Generator().manual_seed(100 + rr)
        ref(torch.randn(4, 8, generator=gg)).pow(2).sum().backward()
        flat = torch.cat([p.grad.flatten() for p in ref.parameters()])
        acc = flat if acc is None else acc + flat
    ok_grad = torch.allclose(grads, acc / world, atol=1e-6)
    tmax = D.global_max(1.0 + rank)
    D.barrier()
    out[rank] = (ids, bool(ok_grad), tmax)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_data_parallel():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    ids0, ok0, t0 = out[0]
    ids1, ok1, t1 = out[1]
    assert ok0 and ok1
    assert t0 == t1 == 2.0                      # max over ranks
    assert len(set(ids0) | set(ids1)) == 8 and not set(ids0) & set(ids1)
    assert ids0 == [24, 25, 26, 27] and ids1 == [28, 29, 30, 31]


def test_single_process_helpers_are_noops():
    sys.path.insert(0, ROOT)
    from msmdfusion_amd import distributed as D
    m = torch.nn.Linear(2, 2)
    assert D.wrap_data_parallel(m) is m
    assert D.global_max(3.5) == 3.5
    assert D.shard_sample_ids(0, 1, 4) == [0, 1, 2, 3]
