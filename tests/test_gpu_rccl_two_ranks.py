"""Two ranks over RCCL (torch backend "nccl" on ROCm), one process per GPU: the data-parallel
recipe of tools/dist_train.sh:8-9 / configs/MSMDFusion_nusc_voxel_LC.py:300,309 on real
devices.  Skips unless two GPUs are visible -- the build box has one, so the first multi-GPU
lease that runs `pytest -m gpu` measures something: gradients of the sparse path's trained
weights, all-reduced over xGMI, against the average of the two ranks' single-process
gradients; the bench entry's step (prefetch -> DDP forward -> clip -> AdamW) leaves both
ranks with identical parameters."""
import os
import socket
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _tiny_sparse_net(dev):
    from msmdfusion_amd import spconv
    torch.manual_seed(0)
    return spconv.SparseSequential(
        spconv.SubMConv3d(32, 32, 3, padding=1, bias=False, indice_key="a"),
        torch.nn.BatchNorm1d(32), torch.nn.ReLU(),
        spconv.SparseConv3d(32, 64, 3, stride=2, padding=1, bias=False)).to(dev)


def _batch(rank, dev):
    from msmdfusion_amd import synthetic as S
    shape = [9, 48, 48]
    idx = torch.from_numpy(S.random_voxel_indices(2500, 2, shape, seed=11 + rank)).to(dev)
    g = torch.Generator().manual_seed(50 + rank)
    return torch.randn(idx.shape[0], 32, generator=g).to(dev), idx, shape


def _loss(net, feat, idx, shape):
    from msmdfusion_amd import spconv
    return net(spconv.SparseConvTensor(feat, idx, shape, 2)).features.square().mean()


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    from msmdfusion_amd import distributed as D
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    r, lr, w = D.init_distributed(device=dev)          # RCCL
    assert (r, w) == (rank, world) and D.rccl_ranks() == world
    net = _tiny_sparse_net(dev)
    ddp = D.wrap_data_parallel(net, device_ids=[rank])
    _loss(ddp, *_batch(rank, dev)).backward()
    grads = torch.cat([p.grad.flatten() for p in net.parameters()]).cpu()
    # every rank also computes both ranks' gradients without DDP
    ref = _tiny_sparse_net(dev)
    acc = None
    for rr in range(world):
        ref.zero_grad()
        _loss(ref, *_batch(rr, dev)).backward()
        flat = torch.cat([p.grad.flatten() for p in ref.parameters()]).cpu()
        acc = flat if acc is None else acc + flat
    err = float((grads - acc / world).abs().max() / (acc / world).abs().max())
    # two optimisation steps through the bench's step object: same parameters on both ranks
    params = list(net.parameters())
    opt = torch.optim.AdamW(params, lr=1e-3, weight_decay=0.01, fused=True)
    for s in range(2):
        f, i, sh = _batch(rank + 2 * s, dev)
        opt.zero_grad(set_to_none=True)
        _loss(ddp, f, i, sh).backward()
        torch.nn.utils.clip_grad_norm_(params, 10.0)
        opt.step()
    flat = torch.cat([p.detach().flatten() for p in params]).cpu()
    tmax = D.global_max(1.0 + rank, device=dev)
    out[rank] = (err, flat, tmax)
    D.shutdown()


@pytest.mark.timeout(600)
def test_two_rank_rccl_data_parallel():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL at world 2); this box has %d"
                    % torch.cuda.device_count())
    import torch.multiprocessing as mp
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    (e0, p0, t0), (e1, p1, t1) = out[0], out[1]
    # fp32 conv sums in another order per rank (stream-K pieces) + ring all-reduce rounding
    assert e0 < 1e-4 and e1 < 1e-4, (e0, e1)
    assert torch.equal(p0, p1)                  # DDP: identical parameters after the steps
    assert t0 == t1 == 2.0
