"""Data-parallel plumbing: one process per GPU, RCCL (torch backend "nccl")
over xGMI for the gradient all-reduce -- the only exchange of this path
(reference: tools/dist_train.sh:8-9, dist_params backend='nccl'
configs/MSMDFusion_nusc_voxel_LC.py:300, MMDistributedDataParallel).

Samples are independent through the whole sparse path, so ranks shard samples
and nothing else; rulebooks, voxel indices and activations never cross GPUs.
The helpers are backend-agnostic so the N>1 logic is covered on CPU with gloo
(tests/test_dist_cpu.py).
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_distributed(backend=None, device=None):
    """Initialise torch.distributed from the torchrun environment (no-op for a
    single process).  backend defaults to nccl (= RCCL) on GPU, gloo on CPU."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def shard_sample_ids(rank, world, samples_per_gpu, step=0):
    """Global sample ids of one rank for one step: disjoint across ranks,
    contiguous per rank (weak scaling: samples_per_gpu is fixed)."""
    base = (step * world + rank) * samples_per_gpu
    return list(range(base, base + samples_per_gpu))


def wrap_data_parallel(model, device_ids=None, find_unused_parameters=False):
    """DDP with bucketed gradient all-reduce overlapped with backward.  The
    reference needs find_unused_parameters=True (LC.py:309) because
    grouped_sp_conv_blocks_2D/_mix are built but never called; callers here
    freeze those blocks instead, which avoids the per-step graph walk."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return model
    return torch.nn.parallel.DistributedDataParallel(
        model, device_ids=device_ids, gradient_as_bucket_view=True,
        find_unused_parameters=find_unused_parameters)


def global_max(value, device=None):
    """max over ranks of a python float (the bench's step-time reduction)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def freeze_unused_fusion_blocks(multimodal_encoder):
    """requires_grad=False for the blocks the forward never calls
    (sparse_multimodal_encoder_painting.py:142-156 vs :413-428)."""
    n = 0
    for name in ("grouped_sp_conv_blocks_2D", "grouped_sp_conv_blocks_mix"):
        for p in getattr(multimodal_encoder, name).parameters():
            p.requires_grad = False
            n += 1
    return n
