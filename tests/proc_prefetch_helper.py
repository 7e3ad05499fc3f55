"""Worker-side half of test_process_prefetch_matches_inline_lc (module-level, so that the
spawned index worker can import it)."""
import torch


def fixed_dummy(c, device):
    return torch.full((1, c), 0.25, device=device)


def build_model(dev):
    import bench
    torch.manual_seed(0)
    model = bench.FusionBackbone().to(dev).train()
    model.path.multimodal_middle_encoder.dummy_embedding_fn = fixed_dummy
    return model


def make_batch(dev, n_az=300, n_virtual=12000):
    from msmdfusion_amd import synthetic as S
    clouds = [torch.from_numpy(S.lidar_sweep(i, n_az=n_az)).to(dev) for i in range(2)]
    virt = [torch.from_numpy(S.virtual_points(i, n=n_virtual)).to(dev) for i in range(2)]
    return clouds, virt


def init():
    dev = torch.device("cuda", torch.cuda.current_device())
    model = build_model(dev)
    clouds, virt = make_batch(dev)
    return lambda: model.prepare(clouds, virt)
