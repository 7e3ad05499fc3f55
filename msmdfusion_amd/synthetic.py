"""Seeded synthetic nuScenes-shaped inputs.

There is no dataset on the build or GPU boxes, so bench.py and the tests use
these generators: a 32-beam ring-scan LiDAR sweep (~28.8k points, ~26.8k inside
the detection range -> ~19k voxels at 0.075 m) and ~50k multi-depth virtual
points on the visible faces of 30 box-shaped objects, 64 channels each (the
layout MSMDFusionDetector.get_foreground2D builds:
mmdet3d/models/detectors/MSMDFusion.py:169-238 -- xyz, 11 semantic, dt, 49
image channels).  numpy only, deterministic per (seed, sample index).
"""
import math

import numpy as np

POINT_CLOUD_RANGE = [-54.0, -54.0, -5.0, 54.0, 54.0, 3.0]
VOXEL_SIZE = [0.075, 0.075, 0.2]
SPARSE_SHAPE = [41, 1440, 1440]
MAX_NUM_POINTS = 10
MAX_VOXELS = (120000, 160000)


def lidar_sweep(seed=0, n_az=1084, beams=32, sweeps=1):
    """One synthetic cloud, float32 [N,5] = x,y,z,intensity,dt."""
    rng = np.random.RandomState(seed)
    clouds = []
    for s in range(sweeps):
        el = np.linspace(math.radians(-30.67), math.radians(10.67), beams)
        az = np.arange(n_az) / n_az * 2 * math.pi + (0.37 * s)
        el, az = [a.ravel() for a in np.meshgrid(el, az, indexing="ij")]
        h = 1.84
        with np.errstate(divide="ignore"):
            r_ground = np.where(el < 0, h / np.maximum(np.tan(-el), 1e-6), 1e9)
        r_obj = 3 + rng.rand(el.size) * 60
        hit = rng.rand(el.size) < 0.45
        r = np.where(hit, np.minimum(r_obj, r_ground), r_ground)
        r = r + rng.randn(el.size) * 0.02
        keep = r < 75
        x = r * np.cos(el) * np.cos(az) + 0.6 * s
        y = r * np.cos(el) * np.sin(az)
        z = r * np.sin(el)
        p = np.stack([x, y, z, rng.rand(el.size), np.full(el.size, 0.05 * s)], 1)[keep]
        clouds.append(p)
    return np.ascontiguousarray(np.concatenate(clouds, 0), dtype=np.float32)


def virtual_points(seed=0, n=50000, nobj=30, channels=64):
    """Virtual (image-derived) points, float32 [n,64]."""
    rng = np.random.RandomState(1000003 + seed)
    per = n // nobj
    out = []
    size = np.array([4.5, 2.0, 1.6])
    for _ in range(nobj):
        c = np.array([rng.rand() * 90 - 45, rng.rand() * 90 - 45, -1.0])
        q = (rng.rand(per, 3) - 0.5) * size
        q[:, 0] = size[0] / 2 * np.sign(q[:, 0]) * (1 - 0.05 * rng.rand(per))
        out.append(q + c)
    xyz = np.concatenate(out, 0)
    m = xyz.shape[0]
    sem = np.zeros((m, 11))
    sem[np.arange(m), rng.randint(0, 10, m)] = 1.0
    sem[:, 10] = rng.rand(m)
    feat = np.concatenate([xyz, sem, np.zeros((m, 1)), rng.rand(m, channels - 15)], 1)
    return np.ascontiguousarray(feat, dtype=np.float32)


def random_voxel_indices(n, batch_size, spatial_shape, seed=0, clustered=True):
    """Unique random (b,z,y,x) int32 rows for unit tests."""
    rng = np.random.RandomState(seed)
    d, h, w = spatial_shape
    if clustered:
        centres = rng.rand(max(n // 40, 1), 3) * [d, h, w]
        pick = rng.randint(0, centres.shape[0], 3 * n)
        pos = centres[pick] + rng.randn(3 * n, 3) * [max(d / 20, 0.7), 2.5, 2.5]
    else:
        pos = rng.rand(3 * n, 3) * [d, h, w]
    pos = np.floor(pos).astype(np.int64)
    ok = (pos >= 0).all(1) & (pos[:, 0] < d) & (pos[:, 1] < h) & (pos[:, 2] < w)
    pos = pos[ok]
    b = rng.randint(0, batch_size, pos.shape[0])
    rows = np.concatenate([b[:, None], pos], 1)
    _, first = np.unique(rows, axis=0, return_index=True)
    rows = rows[np.sort(first)][:n]
    return np.ascontiguousarray(rows, dtype=np.int32)


def seeded_parameters(module, seed=0):
    """Overwrite every parameter of a torch module with values that depend only
    on (seed, parameter NAME, shape) -- numpy's frozen RandomState, not torch's
    generator -- so two implementations with the same state-dict keys (the
    reference's SPPModule and ours) get identical weights without shipping them.
    Conv / linear weights ~ N(0, 1/fan_in); norm weights in [0.5, 1.5]; biases
    in [-0.2, 0.5]."""
    import zlib

    import torch
    with torch.no_grad():
        for name, p in sorted(module.named_parameters()):
            rng = np.random.RandomState((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31))
            if p.dim() > 1:
                fan_in = int(np.prod(p.shape[1:]))
                v = rng.standard_normal(tuple(p.shape)) / math.sqrt(fan_in)
            elif name.endswith("weight"):
                v = 0.5 + rng.rand(*p.shape)
            else:
                v = -0.2 + 0.7 * rng.rand(*p.shape)
            p.copy_(torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)))
    return module
