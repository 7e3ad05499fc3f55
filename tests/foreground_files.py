"""Synthetic virtual-point files in the reference's on-disk format, for the loader tests
and for tests/golden/make_loader_golden.py (same seeds -> same files)."""
import os

import numpy as np

from msmdfusion_amd.loaders import FOREGROUND_DIR, save_foreground

CAMS = 6


def _sweep_payload(rng, with_labels_in_points=False):
    vpix, rpix, vpts, rpts = [], [], [], []
    for cam in range(CAMS):
        nv = 0 if cam == 4 else int(rng.randint(5, 40))      # one camera without virtual points
        nr = int(rng.randint(3, 15))
        for n, pix, pts in ((nv, vpix, vpts), (nr, rpix, rpts)):
            lab = np.zeros((n, 11))
            if n:
                lab[np.arange(n), rng.randint(0, 10, n)] = 1.0
                lab[:, 10] = rng.rand(n)
            rec = np.concatenate([rng.rand(n, 2) * [1600, 900], 1 + rng.rand(n, 1) * 60, lab], 1)
            pix.append(rec)
            xyz = rng.randn(n, 3) * 20
            pts.append(np.concatenate([xyz, lab], 1) if with_labels_in_points else xyz)
    return vpix, rpix, vpts, rpts


def make_tree(root, seed=0):
    """root/samples/LIDAR_TOP/key.bin (+ 3 sweeps under root/sweeps/LIDAR_TOP, the second
    of which has no foreground file) -> the `results` dict a pipeline would carry."""
    rng = np.random.RandomState(seed)
    key = os.path.join(root, "samples", "LIDAR_TOP", "key.pcd.bin")
    save_foreground(os.path.join(root, "samples", FOREGROUND_DIR, "key.pcd.bin.pkl.npy"),
                    *_sweep_payload(rng))
    sweeps = []
    for i in range(3):
        path = os.path.join(root, "sweeps", "LIDAR_TOP", "sweep%d.pcd.bin" % i)
        if i != 1:
            save_foreground(os.path.join(root, "sweeps", FOREGROUND_DIR,
                                         "sweep%d.pcd.bin.pkl.npy" % i),
                            *_sweep_payload(rng, with_labels_in_points=(i == 2)))
        a = rng.rand() * 0.2
        rot = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
        sweeps.append(dict(data_path=path, timestamp=1.5e15 - (i + 1) * 5e4,
                           sensor2lidar_rotation=rot,
                           sensor2lidar_translation=rng.randn(3) * 0.5))
    return dict(pts_filename=key, timestamp=1.5e9, sweeps=sweeps)


def add_lidar_files(results, seed=0):
    """Writes the key frame's and the sweeps' raw .bin point files (5 float32 per point,
    a few points near the sensor) and returns `results` with pts_filename / data_path set."""
    rng = np.random.RandomState(1000 + seed)

    def cloud(path, n):
        p = np.concatenate([rng.randn(n, 3) * [20, 20, 2], rng.rand(n, 1), np.zeros((n, 1))], 1)
        p[:5, :2] = rng.rand(5, 2) - 0.5              # inside remove_close's 1 m box
        os.makedirs(os.path.dirname(path), exist_ok=True)
        p.astype(np.float32).tofile(path)
    cloud(results["pts_filename"], 300)
    for i, sw in enumerate(results["sweeps"]):
        cloud(sw["data_path"], 200 + 10 * i)
    return results
