#!/usr/bin/env python
"""Generates tests/golden/reference_vectors.npz from the REFERENCE's own CPU
code (oracle/_ref/libmsmd_ref.so, built from /root/reference by `make -C oracle
ref`).  Run in the build container only; the .npz is committed so the GPU box
(no /root/reference) can check the oracle against reference outputs.

Inputs are small and seeded; outputs are what the reference returned, raw
(CPU first-touch order).  Only data is stored -- no reference source.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from msmdfusion_amd import synthetic as S  # noqa: E402
from oracle import oracle as O  # noqa: E402

GEOMS = [  # name, subm, ksize, stride, padding  (SURVEY Appendix A geometries)
    ("subm3", True, [3, 3, 3], [1, 1, 1], [1, 1, 1]),
    ("down_p1", False, [3, 3, 3], [2, 2, 2], [1, 1, 1]),
    ("down_p011", False, [3, 3, 3], [2, 2, 2], [0, 1, 1]),
    ("out_311", False, [3, 1, 1], [2, 1, 1], [0, 0, 0]),
    ("down_k3s1p0", False, [3, 3, 3], [1, 1, 1], [0, 0, 0]),
    ("down_k2s2", False, [2, 2, 2], [2, 2, 2], [0, 0, 0]),
    ("subm133", True, [1, 3, 3], [1, 1, 1], [0, 1, 1]),
    ("down_s3p2", False, [3, 3, 3], [3, 3, 3], [2, 2, 2]),
]


def main():
    assert O.have_ref(), "build oracle/_ref first: make -C oracle ref"
    out = {}
    # --- voxelization: LiDAR-like cloud, three parameter sets incl. truncation
    pts = S.lidar_sweep(7, n_az=120)
    out["vox_points"] = pts
    for tag, vs, mp, mv in [("a", S.VOXEL_SIZE, 10, 20000), ("b", [0.3, 0.3, 0.8], 3, 20000),
                            ("c", S.VOXEL_SIZE, 2, 1500)]:
        v, c, n = O.hard_voxelize(pts, vs, S.POINT_CLOUD_RANGE, mp, mv, use_ref=True)
        out[f"vox_{tag}_params"] = np.array(list(vs) + [mp, mv], np.float64)
        out[f"vox_{tag}_voxels"], out[f"vox_{tag}_coors"], out[f"vox_{tag}_num"] = v, c, n
    # --- rulebooks on a clustered random voxel set, batch 2
    shape = [11, 48, 48]
    idx = S.random_voxel_indices(900, 2, shape, seed=21)
    out["rb_indices"], out["rb_shape"] = idx, np.array(shape)
    for name, subm, ks, st, pd in GEOMS:
        oi, pr, nm, osz = O.get_indice_pairs(idx, 2, shape, ks, st, pd, 1, subm, use_ref=True)
        out[f"rb_{name}_out"], out[f"rb_{name}_pairs"], out[f"rb_{name}_num"] = oi, pr, nm
        out[f"rb_{name}_oshape"] = np.array(osz)
    # --- Native conv forward (reference gather / torch::mm / scatter-add)
    rng = np.random.RandomState(3)
    rng_g = np.random.RandomState(4)      # (its own stream: the forward vectors keep their values)
    for name, subm, ks, st, pd in GEOMS[:4]:
        cin, cout = (16, 32) if subm else (32, 16)
        f = rng.randn(idx.shape[0], cin).astype(np.float32)
        kv = int(np.prod(ks))
        w = (rng.randn(kv, cin, cout) / np.sqrt(kv * cin)).astype(np.float32)
        pr, nm = out[f"rb_{name}_pairs"], out[f"rb_{name}_num"]
        n_out = out[f"rb_{name}_out"].shape[0]
        out[f"conv_{name}_feat"], out[f"conv_{name}_w"] = f, w
        out[f"conv_{name}_out"] = O.indice_conv_fwd(f, w, pr, nm, n_out, subm=subm, use_ref=True)
        # backward through the reference's gather / scatter-add functors
        # (indiceConvBackward, spconv_ops.h:363-456)
        g = rng_g.randn(n_out, cout).astype(np.float32)
        din, dw = O.indice_conv_bwd(f, w, g, pr, nm, subm=subm, use_ref=True)
        out[f"conv_{name}_gout"], out[f"conv_{name}_din"], out[f"conv_{name}_dw"] = g, din, dw
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
