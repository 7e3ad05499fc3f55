#!/usr/bin/env python
"""Generates tests/golden/modality_split_vectors.npz by RUNNING the reference's own
Python code for row a14:

    type_assign                                  MSMDFusion.py:27-45   (numba @jit stripped:
                                                 numba is not installed; the body is plain Python)
    MSMDFusionDetector.voxel_modality_split      MSMDFusion.py:251-325

mmdet3d itself cannot be imported here (mmcv / mmdet / spconv / numba are absent), so the two
definitions are pulled out of the reference FILE at run time (ast), compiled as they stand and
called with bare objects that carry an `.indices` tensor.  Nothing of the reference is written
to the repo: the .npz holds seeded voxel sets and what the reference code returned for them on
CPU -- the mix flags and syn_mix_3D / syn_mix_2D, float32-key aliasing and non-cumulative
batch offsets included.  Build container only (/root/reference).

The voxel sets are built so that no float32 key repeats INSIDE a set (the reference's
torch.sort leaves the order of equal keys unspecified; which of two tied rows gets matched
would depend on it) while keys DO collide ACROSS the sets:
  * z in 17..32: keys lie in [2^24, 2^25), float spacing 2.  LiDAR voxels at x = 0 (mod 4) and
    2 (mod 4); virtual-point voxels at x = 1 (mod 4) -- rounded to the multiple of 4 below: a
    FALSE match with the LiDAR voxel there -- and at x = 2 (mod 4): true matches;
  * x >= 1000: a voxel at (y, 1000 + a) has the key of (y + 1, a): false matches across rows;
  * z <= 15: exact keys, ordinary matches.
"""
import ast
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference/mmdet3d/models/detectors/MSMDFusion.py"
OUT = os.path.join(ROOT, "tests", "golden", "modality_split_vectors.npz")


def reference_defs():
    tree = ast.parse(open(REF).read())
    ns = {"torch": torch, "F": F, "np": np, "jit": lambda *a, **k: (lambda f: f)}
    ta = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "type_assign")
    det = next(n for n in tree.body if isinstance(n, ast.ClassDef)
               and n.name == "MSMDFusionDetector")
    vms = next(n for n in det.body if isinstance(n, ast.FunctionDef)
               and n.name == "voxel_modality_split")
    exec(compile(ast.Module(body=[ta, vms], type_ignores=[]), REF, "exec"), ns)
    return ns


def voxel_sets(rng, b):
    """(3D, 2D) int32 [n, 4] rows (b, z, y, x) of one sample, shuffled."""
    rows3, rows2 = [], []
    # spacing-2 zone
    for z in range(17, 33):
        for y in range(100, 104):
            xs = np.arange(200, 420)
            pick3 = xs[(xs % 2 == 0) & (rng.rand(xs.size) < 0.45)]
            pick2 = xs[((xs % 4 == 1) | (xs % 4 == 2)) & (rng.rand(xs.size) < 0.45)]
            rows3 += [(b, z, y, x) for x in pick3]
            rows2 += [(b, z, y, x) for x in pick2]
    # x >= 1000 runs into the next y (any z; here the exact zone so that nothing else aliases)
    for z in (3, 9):
        for a in rng.choice(400, 60, replace=False):
            rows3.append((b, z, 300, 1000 + int(a)))
            rows2.append((b, z, 301, int(a)))
    # exact zone: plain matches and misses
    for z in range(0, 16, 3):
        for y in range(50, 54):
            xs = np.arange(100, 300)
            rows3 += [(b, z, y, x) for x in xs[rng.rand(xs.size) < 0.3]]
            rows2 += [(b, z, y, x) for x in xs[rng.rand(xs.size) < 0.3]]
    r3 = np.unique(np.array(rows3, np.int32), axis=0)
    r2 = np.unique(np.array(rows2, np.int32), axis=0)
    return r3[rng.permutation(r3.shape[0])], r2[rng.permutation(r2.shape[0])]


def float_keys(zyx):
    k = zyx[:, 0].astype(np.float32) * np.float32(1e6)
    k = k + zyx[:, 1].astype(np.float32) * np.float32(1e3)
    return k + zyx[:, 2].astype(np.float32)


def main():
    ns = reference_defs()
    out = {}
    for tag, batch, seed in (("b2", 2, 5), ("b3", 3, 6), ("b1", 1, 7)):
        rng = np.random.RandomState(seed)
        sets = [voxel_sets(rng, b) for b in range(batch)]
        i3 = np.concatenate([s[0] for s in sets])
        i2 = np.concatenate([s[1] for s in sets])
        for b in range(batch):          # the fixture's premise: no key repeats inside a set
            for idx in (i3, i2):
                k = float_keys(idx[idx[:, 0] == b][:, 1:])
                assert np.unique(k).size == k.size, "a float key repeats inside a set"
        v3 = types.SimpleNamespace(indices=torch.from_numpy(i3.copy()))
        v2 = types.SimpleNamespace(indices=torch.from_numpy(i2.copy()))
        v3, v2, s3, s2 = ns["voxel_modality_split"](None, v3, v2, batch)
        c3, c2 = v3.indices.numpy(), v2.indices.numpy()
        assert np.array_equal(c3[:, [0, 2, 3, 4]], i3) and np.array_equal(c2[:, [0, 2, 3, 4]], i2)
        out.update({tag + "_idx3": i3, tag + "_idx2": i2, tag + "_mix3": c3[:, 1].astype(np.int32),
                    tag + "_mix2": c2[:, 1].astype(np.int32),
                    tag + "_syn3": s3.numpy().astype(np.int64),
                    tag + "_syn2": s2.numpy().astype(np.int64)})
        # what the fixture exercises (global rows only where the reference's offsets are right)
        n_false = -1
        if batch <= 2:
            n_false = int((i3[s3.numpy()] != i2[s2.numpy()]).any(1).sum())
        print("%s: %d / %d voxels, %d pairs, %d false matches" % (tag, i3.shape[0], i2.shape[0],
                                                                   s3.shape[0], n_false))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
