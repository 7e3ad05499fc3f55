#!/usr/bin/env python
"""Generates tests/golden/loader_vectors.npz by RUNNING the reference's own loader classes
(mmdet3d/datasets/pipelines/my_loading_multi_proj.py: LoadForeground2D :14-161,
LoadForeground2DFromMultiSweeps :163-338) on the synthetic files tests/foreground_files.py
writes.  The module itself cannot be imported here (mmcv / mmdet absent): the two class
definitions are taken from the reference FILE at run time (ast) and executed as they stand;
the registry decorator and the point-class factory they name are given inert stand-ins
(`PIPELINES.register_module()` returns the class; `get_points_type` returns a holder with
`.tensor`), neither of which takes part in the arithmetic.  Relative paths are used, as the
reference's path handling requires (:126 drops a leading '/').  Only outputs are stored.
"""
import ast
import copy
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import foreground_files as FF  # noqa: E402

REF = "/root/reference/mmdet3d/datasets/pipelines/my_loading_multi_proj.py"
REF_LOADING = "/root/reference/mmdet3d/datasets/pipelines/loading.py"
OUT = os.path.join(ROOT, "tests", "golden", "loader_vectors.npz")


class _Registry:
    def register_module(self):
        return lambda cls: cls


class _Points:
    def __init__(self, tensor, points_dim=None):
        self.tensor = torch.as_tensor(np.asarray(tensor), dtype=torch.float32)


def reference_classes():
    tree = ast.parse(open(REF).read())
    want = ("LoadForeground2D", "LoadForeground2DFromMultiSweeps")
    body = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name in want]
    ns = {"np": np, "os": os, "torch": torch, "PIPELINES": _Registry(),
          "get_points_type": lambda kind: _Points}
    exec(compile(ast.Module(body=body, type_ignores=[]), REF, "exec"), ns)
    return ns


class _FileClient:                       # mmcv.FileClient(backend='disk')
    def __init__(self, **kw):
        pass

    def get(self, path):
        return open(path, "rb").read()


class _PointsBox(_Points):               # BasePoints: holder + new_point / cat / indexing
    def new_point(self, data):
        return _PointsBox(data)

    def cat(self, items):
        return _PointsBox(torch.cat([p.tensor for p in items], 0).numpy())

    def __getitem__(self, item):
        if isinstance(item, np.ndarray) and item.dtype == np.bool_:
            item = torch.from_numpy(item)
        return _PointsBox(self.tensor[item].numpy())


def reference_multisweep_class():
    import types
    tree = ast.parse(open(REF_LOADING).read())
    body = [n for n in tree.body if isinstance(n, ast.ClassDef)
            and n.name == "LoadPointsFromMultiSweeps"]
    ns = {"np": np, "PIPELINES": _Registry(), "BasePoints": _PointsBox,
          "mmcv": types.SimpleNamespace(FileClient=_FileClient,
                                        check_file_exist=lambda p: None)}
    exec(compile(ast.Module(body=body, type_ignores=[]), REF_LOADING, "exec"), ns)
    return ns["LoadPointsFromMultiSweeps"]


def main():
    ns = reference_classes()
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            results = FF.make_tree("data", seed=5)
            single = ns["LoadForeground2D"]()(copy.deepcopy(results))
            info = single["foreground2D_info"]
            for key in ("fg_pixels", "fg_points", "fg_real_pixels", "fg_real_points"):
                for cam, a in enumerate(info[key]):
                    out["single_%s_%d" % (key, cam)] = np.asarray(a)
            multi_loader = ns["LoadForeground2DFromMultiSweeps"](sweeps_num=10)
            multi = multi_loader(single)
            info = multi["foreground2D_info"]
            for key in ("fg_pixels", "fg_real_pixels", "fg_real_points"):
                for cam, a in enumerate(info[key]):
                    out["multi_%s_%d" % (key, cam)] = np.asarray(a)
            for cam, p in enumerate(info["fg_points"]):
                out["multi_fg_points_%d" % cam] = p.tensor.numpy()
            # LiDAR sweeps (loading.py:503-636) over raw .bin files of the same tree
            FF.add_lidar_files(results, seed=5)
            key = np.fromfile(results["pts_filename"], dtype=np.float32).reshape(-1, 5)
            Ref = reference_multisweep_class()
            for tag, kw in (("plain", {}), ("noclose", dict(remove_close=True)),
                            ("one", dict(sweeps_num=1, test_mode=True))):
                res = Ref(use_dim=[0, 1, 2, 3, 4], **kw)(
                    dict(copy.deepcopy(results), points=_PointsBox(key.copy())))
                out["sweeps_" + tag] = res["points"].tensor.numpy()
        finally:
            os.chdir(cwd)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
