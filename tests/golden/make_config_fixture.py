#!/usr/bin/env python
"""Dump the hot-path part of the reference's two model configs to JSON (values
only -- the dicts are facts).  Runs in the build container, where
/root/reference exists; the result is committed as reference_configs.json and
pins msmdfusion_amd/configs.py (tests/test_boundary.py).

    python tests/golden/make_config_fixture.py
"""
import json
import os

REF = "/root/reference/configs"
HOT_KEYS = ["type", "spatial_shapes", "downscale_factors", "fps_num_list", "radius_list",
            "max_cluster_samples_list", "dist_thresh_list", "pts_voxel_layer", "pts_voxel_encoder",
            "pts_middle_encoder", "multimodal_middle_encoder", "pts_backbone", "pts_neck"]


def load(name):
    ns = {}
    exec(compile(open(os.path.join(REF, name)).read(), name, "exec"), ns)   # plain-Python config
    model = {k: ns["model"][k] for k in HOT_KEYS if k in ns["model"]}
    return dict(model=model, samples_per_gpu=ns["data"]["samples_per_gpu"],
                point_cloud_range=ns["point_cloud_range"], voxel_size=ns["voxel_size"],
                optimizer=ns["optimizer"],
                freeze_lidar_components=ns.get("freeze_lidar_components", False))


def main():
    out = {"MSMDFusion_nusc_voxel_LC": load("MSMDFusion_nusc_voxel_LC.py"),
           "transfusion_nusc_voxel_L": load("transfusion_nusc_voxel_L.py")}
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_configs.json")
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    print("wrote", dst)


if __name__ == "__main__":
    main()
