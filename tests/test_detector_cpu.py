"""msmdfusion_amd.detector: the two configs' `model` dicts build through the DETECTORS
registry with the reference's attribute / checkpoint-key names (no GPU needed)."""
import json
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _reference_dicts():
    """The `model` dicts of the two reference configs: msmdfusion_amd.configs restates them
    (python tuples kept), tests/golden/reference_configs.json holds what the reference files
    themselves evaluate to -- equal up to tuple/list (also pinned by test_boundary.py)."""
    from msmdfusion_amd import configs as C
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_configs.json")))
    norm = lambda o: json.loads(json.dumps(o))
    assert norm(C.MSMDFUSION_LC) == fx["MSMDFusion_nusc_voxel_LC"]
    assert norm(C.TRANSFUSION_L) == fx["transfusion_nusc_voxel_L"]
    return {"MSMDFusion_nusc_voxel_LC": C.MSMDFUSION_LC, "transfusion_nusc_voxel_L": C.TRANSFUSION_L}


FX = _reference_dicts()


def test_detectors_build_from_the_reference_model_dicts():
    from msmdfusion_amd import configs as C
    from msmdfusion_amd.detector import build_detector, freeze_lidar_components
    from msmdfusion_amd.registry import DETECTORS
    assert "MSMDFusionDetector" in DETECTORS and "TransFusionDetector" in DETECTORS
    # the reference configs' model dicts, unchanged
    lc = build_detector(FX["MSMDFusion_nusc_voxel_LC"]["model"])
    tl = build_detector(FX["transfusion_nusc_voxel_L"]["model"])
    assert type(lc).__name__ == "MSMDFusionDetector" and type(tl).__name__ == "TransFusionDetector"
    keys = set(lc.state_dict().keys())
    for k in ("pts_middle_encoder.conv_input.0.weight",
              "pts_middle_encoder.encoder_layers.encoder_layer1.0.conv1.weight",
              "multimodal_middle_encoder.gate_control.0.0.weight",
              "multimodal_middle_encoder.downscale_blocks.stage_4.0.weight",
              "conv1x1_blocks.0.0.weight", "conv1x1_blocks.2.1.running_mean",
              "score_net.0.weight", "bev_fusion.conv1x1.0.weight",
              "pts_backbone.blocks.0.0.weight", "pts_neck.deblocks.1.0.weight"):
        assert k in keys, k
    assert not any(k.startswith("_") for k in keys)          # helper objects add no keys
    assert lc.spatial_shapes[0] == [41, 1440, 1440] and lc.fps_num_list == [2048] * 4
    assert lc.pts_voxel_layer.max_num_points == 10 and lc.with_pts_neck and not lc.with_pts_bbox
    assert set(tl.state_dict().keys()) == {k for k in keys if k.startswith(
        ("pts_middle_encoder.", "pts_backbone.", "pts_neck."))}
    # rows=False: the torch / MIOpen dense modules, same keys
    plain = build_detector(dict(FX["MSMDFusion_nusc_voxel_LC"]["model"], rows=False))
    assert set(plain.state_dict().keys()) == keys
    plain.load_state_dict(lc.state_dict())
    # with the head and its train / test cfg (configs/MSMDFusion_nusc_voxel_LC.py:207-268)
    full = build_detector(dict(FX["MSMDFusion_nusc_voxel_LC"]["model"],
                               pts_bbox_head=dict(C._PTS_BBOX_HEAD)),
                          train_cfg=dict(pts=dict(C._TRAIN_CFG_PTS)),
                          test_cfg=dict(pts=dict(C._TEST_CFG_PTS)))
    assert full.with_pts_bbox and "pts_bbox_head.shared_conv.weight" in full.state_dict()
    # freeze_lidar_components (tools/train.py:185-219)
    trained = freeze_lidar_components(lc)
    assert trained and not any("pts_middle_encoder" in n for n in trained)
    assert not any("blocks_2D" in n or "blocks_mix" in n for n in trained)
    assert all(not m.track_running_stats for m in lc.pts_middle_encoder.modules()
               if isinstance(m, torch.nn.BatchNorm1d))
    assert any(n.startswith("multimodal_middle_encoder.aggregation_blocks") for n in trained)


def test_detector_needs_virtual_points_or_images():
    import pytest
    from msmdfusion_amd.detector import build_detector
    lc = build_detector({k: v for k, v in FX["MSMDFusion_nusc_voxel_LC"]["model"].items()
                         if k not in ("pts_backbone", "pts_neck")})
    with pytest.raises(ValueError, match="virtual_points"):
        lc.extract_pts_feat([torch.zeros(4, 5)])
    assert lc.extract_img_feat(None, []) is None


def test_detector_says_which_modality_split_it_runs(caplog):
    """The unchanged reference config builds the exact-key split (reference_quirks=False);
    the detector names the mode once when it is built (logger `msmdfusion_amd`) and warns
    when weights are loaded into the non-reference mode -- a checkpoint trained with the
    reference saw the float32 keys' false 'mixed' voxels (MSMDFusion.py:271-272)."""
    import logging
    import warnings
    from msmdfusion_amd import detector as D
    cfg = {k: v for k, v in FX["MSMDFusion_nusc_voxel_LC"]["model"].items()
           if k not in ("pts_backbone", "pts_neck")}
    D._WARNED.clear()
    with caplog.at_level(logging.INFO, logger="msmdfusion_amd"):
        plain = D.build_detector(cfg)
        D.build_detector(cfg)                                   # (once per process and mode)
        quirks = D.build_detector(dict(cfg, reference_quirks=True))
    said = [r.getMessage() for r in caplog.records if "reference_quirks" in r.getMessage()]
    assert len(said) == 2
    assert "reference_quirks=False" in said[0] and "exact integer keys" in said[0]
    assert "reference_quirks=True" in said[1] and "float32 keys" in said[1]
    sd = plain.state_dict()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        quirks.load_state_dict(sd)                              # the reference mode: silent
        assert not [x for x in w if "reference_quirks" in str(x.message)]
        plain.load_state_dict(sd)
        plain.load_state_dict(sd)                               # said once
        assert len([x for x in w if "reference_quirks=False" in str(x.message)]) == 1
