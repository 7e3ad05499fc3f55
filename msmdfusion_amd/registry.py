"""Minimal stand-in for the mmcv / mmdet3d registries the reference configs go
through (mmdet3d/models/registry.py:1-5, mmdet3d/models/builder.py:1-63,
mmcv.cnn.build_conv_layer / build_norm_layer as used by
mmdet3d/ops/sparse_block.py:159-187).

Only the registry *mechanics* are restated (type-string -> class, kwargs
pass-through) so `configs/MSMDFusion_nusc_voxel_LC.py`'s `model=` sub-dicts for
the hot path build unchanged: dict(type='SparseEncoder', ...),
dict(type='SubMConv3d', indice_key=...), dict(type='BN1d', eps=1e-3, momentum=0.01).
"""
import inspect

from torch import nn


class Registry:
    def __init__(self, name):
        self.name = name
        self._modules = {}

    def get(self, key):
        if key not in self._modules:
            raise KeyError(f"{key} is not in the {self.name} registry")
        return self._modules[key]

    def __contains__(self, key):
        return key in self._modules

    def register_module(self, name=None, module=None, force=False):
        def _register(cls):
            key = name or cls.__name__
            if key in self._modules and not force:
                raise KeyError(f"{key} is already registered in {self.name}")
            self._modules[key] = cls
            return cls
        if module is not None:
            return _register(module)
        return _register

    def build(self, cfg, *args, **kwargs):
        if not isinstance(cfg, dict) or "type" not in cfg:
            raise TypeError(f"cfg must be a dict with a 'type' key, got {cfg!r}")
        cfg = dict(cfg)
        cls = cfg.pop("type")
        if isinstance(cls, str):
            cls = self.get(cls)
        elif not inspect.isclass(cls):
            raise TypeError(f"type must be a str or class, got {type(cls)}")
        return cls(*args, **kwargs, **cfg)


CONV_LAYERS = Registry("conv layer")
NORM_LAYERS = Registry("norm layer")
MIDDLE_ENCODERS = Registry("middle_encoder")
VOXEL_ENCODERS = Registry("voxel_encoder")
BACKBONES = Registry("backbone")
NECKS = Registry("neck")
DETECTORS = Registry("detector")

CONV_LAYERS.register_module("Conv1d", module=nn.Conv1d)
CONV_LAYERS.register_module("Conv2d", module=nn.Conv2d)
CONV_LAYERS.register_module("Conv3d", module=nn.Conv3d)
CONV_LAYERS.register_module("Conv", module=nn.Conv2d)
NORM_LAYERS.register_module("BN", module=nn.BatchNorm2d)
NORM_LAYERS.register_module("BN1d", module=nn.BatchNorm1d)
NORM_LAYERS.register_module("BN2d", module=nn.BatchNorm2d)
NORM_LAYERS.register_module("BN3d", module=nn.BatchNorm3d)

_NORM_ABBR = {"BN": "bn", "BN1d": "bn", "BN2d": "bn", "BN3d": "bn"}


def build_conv_layer(cfg, *args, **kwargs):
    """mmcv.cnn.build_conv_layer: cfg=None means Conv2d."""
    cfg = dict(type="Conv2d") if cfg is None else cfg
    return CONV_LAYERS.build(cfg, *args, **kwargs)


def build_norm_layer(cfg, num_features, postfix=""):
    """mmcv.cnn.build_norm_layer -> (name, layer); name = abbreviation +
    postfix, e.g. 'bn1' (the attribute name BasicBlock registers, which is what
    makes checkpoint keys '...bn1.weight')."""
    cfg = dict(cfg)
    layer_type = cfg.pop("type")
    requires_grad = cfg.pop("requires_grad", True)
    cfg.setdefault("eps", 1e-5)
    layer = NORM_LAYERS.get(layer_type)(num_features, **cfg)
    for p in layer.parameters():
        p.requires_grad = requires_grad
    return _NORM_ABBR.get(layer_type, "norm") + str(postfix), layer


def _register_hot_path():
    """Import the modules whose decorators fill the registries (mmdet3d does
    this from its package __init__)."""
    from . import bev, multimodal_encoder, sparse_encoder, voxel_encoder  # noqa: F401
    from .spconv import conv  # noqa: F401


def build_middle_encoder(cfg):
    _register_hot_path()
    return MIDDLE_ENCODERS.build(cfg)


def build_voxel_encoder(cfg):
    _register_hot_path()
    return VOXEL_ENCODERS.build(cfg)


def build_backbone(cfg):
    """mmdet3d.models.builder.build_backbone for the BEV tail (SECOND)."""
    _register_hot_path()
    return BACKBONES.build(cfg)


def build_detector(cfg, train_cfg=None, test_cfg=None, **injected):
    """mmdet3d.models.build_detector (msmdfusion_amd/detector.py registers the classes)."""
    from . import detector
    return detector.build_detector(cfg, train_cfg=train_cfg, test_cfg=test_cfg, **injected)


def build_neck(cfg):
    """mmdet3d.models.builder.build_neck for the BEV tail (SECONDFPN)."""
    _register_hot_path()
    return NECKS.build(cfg)
