"""Import the reference's own LSS view-transformer Python in THIS container.

TEST INFRASTRUCTURE (fixture generation only; never runs on the GPU box --
/root/reference does not exist there).  The reference modules are loaded by
file path from /root/reference, unmodified, with stub modules standing in for
third-party packages that are not installed here (mmcv, mmdet) and for the
CUDA-only extension.  Nothing is copied: the stubs contain no
reference code, only names.
"""
import importlib.util
import os
import sys
import types

import torch
import torch.nn as nn

REF = os.environ.get('VEON_REFERENCE', '/root/reference')


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__path__ = []  # behave as a package so relative imports resolve
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class _Registry:
    def __init__(self):
        self.classes = {}

    def register_module(self, *a, **k):
        def deco(cls):
            self.classes[cls.__name__] = cls
            return cls
        return deco


def _force_fp32(*a, **k):
    def deco(fn):
        return fn
    return deco


def install_stubs(bev_pool_v2_cpu):
    """``bev_pool_v2_cpu``: the CPU stand-in for the CUDA-only op, with the
    reference signature (bev_pool.py:86-92)."""
    necks = _Registry()
    _mod('mmcv')
    _mod('mmcv.cnn', build_conv_layer=lambda *a, **k: None)
    _mod('mmcv.runner', BaseModule=nn.Module, force_fp32=_force_fp32)
    _mod('mmdet')
    _mod('mmdet.models')
    _mod('mmdet.models.backbones')
    _mod('mmdet.models.backbones.resnet', BasicBlock=nn.Module)
    _mod('mmdet3d')
    _mod('mmdet3d.ops')
    _mod('mmdet3d.ops.bev_pool_v2')
    _mod('mmdet3d.ops.bev_pool_v2.bev_pool', bev_pool_v2=bev_pool_v2_cpu)
    _mod('mmdet3d.models')
    _mod('mmdet3d.models.builder', NECKS=necks)
    _mod('mmdet3d.models.necks')
    return necks


def load(relpath, modname):
    path = os.path.join(REF, relpath)
    spec = importlib.util.spec_from_file_location(modname, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[modname] = m
    spec.loader.exec_module(m)
    return m


def load_view_transformers(bev_pool_v2_cpu):
    """-> (view_transformer_raw module, view_transformer module)."""
    if not os.path.isdir(REF):
        raise RuntimeError('reference tree not present at %s' % REF)
    install_stubs(bev_pool_v2_cpu)
    import warnings
    warnings.filterwarnings('ignore', message='.*torch.range.*')
    raw = load('mmdet3d/models/necks/view_transformer_raw.py',
               'mmdet3d.models.necks.view_transformer_raw')
    vt = load('mmdet3d/models/necks/view_transformer.py',
              'mmdet3d.models.necks.view_transformer')
    return raw, vt
