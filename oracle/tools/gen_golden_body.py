"""Golden vectors for the Conv3d body / prediction heads.

TEST INFRASTRUCTURE (fixture generation, run in the build container only).
Loads the reference's own ``align_net_occ3d.py`` by file path, unmodified, and
runs its ``ResBlock3D``, ``PredHead3DOcc`` and ``PredHead3DSem``
(mmdet3d/models/semantic_net/side_adapter/align_net_occ3d.py:363-534) on seeded
inputs.  mmcv is not installed: ``ConvModule`` is a stand-in written here from
mmcv's documented behaviour (order conv -> norm -> act, sub-module names
``conv`` / ``bn`` / ``activate``, ``bias`` honoured as given) -- so these
vectors pin the reference's WIRING (which conv has a norm / activation / bias,
the identity add, the final ReLU, ``sigmoid - 0.5``), not mmcv itself.

    python oracle/tools/gen_golden_body.py   ->  tests/golden/align_body_tiny.npz
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(HERE))


class ConvModuleStandIn(nn.Module):
    """mmcv.cnn.ConvModule as the reference uses it (Conv3d, BN3d, ReLU)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0,
                 bias='auto', conv_cfg=None, norm_cfg=None,
                 act_cfg=dict(type='ReLU'), **kw):
        super().__init__()
        assert conv_cfg is None or conv_cfg['type'] == 'Conv3d'
        with_norm = norm_cfg is not None
        if bias == 'auto':
            bias = not with_norm
        self.conv = nn.Conv3d(in_channels, out_channels, kernel_size, stride, padding,
                              bias=bias)
        self.bn = None
        if with_norm:
            assert norm_cfg['type'] == 'BN3d'
            self.bn = nn.BatchNorm3d(out_channels)
        else:
            del self.bn
        self.activate = None
        if act_cfg is not None:
            assert act_cfg['type'] == 'ReLU'
            self.activate = nn.ReLU(inplace=act_cfg.get('inplace', True))
        else:
            del self.activate
        self._norm, self._act = with_norm, act_cfg is not None

    def forward(self, x):
        x = self.conv(x)
        if self._norm:
            x = self.bn(x)
        if self._act:
            x = self.activate(x)
        return x


def load_reference():
    ref_import.install_stubs(lambda *a, **k: None)
    ref_import._mod('mmcv.cnn.bricks')
    ref_import._mod('mmcv.cnn.bricks.conv_module', ConvModule=ConvModuleStandIn)
    ref_import._mod('mmdet3d.models.necks.view_transformer_raw',
                    LSSViewTransformerRaw=object)
    ref_import._mod('mmdet3d.utils')
    ref_import._mod('mmdet3d.utils.vis', vis_occ=None)
    pkg = 'refsem'
    ref_import._mod(pkg)
    ref_import._mod(pkg + '.layers', build_fusion_layer_lift=None)
    ref_import._mod(pkg + '.side_adapter')
    return ref_import.load(
        'mmdet3d/models/semantic_net/side_adapter/align_net_occ3d.py',
        pkg + '.side_adapter.align_net_occ3d')


def randomise(mod, gen):
    for m in mod.modules():
        if isinstance(m, nn.BatchNorm3d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=gen) * 0.2)
            m.running_var.copy_(torch.rand(m.num_features, generator=gen) + 0.5)
            m.weight.data.copy_(torch.rand(m.num_features, generator=gen) + 0.5)
            m.bias.data.copy_(torch.randn(m.num_features, generator=gen) * 0.2)


def main():
    ref = load_reference()
    torch.manual_seed(0)
    gen = torch.Generator().manual_seed(1)
    C = 64
    blk = ref.ResBlock3D(channels_in=C, channels_out=C).eval()
    occ = ref.PredHead3DOcc(channels_in=C, channels_out=2).eval()
    sem = ref.PredHead3DSem(channels_in=C, channels_out=24).eval()
    for m in (blk, occ, sem):
        randomise(m, gen)
    x = torch.randn(2, C, 3, 6, 7, generator=gen)
    with torch.no_grad():
        y = blk(x)
        o = occ(y)
        s = sem(y)
    out = {'x': x.numpy(), 'block_out': y.numpy(), 'occ_out': o.numpy(),
           'sem_out': s.numpy()}
    for tag, m in (('block', blk), ('occ', occ), ('sem', sem)):
        for k, v in m.state_dict().items():
            out['%s/%s' % (tag, k)] = v.numpy()
    path = os.path.join(ROOT, 'tests', 'golden', 'align_body_tiny.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, {k: v.shape for k, v in out.items() if not '/' in k})
    print(sorted(k for k in out if k.startswith('sem/'))[:8])


if __name__ == '__main__':
    main()
