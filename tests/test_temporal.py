"""Temporal path: the PyTorch mirrors against vectors made by
the reference's own TemporalFusionMultiFrame / TemporalDeformable
(align_net_occ3d.py:13-204) and SANInVeonTemporal.align_after_lss
(san_in_veon_temporal.py:325-365); generator oracle/tools/gen_golden_temporal.py."""
import numpy as np
import pytest
import torch

from tests.conftest import load_golden
from veon_amd.models.semantic_net import temporal_fusion as tfm


def _fusion_from_golden(g):
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith('tf/')}
    C = sd['t_final.conv.weight'].shape[0]
    seqs = len([k for k in sd if k.startswith('t_fuse_mid.t_fuse.') and k.endswith('conv.weight')])
    net = tfm.TemporalFusionMultiFrame(C, seqs=seqs).eval()
    missing = net.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return net


def test_temporal_fusion_matches_reference_vectors():
    g = load_golden('temporal_tiny')
    net = _fusion_from_golden(g)
    cur = torch.from_numpy(g['tf_cur'])
    prevs = [torch.from_numpy(g['tf_prev%d' % i]) for i in range(2)]
    with torch.no_grad():
        d = net.deform_fusion_layer.t_deform(prevs[0], cur)
        y = net(cur, prevs)
    assert torch.allclose(d, torch.from_numpy(g['deform_out']), atol=2e-5, rtol=1e-5)
    assert torch.allclose(y, torch.from_numpy(g['tf_out']), atol=2e-5, rtol=1e-5)


def test_temporal_fusion_state_dict_names():
    """the reference's parameter names, so temporal_fusion.* checkpoint keys load"""
    net = tfm.TemporalFusionMultiFrame(32, seqs=1)
    keys = set(net.state_dict())
    for k in ('t_final.conv.weight', 't_final.bn.running_mean',
              'before_fusion_layer.offset_conv.conv.weight',
              't_fuse_mid.t_fuse.0.conv.weight',
              'deform_fusion_layer.t_deform.offset_conv.0.bias',
              'deform_fusion_layer.t_deform.offset_conv.2.weight',
              'deform_fusion_layer.t_deform.key_value_proj.weight',
              'deform_fusion_layer.t_deform.query_proj.bias',
              'deform_fusion_layer.t_deform.out_proj.weight',
              'deform_fusion_layer.t_deform.final_norm.running_var'):
        assert k in keys, k
    assert 'deform_fusion_layer.t_deform.offset_conv.2.bias' not in keys


def test_single_past_frame_uses_one_fuse_conv():
    net = tfm.TemporalFusionMultiFrame(8, seqs=1).eval()
    x = torch.randn(1, 8, 2, 3, 4)
    with torch.no_grad():
        y = net(x, [torch.randn(1, 8, 2, 3, 4)])
    assert y.shape == x.shape


def test_align_after_lss_matches_reference_vectors():
    g = load_golden('temporal_tiny')
    grid = {k: [float(v) for v in g['align_grid'][i]] for i, k in enumerate('xyz')}
    ds = tuple(int(v) for v in g['align_ds'])
    occ = torch.from_numpy(g['align_in'])
    metas = [torch.from_numpy(g['align_cur2glob']), torch.from_numpy(g['align_prev2glob'])]
    out = tfm.align_after_lss(occ, metas, grid, ds)
    assert torch.allclose(out, torch.from_numpy(g['align_out']), atol=1e-4, rtol=1e-4)


def test_align_identity_transform_is_a_copy():
    grid = {'x': [-4.0, 4.0, 0.5], 'y': [-3.0, 3.0, 0.5], 'z': [-1.0, 3.0, 0.5]}
    occ = torch.randn(1, 3, 4, 6, 8)
    eye = torch.eye(4)[None, None]
    out = tfm.align_after_lss(occ, [eye, eye], grid, (2, 2, 2))
    assert torch.allclose(out, occ, atol=1e-5)


def test_depth_cache_wire_format(tmp_path):
    """layout and payload of the reference's depth cache
    (veon_depth_cache.py:146-157 writes, loading.py:1259-1262 reads)."""
    import os
    from veon_amd import depth_cache
    home = str(tmp_path)
    toks = ['ab12cd34-CAM_FRONT', 'ab12cd34-CAM_BACK_LEFT']
    depth = torch.rand(1, 2, 8, 22) * 80
    written = depth_cache.store(home, toks, depth)
    assert written == [os.path.join(home, 'ab', 'ab12cd34', t + '.tensor') for t in toks]
    # the reference's reader is a bare torch.load of each file
    raw = torch.load(written[0])
    assert raw.dtype == torch.float32 and raw.shape == (8, 22) and torch.equal(raw, depth[0, 0])
    assert os.path.getsize(written[0]) < 8 * 22 * 4 + 2048    # the map, not the batch
    assert torch.equal(depth_cache.load(home, toks), depth[0])
    # existing files are kept (the reference `continue`s), unless asked
    assert depth_cache.store(home, toks, depth * 2) == []
    assert torch.equal(depth_cache.load(home, toks), depth[0])
    assert len(depth_cache.store(home, toks, depth * 2, overwrite=True)) == 2
    # a file written the way the reference writes it is read back
    other = 'ff00aa11-CAM_BACK'
    os.makedirs(os.path.dirname(depth_cache.cache_path(home, other)))
    torch.save(depth[0][1].cpu(), depth_cache.cache_path(home, other))
    assert torch.equal(depth_cache.load(home, [other])[0], depth[0, 1])
    with pytest.raises(FileNotFoundError):
        depth_cache.load(home, ['00000000-CAM_FRONT'])
