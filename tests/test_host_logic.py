"""CPU-side checks: the C-ABI library loads and exports every declared symbol,
the plugin surface has the reference's names, and the device-agnostic host
logic (geometry / prepare / depth prep in torch ops) reproduces the golden
vectors.  No compute call reaches the HIP library here (no GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from tests.conftest import ROOT, load_golden
from veon_amd import _lib, depth_ops, lss_prepare, synthetic
from veon_amd.models import NECKS, build_neck
from veon_amd.models.necks import (LSSViewTransformer,
                                   LSSViewTransformerBEVDepth,
                                   LSSViewTransformerBEVStereo,
                                   LSSViewTransformerRaw)
from veon_amd.ops.bev_pool_v2 import (QuickCumsumCuda, TRTBEVPoolv2,
                                      bev_pool_v2)
from veon_amd.ops.bev_pool_v2 import bev_pool_v2_ext


def _header_symbols():
    names = set()
    inc = os.path.join(ROOT, 'include')
    for fn in os.listdir(inc):
        if fn.endswith('.h'):
            src = open(os.path.join(inc, fn)).read()
            src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
            names |= set(re.findall(r'\b(veon_[a-z0-9_]+)\s*\(', src))
    return names


def test_library_exports_every_declared_symbol():
    from veon_amd import build
    build.build()
    declared = _header_symbols()
    assert declared, 'no declarations found in include/*.h'
    # the Python binding knows every declared entry point and nothing else
    assert set(_lib.declared_symbols()) == declared
    # both flavours of the library (bf16 / fp16 operands) export all of them
    for flavour, path in _lib.LIB_PATHS.items():
        lib = ctypes.CDLL(path)
        for name in declared:
            assert hasattr(lib, name), '%s lacks %s' % (os.path.basename(path), name)
        lib.veon_abi_version.restype = ctypes.c_int
        assert lib.veon_abi_version() == 2          # host-only calls, no GPU needed
        assert lib.veon_half_mode() == (1 if flavour == 'fp16' else 0)


def test_half_flavour_switch():
    from veon_amd import half
    assert half.dtype() == torch.bfloat16 and half.name() == 'bf16'   # default
    with half.use('fp16'):
        assert half.dtype() == torch.float16 and half.is_half(torch.float16)
        assert not half.is_half(torch.bfloat16) and not half.is_half(None)
        assert _lib.lib().veon_half_mode() == 1
    assert half.dtype() == torch.bfloat16 and _lib.lib().veon_half_mode() == 0
    with pytest.raises(ValueError):
        half.set_half_dtype(torch.float32)
    # operands of the other flavour are refused, never reinterpreted
    _lib.require_half(torch.zeros(2, dtype=torch.bfloat16), None)
    with pytest.raises(_lib.VeonHipError):
        _lib.require_half(torch.zeros(2, dtype=torch.float16))


def test_ops_refuse_cpu_tensors():
    z = torch.zeros(1)
    i = torch.zeros(1, dtype=torch.int32)
    with pytest.raises(_lib.VeonHipError):
        bev_pool_v2_ext.bev_pool_v2_forward(z, z, z, i, i, i, i, i)
    with pytest.raises(_lib.VeonHipError):
        bev_pool_v2(torch.zeros(1, 1, 1, 1, 1), torch.zeros(1, 1, 1, 1, 1), i, i,
                    i, (1, 1, 1, 1, 1), i, i)


def test_ext_validates_dtypes():
    z = torch.zeros(1)
    i = torch.zeros(1, dtype=torch.int32)
    with pytest.raises(TypeError):
        bev_pool_v2_ext.bev_pool_v2_forward(z.double(), z, z, i, i, i, i, i)
    with pytest.raises(TypeError):
        bev_pool_v2_ext.bev_pool_v2_forward(z, z, z, i.long(), i, i, i, i)


def test_plugin_surface_names():
    for cls in (LSSViewTransformer, LSSViewTransformerBEVDepth,
                LSSViewTransformerBEVStereo, LSSViewTransformerRaw):
        assert NECKS.get(cls.__name__) is cls
        for m in ('create_grid_infos', 'create_frustum', 'get_lidar_coor',
                  'init_acceleration_v2', 'voxel_pooling_v2',
                  'voxel_pooling_prepare_v2', 'pre_compute',
                  'view_transform_core', 'view_transform', 'forward'):
            assert callable(getattr(cls, m)), (cls, m)
    for m in ('downsample_depth', 'get_two_hot_depth', 'get_one_hot_depth'):
        assert callable(getattr(LSSViewTransformerRaw, m))
    assert callable(bev_pool_v2) and issubclass(QuickCumsumCuda, torch.autograd.Function)
    assert hasattr(TRTBEVPoolv2, 'symbolic')
    with pytest.raises(KeyError):
        build_neck(dict(type='NoSuchNeck'))


def test_veon_config_builds_unchanged():
    """The img_view_transformer dict of
    configs/veon/veon-temporal-base-512x1408-dav2-nodepthcache.py:71-82."""
    cfg = dict(type='LSSViewTransformerRaw', grid_config=synthetic.GRID_VEON,
               input_size=(512, 1408), sid=False, collapse_z=False,
               out_channels=256, downsample=16, mode='nuscenes',
               loss_depth_weight=0.05, ds_feat=[2, 2, 2])
    vt = build_neck(cfg)
    assert vt.D == 88 and tuple(vt.frustum.shape) == (88, 32, 88, 3)
    assert vt.grid_size.tolist() == [200.0, 200.0, 16.0]
    assert vt.out_channels == 256 and vt.accelerate is False and vt.initial_flag
    assert vt.mode == 'nuscenes' and vt.use_ds


def test_bevdepth_builds_and_mlp_input():
    vt = build_neck(dict(
        type='LSSViewTransformerBEVDepth', grid_config=synthetic.GRID_BEVDET,
        input_size=(256, 704), in_channels=16, out_channels=8,
        depthnet_cfg=dict(use_dcn=False, aspp_mid_channels=8)))
    assert vt.D == 59 and vt.grid_size.tolist() == [128.0, 128.0, 1.0]
    rig = synthetic.make_rig(2, 6, (256, 704))
    mlp = vt.get_mlp_input(*synthetic.rig_inputs(rig))
    assert mlp.shape == (2, 6, 27)
    assert torch.equal(mlp[0, 1, 15:], rig['sensor2ego'][0, 1, :3, :].reshape(-1))
    assert mlp[0, 0, 0] == rig['intrins'][0, 0, 0, 0]


@pytest.mark.parametrize('name', ['lss_small', 'lss_small_b2', 'lss_mid'])
def test_host_prepare_reproduces_reference(name):
    g = load_golden(name)
    grid = {'x': list(g['grid_x']), 'y': list(g['grid_y']),
            'z': list(g['grid_z']), 'depth': list(g['grid_depth'])}
    vt = LSSViewTransformerRaw(grid_config=grid,
                               input_size=tuple(int(v) for v in g['input_size']),
                               out_channels=4, collapse_z=False)
    assert np.array_equal(vt.frustum.numpy(), g['frustum'])
    assert np.array_equal(vt.grid_lower_bound.numpy(), g['grid_lower_bound'])
    assert np.array_equal(vt.grid_interval.numpy(), g['grid_interval'])
    assert np.array_equal(vt.grid_size.numpy(), g['grid_size'])
    inp = [torch.from_numpy(g[k]) for k in ('sensor2ego', 'ego2global', 'intrins',
                                            'post_rots', 'post_trans', 'bda')]
    coor = vt.get_lidar_coor(*inp)
    assert np.array_equal(coor.numpy(), g['coor'])
    out = vt.voxel_pooling_prepare_v2(coor)
    for got, key in zip(out, ('ranks_bev', 'ranks_depth', 'ranks_feat',
                              'interval_starts', 'interval_lengths')):
        assert got.dtype == torch.int32 and got.is_contiguous()
        assert np.array_equal(got.numpy(), g[key]), key
    ds = vt.downsample_depth(torch.from_numpy(g['metric_depth']), 8)
    assert np.array_equal(ds.numpy(), g['ds_depth'])
    th = vt.get_two_hot_depth(ds)
    assert th.shape == g['two_hot'].shape
    np.testing.assert_allclose(th.numpy(), g['two_hot'], rtol=2e-6, atol=1e-9)
    # downsample=True branch == explicit two-step
    th2 = vt.get_two_hot_depth(torch.from_numpy(g['metric_depth'][:, :, ::1, ::1]),
                               downsample=False)
    assert th2.shape[2] == vt.D


def test_prepare_returns_none_when_empty():
    coor = torch.full((1, 1, 2, 2, 2, 3), 1e6)
    lower, interval, size = (torch.tensor(v, dtype=torch.float32) for v in
                             ([0, 0, 0], [1, 1, 1], [4, 4, 2]))
    assert lss_prepare.voxel_pooling_prepare_v2(coor, lower, interval, size) == \
        (None,) * 5


def test_one_hot_depth_is_argmax_of_two_hot():
    vt = LSSViewTransformerRaw(grid_config=synthetic.GRID_VEON,
                               input_size=(64, 176), out_channels=4,
                               collapse_z=False)
    d = 1.0 + 43.0 * torch.rand(1, 2, 4, 11)
    one = vt.get_one_hot_depth(d)
    two = vt.get_two_hot_depth(d)
    assert one.shape == two.shape == (1, 2, vt.D, 4, 11)
    inside = one.sum(2) > 0
    assert torch.equal(one.argmax(2)[inside], two.argmax(2)[inside])


def test_synthetic_rig_is_deterministic_and_sane():
    a = synthetic.make_rig(1, 6, (256, 704))
    b = synthetic.make_rig(1, 6, (256, 704))
    for k in a:
        assert torch.equal(a[k], b[k])
    r = a['sensor2ego'][0, :, :3, :3]
    eye = torch.eye(3).expand(6, 3, 3)
    assert torch.allclose(r @ r.transpose(1, 2), eye, atol=1e-6)
    assert a['post_rots'][0, 0, 0, 0] == pytest.approx(0.44)
    assert a['post_trans'][0, 0].tolist() == [0.0, -140.0, 0.0]


def test_align_body_state_dict_names_and_bn_fold():
    """layers_3d_body.* keys follow mmcv's ConvModule layout so a VEON checkpoint
    loads unchanged; the eval-mode BN fold equals BatchNorm3d."""
    import torch
    from veon_amd.models.semantic_net.align_net_body import AlignBody3D, ConvModule3d
    body = AlignBody3D(embed_dim=64, layer_depth=2)
    keys = set(body.state_dict())
    for i in range(2):
        for c in ('conv1', 'conv2'):
            assert 'layers_3d_body.%d.%s.conv.weight' % (i, c) in keys
            for n in ('weight', 'bias', 'running_mean', 'running_var'):
                assert 'layers_3d_body.%d.%s.bn.%s' % (i, c, n) in keys
    assert not any(k.endswith('conv.bias') for k in keys)
    torch.manual_seed(0)
    m = ConvModule3d(64, 64, act=False).eval()
    m.bn.running_mean.normal_()
    m.bn.running_var.uniform_(0.5, 2.0)
    m.bn.weight.data.uniform_(0.5, 2.0)
    m.bn.bias.data.normal_()
    x = torch.randn(1, 64, 2, 4, 4)
    scale = m.bn.weight / torch.sqrt(m.bn.running_var + m.bn.eps)
    shift = m.bn.bias - m.bn.running_mean * scale
    with torch.no_grad():
        want = m(x)
        got = m.conv(x) * scale.view(1, -1, 1, 1, 1) + shift.view(1, -1, 1, 1, 1)
    assert torch.allclose(got, want, atol=1e-5)


def test_align_body_and_heads_reproduce_reference_vectors():
    """ResBlock3D / PredHead3DOcc / PredHead3DSem mirrors load the reference
    modules' state_dicts strictly and reproduce the outputs of the reference's
    own align_net_occ3d.py (run with a ConvModule stand-in, see
    oracle/tools/gen_golden_body.py): pins the wiring -- norm / act / bias per
    conv, identity add, final ReLU, sigmoid - 0.5."""
    import torch
    from tests.conftest import load_golden
    from veon_amd.models.semantic_net.align_net_body import (PredHead3DOcc,
                                                            PredHead3DSem, ResBlock3D)
    g = load_golden('align_body_tiny')

    def sd(tag):
        return {k[len(tag) + 1:]: torch.from_numpy(g[k]) for k in g
                if k.startswith(tag + '/')}
    blk, occ, sem = ResBlock3D(64, 64).eval(), PredHead3DOcc(64, 2).eval(), \
        PredHead3DSem(64, 24).eval()
    blk.load_state_dict(sd('block'), strict=True)
    occ.load_state_dict(sd('occ'), strict=True)
    sem.load_state_dict(sd('sem'), strict=True)
    x = torch.from_numpy(g['x'])
    with torch.no_grad():
        y = blk(x)
        assert torch.allclose(y, torch.from_numpy(g['block_out']), atol=1e-5)
        assert torch.allclose(occ(y), torch.from_numpy(g['occ_out']), atol=1e-5)
        assert torch.allclose(sem(y), torch.from_numpy(g['sem_out']), atol=1e-5)


def test_fused_semantic_inference_equals_reference_order():
    """Classify-then-upsample == upsample-then-classify (both maps are linear,
    interpolation weights sum to one): san_in_veon_temporal.py:196-201, 257-259."""
    import torch
    from veon_amd.models.semantic_net import (semantic_inference_3d,
                                              semantic_inference_3d_fused)
    torch.manual_seed(0)
    W = torch.randn(17, 64) * 3
    feat = torch.rand(2, 64, 3, 5, 6) - 0.5
    a = semantic_inference_3d(W, feat, (6, 10, 12))
    b = semantic_inference_3d_fused(W, feat, (6, 10, 12))
    assert a.shape == b.shape == (2, 17, 6, 10, 12)
    assert torch.allclose(a, b, atol=2e-5, rtol=1e-5)


def test_bevstereo_cost_volume_matches_reference_vectors():
    """DepthNet(stereo=True): gen_grid / calculate_cost_volumn reproduce the
    reference's own methods (view_transformer.py:543-601, vectors from
    oracle/tools/gen_golden_stereo.py); the stereo forward runs end to end."""
    import torch
    from tests.conftest import load_golden
    from veon_amd.models.necks.view_transformer import DepthNet
    g = load_golden('stereo_cost_volume')
    t = {k: torch.from_numpy(v) for k, v in g.items() if v.ndim > 0}
    net = DepthNet(16, 32, 8, 7, use_dcn=False, use_aspp=False, stereo=True,
                   bias=float(g['bias'])).eval()
    metas = dict(frustum=t['frustum'], post_trans=t['post_trans'],
                 post_rots=t['post_rots'], k2s_sensor=t['k2s_sensor'],
                 intrins=t['intrins'], cv_feat_list=[t['prev'], t['curr']],
                 downsample=16, cv_downsample=4)
    D, H, W, _ = t['frustum'].shape
    with torch.no_grad():
        grid = net.gen_grid(metas, 1, 2, D, H, W, H * 4, W * 4)
        cv = net.calculate_cost_volumn(metas)
    assert torch.allclose(grid, t['grid'], atol=1e-5)
    assert torch.allclose(cv, t['cost_volume'], atol=1e-6)
    # forward with the cost volume (features at downsample 16 = cv res / 4) and
    # with the "no previous frame" zeros branch
    x = torch.randn(2, 16, 2, 3)
    mlp = torch.randn(1, 2, 27)
    with torch.no_grad():
        y = net(x, mlp, metas)
        metas0 = dict(metas, cv_feat_list=[None, t['curr']])
        y0 = net(x, mlp, metas0)
    assert y.shape == y0.shape == (2, 7 + 8, 2, 3)


def test_fusion_layer_and_decoder_state_dict_match_reference_vectors():
    """CatFusionLift reproduces the reference layer (layers.py:154-199) and the
    AlignNetOcc3D mirror loads the reference decoder's state_dict strictly
    (oracle/tools/gen_golden_align_net.py)."""
    import torch
    from tests.conftest import load_golden
    from veon_amd.models.semantic_net import AlignNetOcc3D
    g = load_golden('align_net_tiny')
    t = {k: torch.from_numpy(v) for k, v in g.items()}
    net = AlignNetOcc3D(clip_dim=32, hsa_dim=16, embed_dim=64, clip_outdim=24,
                        layer_lifting_map=['2->0->0'], fusion_type='cat_fusion',
                        layer_depth=2).eval()
    net.load_state_dict({k[3:]: v for k, v in t.items() if k.startswith('sd/')},
                        strict=True)
    with torch.no_grad():
        got = net.fusion_layers['layer_0'](t['supp'], t['clip2'], (4, 11))
    assert torch.allclose(got, t['cat_fusion_out'], atol=1e-5)


def _hsa_from_golden(g, device='cpu'):
    import torch
    from veon_amd.models.semantic_net.hsa_network import HighresSideAdaptorNetwork
    t = {k: torch.from_numpy(v).to(device) for k, v in g.items()}
    net = HighresSideAdaptorNetwork.build(
        dim=64, clip_dim=32, mlp_dim=64, input_size=(32, 48), patch_shape=(8, 8),
        num_heads=2, fusion_map=('0->1->1', '1->2->-1'), manip_dim_head=8,
        manip_attn_layers=3, manip_add_layers=2, manip_supp_dim=16)
    net.load_state_dict({k[3:]: v for k, v in t.items() if k.startswith('sd/')},
                        strict=True)
    return net.to(device).eval(), t


def test_hsa_network_matches_reference_vectors():
    """HighresSideAdaptorNetwork mirror vs the reference's own
    highres_side_adaptor.py (oracle/tools/gen_golden_hsa.py): strict state_dict,
    ConvBlock, attention biases and supp features."""
    import torch
    from tests.conftest import load_golden
    net, t = _hsa_from_golden(load_golden('hsa_tiny'))
    with torch.no_grad():
        cb = net.hsa_net_body[0].ff(t['tokens'], (4, 6))
        _, attns, supp = net(t['image'], {1: t['clip1'], 2: t['clip2']})
    assert torch.allclose(cb, t['convblock_out'], atol=1e-5)
    assert torch.allclose(attns, t['attns'], atol=2e-4, rtol=1e-4)
    assert torch.allclose(supp, t['supp'], atol=1e-5)


def test_conv_tile_choice_rule():
    """The conv launcher's tile (csrc/conv3d.hip: conv_pick_tile), host-only: big grids
    keep the measured choices, under-filled grids take the small tiles the sweep found
    (profiles/r02_conv_tile_sweep.txt)."""
    lib = _lib.lib()

    def tile(kd, B, Z, Y, X, Cin, Cout, stride=1):
        v = lib.veon_conv_tile_choice(kd, B, Z, Y, X, Cin, Cout, stride)
        return (v & 0xffff, v >> 16)
    assert tile(3, 1, 8, 100, 100, 256, 256) == (336, 256)       # the Conv3d body
    assert tile(1, 6, 1, 9, 25, 768, 768, 2) == (64, 128)        # DPT stride-2 conv: 48 -> 204 tiles
    assert tile(1, 6, 1, 32, 88, 384, 384) == (192, 192)         # HSA at 256x704: grid fills
    assert tile(1, 6, 1, 16, 44, 384, 384)[0] <= 128             # a quarter of that: small tile
    assert tile(1, 6, 1, 144, 400, 128, 64)[1] == 64             # 64-feature class
    assert tile(1, 6, 1, 72, 200, 128, 128) == (128, 128)        # large 128-feature conv
    assert lib.veon_conv_tile_choice(1, 6, 1, 9, 25, 100, 128, 1) == -1   # Cin % 64


def test_no_low_half_op_sel_packed_f32_beside_mfma_in_the_vit_kernels():
    """v_pk_*_f32 with an op_sel that feeds the LOW half from the HIGH dword read
    its operand as 0.0 in lanes 48-63 when the same wave issues MFMAs (stand-alone
    reproducer tools/ubench/pk_opsel_repro.hip); hipcc emits that form on its own from
    scalar code.  The ISA of vit_block.hip (GEMM, attention: where it happened) must not
    contain it in any kernel that has MFMAs.  (tools/check_pk_opsel.py checks every source
    in both flavours; that takes minutes.)"""
    import importlib.util
    import shutil
    if shutil.which('hipcc') is None:
        pytest.skip('no hipcc')
    spec = importlib.util.spec_from_file_location(
        'check_pk_opsel', os.path.join(ROOT, 'tools', 'check_pk_opsel.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.offenders(only={'vit_block.hip'}) == {}
