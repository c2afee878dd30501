"""The fused tail of the occupancy path (csrc/occ_head.hip) against the reference's
op sequence: F.interpolate(trilinear, align_corners=False) of both logit volumes
(san_in_veon_temporal.py:196-211), softmax / max / threshold / where / permute of
VEONTemporal.simple_test (detectors/veon_temporal.py:219-227)."""
import pytest
import torch
import torch.nn.functional as F

from veon_amd import conv3d_ops

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def reference(sem_low, bin_low, occ_size):
    sem = F.interpolate(sem_low, size=occ_size, mode='trilinear', align_corners=False)
    binv = F.interpolate(bin_low, size=occ_size, mode='trilinear', align_corners=False)
    score, cls = torch.softmax(sem, dim=1).max(dim=1)
    keep = (score > 0.0) & (torch.softmax(binv, dim=1)[:, 0] > 0.5)
    occ = torch.where(keep, cls, torch.full_like(cls, sem.shape[1]))
    return sem, binv, occ.permute(0, 3, 2, 1).contiguous()


@pytest.mark.parametrize('B,Q,low,size', [(1, 17, (8, 100, 100), (16, 200, 200)),
                                          (2, 5, (3, 7, 5), (6, 14, 10)),
                                          (1, 40, (2, 5, 9), (5, 11, 20))])   # ratios != 2
def test_fused_tail_matches_the_op_sequence(B, Q, low, size):
    torch.manual_seed(0)
    sem_low = torch.randn(B, Q, *low, device=DEV) * 3
    bin_low = torch.randn(B, 2, *low, device=DEV)
    sem, binv, occ = conv3d_ops.occ_classify(sem_low, bin_low, size)
    rs, rb, ro = reference(sem_low, bin_low, size)
    # fp32: the kernel is built without FMA contraction, ATen's with -> last-bit level
    torch.testing.assert_close(sem, rs, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(binv, rb, rtol=1e-5, atol=1e-5)
    assert occ.shape == ro.shape and occ.dtype == torch.int64
    # labels: identical except where two logits tie at rounding level
    assert (occ != ro).float().mean().item() < 1e-4
    assert occ.max().item() <= Q


def test_strided_inputs_are_read_in_place():
    """The class logits arrive as a channels-last slice of the padded GEMM rows, the
    occupancy logits as a channel slice of an unpacked volume."""
    torch.manual_seed(1)
    B, Q, (z, y, x) = 1, 17, (4, 10, 12)
    rows = torch.randn(B, z + 2, y + 2, x + 2, 24, device=DEV)
    sem_low = rows[:, 1:-1, 1:-1, 1:-1, :Q].permute(0, 4, 1, 2, 3)
    bin_low = torch.randn(B, z, 8, y, x, device=DEV).transpose(1, 2)[:, :2]
    assert not bin_low.is_contiguous()
    assert not sem_low.is_contiguous()
    size = (8, 20, 24)
    sem, binv, occ = conv3d_ops.occ_classify(sem_low, bin_low, size)
    rs, rb, ro = reference(sem_low.contiguous(), bin_low.contiguous(), size)
    torch.testing.assert_close(sem, rs, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(binv, rb, rtol=1e-5, atol=1e-5)
    assert (occ != ro).float().mean().item() < 1e-3


def test_path_tail_uses_the_fused_kernel():
    from veon_amd import _lib
    from veon_amd.models.veon_occ import VeonOccupancyPath
    grid = {'x': [-10.0, 10.0, 1.0], 'y': [-10.0, 10.0, 1.0], 'z': [-1.0, 3.0, 1.0],
            'depth': [1.0, 13.0, 1.0]}
    net = VeonOccupancyPath(input_size=(64, 176), num_cam=2, clip_width=64, clip_layers=4,
                            clip_heads=1, clip_first_tail=2, clip_proj_dim=64, embed_dim=64,
                            n_classes=5, occ_size=(4, 20, 20), hsa_dim=64,
                            hsa_fusion_map=('0->1->1', '1->2->2'), grid_config=grid,
                            two_streams=False, clip_image=64).to(DEV).eval()
    bin_low = torch.randn(1, 2, 2, 10, 10, device=DEV)
    feat = torch.randn(1, 64, 2, 10, 10, device=DEV)
    before = _lib.CALLS.get('veon_occ_classify', 0)
    with torch.no_grad():
        out = net._classify(bin_low, feat)
        low = torch.einsum('qc,bczhw->bqzhw', net.ov_classifier_weight, feat)
    assert _lib.CALLS.get('veon_occ_classify', 0) == before + 1
    rs, rb, ro = reference(low, bin_low, (4, 20, 20))
    torch.testing.assert_close(out['sem_occ'], rs, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(out['bin_occ'], rb, rtol=1e-5, atol=1e-5)
    assert (out['occ_pred_cls'] != ro).float().mean().item() < 1e-3
