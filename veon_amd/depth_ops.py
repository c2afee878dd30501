"""Depth preparation for the lift: block-min downsample and soft two-hot bins.

Host-side mirror of ``LSSViewTransformerRaw.downsample_depth`` /
``get_two_hot_depth`` (mmdet3d/models/necks/view_transformer_raw.py:393-429).
ROCm tensors go to the HIP kernel (csrc/depth_twohot.hip) once registered via
``_HIP``; CPU tensors use the same arithmetic in torch ops (the reference's own
device-agnostic behaviour, kept for host logic and CPU tests).
"""
import torch

_HIP = None  # veon_amd.depth_ops_hip registers itself here on import


def _hip():
    if _HIP is None:
        from ._lib import VeonHipError
        raise VeonHipError('veon_amd.depth_ops_hip is not loaded: ROCm tensors '
                           'have no non-native path')
    return _HIP


def downsample_depth_torch(depths, downsample):
    B, N, H, W = depths.shape
    ds = downsample
    d = depths.view(B * N, H // ds, ds, W // ds, ds).permute(0, 1, 3, 2, 4)
    d = d.reshape(-1, ds * ds)
    d = torch.where(d == 0.0, torch.full_like(d, 1e5), d).min(dim=-1).values
    return d.view(B, N, H // ds, W // ds)


def two_hot_depth_torch(depths, D, lo, step, gamma=4, min_gap=-16.0):
    B, N, H, W = depths.shape
    centers = torch.arange(D + 1, device=depths.device) * step + (lo + step / 2)
    gap = -(depths.reshape(B * N, H, W, 1) - centers.view(1, 1, 1, -1)).abs() * gamma
    # forward value of the reference's straight-through clamp (:421-422)
    gap = torch.where(gap >= min_gap, gap, torch.full_like(gap, min_gap))
    dist = torch.softmax(gap, dim=-1)[..., :-1]
    return dist.view(B, N, H, W, D).permute(0, 1, 4, 2, 3)


def downsample_depth(depths, downsample):
    if depths.is_cuda:
        return _hip().downsample_depth(depths, downsample)
    return downsample_depth_torch(depths, downsample)


def two_hot_depth(depths, D, lo, step, gamma=4):
    if depths.is_cuda:
        return _hip().two_hot_depth(depths, D, lo, step, gamma)
    return two_hot_depth_torch(depths, D, lo, step, gamma)


def two_hot_depth_fused(depths, downsample, D, lo, step, gamma=4):
    """downsample_depth + get_two_hot_depth in one pass on a ROCm device
    (AlignNetOcc3D.prepare_depth, align_net_occ3d.py:320-326)."""
    if depths.is_cuda:
        return _hip().two_hot_depth(depths, D, lo, step, gamma,
                                    fused_downsample=downsample)
    return two_hot_depth_torch(downsample_depth_torch(depths, downsample), D,
                               lo, step, gamma)


from . import depth_ops_hip  # noqa: E402,F401  (registers the device path)
