"""HIP depth preparation (csrc/depth_ops.hip) behind ``veon_amd.depth_ops``."""
import sys

import torch

from . import _lib, depth_ops


def downsample_depth(depths, downsample):
    dev = _lib.require_device(depths)
    B, N, H, W = depths.shape
    ds = int(downsample)
    src = depths.contiguous().float()
    out = torch.empty((B, N, H // ds, W // ds), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        st = _lib.lib().veon_downsample_depth(
            B * N, H, W, ds, _lib.ptr(src), _lib.ptr(out), _lib.stream_ptr(dev))
    _lib.check(st, 'veon_downsample_depth')
    return out


def two_hot_depth(depths, D, lo, step, gamma=4, fused_downsample=0):
    """(B,N,H,W) -> (B,N,D,H,W); with ``fused_downsample=ds`` the input is at
    ds x the output resolution and the block-min is fused in."""
    dev = _lib.require_device(depths)
    B, N, H, W = depths.shape
    ds = int(fused_downsample)
    if ds:
        H, W = H // ds, W // ds
    src = depths.contiguous().float()
    out = torch.empty((B, N, D, H, W), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        st = _lib.lib().veon_two_hot_depth(
            B * N, H, W, ds, D, float(lo), float(step), float(gamma),
            _lib.ptr(src), _lib.ptr(out), _lib.stream_ptr(dev))
    _lib.check(st, 'veon_two_hot_depth')
    return out


def two_hot_windows(depths, D, lo, step, gamma=4, eps=0.0, fused_downsample=0):
    """(B,N,H,W) metric depth -> ``depth_ops.TwoHotWindows`` (csrc k_two_hot_window):
    the compact exact form of ``two_hot_depth``; the dense tensor is never written."""
    dev = _lib.require_device(depths)
    B, N, H, W = depths.shape
    ds = int(fused_downsample)
    if ds:
        H, W = H // ds, W // ds
    L = _lib.lib()
    K = L.veon_two_hot_window_slots(int(D), float(step), float(gamma))
    if K <= 0 or K != depth_ops.two_hot_window_slots(D, step, gamma):
        raise _lib.VeonHipError('two-hot window slots: native %d, host %d'
                                % (K, depth_ops.two_hot_window_slots(D, step, gamma)))
    src = depths.contiguous().float()
    win = torch.empty((B, N, H, W, 2), dtype=torch.int32, device=dev)
    wts = torch.empty((B, N, H, W, K), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        st = L.veon_two_hot_window(
            B * N, H, W, ds, int(D), float(lo), float(step), float(gamma), float(eps), K,
            _lib.ptr(src), _lib.ptr(win), _lib.ptr(wts), _lib.stream_ptr(dev))
    _lib.check(st, 'veon_two_hot_window')
    return depth_ops.TwoHotWindows(win, wts, D, eps)


depth_ops._HIP = sys.modules[__name__]
