"""ctypes binding of oracle/lss_oracle.c -- TEST INFRASTRUCTURE ONLY.

All functions take / return numpy arrays (C-contiguous, float32 / int32).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'liblss_oracle.so')
_lib = None

_f = ctypes.POINTER(ctypes.c_float)
_i = ctypes.POINTER(ctypes.c_int)


def build(force=False):
    """Compile the C oracle with gcc (see oracle/Makefile)."""
    src = os.path.join(_HERE, 'lss_oracle.c')
    if (not force and os.path.exists(_SO)
            and (not os.path.exists(src)
                 or os.path.getmtime(_SO) >= os.path.getmtime(src))):
        return _SO
    subprocess.check_call(['make', '-C', _HERE, '-s'])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_voxel_prepare.restype = ctypes.c_int64
    return _lib


def _fp(a):
    return a.ctypes.data_as(_f)


def _ip(a):
    return a.ctypes.data_as(_i)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def bev_pool_v2_fwd(depth, feat, ranks_depth, ranks_feat, ranks_bev,
                    interval_starts, interval_lengths, n_voxels):
    """-> out [n_voxels, C] (zero where no interval lands)."""
    depth, feat = _f32(depth), _f32(feat)
    c = feat.shape[-1]
    out = np.zeros((n_voxels, c), np.float32)
    rd, rf, rb = _i32(ranks_depth), _i32(ranks_feat), _i32(ranks_bev)
    is_, il = _i32(interval_starts), _i32(interval_lengths)
    lib().oracle_bev_pool_v2_fwd(
        ctypes.c_int(c), ctypes.c_int(len(is_)), _fp(depth), _fp(feat),
        _ip(rd), _ip(rf), _ip(rb), _ip(is_), _ip(il), _fp(out))
    return out


def bev_pool_v2_bwd(out_grad, depth, feat, ranks_depth, ranks_feat, ranks_bev,
                    interval_starts_bp, interval_lengths_bp):
    """Intervals are over the ranks_feat-sorted point list (bev_pool.py:47-57).
    -> (depth_grad like depth, feat_grad like feat)."""
    out_grad, depth, feat = _f32(out_grad), _f32(depth), _f32(feat)
    c = feat.shape[-1]
    dg = np.zeros_like(depth)
    fg = np.zeros_like(feat)
    rd, rf, rb = _i32(ranks_depth), _i32(ranks_feat), _i32(ranks_bev)
    is_, il = _i32(interval_starts_bp), _i32(interval_lengths_bp)
    lib().oracle_bev_pool_v2_bwd(
        ctypes.c_int(c), ctypes.c_int(len(is_)), _fp(out_grad), _fp(depth),
        _fp(feat), _ip(rd), _ip(rf), _ip(rb), _ip(is_), _ip(il), _fp(dg),
        _fp(fg))
    return dg, fg


def get_lidar_coor(frustum, post_rots_inv, post_trans, combine, trans, bda):
    """frustum (D,H,W,3); post_rots_inv, combine (B,N,3,3); post_trans, trans
    (B,N,3); bda (B,3,3) -> coor (B,N,D,H,W,3)."""
    frustum = _f32(frustum)
    D, H, W, _ = frustum.shape
    post_rots_inv, combine = _f32(post_rots_inv), _f32(combine)
    post_trans, trans, bda = _f32(post_trans), _f32(trans), _f32(bda)
    B, N = combine.shape[:2]
    coor = np.empty((B, N, D, H, W, 3), np.float32)
    lib().oracle_get_lidar_coor(
        ctypes.c_int(B), ctypes.c_int(N), ctypes.c_int(D), ctypes.c_int(H),
        ctypes.c_int(W), _fp(frustum), _fp(post_rots_inv), _fp(post_trans),
        _fp(combine), _fp(trans), _fp(bda), _fp(coor))
    return coor


def voxel_prepare(coor, lower, interval, gsize):
    """coor (B,N,D,H,W,3) -> (ranks_bev, ranks_depth, ranks_feat,
    interval_starts, interval_lengths) int32, canonical (stable) order."""
    coor = _f32(coor)
    B, N, D, H, W, _ = coor.shape
    P = B * N * D * H * W
    rb = np.empty(max(P, 1), np.int32)
    rd = np.empty(max(P, 1), np.int32)
    rf = np.empty(max(P, 1), np.int32)
    is_ = np.empty(max(P, 1), np.int32)
    il = np.empty(max(P, 1), np.int32)
    ni = ctypes.c_int64(0)
    lower, interval, gsize = _f32(lower), _f32(interval), _f32(gsize)
    kept = lib().oracle_voxel_prepare(
        ctypes.c_int(B), ctypes.c_int(N), ctypes.c_int(D), ctypes.c_int(H),
        ctypes.c_int(W), _fp(coor), _fp(lower), _fp(interval), _fp(gsize),
        _ip(rb), _ip(rd), _ip(rf), _ip(is_), _ip(il), ctypes.byref(ni))
    n = ni.value
    return (rb[:kept].copy(), rd[:kept].copy(), rf[:kept].copy(),
            is_[:n].copy(), il[:n].copy())


def downsample_depth(depths, ds):
    """depths (B,N,H,W) -> (B,N,H/ds,W/ds)."""
    depths = _f32(depths)
    B, N, H, W = depths.shape
    out = np.empty((B, N, H // ds, W // ds), np.float32)
    lib().oracle_downsample_depth(
        ctypes.c_int(B * N), ctypes.c_int(H), ctypes.c_int(W),
        ctypes.c_int(ds), _fp(depths), _fp(out))
    return out


def two_hot_depth(depths, D, lo, step, gamma=4.0):
    """depths (B,N,H,W) -> (B,N,D,H,W)."""
    depths = _f32(depths)
    B, N, H, W = depths.shape
    out = np.empty((B, N, D, H, W), np.float32)
    lib().oracle_two_hot_depth(
        ctypes.c_int(B * N), ctypes.c_int(H), ctypes.c_int(W), ctypes.c_int(D),
        ctypes.c_float(lo), ctypes.c_float(step), ctypes.c_float(gamma),
        _fp(depths), _fp(out))
    return out


def permute_to_bczyx(vol_bzyxc):
    """(B,Z,Y,X,C) -> (B,C,Z,Y,X)."""
    v = _f32(vol_bzyxc)
    B, Z, Y, X, C = v.shape
    out = np.empty((B, C, Z, Y, X), np.float32)
    lib().oracle_permute_to_bczyx(
        ctypes.c_int(B), ctypes.c_int64(Z * Y * X), ctypes.c_int(C), _fp(v),
        _fp(out))
    return out


def maxpool3d(vol_bczyx, ds):
    """(B,C,Z,Y,X), ds=(dz,dy,dx) -> (B,C,Z/dz,Y/dy,X/dx)."""
    v = _f32(vol_bczyx)
    B, C, Z, Y, X = v.shape
    dz, dy, dx = ds
    out = np.empty((B, C, Z // dz, Y // dy, X // dx), np.float32)
    lib().oracle_maxpool3d(
        ctypes.c_int(B * C), ctypes.c_int(Z), ctypes.c_int(Y), ctypes.c_int(X),
        ctypes.c_int(dz), ctypes.c_int(dy), ctypes.c_int(dx), _fp(v), _fp(out))
    return out


def canonicalise(ranks_bev, ranks_depth, ranks_feat):
    """Stable re-sort of (possibly unstable-argsorted) reference output so the
    order inside an interval is ascending ranks_depth (SURVEY 7 'hard parts')."""
    rb = np.asarray(ranks_bev).astype(np.int64)
    rd = np.asarray(ranks_depth).astype(np.int64)
    order = np.lexsort((rd, rb))
    return (np.asarray(ranks_bev)[order], np.asarray(ranks_depth)[order],
            np.asarray(ranks_feat)[order])
