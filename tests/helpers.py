"""Shared test helpers (oracle-side)."""
import hashlib

import numpy as np
import torch

from oracle import c_oracle, lss_torch
from veon_amd import synthetic


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def grid_np(grid_config):
    lower, interval, gsize = lss_torch.grid_infos(grid_config)
    return lower.numpy(), interval.numpy(), gsize.numpy()


def oracle_coor(grid_config, input_size, n_cams, batch=1, rig=None):
    """Bit-reproducible frustum coordinates from the C oracle."""
    rig = rig or synthetic.make_rig(batch, n_cams, input_size)
    fr = lss_torch.make_frustum(grid_config['depth'], input_size, 16)
    pri, comb, trans = lss_torch.camera_matrices(
        rig['sensor2ego'], rig['intrins'], rig['post_rots'])
    coor = c_oracle.get_lidar_coor(fr.numpy(), pri.numpy(),
                                   rig['post_trans'].numpy(), comb.numpy(),
                                   trans.numpy(), rig['bda'].numpy())
    return coor, rig, fr


def oracle_ranks(grid_config, input_size, n_cams, batch=1):
    coor, rig, fr = oracle_coor(grid_config, input_size, n_cams, batch)
    lower, interval, gsize = grid_np(grid_config)
    ranks = c_oracle.voxel_prepare(coor, lower, interval, gsize)
    return ranks, coor, rig, fr, gsize


def bp_intervals(ranks_feat_sorted):
    """Intervals of a ranks_feat-sorted list (bev_pool.py:50-57)."""
    rf = np.asarray(ranks_feat_sorted)
    kept = np.ones(len(rf), bool)
    kept[1:] = rf[1:] != rf[:-1]
    starts = np.nonzero(kept)[0].astype(np.int32)
    lengths = np.diff(np.append(starts, len(rf))).astype(np.int32)
    return starts, lengths


def oracle_backward(out_grad_bzyxc, depth, feat_nhwc, rd, rf, rb):
    """QuickCumsumCuda.backward on the oracle: stable feat sort + C grad."""
    order = np.argsort(rf, kind='stable')
    rd, rf, rb = rd[order], rf[order], rb[order]
    st, ln = bp_intervals(rf)
    return c_oracle.bev_pool_v2_bwd(out_grad_bzyxc, depth, feat_nhwc, rd, rf,
                                    rb, st, ln)


def t(a, device='cpu'):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)
