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


def named_init_(module, prefix, seed=0, boost=None, skip=()):
    """Deterministic, NAME-keyed initialisation of every floating-point entry of
    ``module.state_dict()``: the tensor named ``k`` is drawn from a generator seeded
    with crc32(prefix + k) ^ seed.  A reference module and its state-dict-compatible
    mirror initialised this way hold identical weights without the weights ever being
    stored -- what lets a golden vector at PRODUCTION widths (hundreds of MB of
    weights) stay a few hundred KiB (tests/golden/path_prod.npz).  Scales keep a deep
    pre-norm network well conditioned: matrices N(0, 0.8 / sqrt(fan_in)), norm scales
    1 + 0.1 N, running variances 0.5 + U, everything else 0.02 N.  ``boost``: {name
    suffix: factor} multiplies the drawn tensor (e.g. a head's last layer, so that its
    output spans a useful range)."""
    import zlib

    import torch
    with torch.no_grad():
        for k, t in module.state_dict().items():
            if not t.is_floating_point() or any(sk in k for sk in skip):
                continue
            g = torch.Generator().manual_seed((zlib.crc32((prefix + k).encode()) ^ seed)
                                              & 0x7fffffff)
            if k.endswith('running_var'):
                v = 0.5 + torch.rand(t.shape, generator=g)
            elif k.endswith('running_mean'):
                v = 0.1 * torch.randn(t.shape, generator=g)
            elif t.dim() >= 2 and t.shape[0] > 1 and t[0].numel() > 1:
                fan_in = t[0].numel()
                v = torch.randn(t.shape, generator=g) * (0.8 / fan_in ** 0.5)
            elif t.dim() == 1 and (k.endswith('weight') or k.endswith('gamma')):
                v = 1.0 + 0.1 * torch.randn(t.shape, generator=g)
            else:
                v = 0.02 * torch.randn(t.shape, generator=g)
            for suffix, factor in (boost or {}).items():
                if k.endswith(suffix):
                    v = v * factor
            t.copy_(v.to(t.dtype))
    return module
