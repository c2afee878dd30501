"""Generate tests/golden/* by running the reference's own Python here.

Run from the repo root in the build container (needs /root/reference):
    python oracle/tools/gen_golden.py
The outputs are DATA (inputs + expected outputs) and are committed; this
script is committed beside them so they can be regenerated and audited.  The
reference never travels to the GPU box; tests only read the fixtures.

What the reference computes here (its own code, unmodified, on CPU):
  * create_frustum / create_grid_infos / get_lidar_coor /
    voxel_pooling_prepare_v2 / downsample_depth / get_two_hot_depth /
    forward's 2x2x2 max-pool   (view_transformer_raw.py, view_transformer.py)
What it cannot compute: the pool arithmetic itself (CUDA-only op);
for `pooled`/`forward_out` the reference Python runs with this repo's CPU
restatement plugged in as `bev_pool_v2`, so those two arrays pin the
reference's *wiring* (permute, collapse_z, max-pool), not the pool sum --
that is pinned by the known-answer test (kat_bev_pool_v2.npz).
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import c_oracle, lss_torch  # noqa: E402
from oracle.tools import ref_import  # noqa: E402
from veon_amd import synthetic  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')


def cpu_bev_pool_v2(depth, feat, ranks_depth, ranks_feat, ranks_bev,
                    bev_feat_shape, interval_starts, interval_lengths):
    return lss_torch.pool(depth.float(), feat.float(), ranks_depth, ranks_feat,
                          ranks_bev, bev_feat_shape)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def np_(t):
    return t.detach().cpu().numpy()


def kat():
    """The reference's in-module known-answer test, bev_pool.py:145-176."""
    np.savez(
        os.path.join(GOLD, 'kat_bev_pool_v2.npz'),
        depth=np.array([0.3, 0.4, 0.2, 0.1, 0.7, 0.6, 0.8, 0.9],
                       np.float32).reshape(1, 1, 2, 2, 2),
        feat=np.ones((1, 1, 2, 2, 2), np.float32),
        ranks_depth=np.array([0, 4, 1, 6], np.int32),
        ranks_feat=np.array([0, 0, 1, 2], np.int32),
        ranks_bev=np.array([0, 0, 1, 1], np.int32),
        bev_feat_shape=np.array([1, 1, 2, 2, 2], np.int64),
        interval_starts=np.array([0, 2], np.int32),
        interval_lengths=np.array([2, 2], np.int32),
        expect_sum=np.float32(4.4),
        expect_depth_grad=np.array([2., 2., 0., 0., 2., 0., 2., 0.],
                                   np.float32).reshape(1, 1, 2, 2, 2),
        expect_feat_grad=np.array([1.0, 1.0, 0.4, 0.4, 0.8, 0.8, 0., 0.],
                                  np.float32).reshape(1, 1, 2, 2, 2),
    )


def perturbed_rig(batch, n_cams, input_size, seed):
    """Synthetic rig + seeded perturbations so that post_rots / bda are not
    multiples of the identity (exercises every matrix entry)."""
    rig = synthetic.make_rig(batch, n_cams, input_size)
    g = torch.Generator().manual_seed(seed)
    rig['post_rots'][:, :, :2, :2] += 0.01 * torch.randn(batch, n_cams, 2, 2, generator=g)
    rig['post_trans'][:, :, :2] += 2.0 * torch.randn(batch, n_cams, 2, generator=g)
    ang = 0.3 * torch.randn(batch, generator=g)
    for b in range(batch):
        c, s = torch.cos(ang[b]), torch.sin(ang[b])
        rig['bda'][b] = torch.tensor([[c, -s, 0.], [s, c, 0.], [0., 0., 1.02]])
    rig['sensor2ego'][:, :, :3, 3] += 0.05 * torch.randn(batch, n_cams, 3, generator=g)
    return rig


def small_case(raw, name, batch, seed, grid_config, input_size, C, ds_feat,
               perturb):
    n_cams = 6
    vt = raw.LSSViewTransformerRaw(
        grid_config=grid_config, input_size=input_size, downsample=16,
        out_channels=C, collapse_z=False, ds_feat=ds_feat)
    rig = (perturbed_rig(batch, n_cams, input_size, seed) if perturb
           else synthetic.make_rig(batch, n_cams, input_size))
    inp = synthetic.rig_inputs(rig)
    hf, wf = input_size[0] // 16, input_size[1] // 16
    coor = vt.get_lidar_coor(*inp)
    rb, rd, rf, ist, iln = vt.voxel_pooling_prepare_v2(coor)
    rbc, rdc, rfc = c_oracle.canonicalise(np_(rb), np_(rd), np_(rf))
    g = torch.Generator().manual_seed(seed + 100)
    # VEON-style depth: metric depth at 8x the feature resolution -> two-hot
    metric = 1.0 + 50.0 * torch.rand(batch, n_cams, hf * 8, wf * 8, generator=g)
    metric[metric < 6.0] = 0.0   # holes, exercised by the non-zero block-min
    ds_depth = vt.downsample_depth(metric, 8)
    two_hot = vt.get_two_hot_depth(ds_depth)
    feat = torch.randn(batch, n_cams, C, hf, wf, generator=g)
    pooled = vt.voxel_pooling_v2(coor, two_hot.contiguous(), feat)
    fwd = vt.forward([feat] + list(inp), two_hot.contiguous())
    np.savez_compressed(
        os.path.join(GOLD, name + '.npz'),
        grid_x=np.array(grid_config['x'], np.float64),
        grid_y=np.array(grid_config['y'], np.float64),
        grid_z=np.array(grid_config['z'], np.float64),
        grid_depth=np.array(grid_config['depth'], np.float64),
        input_size=np.array(input_size), downsample=np.array(16),
        ds_feat=np.array(ds_feat),
        D=np.array(vt.D), frustum=np_(vt.frustum),
        grid_lower_bound=np_(vt.grid_lower_bound),
        grid_interval=np_(vt.grid_interval), grid_size=np_(vt.grid_size),
        sensor2ego=np_(rig['sensor2ego']), ego2global=np_(rig['ego2global']),
        intrins=np_(rig['intrins']), post_rots=np_(rig['post_rots']),
        post_trans=np_(rig['post_trans']), bda=np_(rig['bda']),
        coor=np_(coor),
        ranks_bev_raw=np_(rb), ranks_depth_raw=np_(rd), ranks_feat_raw=np_(rf),
        ranks_bev=rbc, ranks_depth=rdc, ranks_feat=rfc,
        interval_starts=np_(ist), interval_lengths=np_(iln),
        metric_depth=np_(metric), ds_depth=np_(ds_depth), two_hot=np_(two_hot),
        feat=np_(feat), pooled=np_(pooled), forward_out=np_(fwd),
    )
    print(name, 'P=%d kept=%d I=%d' % (coor.numel() // 3, len(rbc), len(ist)))


def full_case(raw, vtmod, tag, cls, grid_config, input_size, n_cams, C):
    """Full BASELINE shapes: hashes of the reference prepare run on the C
    oracle's coordinates (bit-reproducible on any host), plus statistics of the
    reference's own coordinates."""
    if cls == 'raw':
        vt = raw.LSSViewTransformerRaw(
            grid_config=grid_config, input_size=input_size, downsample=16,
            out_channels=C, collapse_z=False)
    else:
        vt = vtmod.LSSViewTransformer(
            grid_config=grid_config, input_size=input_size, downsample=16,
            in_channels=8, out_channels=C)
    rig = synthetic.make_rig(1, n_cams, input_size)
    inp = synthetic.rig_inputs(rig)
    ref_coor = vt.get_lidar_coor(*inp)
    pri, comb, trans = lss_torch.camera_matrices(
        rig['sensor2ego'], rig['intrins'], rig['post_rots'])
    orc_coor = c_oracle.get_lidar_coor(
        np_(vt.frustum), np_(pri), np_(rig['post_trans']), np_(comb),
        np_(trans), np_(rig['bda']))
    rb, rd, rf, ist, iln = vt.voxel_pooling_prepare_v2(torch.from_numpy(orc_coor))
    rbc, rdc, rfc = c_oracle.canonicalise(np_(rb), np_(rd), np_(rf))
    rb2 = vt.voxel_pooling_prepare_v2(ref_coor)[0]
    refc = np_(ref_coor).astype(np.float64)
    sub = np_(ref_coor).reshape(-1, 3)[::997]
    out = dict(
        tag=tag, cls=cls, grid_config=grid_config, input_size=list(input_size),
        n_cams=n_cams, C=C, D=int(vt.D), P=int(ref_coor.numel() // 3),
        P_kept=int(len(rbc)), n_intervals=int(len(ist)),
        max_interval=int(np_(iln).max()),
        P_kept_from_ref_coor=int(len(rb2)),
        sha_ranks_bev=sha(rbc.astype(np.int32)),
        sha_ranks_depth=sha(rdc.astype(np.int32)),
        sha_ranks_feat=sha(rfc.astype(np.int32)),
        sha_interval_starts=sha(np_(ist).astype(np.int32)),
        sha_interval_lengths=sha(np_(iln).astype(np.int32)),
        sha_oracle_coor=sha(orc_coor),
        ref_coor_sum=[float(refc[..., i].sum()) for i in range(3)],
        ref_coor_abs_sum=[float(np.abs(refc[..., i]).sum()) for i in range(3)],
        ref_vs_oracle_coor_max_abs_diff=float(
            np.abs(refc - orc_coor.astype(np.float64)).max()),
        ref_vs_oracle_coor_n_diff=int((np_(ref_coor) != orc_coor).sum()),
    )
    np.save(os.path.join(GOLD, 'coor_sub_%s.npy' % tag), sub)
    print(tag, {k: out[k] for k in ('P', 'P_kept', 'n_intervals', 'max_interval',
                                    'P_kept_from_ref_coor',
                                    'ref_vs_oracle_coor_max_abs_diff',
                                    'ref_vs_oracle_coor_n_diff')})
    return out


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    raw, vtmod = ref_import.load_view_transformers(cpu_bev_pool_v2)
    kat()
    small = {'x': [-40, 40, 4.0], 'y': [-40, 40, 4.0], 'z': [-1, 5.4, 1.6],
             'depth': [1.0, 33.0, 4.0]}
    small_case(raw, 'lss_small', 1, 1, small, (64, 176), 8, [2, 2, 2], False)
    small_case(raw, 'lss_small_b2', 2, 2, small, (64, 176), 12, [2, 2, 2], True)
    mid = {'x': [-40, 40, 1.6], 'y': [-40, 40, 1.6], 'z': [-1, 5.4, 0.8],
           'depth': [1.0, 45.0, 2.0]}
    small_case(raw, 'lss_mid', 1, 3, mid, (128, 352), 16, [2, 2, 2], True)
    full = [
        full_case(raw, vtmod, 'S1', 'vt', synthetic.GRID_BEVDET, (256, 704), 1, 64),
        full_case(raw, vtmod, 'S2', 'vt', synthetic.GRID_S2, (256, 704), 6, 80),
        full_case(raw, vtmod, 'SV', 'raw', synthetic.GRID_VEON, (512, 1408), 6, 256),
    ]
    with open(os.path.join(GOLD, 'lss_full.json'), 'w') as f:
        json.dump(full, f, indent=1)


if __name__ == '__main__':
    main()
