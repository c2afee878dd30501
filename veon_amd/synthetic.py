"""Synthetic nuScenes-shaped inputs for the lift hot path (no dataset needed).

A fixed six-camera surround rig (yaw 55/0/-55/110/180/-110 deg, the nuScenes
camera order CAM_FRONT_LEFT, CAM_FRONT, CAM_FRONT_RIGHT, CAM_BACK_LEFT,
CAM_BACK, CAM_BACK_RIGHT of configs/veon/*.py ``data_config['cams']``) with
nuScenes-like intrinsics, and the test-time image augmentation of
mmdet3d/datasets/pipelines/loading.py:1174-1186, 1125-1136
(``post_rot = (W_in/1600) * I``, ``post_tran = (0, -(int(900*s) - H_in), 0)``).
Everything is float32 and derived from closed-form numbers, so the same rig is
reproduced bit-for-bit on any host.
"""
import math

import torch

SRC_SIZE = (900, 1600)  # nuScenes source image (H, W), configs/veon data_config

# (yaw_deg, fx, tx, ty, tz): ego frame x forward / y left / z up
_RIG = (
    (55.0, 1266.4, 1.52, 0.49, 1.51),
    (0.0, 1266.4, 1.70, 0.02, 1.51),
    (-55.0, 1260.8, 1.55, -0.49, 1.50),
    (110.0, 1256.7, 1.04, 0.48, 1.56),
    (180.0, 809.2, 0.03, 0.00, 1.57),
    (-110.0, 1259.5, 1.01, -0.48, 1.56),
)

GRID_VEON = {  # configs/veon/veon-temporal-base-512x1408-dav2-nodepthcache.py:33-38
    'x': [-40, 40, 0.4], 'y': [-40, 40, 0.4], 'z': [-1, 5.4, 0.4],
    'depth': [1.0, 45.0, 0.5]}
GRID_S2 = {  # BASELINE.json configs[1]: D=59, 200x200x16
    'x': [-40, 40, 0.4], 'y': [-40, 40, 0.4], 'z': [-1, 5.4, 0.4],
    'depth': [1.0, 60.0, 1.0]}
GRID_BEVDET = {  # configs/bevdet/bevdet-r50.py:54-59: 128x128x1
    'x': [-51.2, 51.2, 0.8], 'y': [-51.2, 51.2, 0.8], 'z': [-5, 3, 8],
    'depth': [1.0, 60.0, 1.0]}


def _cam_rotation(yaw_deg):
    """Camera (x right, y down, z forward) -> ego (x fwd, y left, z up), yawed."""
    a = math.radians(yaw_deg)
    c, s = math.cos(a), math.sin(a)
    # columns: images of camera x, y, z axes in the ego frame
    return [[s, 0.0, c],
            [-c, 0.0, s],
            [0.0, -1.0, 0.0]]


def make_rig(batch=1, n_cams=6, input_size=(256, 704), dtype=torch.float32):
    """-> dict(sensor2ego (B,N,4,4), ego2global (B,N,4,4), intrins (B,N,3,3),
    post_rots (B,N,3,3), post_trans (B,N,3), bda (B,3,3))."""
    h_in, w_in = input_size
    s2e = torch.zeros(n_cams, 4, 4, dtype=torch.float64)
    k = torch.zeros(n_cams, 3, 3, dtype=torch.float64)
    for i in range(n_cams):
        yaw, fx, tx, ty, tz = _RIG[i % len(_RIG)]
        s2e[i, :3, :3] = torch.tensor(_cam_rotation(yaw), dtype=torch.float64)
        s2e[i, :3, 3] = torch.tensor([tx, ty, tz], dtype=torch.float64)
        s2e[i, 3, 3] = 1.0
        k[i] = torch.tensor([[fx, 0.0, 816.3 - 2.0 * i],
                             [0.0, fx, 491.5 + 1.5 * i],
                             [0.0, 0.0, 1.0]], dtype=torch.float64)
    scale = float(w_in) / float(SRC_SIZE[1])
    new_h = int(SRC_SIZE[0] * scale)
    new_w = int(SRC_SIZE[1] * scale)
    crop_h = new_h - h_in
    crop_w = int(max(0, new_w - w_in) / 2)
    post_rot = torch.eye(3, dtype=torch.float64)
    post_rot[0, 0] = scale
    post_rot[1, 1] = scale
    post_tran = torch.tensor([-float(crop_w), -float(crop_h), 0.0],
                             dtype=torch.float64)

    def rep(t, per_cam=True):
        t = t.to(dtype)
        if per_cam:
            return t.unsqueeze(0).expand(batch, *t.shape).contiguous()
        return t

    return dict(
        sensor2ego=rep(s2e),
        ego2global=rep(torch.eye(4, dtype=torch.float64).expand(n_cams, 4, 4)),
        intrins=rep(k),
        post_rots=rep(post_rot.expand(n_cams, 3, 3)),
        post_trans=rep(post_tran.expand(n_cams, 3)),
        bda=torch.eye(3, dtype=dtype).unsqueeze(0).expand(batch, 3, 3).contiguous(),
    )


def rig_inputs(rig):
    """The ``input[1:7]`` tuple the view transformers take
    (view_transformer_raw.py:539)."""
    return (rig['sensor2ego'], rig['ego2global'], rig['intrins'],
            rig['post_rots'], rig['post_trans'], rig['bda'])


def make_depth_feat(batch, n_cams, D, C, hf, wf, seed=0, device='cpu',
                    two_hot=False, depth_cfg=None):
    """Seeded op inputs: depth (B,N,D,Hf,Wf) = softmax(randn) over D (BEVDet
    style) and feat (B,N,C,Hf,Wf) = randn."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    depth = torch.randn(batch, n_cams, D, hf, wf, generator=g).softmax(dim=2)
    feat = torch.randn(batch, n_cams, C, hf, wf, generator=g)
    return depth.to(device), feat.to(device)
