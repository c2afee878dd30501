"""Pure-PyTorch restatement of the lift hot path -- TEST INFRASTRUCTURE ONLY.

This is the "pure-PyTorch scatter_add CPU path" BASELINE.json asks to time on
the GPU box's host cores (bench.py ``cpu_baseline``, kind "port") and a second,
vectorised checker beside the serial C oracle.  Nothing under ``veon_amd/`` may
import it.

The reference has no CPU pool; the pool half restates the kernel
semantics of mmdet3d/ops/bev_pool_v2/src/bev_pool_cuda.cu:21-48 with
``index_add_``; the index half restates
mmdet3d/models/necks/view_transformer_raw.py:91-158, 244-302.
"""
import torch


def grid_infos(grid_config):
    """view_transformer_raw.py:74-89 -- float32 lower bound / interval / size."""
    axes = [grid_config[k] for k in ('x', 'y', 'z')]
    lower = torch.tensor([a[0] for a in axes], dtype=torch.float32)
    interval = torch.tensor([a[2] for a in axes], dtype=torch.float32)
    size = torch.tensor([(a[1] - a[0]) / a[2] for a in axes],
                        dtype=torch.float32)
    return lower, interval, size


def make_frustum(depth_cfg, input_size, downsample):
    """view_transformer_raw.py:91-119 (sid=False) -> (D,Hf,Wf,3) float32 of
    (x_pix, y_pix, depth)."""
    h_in, w_in = input_size
    hf, wf = h_in // downsample, w_in // downsample
    d = torch.arange(*depth_cfg, dtype=torch.float32)
    xs = torch.linspace(0, w_in - 1, wf, dtype=torch.float32)
    ys = torch.linspace(0, h_in - 1, hf, dtype=torch.float32)
    D = d.numel()
    out = torch.empty(D, hf, wf, 3, dtype=torch.float32)
    out[..., 0] = xs.view(1, 1, wf)
    out[..., 1] = ys.view(1, hf, 1)
    out[..., 2] = d.view(D, 1, 1)
    return out


def camera_matrices(sensor2ego, cam2imgs, post_rots):
    """The per-camera 3x3 algebra of get_lidar_coor (:145,151):
    inv(post_rots) and sensor2ego[:3,:3] @ inv(cam2imgs)."""
    post_rots_inv = torch.inverse(post_rots)
    combine = sensor2ego[:, :, :3, :3].matmul(torch.inverse(cam2imgs))
    trans = sensor2ego[:, :, :3, 3].contiguous()
    return post_rots_inv, combine, trans


def lidar_coor(frustum, sensor2ego, cam2imgs, post_rots, post_trans, bda):
    """view_transformer_raw.py:121-158 with broadcasting multiplies instead of
    batched matmul (same k-ascending accumulation)."""
    B, N = sensor2ego.shape[:2]
    pri, comb, trans = camera_matrices(sensor2ego, cam2imgs, post_rots)

    def mv(m, p):  # m (...,3,3) broadcast over points p (...,3)
        acc = m[..., 0] * p[..., 0:1]
        acc = acc + m[..., 1] * p[..., 1:2]
        acc = acc + m[..., 2] * p[..., 2:3]
        return acc

    p = frustum.view(1, 1, *frustum.shape) - post_trans.view(B, N, 1, 1, 1, 3)
    p = mv(pri.view(B, N, 1, 1, 1, 3, 3), p)
    p = torch.cat((p[..., :2] * p[..., 2:3], p[..., 2:3]), -1)
    p = mv(comb.view(B, N, 1, 1, 1, 3, 3), p)
    p = p + trans.view(B, N, 1, 1, 1, 3)
    p = mv(bda.view(B, 1, 1, 1, 1, 3, 3), p)
    return p


def voxel_prepare(coor, lower, interval, gsize):
    """view_transformer_raw.py:244-302, canonical stable order.
    Returns 5 int32 tensors or 5x None when nothing survives."""
    B, N, D, H, W, _ = coor.shape
    P = B * N * D * H * W
    vox = ((coor - lower.to(coor)) / interval.to(coor)).long().view(P, 3)
    ok = ((vox >= 0) & (vox.float() < gsize.to(coor).view(1, 3))).all(1)
    idx = torch.nonzero(ok).squeeze(1)
    if idx.numel() == 0:
        return (None,) * 5
    vox = vox[idx]
    b = idx // (N * D * H * W)
    key = b.float() * (gsize[2] * gsize[1] * gsize[0])
    key = key + vox[:, 2].float() * (gsize[1] * gsize[0])
    key = key + (vox[:, 1].float() * gsize[0] + vox[:, 0].float())
    key, order = torch.sort(key, stable=True)
    rd = idx[order]
    rf = (rd // (D * H * W)) * (H * W) + rd % (H * W)
    rb = key.int()
    _, counts = torch.unique_consecutive(rb, return_counts=True)
    starts = torch.cumsum(counts, 0) - counts
    return (rb.contiguous(), rd.int().contiguous(), rf.int().contiguous(),
            starts.int().contiguous(), counts.int().contiguous())


def pool(depth, feat_nhwc, ranks_depth, ranks_feat, ranks_bev, bev_feat_shape):
    """out[ranks_bev] += depth[ranks_depth] * feat[ranks_feat]
    (bev_pool_cuda.cu:21-48) via index_add_, then the
    permute(0,4,1,2,3).contiguous() of bev_pool.py:91."""
    B, Z, Y, X, C = bev_feat_shape
    contrib = depth.reshape(-1)[ranks_depth.long()].unsqueeze(1) * \
        feat_nhwc.reshape(-1, C)[ranks_feat.long()]
    out = torch.zeros(B * Z * Y * X, C, dtype=feat_nhwc.dtype)
    out.index_add_(0, ranks_bev.long(), contrib)
    return out.view(B, Z, Y, X, C).permute(0, 4, 1, 2, 3).contiguous()


def maxpool(vol, ds):
    """view_transformer_raw.py:549-553."""
    B, C, Z, Y, X = vol.shape
    dz, dy, dx = ds
    v = vol.view(B, C, Z // dz, dz, Y // dy, dy, X // dx, dx)
    return v.amax(dim=(3, 5, 7))


def two_hot_depth(depths, D, lo, step, gamma=4.0):
    """view_transformer_raw.py:406-429, (B,N,H,W) -> (B,N,D,H,W)."""
    B, N, H, W = depths.shape
    centers = torch.arange(D + 1) * step + (lo + step / 2)
    gap = -(depths.reshape(B * N, H, W, 1) - centers.view(1, 1, 1, -1)).abs() * gamma
    gap = torch.clamp_min(gap, -16.0)
    dist = torch.softmax(gap, dim=-1)[..., :-1]
    return dist.view(B, N, H, W, D).permute(0, 1, 4, 2, 3)


def lift(frustum, grid, cams, depth, feat_nchw, ds=None, ranks=None):
    """One pass of the hot path on CPU: geometry -> prepare -> pool -> permute
    (-> max-pool).  ``cams`` = (sensor2ego, cam2imgs, post_rots, post_trans,
    bda).  ``ranks`` short-circuits the prepare (the accelerate=True path)."""
    lower, interval, gsize = grid
    if ranks is None:
        coor = lidar_coor(frustum, *cams)
        ranks = voxel_prepare(coor, lower, interval, gsize)
    rb, rd, rf, _, _ = ranks
    B = depth.shape[0]
    C = feat_nchw.shape[2]
    shape = (B, int(gsize[2]), int(gsize[1]), int(gsize[0]), C)
    vol = pool(depth, feat_nchw.permute(0, 1, 3, 4, 2), rd, rf, rb, shape)
    if ds is not None:
        vol = maxpool(vol, ds)
    return vol
