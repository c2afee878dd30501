"""Index half of the lift: frustum geometry and voxel-rank preparation.

Host-side mirror of ``get_lidar_coor`` and ``voxel_pooling_prepare_v2``
(mmdet3d/models/necks/view_transformer_raw.py:121-158, 244-302; identical in
view_transformer.py:114-152, 202-260).

Two implementations sit behind the same functions:

* tensors on a ROCm device -> the HIP kernels of libveon_hip.so
  (csrc/lss_prepare.hip) -- the product path, raises if the library is missing;
* CPU tensors -> the same arithmetic in torch ops.  The reference's prepare is
  itself device-agnostic torch code, so this is the reference's own CPU
  behaviour, kept for host-side logic and CPU tests; it is NOT a fallback for
  the device path: a ROCm tensor always goes to the HIP kernels and raises if
  they are unavailable (``_hip()``), it can never reach the torch ops.

Both emit the canonical *stable* order (ascending ``ranks_depth`` inside an
interval); the reference's unstable argsort leaves that order unspecified.
"""
import torch

from .ops.bev_pool_v2.bev_pool import mark_sorted

_HIP_PREPARE = None  # veon_amd.lss_prepare_hip registers itself here on import


def _hip():
    """The device implementation; a ROCm tensor never takes any other path."""
    if _HIP_PREPARE is None:
        from ._lib import VeonHipError
        raise VeonHipError('veon_amd.lss_prepare_hip is not loaded: ROCm tensors '
                           'have no non-native path')
    return _HIP_PREPARE


def camera_matrices(sensor2ego, cam2imgs, post_rots):
    """The (B,N) 3x3 algebra of get_lidar_coor (:145, :151): 6 small matrices,
    not hot, left to torch."""
    post_rots_inv = torch.inverse(post_rots)
    combine = sensor2ego[:, :, :3, :3].matmul(torch.inverse(cam2imgs))
    trans = sensor2ego[:, :, :3, 3]
    return post_rots_inv.contiguous(), combine.contiguous(), trans.contiguous()


def _mat_vec(m, p):
    """((0 + m0*x) + m1*y) + m2*z per row -- the k-ascending accumulation of
    torch's small-matrix bmm on CPU, which the HIP kernel and the C oracle
    reproduce operation for operation."""
    acc = m[..., 0] * p[..., 0:1]
    acc = acc + m[..., 1] * p[..., 1:2]
    acc = acc + m[..., 2] * p[..., 2:3]
    return acc


def lidar_coor_from_matrices_torch(frustum, post_rots_inv, post_trans, combine,
                                   trans, bda):
    B, N = combine.shape[:2]
    fr = frustum.to(combine)
    p = fr.view(1, 1, *fr.shape) - post_trans.view(B, N, 1, 1, 1, 3)
    p = _mat_vec(post_rots_inv.view(B, N, 1, 1, 1, 3, 3), p)
    p = torch.cat((p[..., :2] * p[..., 2:3], p[..., 2:3]), -1)
    p = _mat_vec(combine.view(B, N, 1, 1, 1, 3, 3), p)
    p = p + trans.view(B, N, 1, 1, 1, 3)
    p = _mat_vec(bda.view(B, 1, 1, 1, 1, 3, 3), p)
    return p


def lidar_coor_from_matrices(frustum, post_rots_inv, post_trans, combine, trans,
                             bda):
    """Per-point half of get_lidar_coor (:144-155) given the camera matrices.
    Bit-identical to the reference's CPU result for identical matrices."""
    if combine.is_cuda:
        return _hip().lidar_coor_from_matrices(
            frustum, post_rots_inv, post_trans, combine, trans, bda)
    return lidar_coor_from_matrices_torch(frustum, post_rots_inv, post_trans,
                                          combine, trans, bda)


def get_lidar_coor(frustum, sensor2ego, cam2imgs, post_rots, post_trans, bda):
    """view_transformer_raw.py:121-158.  The two ``torch.inverse`` calls run on
    the tensors' device exactly as in the reference (their last bits are
    backend-dependent there too: LAPACK on CPU, rocSOLVER on a GPU)."""
    pri, comb, trans = camera_matrices(sensor2ego, cam2imgs, post_rots)
    return lidar_coor_from_matrices(frustum, pri, post_trans, comb, trans, bda)


def voxel_pooling_prepare_v2_torch(coor, lower, interval, gsize):
    B, N, D, H, W, _ = coor.shape
    P = B * N * D * H * W
    dev = coor.device
    lower, interval, gsize = lower.to(coor), interval.to(coor), gsize.to(coor)
    vox = ((coor - lower) / interval).long().view(P, 3)        # trunc (:267-269)
    inside = ((vox >= 0) & (vox.float() < gsize.view(1, 3))).all(1)  # (:275-277)
    idx = torch.nonzero(inside).squeeze(1)
    if idx.numel() == 0:
        return None, None, None, None, None
    vox = vox[idx]
    b = torch.div(idx, N * D * H * W, rounding_mode='floor')
    # float32 rank, formed exactly as the reference does (:283-286)
    key = b.float() * (gsize[2] * gsize[1] * gsize[0])
    key = key + vox[:, 2].float() * (gsize[1] * gsize[0])
    key = key + (vox[:, 1].float() * gsize[0] + vox[:, 0].float())
    key, order = torch.sort(key, stable=True)
    ranks_depth = idx[order]
    ranks_feat = torch.div(ranks_depth, D * H * W, rounding_mode='floor') * \
        (H * W) + ranks_depth % (H * W)
    ranks_bev = key.int()
    _, counts = torch.unique_consecutive(ranks_bev, return_counts=True)
    starts = (torch.cumsum(counts, 0) - counts).int().contiguous()
    first, last = int(ranks_bev[0]), int(ranks_bev[-1])
    mark_sorted(starts, first, last)
    return (ranks_bev.contiguous(), ranks_depth.int().contiguous(),
            ranks_feat.int().contiguous(), starts,
            counts.int().contiguous())


def voxel_pooling_prepare_v2(coor, lower, interval, gsize):
    if coor.is_cuda:
        return _hip().voxel_pooling_prepare_v2(coor, lower, interval, gsize)
    return voxel_pooling_prepare_v2_torch(coor, lower, interval, gsize)


def prepare_from_matrices(frustum, post_rots_inv, post_trans, combine, trans,
                          bda, lower, interval, gsize):
    """get_lidar_coor's per-point half + voxel_pooling_prepare_v2 in one go
    (on a ROCm device the coordinates are never materialised)."""
    if combine.is_cuda:
        return _hip().prepare_from_matrices(
            frustum, post_rots_inv, post_trans, combine, trans, bda, lower,
            interval, gsize)
    coor = lidar_coor_from_matrices_torch(frustum, post_rots_inv, post_trans,
                                          combine, trans, bda)
    return voxel_pooling_prepare_v2_torch(coor, lower, interval, gsize)


from . import lss_prepare_hip  # noqa: E402,F401  (registers the device path)
