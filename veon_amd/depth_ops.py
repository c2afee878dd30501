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


def two_hot_window_slots(D, step, gamma=4):
    """K = 1 (tail) + the longest possible window of unclamped bins
    (|d - c_k| <= 16/gamma, centres ``step`` apart); csrc veon_two_hot_window_slots."""
    return 1 + min(int(D), int(32.0 / (float(gamma) * float(step))) + 2)


class TwoHotWindows:
    """The soft two-hot depth distribution of ``get_two_hot_depth``
    (view_transformer_raw.py:406-429) in compact, EXACT form -- the (B,N,D,H,W)
    tensor is never written.  A pixel's weights are distinct only
    on the window of bins whose logit is not clamped at -16, and one value (the
    tail) everywhere else:

        wts (B,N,H,W,K) fp32   [..., 0] = tail, [..., 1 + j] = weight of bin k0 + j
        win (B,N,H,W,2) int32  [..., 0] = k0 | nk << 16
                               [..., 1] = q0 | nq << 16 | t << 31: the bins with
                               weight >= eps, t = (tail >= eps)

    The view transformer's sync-free lift takes it in place of the dense ``depth``
    (``LSSViewTransformerRaw.forward(input, windows)``): the prepare keeps the points
    whose bin passed the threshold and the pool reads ``wts`` through compact
    ``ranks_depth``.  ``eps = 0``: the dense lift to the bit."""

    def __init__(self, win, wts, D, eps):
        self.win, self.wts = win, wts
        self.D, self.K, self.eps = int(D), int(wts.shape[-1]), float(eps)

    @property
    def shape(self):
        B, N, H, W = self.wts.shape[:4]
        return (B, N, self.D, H, W)

    @property
    def is_cuda(self):
        return self.wts.is_cuda

    @property
    def device(self):
        return self.wts.device

    def _windows(self):
        k = torch.arange(self.D, device=self.wts.device)
        k0 = (self.win[..., 0] & 0xffff)[..., None]
        nk = (self.win[..., 0] >> 16)[..., None]
        j = k - k0
        return k, j, (j >= 0) & (j < nk)

    def kept(self):
        """bool (B,N,D,H,W): the bins the lift keeps (weight >= eps)."""
        k, _, inwin = self._windows()
        y = self.win[..., 1]
        q0 = (y & 0xffff)[..., None]
        nq = ((y >> 16) & 0x7fff)[..., None]
        tail = (y < 0)[..., None]
        keep = ((k >= q0) & (k < q0 + nq)) | (tail & ~inwin)
        return keep.permute(0, 1, 4, 2, 3)

    def dense(self, thresholded=False):
        """The (B,N,D,H,W) tensor ``get_two_hot_depth`` returns (tests, fallbacks);
        ``thresholded``: zero where the lift drops the point."""
        _, j, inwin = self._windows()
        w = torch.gather(self.wts[..., 1:], -1, j.clamp(0, self.K - 2))
        out = torch.where(inwin, w, self.wts[..., :1]).permute(0, 1, 4, 2, 3)
        if thresholded:
            out = torch.where(self.kept(), out, torch.zeros_like(out))
        return out.contiguous()


def two_hot_windows_torch(depths, D, lo, step, gamma=4, eps=0.0, min_gap=-16.0):
    """``TwoHotWindows`` by torch ops (CPU mirror of csrc k_two_hot_window)."""
    B, N, H, W = depths.shape
    K = two_hot_window_slots(D, step, gamma)
    centers = torch.arange(D + 1, device=depths.device) * step + (lo + step / 2)
    gap = -(depths.reshape(B, N, H, W, 1) - centers.view(1, 1, 1, 1, -1)).abs() * gamma
    un = gap >= min_gap
    gap = torch.where(un, gap, torch.full_like(gap, min_gap))
    mx = gap.max(-1, keepdim=True).values
    e = torch.exp(gap - mx)
    s = e.sum(-1, keepdim=True)
    w = e / s
    tail = torch.exp(min_gap - mx) / s
    und = un[..., :D]
    k = torch.arange(D, device=depths.device)
    big = torch.full_like(k, D)
    k0 = torch.where(und, k, big).min(-1).values
    k1 = torch.where(und, k, torch.full_like(k, -1)).max(-1).values
    nk = (k1 - k0 + 1).clamp(min=0, max=K - 1)
    j = torch.arange(K - 1, device=depths.device)
    idx = (k0[..., None] + j).clamp(max=D - 1)
    ww = torch.where(j < nk[..., None], torch.gather(w[..., :D], -1, idx),
                     torch.zeros((), device=depths.device))
    ok = (ww >= eps) & (j < nk[..., None])
    q0 = torch.where(ok, j, torch.full_like(j, K)).min(-1).values
    q1 = torch.where(ok, j, torch.full_like(j, -1)).max(-1).values
    nq = (q1 - q0 + 1).clamp(min=0)
    q0 = torch.where(nq > 0, q0, torch.zeros_like(q0))
    t = (tail[..., 0] >= eps).to(torch.int64) << 31
    y = (k0 + q0) | (nq << 16) | t
    y = torch.where(y >= 2 ** 31, y - 2 ** 32, y)
    win = torch.stack((k0 | (nk << 16), y), -1).to(torch.int32)
    wts = torch.cat((tail, ww), -1).float().contiguous()
    return TwoHotWindows(win.contiguous(), wts, D, eps)


def two_hot_windows(depths, D, lo, step, gamma=4, eps=0.0, downsample=0):
    """Metric depth (B,N,H,W) -> ``TwoHotWindows``; ``downsample = ds`` fuses the
    block-min of ``downsample_depth`` (input at ds x the output resolution)."""
    if depths.is_cuda:
        return _hip().two_hot_windows(depths, D, lo, step, gamma, eps, downsample)
    if downsample:
        depths = downsample_depth_torch(depths, downsample)
    return two_hot_windows_torch(depths, D, lo, step, gamma, eps)


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
