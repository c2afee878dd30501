"""Temporal path of VEON's 3D alignment network -- mirrors of

* ``TemporalFusionMultiFrame`` and its parts ``BeforeFusionLayer``,
  ``TemporalFusionMultiFrameMiddle3x3Seq``, ``TemporalDeformable``,
  ``TemporalFusionDeformMiddle``
  (mmdet3d/models/semantic_net/side_adapter/align_net_occ3d.py:13-204), with the
  reference's attribute names so a VEON checkpoint's ``temporal_fusion.*`` keys
  load unchanged;
* ``SANInVeonTemporal.align_after_lss``
  (mmdet3d/models/semantic_net/san_in_veon_temporal.py:325-365): the rigid warp
  of a past frame's lifted volume into the current ego frame.

The PyTorch formulation below is the definition (CPU, training); it is pinned by
vectors generated from the reference's own classes
(oracle/tools/gen_golden_temporal.py).  Two quirks of the reference are kept on
purpose because trained weights depend on them: ``TemporalDeformable`` stacks its
base grid as (z, y, x) while ``grid_sample`` reads the last axis as (x, y, z), so
the W-axis sampling position is driven by the voxel's z index and vice versa; and
the offsets are divided by (D, H, W) in normalised units.

On a ROCm device at inference the whole fusion runs on ``PaddedVolume`` rows
(csrc/conv3d.hip, csrc/temporal.hip): 3x3x3 convs on the implicit-GEMM MFMA
kernel with BN/ReLU/GELU epilogues, 1x1x1 projections as GEMMs, the deformable
sampling + 8-sample attention in one gather kernel, and the warp as a trilinear
gather on the channels-last grid (the zero halo is the zero padding).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import conv3d_ops, vit_ops
from .._native_cache import NativeCacheMixin
from .align_net_body import ConvModule3d
from ... import half as _half


def _conv_bn(cin, cout):
    return ConvModule3d(cin, cout, 3, 1, 1, bias=False, norm=True, act=True)


def _fast(x):
    return isinstance(x, conv3d_ops.PaddedVolume)


def _conv_hip(cm, vol, out=None):
    """ConvModule3d (3x3x3 + BN + ReLU) on a PaddedVolume."""
    if cm.__dict__.get('_hip3') is None:
        cm.__dict__['_hip3'] = cm.folded(pack=True)
    w, sc, sh = cm.__dict__['_hip3']
    return conv3d_ops.conv3d_k3(vol, w, sc, sh, relu=cm.activate is not None, out=out)


def _cat(vols):
    """Channel concatenation of PaddedVolumes (halo rows stay zero)."""
    B, _, Z, Y, X = vols[0].shape
    C = sum(v.shape[1] for v in vols)
    out = conv3d_ops.PaddedVolume.__new__(conv3d_ops.PaddedVolume)
    out.shape = (B, C, Z, Y, X)
    out.guard, out.M = vols[0].guard, vols[0].M
    out.storage = torch.cat([v.storage for v in vols], dim=1)
    out.rows = out.storage[out.guard:out.guard + out.M]
    return out


class BeforeFusionLayer(nn.Module):
    """One shared conv applied to every frame (align_net_occ3d.py:77-86)."""

    def __init__(self, channels):
        super().__init__()
        self.offset_conv = _conv_bn(channels, channels)

    def forward(self, feat_list):
        if _fast(feat_list[0]):
            return [_conv_hip(self.offset_conv, f) for f in feat_list]
        return [self.offset_conv(f) for f in feat_list]


class TemporalFusionMultiFrameMiddle3x3Seq(nn.Module):
    """Folds the past frames oldest-first, two at a time, then the current one
    (align_net_occ3d.py:25-46).  Returns (reference feature, folded past)."""

    def __init__(self, channels, seqs=2):
        super().__init__()
        self.t_fuse = nn.ModuleList([_conv_bn(channels * 2, channels)
                                     for _ in range(seqs)])

    def forward(self, cur_occ_feat, prev_occ_feats):
        fast = _fast(cur_occ_feat)
        cat = _cat if fast else (lambda ts: torch.cat(ts, dim=1))
        conv = (lambda cm, v: _conv_hip(cm, v)) if fast else (lambda cm, v: cm(v))
        past = prev_occ_feats[-1]
        idx = 0
        for frame in prev_occ_feats[-2::-1]:
            past = conv(self.t_fuse[idx], cat([frame, past]))
            idx += 1
        ref = conv(self.t_fuse[idx], cat([cur_occ_feat, past]))
        return ref, past


class TemporalDeformable(NativeCacheMixin, nn.Module):
    """Deformable cross-frame attention (align_net_occ3d.py:89-204): queries and
    sampling offsets from one volume, keys/values sampled trilinearly from the
    other at ``num_samples`` points per head, softmax over the samples."""

    def __init__(self, channels, num_heads=4, num_samples=8):
        super().__init__()
        assert channels % num_heads == 0
        self.channels, self.num_heads, self.num_samples = channels, num_heads, num_samples
        self.head_dim = channels // num_heads
        self.offset_conv = nn.Sequential(
            nn.Conv3d(channels, channels, 3, padding=1),
            nn.GELU(),
            nn.Conv3d(channels, num_heads * num_samples * 3, 3, padding=1, bias=False),
            nn.Tanh())
        self.key_value_proj = nn.Conv3d(channels, channels * 2, 1)
        self.query_proj = nn.Conv3d(channels, channels, 1)
        self.out_proj = nn.Conv3d(channels, channels, 1)
        self.final_norm = nn.BatchNorm3d(channels)
        self.out_activate = nn.ReLU()

    # ------------------------------------------------------------ definition
    def sampling_grid(self, offsets):
        """offsets (N, S, 3, D, H, W) in tanh units -> grid_sample grid
        (N, S*D, H, W, 3).  Component a of the last axis is built from the a-th
        of (z, y, x) and divided by the a-th of (D, H, W) -- the reference's
        order, which grid_sample then reads as (x, y, z)."""
        N, S, _, D, H, W = offsets.shape
        dev, dt = offsets.device, offsets.dtype
        axes = [torch.linspace(-1, 1, n, device=dev, dtype=dt) for n in (D, H, W)]
        base = torch.stack(torch.meshgrid(*axes, indexing='ij'), dim=0)  # (3,D,H,W)
        size = torch.tensor([D, H, W], device=dev, dtype=dt).view(1, 1, 3, 1, 1, 1)
        g = (base.view(1, 1, 3, D, H, W) + offsets / size).clamp(-1, 1)
        return g.permute(0, 1, 3, 4, 5, 2).reshape(N, S * D, H, W, 3)

    def attend(self, kv, query, offsets):
        """kv (B, 2C, D,H,W) laid out per head as [key | value]; query (B,C,...);
        offsets (B, heads*S*3, ...) after tanh -> fused (B, C, D, H, W)."""
        B, C, D, H, W = query.shape
        nh, S, hd = self.num_heads, self.num_samples, self.head_dim
        grid = self.sampling_grid(offsets.reshape(B * nh, S, 3, D, H, W))
        got = F.grid_sample(kv.reshape(B * nh, 2 * hd, D, H, W), grid, mode='bilinear',
                            padding_mode='border', align_corners=True)
        got = got.view(B, nh, 2 * hd, S, D, H, W)
        key, value = got[:, :, :hd], got[:, :, hd:]
        q = query.view(B, nh, hd, 1, D, H, W) * hd ** -0.5
        attn = (q * key).sum(dim=2, keepdim=True).softmax(dim=3)
        return (attn * value).sum(dim=3).reshape(B, C, D, H, W)

    def forward(self, feat_prev, feat_curr):
        if _fast(feat_curr):
            return self.hip_forward(feat_prev, feat_curr)
        fused = self.attend(self.key_value_proj(feat_prev), self.query_proj(feat_curr),
                            self.offset_conv(feat_curr))
        return self.out_activate(self.final_norm(self.out_proj(fused)))

    # ------------------------------------------------------------- MFMA path
    _native_cache = ('_hip',)

    def _packed(self):
        p = self.__dict__.get('_hip')
        if p is None:
            C = self.channels
            bf = lambda w: w.detach().reshape(w.shape[0], -1).to(_half.dtype()).contiguous()
            c1, c2 = self.offset_conv[0], self.offset_conv[2]
            bn = self.final_norm
            g = bn.weight.detach().float() / torch.sqrt(bn.running_var.float() + bn.eps)
            p = dict(
                w1=conv3d_ops.pack_weight(c1.weight), b1=c1.bias.detach().float().contiguous(),
                w2=conv3d_ops.pack_weight(c2.weight),
                wkv=bf(self.key_value_proj.weight),
                bkv=self.key_value_proj.bias.detach().float().contiguous(),
                wq=bf(self.query_proj.weight),
                bq=self.query_proj.bias.detach().float().contiguous(),
                wo=bf(self.out_proj.weight), go=g.contiguous(),
                so=((self.out_proj.bias.detach().float() - bn.running_mean.float()) * g
                    + bn.bias.detach().float()).contiguous())
            self.__dict__['_hip'] = p
        return p

    def project_kv(self, feat_prev):
        """key/value projection of the sampled volume; both calls of
        ``TemporalFusionDeformMiddle`` share it."""
        p = self._packed()
        kv = feat_prev.like(2 * self.channels)
        vit_ops.linear(feat_prev.rows, p['wkv'], p['bkv'], vit_ops.EPI_BF16, out=kv.rows)
        return kv

    def hip_forward(self, feat_prev, feat_curr, kv=None):
        p = self._packed()
        if kv is None:
            kv = self.project_kv(feat_prev)
        q = feat_curr.like()
        vit_ops.linear(feat_curr.rows, p['wq'], p['bq'], vit_ops.EPI_BF16, out=q.rows)
        hid = conv3d_ops.conv3d_k3(feat_curr, p['w1'], None, p['b1'], act='gelu')
        off = conv3d_ops.conv3d_k3(hid, p['w2'])          # tanh is taken in the gather
        fused = conv3d_ops.deform_attention(kv, q, off, self.num_heads, self.num_samples)
        out = feat_curr.like()
        vit_ops.linear(fused.rows, p['wo'], p['so'], vit_ops.EPI_AFFINE_RELU, out=out.rows,
                       gamma=p['go'])
        conv3d_ops.zero_halo(out)   # the GEMM wrote relu(shift) into the halo rows
        return out


class TemporalFusionDeformMiddle(nn.Module):
    """(align_net_occ3d.py:13-22) concatenates the reference feature with its two
    deformable views: of the current frame and of the folded past."""

    def __init__(self, channels):
        super().__init__()
        self.t_deform = TemporalDeformable(channels=channels)

    def forward(self, mid_feat, occ_feat, prev_feat):
        if _fast(mid_feat):
            kv = self.t_deform.project_kv(mid_feat)
            a = self.t_deform.hip_forward(mid_feat, occ_feat, kv)
            b = self.t_deform.hip_forward(mid_feat, prev_feat, kv)
            return _cat([mid_feat, a, b])
        a = self.t_deform(mid_feat, occ_feat)
        b = self.t_deform(mid_feat, prev_feat)
        return torch.cat([mid_feat, a, b], dim=1)


class TemporalFusionMultiFrame(nn.Module):
    """(align_net_occ3d.py:49-74) current volume + aligned past volumes -> one
    volume of the same shape.  Accepts (B,C,Z,Y,X) tensors, or PaddedVolumes on
    the MFMA path (``hip_ok``)."""

    def __init__(self, channels, seqs=2):
        super().__init__()
        self.t_final = _conv_bn(channels * 3, channels)
        self.before_fusion_layer = BeforeFusionLayer(channels)
        self.t_fuse_mid = TemporalFusionMultiFrameMiddle3x3Seq(channels, seqs=seqs)
        self.deform_fusion_layer = TemporalFusionDeformMiddle(channels)

    def hip_ok(self, x):
        c = self.t_final.conv.out_channels
        return (not self.training and not torch.is_grad_enabled() and c % 64 == 0
                and (c // self.deform_fusion_layer.t_deform.num_heads) in (32, 64)
                and self.deform_fusion_layer.t_deform.num_samples == 8
                and (_fast(x) or x.is_cuda))

    def forward(self, cur_occ_feat, prev_occ_feats):
        feats = self.before_fusion_layer([cur_occ_feat] + list(prev_occ_feats))
        cur, prevs = feats[0], feats[1:]
        ref, past = self.t_fuse_mid(cur, prevs)
        fused = self.deform_fusion_layer(ref, cur, past)
        if _fast(fused):
            return _conv_hip(self.t_final, fused)
        return self.t_final(fused)

    def forward_fast(self, cur_occ_feat, prev_occ_feats):
        """fp32 (B,C,Z,Y,X) ROCm tensors in and out, MFMA path inside."""
        vols = [conv3d_ops.pack(t) for t in [cur_occ_feat] + list(prev_occ_feats)]
        return conv3d_ops.unpack(self.forward(vols[0], vols[1:]))


# --------------------------------------------------------------------- the warp
def voxel_centres(grid_config, ds_feat, shape):
    """First voxel centre and step (x, y, z) of the max-pooled grid
    (san_in_veon_temporal.py:326-340); ``ds_feat`` is given in (z, y, x) order."""
    step = [grid_config[s][2] * ds_feat[i] for i, s in enumerate(['z', 'y', 'x'])][::-1]
    first = [grid_config[s][0] + st / 2 for s, st in zip(['x', 'y', 'z'], step)]
    return first, step


def prev_from_cur(adj_metas):
    """(B,4,4) current-ego -> past-ego transforms (san_in_veon_temporal.py:343-345)."""
    cur2glob, prev2glob = adj_metas
    return torch.linalg.inv(prev2glob[:, 0]) @ cur2glob[:, 0]


def align_after_lss(occ_feat, adj_metas, grid_config, ds_feat):
    """Resample a past frame's volume (B,C,Z,Y,X) at the positions the current
    frame's voxel centres have in the past ego frame; trilinear, zeros outside
    (san_in_veon_temporal.py:325-365).  PaddedVolume in -> PaddedVolume out."""
    if _fast(occ_feat):
        B, C, Z, Y, X = occ_feat.shape
        first, step = voxel_centres(grid_config, ds_feat, (Z, Y, X))
        dev = occ_feat.device
        A = conv3d_ops.warp_affine(adj_metas[0].to(dev), adj_metas[1].to(dev), first, step)
        return conv3d_ops.warp_volume(occ_feat, A)
    B, C, Z, Y, X = occ_feat.shape
    dev, dt = occ_feat.device, occ_feat.dtype
    first, step = voxel_centres(grid_config, ds_feat, (Z, Y, X))
    axes = [torch.arange(n, device=dev) * st + f0
            for n, st, f0 in zip((X, Y, Z), step, first)]
    # reference order: the flattened list runs x-major (meshgrid 'ij' over x,y,z)
    pts = torch.stack(torch.meshgrid(*axes, indexing='ij'), dim=-1).to(dt)  # (X,Y,Z,3)
    T = prev_from_cur(adj_metas).to(dt)
    moved = torch.einsum('xyzj,bij->bxyzi', pts, T[:, :3, :3]) + T[:, None, None, None, :3, 3]
    moved = moved.permute(0, 3, 2, 1, 4)                                   # (B,Z,Y,X,3)
    lo = pts[0, 0, 0]
    span = pts[-1, -1, -1] - lo
    grid = (moved - lo) / span * 2 - 1
    return F.grid_sample(occ_feat, grid, mode='bilinear', padding_mode='zeros',
                         align_corners=True)
