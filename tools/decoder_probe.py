#!/usr/bin/env python
"""Where does the time of VeonOccupancyPath's 3-D part go?"""
import os
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from veon_amd import synthetic  # noqa: E402
from veon_amd.models.veon_occ import VeonOccupancyPath  # noqa: E402
from veon_amd.models.semantic_net import semantic_inference_3d_fused  # noqa: E402
from tools.hotpath_bench import timeit  # noqa: E402

dev = 'cuda:0'
size = (256, 704)
torch.manual_seed(0)
net = VeonOccupancyPath(input_size=size).to(dev).eval()
geom = [t.to(dev) for t in synthetic.rig_inputs(synthetic.make_rig(1, 6, size))]
images = torch.randn(1, 6, 3, *size, device=dev)
img = images.flatten(0, 1)
dec = net.occ_decoder
with torch.no_grad():
    feats, supp = net.clip_features(img)
    depth = net.estimate_depth(img)
    metas = list(geom[:5]) + [geom[5][None]]
    print('prepare_depth %.3f ms' % timeit(lambda: dec.prepare_depth(depth)))
    print('prepare_meta  %.3f ms' % timeit(lambda: dec.prepare_meta(metas)))
    d2, m2 = dec.prepare_depth(depth), dec.prepare_meta(metas)
    fl = dec.fusion_layers['layer_0']
    print('CatFusionLift %.3f ms' % timeit(lambda: fl(supp, feats[12], (16, 44))))
    fused = dec.prepare_feat_for_lifting(fl(supp, feats[12], (16, 44)))
    vol = dec._lift_volume(1, 256, dev)
    print('lift          %.3f ms' % timeit(lambda: net.view_transformer([fused] + m2, d2, out_volume=vol)))
    body = dec.__dict__['_body']
    print('body          %.3f ms' % timeit(lambda: body(vol, return_volume=True)))
    x = body(vol, return_volume=True)
    print('occ head      %.3f ms' % timeit(lambda: dec.occupancy_pred(x)))
    print('sem head      %.3f ms' % timeit(lambda: dec.feat_pred(x, return_volume=True)))
    feat = dec.feat_pred(x, return_volume=True)
    print('classifier+up %.3f ms' % timeit(lambda: semantic_inference_3d_fused(net.ov_classifier_weight, feat, net.occ_size)))
    sem = semantic_inference_3d_fused(net.ov_classifier_weight, feat, net.occ_size)
    b = F.interpolate(dec.occupancy_pred(x), size=net.occ_size, mode='trilinear', align_corners=False)

    def post():
        score, cls = torch.softmax(sem, dim=1).max(dim=1)
        keep = (score > 0.0) & (torch.softmax(b, dim=1)[:, 0] > 0.5)
        return torch.where(keep, cls, torch.full_like(cls, 17)).permute(0, 3, 2, 1).contiguous()
    print('post (softmax/argmax/where) %.3f ms' % timeit(post))
