"""Golden vectors for the temporal path.

TEST INFRASTRUCTURE (fixture generation, run in the build container only).
Loads the reference's own files by path, unmodified:

* ``side_adapter/align_net_occ3d.py`` -> ``TemporalFusionMultiFrame`` with its
  ``BeforeFusionLayer``, ``TemporalFusionMultiFrameMiddle3x3Seq`` and
  ``TemporalDeformable`` (:13-204), run on seeded volumes.  mmcv's ``ConvModule``
  is the stand-in of gen_golden_body.py (conv -> norm -> act, documented order).
* ``san_in_veon_temporal.py`` -> ``SANInVeonTemporal.align_after_lss`` (:325-365),
  called unbound on a namespace carrying ``grid_config`` / ``ds_feat``.  The
  module's third-party imports (open_clip, detectron2) and sibling packages are
  name-only stubs; the method itself uses torch only.

    python oracle/tools/gen_golden_temporal.py  ->  tests/golden/temporal_tiny.npz
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402
import gen_golden_body  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(HERE))


class _MetaArch:
    def register(self):
        return lambda cls: cls


def load_san():
    ident = lambda fn: fn
    ref_import._mod('open_clip')
    ref_import._mod('detectron2')
    ref_import._mod('detectron2.config', configurable=ident)
    ref_import._mod('detectron2.modeling', META_ARCH_REGISTRY=_MetaArch())
    ref_import._mod('detectron2.modeling.postprocessing', sem_seg_postprocess=None)
    ref_import._mod('detectron2.structures', ImageList=None)
    ref_import._mod('detectron2.utils')
    ref_import._mod('detectron2.utils.memory', retry_if_cuda_oom=ident)
    pkg = 'refsan'
    ref_import._mod(pkg)
    ref_import._mod(pkg + '.clip_utils', ClipOutput=None, FeatureExtractor=None,
                    LearnableBgOvClassifier=None, PredefinedOvClassifier=None,
                    RecWithAttnbiasHead=None, get_predefined_templates=None)
    ref_import._mod(pkg + '.side_adapter', build_side_adapter_network_in_veon=None,
                    build_hsa_network=None)
    ref_import._mod(pkg + '.side_adapter.align_net_occ3d', AlignNetOcc3D=None)
    return ref_import.load('mmdet3d/models/semantic_net/san_in_veon_temporal.py',
                           pkg + '.san_in_veon_temporal')


def rigid(gen, angle, shift):
    a = (torch.rand(1, generator=gen).item() - 0.5) * 2 * angle
    m = torch.eye(4)
    m[0, 0], m[0, 1], m[1, 0], m[1, 1] = np.cos(a), -np.sin(a), np.sin(a), np.cos(a)
    m[:3, 3] = (torch.rand(3, generator=gen) - 0.5) * 2 * shift
    return m


def main():
    out = {}
    gen = torch.Generator().manual_seed(7)
    torch.manual_seed(7)

    # ---- TemporalFusionMultiFrame -------------------------------------------
    ref = gen_golden_body.load_reference()
    C, T = 32, 2
    tf = ref.TemporalFusionMultiFrame(channels=C, seqs=T).eval()
    gen_golden_body.randomise(tf, gen)
    # BatchNorm3d of the deformable layer is a plain nn.BatchNorm3d: covered by
    # randomise().  Larger offset weights so the sampling points move.
    with torch.no_grad():
        tf.deform_fusion_layer.t_deform.offset_conv[2].weight.mul_(6.0)
    cur = torch.randn(2, C, 3, 6, 7, generator=gen)
    prevs = [torch.randn(2, C, 3, 6, 7, generator=gen) for _ in range(T)]
    with torch.no_grad():
        y = tf(cur, [p.clone() for p in prevs])
        d = tf.deform_fusion_layer.t_deform(prevs[0], cur)
    out.update(tf_cur=cur.numpy(), tf_out=y.numpy(), deform_out=d.numpy())
    for i, p in enumerate(prevs):
        out['tf_prev%d' % i] = p.numpy()
    for k, v in tf.state_dict().items():
        out['tf/' + k] = v.numpy()

    # ---- align_after_lss ----------------------------------------------------
    san = load_san()
    grid_config = {'x': [-4.0, 4.0, 0.5], 'y': [-3.0, 3.0, 0.5], 'z': [-1.0, 3.0, 0.5]}
    ds_feat = (2, 2, 2)
    me = types.SimpleNamespace(grid_config=grid_config, ds_feat=ds_feat)
    B, Ca = 2, 5
    occ = torch.randn(B, Ca, 4, 6, 8, generator=gen)
    cur2glob = torch.stack([rigid(gen, 0.3, 5.0) for _ in range(B)])[:, None]
    prev2glob = torch.stack([cur2glob[b, 0] @ rigid(gen, 0.15, 1.2) for b in range(B)])[:, None]
    with torch.no_grad():
        warped = san.SANInVeonTemporal.align_after_lss(me, occ, [cur2glob, prev2glob])
    out.update(align_in=occ.numpy(), align_cur2glob=cur2glob.numpy(),
               align_prev2glob=prev2glob.numpy(), align_out=warped.numpy(),
               align_grid=np.array([grid_config[k] for k in 'xyz'], dtype=np.float64),
               align_ds=np.array(ds_feat))
    path = os.path.join(ROOT, 'tests', 'golden', 'temporal_tiny.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, {k: v.shape for k, v in out.items() if '/' not in k})
    print('nonzero warped fraction', float((warped != 0).float().mean()))


if __name__ == '__main__':
    main()
