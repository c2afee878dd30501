from .align_net_body import (AlignBody3D, ConvModule3d, PredHead3DOcc,
                             PredHead3DSem, ResBlock3D, semantic_inference_3d,
                             semantic_inference_3d_fused)
from .clip_blocks import ClipRecHead, ClipVisualTrunk, ResidualAttentionBlock

__all__ = ['ResidualAttentionBlock', 'ClipVisualTrunk', 'ClipRecHead', 'ResBlock3D',
           'ConvModule3d', 'AlignBody3D', 'PredHead3DOcc', 'PredHead3DSem',
           'semantic_inference_3d', 'semantic_inference_3d_fused']
