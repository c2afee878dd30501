from .align_net_body import (AlignBody3D, ConvModule3d, PredHead3DOcc,
                             PredHead3DSem, ResBlock3D, classifier_logits_low,
                             semantic_inference_3d, semantic_inference_3d_fused)
from .align_net_occ3d import AlignNetOcc3D
from .fusion_layers import (AddFusionLift, CatFusionLift, LayerNorm,
                            build_fusion_layer_lift)
from .clip_blocks import ClipRecHead, ClipVisualTrunk, ResidualAttentionBlock
from .side_adapter import (MLPMaskDecoder, RegionwiseSideAdapterNetwork, SideAdapterViT,
                           semantic_branch_2d, semantic_inference_2d_w_embed)

__all__ = ['AlignNetOcc3D', 'CatFusionLift', 'AddFusionLift', 'LayerNorm',
           'build_fusion_layer_lift', 'ResidualAttentionBlock', 'ClipVisualTrunk', 'ClipRecHead', 'ResBlock3D',
           'ConvModule3d', 'AlignBody3D', 'PredHead3DOcc', 'PredHead3DSem',
           'semantic_inference_3d', 'semantic_inference_3d_fused', 'RegionwiseSideAdapterNetwork',
           'MLPMaskDecoder', 'SideAdapterViT', 'semantic_branch_2d',
           'semantic_inference_2d_w_embed']
