from .dinov2 import DinoVisionTransformer, DINOv2, DINOv2Adaptor
from .dpt import DepthAnythingV2Adaptor, DPTHead

__all__ = ['DinoVisionTransformer', 'DINOv2', 'DINOv2Adaptor', 'DPTHead',
           'DepthAnythingV2Adaptor']
