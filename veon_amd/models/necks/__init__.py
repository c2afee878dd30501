from .view_transformer import (LSSViewTransformer, LSSViewTransformerBEVDepth,
                               LSSViewTransformerBEVStereo)
from .view_transformer_raw import LSSViewTransformerRaw

__all__ = ['LSSViewTransformer', 'LSSViewTransformerBEVDepth',
           'LSSViewTransformerBEVStereo', 'LSSViewTransformerRaw']
