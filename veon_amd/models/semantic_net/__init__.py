from .clip_blocks import ClipVisualTrunk, ResidualAttentionBlock

__all__ = ['ResidualAttentionBlock', 'ClipVisualTrunk']
