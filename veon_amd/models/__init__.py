from .builder import NECKS, build_neck
from . import necks  # noqa: F401  (registers the view transformers)
from . import depth_anything  # noqa: F401  (registers DepthAnythingV2Adaptor)

__all__ = ['NECKS', 'build_neck']
