from .builder import NECKS, build_neck
from . import necks  # noqa: F401  (registers the view transformers)

__all__ = ['NECKS', 'build_neck']
