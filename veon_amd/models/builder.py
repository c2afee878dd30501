"""Plugin seam: the ``NECKS`` registry and ``build_neck(cfg)``.

Mirror of mmdet3d/models/builder.py:40-45.  When mmdet3d (and therefore mmcv)
is importable the view transformers register into mmdet3d's own ``NECKS``
registry with ``force=True``, so ``configs/veon/*.py`` --
``img_view_transformer=dict(type='LSSViewTransformerRaw', ...)`` -- build this
implementation unchanged.  Otherwise (this image has neither package) a minimal
registry with the same ``register_module()`` / ``build(cfg)`` calls is used.
"""
import inspect


class _Registry:
    """The subset of mmcv.utils.Registry the hot path touches."""

    def __init__(self, name):
        self.name = name
        self.module_dict = {}

    def register_module(self, name=None, force=False, module=None):
        def _register(cls):
            key = name or cls.__name__
            if key in self.module_dict and not force:
                raise KeyError('%s is already registered in %s' % (key, self.name))
            self.module_dict[key] = cls
            return cls
        if module is not None:
            return _register(module)
        return _register

    def get(self, key):
        return self.module_dict.get(key)

    def build(self, cfg, default_args=None):
        if not isinstance(cfg, dict) or 'type' not in cfg:
            raise TypeError('cfg must be a dict with a "type" key, got %r' % (cfg,))
        args = dict(cfg)
        if default_args:
            for k, v in default_args.items():
                args.setdefault(k, v)
        obj_type = args.pop('type')
        if isinstance(obj_type, str):
            cls = self.get(obj_type)
            if cls is None:
                raise KeyError('%s is not in the %s registry' % (obj_type, self.name))
        elif inspect.isclass(obj_type):
            cls = obj_type
        else:
            raise TypeError('type must be a str or class, got %r' % (obj_type,))
        return cls(**args)


try:  # pragma: no cover - mmdet3d is not installed in the build image
    from mmdet3d.models.builder import NECKS as _MMDET3D_NECKS
    NECKS = _MMDET3D_NECKS
    HAVE_MMDET3D = True
except Exception:  # ImportError, or mmcv version asserts
    NECKS = _Registry('neck')
    HAVE_MMDET3D = False


def register_neck():
    """Decorator used by the view transformers: override mmdet3d's class of the
    same name when mmdet3d is present."""
    return NECKS.register_module(force=True)


def build_neck(cfg):
    """mmdet3d/models/builder.py:40-45."""
    return NECKS.build(cfg)
