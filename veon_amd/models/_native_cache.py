"""Invalidation of the packed / folded weight copies the native path keeps.

A module that caches derived tensors (bf16-packed weights, BN folds, scratch
grids) in ``self.__dict__`` lists the keys in ``_native_cache``; they are dropped
whenever the parameters may have changed or moved: ``train()`` / ``eval()``,
``load_state_dict`` and ``_apply`` (``.to()``, ``.cuda()``, ``.half()`` ...).
"""


class NativeCacheMixin:
    _native_cache = ()

    def _drop_native_cache(self):
        for key in self._native_cache:
            self.__dict__.pop(key, None)

    def train(self, mode=True):
        self._drop_native_cache()
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kwargs):
        self._drop_native_cache()
        return super()._load_from_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        self._drop_native_cache()
        return super()._apply(fn, *args, **kwargs)
