"""``@persistent_class``: records constructor arguments on the instance.

The reference version (torch_utils/persistence.py:37-133) additionally embeds module
source in pickles; callers on the hot path only rely on ``init_args`` / ``init_kwargs``
(reenact_avatar_next3d.py:158), which is what is kept here."""
import copy
import functools


def persistent_class(cls):
    orig_init = cls.__init__

    @functools.wraps(orig_init)
    def __init__(self, *args, **kwargs):
        if not hasattr(self, '_init_args'):  # outermost constructor wins
            self._init_args = copy.deepcopy(args)
            self._init_kwargs = copy.deepcopy(kwargs)
        orig_init(self, *args, **kwargs)

    cls.__init__ = __init__
    cls.init_args = property(lambda self: copy.deepcopy(self._init_args))
    cls.init_kwargs = property(lambda self: copy.deepcopy(self._init_kwargs))
    cls._persistent = True
    return cls


def is_persistent(obj):
    return bool(getattr(obj, '_persistent', False))
