"""EasyDict and name->object lookup.  Mirrors the reference's dnnlib.util API for the
functions the hot path reaches (dnnlib/util.py:42-56, 231-310)."""
import importlib

_PACKAGE = __name__.split('.')[0]  # 'invertavatar_amd'


class EasyDict(dict):
    """dict whose keys are also attributes."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name) from None

    def __setattr__(self, name, value):
        self[name] = value

    def __delattr__(self, name):
        del self[name]


def _resolve(dotted):
    parts = dotted.split('.')
    errors = []
    # Longest module prefix first; reference module paths ('training_avatar_texture....') are
    # also tried below this package, so pickled rendering_kwargs keep working without aliases.
    for prefix in ('', _PACKAGE + '.'):
        for cut in range(len(parts) - 1, 0, -1):
            mod_name = prefix + '.'.join(parts[:cut])
            try:
                obj = importlib.import_module(mod_name)
            except ImportError as exc:
                errors.append(exc)
                continue
            try:
                for attr in parts[cut:]:
                    obj = getattr(obj, attr)
                return obj
            except AttributeError as exc:
                errors.append(exc)
    raise ImportError(f'cannot resolve {dotted!r}: {errors[-1] if errors else "no candidates"}')


def get_obj_by_name(name):
    return _resolve(name)


def call_func_by_name(*args, func_name=None, **kwargs):
    assert func_name is not None
    fn = get_obj_by_name(func_name)
    assert callable(fn)
    return fn(*args, **kwargs)


def construct_class_by_name(*args, class_name=None, **kwargs):
    return call_func_by_name(*args, func_name=class_name, **kwargs)
