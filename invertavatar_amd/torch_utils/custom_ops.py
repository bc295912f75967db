"""Plugin loader.  The reference JIT-compiles three pybind11 CUDA plugins here
(torch_utils/custom_ops.py:61-157); this backend has ONE prebuilt C-ABI library, so
``get_plugin`` keeps its signature and returns a thin object that exposes the plugin's
function on top of libia_hip.so."""
from .. import _lib

verbosity = 'brief'  # 'none' | 'brief' | 'full', as in the reference

_cached_plugins = {}


class _Plugin:
    """Namespace whose attributes are the Python-level plugin entry points."""

    def __init__(self, name, functions):
        self.__name__ = name
        self.__dict__.update(functions)


def get_plugin(module_name, sources=None, headers=None, source_dir=None, **build_kwargs):
    """Return the plugin object for `module_name` ('bias_act_plugin', 'upfirdn2d_plugin',
    'filtered_lrelu_plugin').  `sources`/`headers`/`build_kwargs` are accepted for signature
    compatibility; the HIP library is built by invertavatar_amd.build."""
    assert verbosity in ('none', 'brief', 'full')
    if module_name in _cached_plugins:
        return _cached_plugins[module_name]
    if verbosity != 'none':
        print(f'Setting up PyTorch plugin "{module_name}"... ', end='', flush=True)
    try:
        _lib.load()
        from .ops import _plugins
        functions = _plugins.TABLE[module_name]
    except Exception:
        if verbosity != 'none':
            print('Failed!')
        raise
    if verbosity != 'none':
        print('Done.')
    plugin = _Plugin(module_name, functions)
    _cached_plugins[module_name] = plugin
    return plugin
