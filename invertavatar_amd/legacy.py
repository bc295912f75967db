"""Loading the reference's network pickles (API of the reference's legacy.py:24-60, ``load_network_pkl``) on this backend.

A reference pickle stores every ``@persistence.persistent_class`` object as a call
``torch_utils.persistence._reconstruct_persistent_obj(meta)`` with ``meta = {type: 'class', version, module_src, class_name,
state}`` (torch_utils/persistence.py:112-123, :186-207): the reference re-creates the object by EXECUTING the pickled module
source.  This loader never executes pickled SOURCE (globals other than this package's classes, torch.nn layers and the
tensor / ndarray reconstruction helpers are refused, see _SAFE_GLOBALS; a pickle is still untrusted input): it reads the same byte stream, keeps ``class_name`` / ``_init_args`` /
``_init_kwargs`` / parameters / buffers of each persistent object in a ``PickledModule`` shell, and then builds THIS backend's
class of the same name from the recorded constructor arguments and loads the tensors by name -- the documented upgrade recipe of
the reference (persistence.py:84-90, reenact_avatar_next3d.py:158-161: ``G_new = TriPlaneGenerator(*G.init_args,
**G.init_kwargs); misc.copy_params_and_buffers(G, G_new, require_all=True)``) done at load time.  Plain (non-persistent) classes
pickled by reference to a reference module path (``encoder_inversion.models.uvnet.inversionNet`` ...) resolve to this package's
module of the same path.  TensorFlow-era pickles and discriminators (training side) are out of scope: the former raise, the
latter stay shells."""
import collections
import copy
import importlib
import io
import pickle

import torch

from . import dnnlib

_PACKAGE = __name__.rsplit('.', 1)[0]

# persistent class name -> this backend's class
CLASS_REGISTRY = {
    'TriPlaneGenerator': f'{_PACKAGE}.training_avatar_texture.triplane_v20.TriPlaneGenerator',
}


class PickledModule:
    """Shell of one persistent object of a reference pickle: what was recorded, nothing executed."""

    def __init__(self, meta):
        self.class_name = meta['class_name']
        self.version = meta.get('version')
        self.state = dict(meta['state'] or {})

    @property
    def init_args(self):
        return copy.deepcopy(self.state.get('_init_args', ()))

    @property
    def init_kwargs(self):
        return dnnlib.util.EasyDict(copy.deepcopy(self.state.get('_init_kwargs', {})))

    def __repr__(self):
        return f'PickledModule({self.class_name})'


def _reconstruct_persistent_obj(meta):
    if meta.get('type') != 'class':
        raise pickle.UnpicklingError(f"persistent object of type {meta.get('type')!r} is not supported")
    return PickledModule(meta)


def _module_dicts(obj):
    """(_parameters, _buffers, _modules, non-persistent buffer names) of a shell or of a regular torch module."""
    st = obj.state if isinstance(obj, PickledModule) else obj.__dict__
    return (st.get('_parameters') or {}, st.get('_buffers') or {}, st.get('_modules') or {}, st.get('_non_persistent_buffers_set') or set())


def collect_state_dict(obj, prefix=''):
    """Flat {name: tensor} of a tree of shells and regular modules (the names torch's state_dict() would give)."""
    out = collections.OrderedDict()
    params, buffers, modules, transient = _module_dicts(obj)
    for name, p in params.items():
        if p is not None:
            out[prefix + name] = p.detach()
    for name, b in buffers.items():
        if b is not None and name not in transient:
            out[prefix + name] = b.detach()
    for name, child in modules.items():
        if child is not None:
            out.update(collect_state_dict(child, f'{prefix}{name}.'))
    return out


def _resolve_class(class_name):
    target = CLASS_REGISTRY.get(class_name)
    if target is None:
        return None
    mod, attr = target.rsplit('.', 1)
    return getattr(importlib.import_module(mod), attr)


def _convert_args(value):
    """Shells inside recorded constructor arguments (e.g. inversionNet(generator=<TriPlaneGenerator>)) become built modules."""
    if isinstance(value, PickledModule):
        return build_module(value)
    if isinstance(value, dict):
        return type(value)((k, _convert_args(v)) for k, v in value.items())
    if isinstance(value, (list, tuple)):
        return type(value)(_convert_args(v) for v in value)
    return value


_PLAIN_ATTRS = ('neural_rendering_resolution', 'rendering_kwargs', 'fill_mouth')


def build_module(shell):
    """This backend's module for a shell: constructor arguments as recorded, tensors loaded by name (strict)."""
    cls = _resolve_class(shell.class_name)
    if cls is None:
        raise KeyError(f'no class registered for persistent class {shell.class_name!r}')
    obj = cls(*_convert_args(shell.init_args), **_convert_args(dict(shell.init_kwargs)))
    obj.load_state_dict(collect_state_dict(shell), strict=True)
    for name in _PLAIN_ATTRS:          # attributes the scripts assign after construction and expect to survive the pickle
        if name in shell.state:
            setattr(obj, name, copy.deepcopy(shell.state[name]))
    obj.train(bool(shell.state.get('training', True)))
    return obj


def _adopt_children(module):
    """Replace shells hanging below a regular module (inversionNet.generator) by built modules, recursively."""
    for name, child in list(module._modules.items()):
        if isinstance(child, PickledModule):
            module._modules[name] = _materialize(child)
        elif isinstance(child, torch.nn.Module):
            _adopt_children(child)
    return module


def _materialize(obj):
    if isinstance(obj, PickledModule):
        return build_module(obj) if _resolve_class(obj.class_name) is not None else obj
    if isinstance(obj, torch.nn.Module):
        return _adopt_children(obj)
    return obj


def _load_storage_from_bytes(b):
    """Stand-in for torch.storage._load_from_bytes inside network pickles: same bytes, restricted (weights-only) unpickler."""
    return torch.load(io.BytesIO(b), weights_only=True)


class _Unpickler(pickle.Unpickler):
    """Resolves only (a) classes DEFINED in this package's mirror of the reference's module paths, (b) torch.nn layer classes,
    (c) the reconstruction helpers of _SAFE_GLOBALS.  Names are never walked: protocol >= 4 lets a pickle name
    'module', 'attr.attr.attr' and pickle.Unpickler.find_class follows the dots through whatever the module imported
    ('invertavatar_amd.hipops', '_os.system'), so a dotted name is refused outright and a resolved object must be a class
    whose __module__ lies below the prefix it was asked for (an imported `os`, `subprocess`, `importlib` never qualifies)."""

    @staticmethod
    def _class_defined_below(module, name, prefix):
        try:
            obj = getattr(importlib.import_module(module), name)
        except (ImportError, AttributeError):
            return None
        owner = getattr(obj, '__module__', None)
        if isinstance(obj, type) and isinstance(owner, str) and (owner == prefix or owner.startswith(prefix + '.')):
            return obj
        return None

    def find_class(self, module, name):
        if '.' in name:
            raise pickle.UnpicklingError(f'dotted global {module}:{name} is refused by the network-pickle loader')
        if module == 'torch_utils.persistence' and name == '_reconstruct_persistent_obj':
            return _reconstruct_persistent_obj
        if module == 'dnnlib.tflib.network':
            raise pickle.UnpicklingError('TensorFlow-era StyleGAN pickles are outside this backend (legacy.py:27-33 converts them in the reference)')
        if module.split('.')[0] == 'dnnlib':
            if name != 'EasyDict':
                raise pickle.UnpicklingError(f'global {module}.{name} is not on the allow-list of the network-pickle loader')
            return dnnlib.util.EasyDict
        own = module == _PACKAGE or module.startswith(_PACKAGE + '.')       # (pickles written by this backend itself)
        # classes pickled by reference to a module path of the reference repository: this package mirrors the paths
        cls = self._class_defined_below(module if own else f'{_PACKAGE}.{module}', name, _PACKAGE)
        if cls is not None:
            return cls
        if module.startswith('torch.nn.modules.'):                            # plain torch.nn layers inside non-persistent networks
            cls = self._class_defined_below(module, name, 'torch.nn.modules')
            if cls is not None and issubclass(cls, torch.nn.Module):
                return cls
        # Everything else must be on the allow-list of reconstruction helpers a network pickle legitimately needs: a pickle that
        # names any other global (os.system, builtins.eval, subprocess ...) is refused instead of resolved.
        if (module, name) == ('torch.storage', '_load_from_bytes'):
            # Tensor.__reduce_ex__ of legacy-format tensors.  torch's helper is torch.load(BytesIO(b), weights_only=False), i.e. the
            # STOCK unpickler on a nested payload -- an allow-list bypass.  The nested payload is a storage: the
            # weights-only loader reads exactly that and nothing else.
            return _load_storage_from_bytes
        if (module, name) in _SAFE_GLOBALS or (module in ('torch', 'torch.storage') and name.endswith('Storage')):
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f'global {module}.{name} is not on the allow-list of the network-pickle loader')


_SAFE_GLOBALS = {
    ('collections', 'OrderedDict'), ('builtins', 'set'), ('builtins', 'frozenset'), ('builtins', 'slice'), ('builtins', 'complex'),
    ('builtins', 'bytearray'), ('builtins', 'range'),
    ('torch._utils', '_rebuild_tensor'), ('torch._utils', '_rebuild_tensor_v2'), ('torch._utils', '_rebuild_parameter'),
    ('torch._utils', '_rebuild_parameter_with_state'), ('torch._utils', '_rebuild_device_tensor_from_numpy'),
    ('torch', 'Size'), ('torch', 'device'), ('torch', 'dtype'), ('torch', 'Tensor'),
    ('torch.nn.parameter', 'Parameter'), ('torch._tensor', '_rebuild_from_type_v2'),
    ('torch', 'float32'), ('torch', 'float16'), ('torch', 'float64'), ('torch', 'bfloat16'), ('torch', 'int64'), ('torch', 'int32'),
    ('torch', 'int16'), ('torch', 'int8'), ('torch', 'uint8'), ('torch', 'bool'),
    ('numpy.core.multiarray', '_reconstruct'), ('numpy._core.multiarray', '_reconstruct'), ('numpy.core.multiarray', 'scalar'),
    ('numpy._core.multiarray', 'scalar'), ('numpy', 'ndarray'), ('numpy', 'dtype'),
    ('numpy.core.numeric', '_frombuffer'), ('numpy._core.numeric', '_frombuffer'),
}


def load_network_pkl(f, force_fp16=False):
    """dict with 'G', 'G_ema', 'D', 'training_set_kwargs', 'augment_pipe' as the reference returns it; networks this backend
    implements come back as its modules, others (discriminator) as PickledModule shells."""
    data = _Unpickler(f).load()
    if not isinstance(data, dict):
        raise pickle.UnpicklingError(f'expected a dict of networks, got {type(data).__name__}')
    for key in list(data.keys()):
        data[key] = _materialize(data[key])
    data.setdefault('training_set_kwargs', None)
    data.setdefault('augment_pipe', None)
    nets = [data.get(k) for k in ('G', 'G_ema')]
    assert any(isinstance(n, torch.nn.Module) for n in nets), 'no generator in the pickle'
    assert isinstance(data['training_set_kwargs'], (dict, type(None)))
    if force_fp16:
        for key in ('G', 'G_ema'):
            old = data.get(key)
            if not isinstance(old, torch.nn.Module) or not hasattr(old, 'init_kwargs'):
                continue
            kwargs = copy.deepcopy(old.init_kwargs)
            target = kwargs.get('synthesis_kwargs', kwargs)
            target['num_fp16_res'] = 4
            target['conv_clamp'] = 256
            if kwargs != old.init_kwargs:
                from .torch_utils import misc
                new = type(old)(**kwargs).eval().requires_grad_(False)
                misc.copy_params_and_buffers(old, new, require_all=True)
                data[key] = new
    return data
