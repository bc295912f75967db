"""legacy.load_network_pkl: pickles in the REFERENCE's persistence format load as this backend's modules.

The fixture pickle is written by this test in the reference's byte format -- every persistent object reduced to
torch_utils.persistence._reconstruct_persistent_obj(meta) with an (empty) module_src -- from a generator of this backend, so no
reference source travels; when /root/reference is present (build container) a pickle made by the reference's own classes is
loaded as well."""
import io
import os
import pickle
import sys
import types

import pytest
import torch

from invertavatar_amd import legacy, synthetic
from invertavatar_amd.training_avatar_texture.triplane_v20 import TriPlaneGenerator


def _reference_format_pickle(obj_dict):
    """pickle.dumps(obj_dict) where every persistent module is written the way torch_utils/persistence.py:112-123 writes it."""
    fake = types.ModuleType('torch_utils.persistence')

    def _reconstruct_persistent_obj(meta):      # only its qualified name is pickled
        raise RuntimeError('not called when dumping')
    _reconstruct_persistent_obj.__module__ = 'torch_utils.persistence'
    _reconstruct_persistent_obj.__qualname__ = '_reconstruct_persistent_obj'
    fake._reconstruct_persistent_obj = _reconstruct_persistent_obj
    saved = {k: sys.modules.get(k) for k in ('torch_utils', 'torch_utils.persistence')}
    sys.modules['torch_utils'] = types.ModuleType('torch_utils')
    sys.modules['torch_utils.persistence'] = fake

    class Pickler(pickle.Pickler):
        def reducer_override(self, obj):
            if isinstance(obj, torch.nn.Module) and getattr(type(obj), '_persistent', False):
                state = {k: v for k, v in obj.__dict__.items() if not (k.startswith('_') and k[1:] in ('packed', 'pre', 'scaled', 'style_batcher'))}
                meta = dict(type='class', version=6, module_src='', class_name=type(obj).__name__, state=state)
                return _reconstruct_persistent_obj, (meta,)
            return NotImplemented
    try:
        buf = io.BytesIO()
        Pickler(buf).dump(obj_dict)
        return buf.getvalue()
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def _render(g):
    frames, nrr = [3], 32
    with torch.no_grad():
        ws = g.mapping(synthetic.latent(0, 1), synthetic.conditioning_camera(), truncation_psi=0.7, truncation_cutoff=14)
        return g.synthesis(ws, synthetic.camera_labels(frames), {'uvcoords_image': synthetic.uv_conditions(frames)}, neural_rendering_resolution=nrr,
                           noise_mode='const', evaluation=True, jitter=synthetic.jitter(frames, nrr * nrr))['image']


def test_reference_format_pickle_loads_as_backend_modules():
    g = TriPlaneGenerator(**synthetic.generator_kwargs('small')).eval().requires_grad_(False)
    synthetic.fill_parameters(g)
    g.neural_rendering_resolution = 64
    blob = _reference_format_pickle(dict(G=g, G_ema=g, D=None, training_set_kwargs=dict(path='x')))
    assert b'_reconstruct_persistent_obj' in blob
    data = legacy.load_network_pkl(io.BytesIO(blob))
    assert set(data) >= {'G', 'G_ema', 'D', 'training_set_kwargs', 'augment_pipe'} and data['augment_pipe'] is None
    new = data['G_ema']
    assert type(new) is TriPlaneGenerator and not new.training and new.neural_rendering_resolution == 64
    assert new.init_kwargs['rendering_kwargs']['depth_resolution'] == 48
    for (n0, t0), (n1, t1) in zip(g.state_dict().items(), new.state_dict().items()):
        assert n0 == n1 and torch.equal(t0, t1)
    assert torch.equal(_render(new), _render(g))
    # the scripts' "reload modules" recipe (reenact_avatar_next3d.py:158-161) works on the loaded object
    from invertavatar_amd.torch_utils import misc
    again = TriPlaneGenerator(*new.init_args, **new.init_kwargs).eval().requires_grad_(False)
    misc.copy_params_and_buffers(new, again, require_all=True)


def test_unknown_persistent_classes_stay_shells_and_tf_pickles_are_refused():
    class Fake(torch.nn.Module):
        _persistent = True
    Fake.__name__ = 'DualDiscriminator'
    d = Fake()
    d.register_buffer('x', torch.ones(3))
    g = TriPlaneGenerator(**synthetic.generator_kwargs('small'))
    data = legacy.load_network_pkl(io.BytesIO(_reference_format_pickle(dict(G=g, D=d))))
    assert isinstance(data['D'], legacy.PickledModule) and data['D'].class_name == 'DualDiscriminator'
    assert torch.equal(legacy.collect_state_dict(data['D'])['x'], torch.ones(3))


@pytest.mark.skipif(not os.path.isdir('/root/reference/training_avatar_texture'), reason='needs the reference checkout (build container only)')
def test_pickle_written_by_the_reference_itself():
    """The real thing: the reference's TriPlaneGenerator pickled by the reference's persistence (module source embedded),
    loaded here without executing that source."""
    import subprocess
    code = r'''
import sys, types, pickle, io
sys.path.insert(0, '/root/reference'); sys.path.insert(0, %r)
import make_golden as mg
mg.install_stubs()
import torch
from invertavatar_amd import synthetic
from training_avatar_texture.triplane_v20 import TriPlaneGenerator
g = TriPlaneGenerator(**synthetic.generator_kwargs('small')).eval().requires_grad_(False)
synthetic.fill_parameters(g)
g.neural_rendering_resolution = 64
sys.stdout.buffer.write(pickle.dumps(dict(G=g, G_ema=g, D=None)))
''' % os.path.join(os.path.dirname(__file__), 'golden')
    blob = subprocess.run([sys.executable, '-c', code], capture_output=True, check=True).stdout
    assert b'module_src' in blob
    data = legacy.load_network_pkl(io.BytesIO(blob))
    new = data['G_ema']
    assert type(new) is TriPlaneGenerator and new.neural_rendering_resolution == 64
    ref = TriPlaneGenerator(**synthetic.generator_kwargs('small')).eval().requires_grad_(False)
    synthetic.fill_parameters(ref)
    assert all(torch.equal(a, b) for a, b in zip(ref.state_dict().values(), new.state_dict().values()))


def test_pickle_naming_a_foreign_global_is_refused():
    """the unpickler resolves only allow-listed globals -- a crafted pickle cannot reach os.system & co."""
    import io
    import os
    import pickle
    import pytest
    from invertavatar_amd import legacy

    class Evil:
        def __reduce__(self):
            return (os.system, ('true',))
    blob = pickle.dumps({'G_ema': Evil()})
    with pytest.raises(pickle.UnpicklingError, match='allow-list'):
        legacy.load_network_pkl(io.BytesIO(blob))


def _stack_global_pickle(module, name, arg):
    """Protocol-4 pickle of `module:name(arg)`; `name` may be dotted (STACK_GLOBAL lets find_class walk attributes)."""
    import pickle
    import struct

    def short_unicode(s):
        b = s.encode()
        return pickle.SHORT_BINUNICODE + struct.pack('<B', len(b)) + b
    return (pickle.PROTO + b'\x04' + short_unicode(module) + short_unicode(name) + pickle.STACK_GLOBAL
            + short_unicode(arg) + pickle.TUPLE1 + pickle.REDUCE + pickle.STOP)


def test_pickle_walking_dotted_names_through_an_allowed_module_is_refused(tmp_path):
    """('invertavatar_amd.hipops', '_os.system') and ('torch.nn.modules.module', 'torch.hub.os.getenv') used to resolve
    (find_class walks dotted names through whatever the module imported); so did a plain attribute that is not a class."""
    import io
    import pickle
    import pytest
    from invertavatar_amd import legacy

    marker = tmp_path / 'pwned'
    for module, name in (('invertavatar_amd.hipops', 'os.system'), ('invertavatar_amd.hipops', '_os.system'),
                         ('torch.nn.modules.module', 'torch.hub.os.system'), ('training.networks_stylegan2', 'np.os.system'),
                         ('invertavatar_amd.legacy', 'importlib'), ('invertavatar_amd.legacy', 'pickle'),
                         ('torch.nn.modules.module', 'warnings'), ('dnnlib.util', 'importlib'), ('dnnlib', 'call_func_by_name')):
        blob = _stack_global_pickle(module, name, f'touch {marker}')
        with pytest.raises(pickle.UnpicklingError):
            legacy._Unpickler(io.BytesIO(blob)).load()
        assert not marker.exists()
    # the classes a network pickle legitimately names still resolve, by reference path and by this package's path
    from invertavatar_amd.training.networks_stylegan2 import MappingNetwork
    u = legacy._Unpickler(io.BytesIO(b''))
    assert u.find_class('training.networks_stylegan2', 'MappingNetwork') is MappingNetwork
    assert u.find_class('invertavatar_amd.training.networks_stylegan2', 'MappingNetwork') is MappingNetwork
    import torch
    assert u.find_class('torch.nn.modules.conv', 'Conv2d') is torch.nn.Conv2d
    assert u.find_class('dnnlib.util', 'EasyDict') is legacy.dnnlib.util.EasyDict


def test_nested_payload_through_load_from_bytes_is_refused(tmp_path):
    """torch.storage._load_from_bytes is torch.load(..., weights_only=False) -- the stock unpickler on a nested payload.
    A pickle that hands it a nested pickle of os.system(...) must not run it; a pickle that hands it a real storage must still load."""
    import io
    import os
    import pickle
    import pytest
    import torch
    from invertavatar_amd import legacy

    marker = tmp_path / 'pwned'

    class Evil:
        def __reduce__(self):
            return (os.system, (f'touch {marker}',))

    class Carrier:
        def __init__(self, payload):
            self.payload = payload

        def __reduce__(self):
            return (torch.storage._load_from_bytes, (self.payload,))
    for nested in (pickle.dumps(Evil(), protocol=4), pickle.dumps(Evil(), protocol=2)):
        blob = pickle.dumps({'G_ema': Carrier(nested)}, protocol=4)
        with pytest.raises(Exception):
            legacy._Unpickler(io.BytesIO(blob)).load()
        assert not marker.exists()
    # a nested torch.save in the legacy format whose pickle stream names a foreign global
    buf = io.BytesIO()
    torch.save({'x': Evil()}, buf, _use_new_zipfile_serialization=False)
    blob = pickle.dumps({'G_ema': Carrier(buf.getvalue())}, protocol=4)
    with pytest.raises(Exception):
        legacy._Unpickler(io.BytesIO(blob)).load()
    assert not marker.exists()
    # the legitimate use: tensors pickled with the plain pickle module reduce to _load_from_bytes(storage bytes)
    t = torch.arange(12, dtype=torch.float32).reshape(3, 4)
    back = legacy._Unpickler(io.BytesIO(pickle.dumps({'t': t, 'h': t.half()[1:]}))).load()
    assert torch.equal(back['t'], t) and torch.equal(back['h'], t.half()[1:])
