"""The reference's top-level module names, served by this package.

The reference's scripts import its packages by their short names (`reenact_avatar_next3d.py:5-19`, `eval_seq.py:7-16`,
`eval_updated_os.py`): `import legacy`, `import dnnlib`, `from torch_utils import misc`,
`from training_avatar_texture.triplane_v20 import TriPlaneGenerator`, `from encoder_inversion.models.uvnet import inversionNet`,
`from data_preprocess.FaceVerse.renderer import Faceverse_manager` ...  `install_aliases()` makes those statements resolve
to the modules of `invertavatar_amd` — the SAME module objects, so `training_avatar_texture.triplane_v20.TriPlaneGenerator is
invertavatar_amd.training_avatar_texture.triplane_v20.TriPlaneGenerator`, `isinstance` works across both spellings and the
mirror's own relative imports (`from .. import hipops`) keep resolving inside `invertavatar_amd`.

Aliasing only the top-level packages in `sys.modules` is NOT enough: `from training_avatar_texture.triplane_v20
import X` makes the import system load the sub-module again under the short name, with `__package__ =
'training_avatar_texture'`, and its `from .. import hipops` then fails with "attempted relative import beyond top-level
package".  A meta-path finder placed in front of the path finders answers every `name` / `name.sub.module` below an aliased
top-level name with the already imported `invertavatar_amd.name.sub.module` instead.
"""
import importlib
import importlib.abc
import importlib.machinery
import importlib.util
import pkgutil
import sys

_PACKAGE = __name__.rsplit('.', 1)[0]      # 'invertavatar_amd'

# Top-level names of the reference repository that this package mirrors.
TOP_LEVEL = ('torch_utils', 'dnnlib', 'training', 'training_avatar_texture', 'encoder_inversion', 'data_preprocess',
             'camera_utils', 'legacy')


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, target):
        self.target = target

    def create_module(self, spec):
        module = importlib.import_module(self.target)     # the mirror's module object itself, not a copy
        self.real_spec = module.__spec__
        return module

    def exec_module(self, module):
        # Already executed under its real name.  The import system has just stamped the alias spec on it; put the real one
        # back so that __spec__.parent keeps agreeing with __package__ (lazy relative imports, importlib.reload).
        module.__spec__ = self.real_spec


class AliasFinder(importlib.abc.MetaPathFinder):
    """`torch_utils.ops.bias_act` -> the module object of `invertavatar_amd.torch_utils.ops.bias_act`."""

    def __init__(self, names=TOP_LEVEL):
        self.names = frozenset(names)

    def find_spec(self, fullname, path=None, target=None):
        if fullname.split('.', 1)[0] not in self.names:
            return None
        real = f'{_PACKAGE}.{fullname}'
        try:
            real_spec = importlib.util.find_spec(real)
        except (ImportError, ValueError):
            real_spec = None
        if real_spec is None:
            return None                                   # not mirrored: let the normal finders (or ImportError) answer
        return importlib.machinery.ModuleSpec(fullname, _AliasLoader(real), origin=real_spec.origin,
                                              is_package=real_spec.submodule_search_locations is not None)


def _installed():
    return next((f for f in sys.meta_path if isinstance(f, AliasFinder)), None)


def install_aliases(eager=False, names=TOP_LEVEL):
    """Serve the reference's module names from this package.  Idempotent.

    eager=True also imports every sub-module of the mirrored packages now (what a `pkgutil.walk_packages` + `sys.modules`
    loop would do), so that a missing optional dependency of any mirrored module shows up here rather than at first use.
    A reference checkout earlier on `sys.path` does not shadow the aliases: the finder sits in front of the path finders;
    modules of the reference that were imported BEFORE this call stay what they are (and are reported).
    """
    clash = sorted(n for n in sys.modules
                   if n.split('.', 1)[0] in names and not getattr(sys.modules[n], '__name__', '').startswith(_PACKAGE + '.'))
    if clash:
        raise ImportError(f'install_aliases() must run before the reference\'s own modules are imported; already loaded: {clash[:5]}')
    if _installed() is None:
        sys.meta_path.insert(0, AliasFinder(names))
    if eager:
        for top in names:
            mod = importlib.import_module(top)
            for info in pkgutil.walk_packages(getattr(mod, '__path__', []), prefix=f'{_PACKAGE}.{top}.'):
                importlib.import_module(info.name[len(_PACKAGE) + 1:])
    return [n for n in names]


def remove_aliases():
    """Undo install_aliases(): drop the finder and every short-name entry it created (tests)."""
    finder = _installed()
    if finder is None:
        return
    sys.meta_path.remove(finder)
    for n in [n for n in sys.modules if n.split('.', 1)[0] in finder.names]:
        if getattr(sys.modules[n], '__name__', '').startswith(_PACKAGE + '.'):
            del sys.modules[n]
