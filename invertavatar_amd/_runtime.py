"""Per-module runtime state kept OUTSIDE the module.

Streams, events, captured graphs and tensors that only live for one frame must not sit in ``nn.Module.__dict__``: they would
be walked by ``copy.deepcopy`` / ``pickle`` / ``torch.save`` (the calls the reference's scripts make on generators,
reenact_avatar_next3d.py:158, legacy.load_network_pkl) and HIP streams can be neither copied nor pickled.  ``state(module)``
returns a plain namespace owned by a weak map: it disappears with the module and is never serialised."""
import types
import weakref

_STATE = weakref.WeakKeyDictionary()


def state(module):
    st = _STATE.get(module)
    if st is None:
        st = _STATE[module] = types.SimpleNamespace()
    return st


class DeviceCache:
    """Base of the per-layer caches of kernel-side tensors (packed weights, style tables): a copy or an unpickled instance
    starts empty and refills itself on first use."""

    def __reduce__(self):
        return (type(self), ())
