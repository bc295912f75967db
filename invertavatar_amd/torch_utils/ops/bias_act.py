"""Fused bias + activation (reference API: torch_utils/ops/bias_act.py:54-88).

Device tensors run the HIP kernel ``ia_bias_act``; CPU tensors (and impl='ref') run the
plain-torch definition, exactly as the reference dispatches (bias_act.py:86-88).  A device
tensor never falls back to torch ops: if libia_hip.so is missing the call raises."""
import math

import torch

from .. import custom_ops, misc
from ... import dnnlib

_E = dnnlib.EasyDict
_SQRT2 = math.sqrt(2)

# Same table keys as the reference (bias_act.py:23-33): cuda_idx doubles as the ia_act id.
activation_funcs = {
    'linear':   _E(func=lambda x, **_: x,                                        def_alpha=0,   def_gain=1,      cuda_idx=1, ref='',  has_2nd_grad=False),
    'relu':     _E(func=lambda x, **_: torch.relu(x),                            def_alpha=0,   def_gain=_SQRT2, cuda_idx=2, ref='y', has_2nd_grad=False),
    'lrelu':    _E(func=lambda x, alpha, **_: torch.nn.functional.leaky_relu(x, alpha), def_alpha=0.2, def_gain=_SQRT2, cuda_idx=3, ref='y', has_2nd_grad=False),
    'tanh':     _E(func=lambda x, **_: torch.tanh(x),                            def_alpha=0,   def_gain=1,      cuda_idx=4, ref='y', has_2nd_grad=True),
    'sigmoid':  _E(func=lambda x, **_: torch.sigmoid(x),                         def_alpha=0,   def_gain=1,      cuda_idx=5, ref='y', has_2nd_grad=True),
    'elu':      _E(func=lambda x, **_: torch.nn.functional.elu(x),               def_alpha=0,   def_gain=1,      cuda_idx=6, ref='y', has_2nd_grad=True),
    'selu':     _E(func=lambda x, **_: torch.nn.functional.selu(x),              def_alpha=0,   def_gain=1,      cuda_idx=7, ref='y', has_2nd_grad=True),
    'softplus': _E(func=lambda x, **_: torch.nn.functional.softplus(x),          def_alpha=0,   def_gain=1,      cuda_idx=8, ref='y', has_2nd_grad=True),
    'swish':    _E(func=lambda x, **_: torch.sigmoid(x) * x,                     def_alpha=0,   def_gain=_SQRT2, cuda_idx=9, ref='x', has_2nd_grad=True),
}

_plugin = None
_null = torch.empty([0])


def _init():
    global _plugin
    if _plugin is None:
        _plugin = custom_ops.get_plugin(module_name='bias_act_plugin', sources=['bias_act.hip'], headers=['ia_hip.h'])
    return True


def _resolve(act, alpha, gain, clamp):
    assert clamp is None or clamp >= 0
    spec = activation_funcs[act]
    return (spec,
            float(spec.def_alpha if alpha is None else alpha),
            float(spec.def_gain if gain is None else gain),
            float(-1 if clamp is None else clamp))


def bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None, impl='cuda'):
    assert isinstance(x, torch.Tensor)
    assert impl in ['ref', 'cuda']
    if impl == 'cuda' and x.device.type == 'cuda' and _init():
        return _bias_act_cuda(dim=dim, act=act, alpha=alpha, gain=gain, clamp=clamp).apply(x, b)
    return _bias_act_ref(x=x, b=b, dim=dim, act=act, alpha=alpha, gain=gain, clamp=clamp)


@misc.profiled_function
def _bias_act_ref(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None):
    spec, alpha, gain, clamp = _resolve(act, alpha, gain, clamp)
    if b is not None:
        assert isinstance(b, torch.Tensor) and b.ndim == 1
        assert 0 <= dim < x.ndim and b.shape[0] == x.shape[dim]
        view = [1] * x.ndim
        view[dim] = -1
        x = x + b.reshape(view)
    x = spec.func(x, alpha=alpha)
    if gain != 1:
        x = x * gain
    if clamp >= 0:
        x = x.clamp(-clamp, clamp)
    return x


_cache = {}


def _bias_act_cuda(dim=1, act='linear', alpha=None, gain=None, clamp=None):
    """autograd.Function bound to one (dim, act, alpha, gain, clamp); cached like the reference (:128-209)."""
    spec, alpha, gain, clamp = _resolve(act, alpha, gain, clamp)
    key = (dim, act, alpha, gain, clamp)
    if key in _cache:
        return _cache[key]
    keep_x = 'x' in spec.ref or spec.has_2nd_grad
    trivial = act == 'linear' and gain == 1 and clamp < 0

    def layout(t):
        return torch.channels_last if t.ndim > 2 and t.stride(1) == 1 else torch.contiguous_format

    class BiasActCuda(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, b):
            ctx.memory_format = layout(x)
            x = x.contiguous(memory_format=ctx.memory_format)
            b = b.contiguous() if b is not None else _null
            y = x
            if not trivial or b is not _null:
                y = _plugin.bias_act(x, b, _null, _null, _null, 0, dim, spec.cuda_idx, alpha, gain, clamp)
            ctx.save_for_backward(x if keep_x else _null, b if keep_x else _null, y if 'y' in spec.ref else _null)
            return y

        @staticmethod
        def backward(ctx, dy):
            dy = dy.contiguous(memory_format=ctx.memory_format)
            x, b, y = ctx.saved_tensors
            dx = db = None
            if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
                dx = dy if trivial else BiasActCudaGrad.apply(dy, x, b, y)
            if ctx.needs_input_grad[1]:
                db = dx.sum([i for i in range(dx.ndim) if i != dim])
            return dx, db

    class BiasActCudaGrad(torch.autograd.Function):
        @staticmethod
        def forward(ctx, dy, x, b, y):
            ctx.memory_format = layout(dy)
            dx = _plugin.bias_act(dy, b, x, y, _null, 1, dim, spec.cuda_idx, alpha, gain, clamp)
            ctx.save_for_backward(dy if spec.has_2nd_grad else _null, x, b, y)
            return dx

        @staticmethod
        def backward(ctx, d_dx):
            d_dx = d_dx.contiguous(memory_format=ctx.memory_format)
            dy, x, b, y = ctx.saved_tensors
            d_dy = d_x = d_b = None
            if ctx.needs_input_grad[0]:
                d_dy = BiasActCudaGrad.apply(d_dx, x, b, y)
            if spec.has_2nd_grad and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
                d_x = _plugin.bias_act(d_dx, b, x, y, dy, 2, dim, spec.cuda_idx, alpha, gain, clamp)
            if spec.has_2nd_grad and ctx.needs_input_grad[2]:
                d_b = d_x.sum([i for i in range(d_x.ndim) if i != dim])
            return d_dy, d_x, d_b, None

    _cache[key] = BiasActCuda
    return BiasActCuda
