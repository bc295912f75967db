"""Fused bias + activation (reference API: torch_utils/ops/bias_act.py:54-88).

Device tensors run the HIP kernel ``ia_bias_act``; CPU tensors (and impl='ref') run the
plain-torch definition, exactly as the reference dispatches (bias_act.py:86-88).  A device
tensor never falls back to torch ops: if libia_hip.so is missing the call raises."""
import dataclasses
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
        _, a, g, c = _resolve(act, alpha, gain, clamp)
        return _BiasAct.apply(x, b, _ActConfig(act, dim, a, g, c))
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


@dataclasses.dataclass(frozen=True)
class _ActConfig:
    """Everything but the tensors: resolved once per call, handed to the autograd Functions as one non-tensor argument."""
    name: str
    dim: int
    alpha: float
    gain: float
    clamp: float

    @property
    def spec(self):
        return activation_funcs[self.name]

    @property
    def is_identity(self):            # y = x: nothing to launch when there is no bias either
        return self.name == 'linear' and self.gain == 1 and self.clamp < 0

    def kernel(self, order, t, b, x, y, dy):
        """ia_bias_act: order 0 = forward on t; 1 = d/dx applied to the incoming gradient t; 2 = second-order term (needs dy)."""
        return _plugin.bias_act(t, b, x, y, dy, order, self.dim, self.spec.cuda_idx, self.alpha, self.gain, self.clamp)

    def bias_grad(self, g):
        return g.sum([i for i in range(g.ndim) if i != self.dim])


def _layout_of(t):
    return torch.channels_last if t.ndim > 2 and t.stride(1) == 1 else torch.contiguous_format


class _BiasAct(torch.autograd.Function):
    """y = act(x + b) * gain, clamped.  Saves what the activation's derivative is written in (`ref`: 'x', 'y' or nothing)."""

    @staticmethod
    def forward(ctx, x, b, cfg):
        ctx.cfg, ctx.layout = cfg, _layout_of(x)
        x = x.contiguous(memory_format=ctx.layout)
        b = _null if b is None else b.contiguous()
        y = x if (cfg.is_identity and b is _null) else cfg.kernel(0, x, b, _null, _null, _null)
        needs_x = 'x' in cfg.spec.ref or cfg.spec.has_2nd_grad
        ctx.save_for_backward(x if needs_x else _null, b if needs_x else _null, y if 'y' in cfg.spec.ref else _null)
        return y

    @staticmethod
    def backward(ctx, grad_y):
        cfg = ctx.cfg
        want_x, want_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if not (want_x or want_b):
            return None, None, None
        x, b, y = ctx.saved_tensors
        grad_y = grad_y.contiguous(memory_format=ctx.layout)
        grad_x = grad_y if cfg.is_identity else _BiasActGrad.apply(grad_y, x, b, y, cfg)
        return grad_x, (cfg.bias_grad(grad_x) if want_b else None), None


class _BiasActGrad(torch.autograd.Function):
    """g -> g * act'(x + b) * gain (zero where the clamp is active): linear in g, so its own gradient w.r.t. g is itself; the
    derivative w.r.t. x / b exists for the activations with a second derivative (order-2 kernel)."""

    @staticmethod
    def forward(ctx, g, x, b, y, cfg):
        ctx.cfg, ctx.layout = cfg, _layout_of(g)
        out = cfg.kernel(1, g, b, x, y, _null)
        ctx.save_for_backward(g if cfg.spec.has_2nd_grad else _null, x, b, y)
        return out

    @staticmethod
    def backward(ctx, gg):
        cfg = ctx.cfg
        g, x, b, y = ctx.saved_tensors
        gg = gg.contiguous(memory_format=ctx.layout)
        d_g = _BiasActGrad.apply(gg, x, b, y, cfg) if ctx.needs_input_grad[0] else None
        d_x = d_b = None
        if cfg.spec.has_2nd_grad and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
            d_x = cfg.kernel(2, gg, b, x, y, g)
            if ctx.needs_input_grad[2]:
                d_b = cfg.bias_grad(d_x)
        return d_g, d_x, d_b, None, None
