"""fma(a, b, c) = a * b + c (reference: torch_utils/ops/fma.py:17-28; used by the non-fused
modulated conv when the generator is left in train() mode)."""
import torch


def fma(a, b, c):
    return _FMA.apply(a, b, c)


def _unbroadcast(x, shape):
    extra = x.ndim - len(shape)
    assert extra >= 0
    dims = [i for i in range(x.ndim) if x.shape[i] > 1 and (i < extra or shape[i - extra] == 1)]
    if dims:
        x = x.sum(dim=dims, keepdim=True)
    if extra:
        x = x.reshape(-1, *x.shape[extra + 1:])
    assert x.shape == shape
    return x


class _FMA(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, c):
        ctx.save_for_backward(a, b)
        ctx.c_shape = c.shape
        return torch.addcmul(c, a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        da = _unbroadcast(g * b, a.shape) if ctx.needs_input_grad[0] else None
        db = _unbroadcast(g * a, b.shape) if ctx.needs_input_grad[1] else None
        dc = _unbroadcast(g, ctx.c_shape) if ctx.needs_input_grad[2] else None
        return da, db, dc
