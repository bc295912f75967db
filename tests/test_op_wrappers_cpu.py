"""Host logic of the op wrappers (torch_utils/ops/{bias_act,upfirdn2d}.py) without a GPU: the autograd Functions that sit between
the public entry points and the plugin-shaped HIP entry are driven with a stand-in `_plugin` that evaluates the SAME contract
(ia_bias_act orders 0 / 1 / 2, ia_upfirdn2d) in plain torch, and their first / second derivatives are compared with autograd through
the plain-torch definitions.  (The kernels themselves are compared with the reference's vectors in tests/test_ops_gpu.py.)"""
import types

import pytest
import torch

from invertavatar_amd.torch_utils.ops import bias_act as BA
from invertavatar_amd.torch_utils.ops import upfirdn2d as UF
from conftest import rnd


def _fake_bias_act_plugin():
    def bias_act(t, b, x, y, dy, order, dim, act_idx, alpha, gain, clamp):
        name = next(k for k, v in BA.activation_funcs.items() if v.cuda_idx == act_idx)
        spec = BA.activation_funcs[name]
        with torch.enable_grad():
            if order == 0:
                return BA._bias_act_ref(t, b if b.numel() else None, dim=dim, act=name, alpha=alpha, gain=gain, clamp=clamp if clamp >= 0 else None).detach()
            if name == 'linear':     # nothing is saved for it (ref = ''): slope = gain; the clamp mask needs y and is therefore not applied (bias_act.cu)
                return (t * gain).detach() if order == 1 else torch.zeros_like(t)
            # reconstruct the pre-activation from whatever was saved (x + b, or invert through y for the 'y' activations by re-running from x)
            assert x.numel() or y.numel() or name == 'linear'
            if x.numel():
                pre = x.detach()
                if b.numel():
                    shape = [1] * pre.ndim
                    shape[dim] = -1
                    pre = pre + b.reshape(shape)
            else:       # only y saved (relu / lrelu): the derivative is a function of sign(y)
                pre = y.detach() / gain
                if name == 'lrelu':
                    pre = torch.where(pre < 0, pre / alpha, pre)
            pre = pre.double().requires_grad_(True)
            out = spec.func(pre, alpha=alpha) * gain
            if clamp >= 0:
                out = out.clamp(-clamp, clamp)
            g1, = torch.autograd.grad(out.sum(), pre, create_graph=True)
            if clamp >= 0 and y.numel():       # (the pre-activation rebuilt from a clamped y sits ON the bound: mask from y, as the kernel does)
                g1 = g1 * (y.abs() < clamp)
            if order == 1:
                return (t.double() * g1).to(t.dtype).detach()
            g2, = torch.autograd.grad(g1.sum(), pre, allow_unused=True)
            g2 = torch.zeros_like(pre) if g2 is None else g2
            return (t.double() * dy.double() * g2).to(t.dtype).detach()
    return types.SimpleNamespace(bias_act=bias_act)


@pytest.mark.parametrize('act', ['linear', 'lrelu', 'relu', 'tanh', 'sigmoid', 'swish', 'softplus'])
@pytest.mark.parametrize('clamp', [None, 0.8])
def test_bias_act_function_derivatives(act, clamp, monkeypatch):
    if act == 'linear' and clamp is not None:
        pytest.skip('linear saves neither x nor y: its gradient kernel cannot see the clamp (same in the reference plugin)')
    monkeypatch.setattr(BA, '_plugin', _fake_bias_act_plugin())
    x0, b0 = rnd(1, 2, 5, 4, 3).double() * 1.5, rnd(2, 5).double()
    _, a, g, c = BA._resolve(act, None, 0.9, clamp)
    cfg = BA._ActConfig(act, 1, a, g, c)
    x, b = x0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
    xr, br = x0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
    y = BA._BiasAct.apply(x, b, cfg)
    yr = BA._bias_act_ref(xr, br, dim=1, act=act, gain=0.9, clamp=clamp)
    assert torch.allclose(y, yr, atol=1e-12)
    w = rnd(3, *y.shape).double()
    gx, gb = torch.autograd.grad((y * w).sum(), (x, b), create_graph=True)
    gxr, gbr = torch.autograd.grad((yr * w).sum(), (xr, br), create_graph=True)
    assert torch.allclose(gx, gxr, atol=1e-10) and torch.allclose(gb, gbr, atol=1e-10)
    if BA.activation_funcs[act].has_2nd_grad:
        v = rnd(4, *gx.shape).double()
        hx, hb = torch.autograd.grad((gx * v).sum(), (x, b))
        hxr, hbr = torch.autograd.grad((gxr * v).sum(), (xr, br))
        assert torch.allclose(hx, hxr, atol=1e-9) and torch.allclose(hb, hbr, atol=1e-9)


def test_bias_act_identity_launches_nothing(monkeypatch):
    def boom(*a, **k):
        raise AssertionError('no kernel launch expected for y = x')
    monkeypatch.setattr(BA, '_plugin', types.SimpleNamespace(bias_act=boom))
    x = rnd(5, 2, 3, 4, 4).requires_grad_(True)
    y = BA._BiasAct.apply(x, None, BA._ActConfig('linear', 1, 0.0, 1.0, -1.0))
    g, = torch.autograd.grad(y.sum(), x)
    assert torch.equal(y, x) and torch.equal(g, torch.ones_like(x))


def _fake_upfirdn2d_plugin():
    def upfirdn2d(x, f, upx, upy, dnx, dny, px0, px1, py0, py1, flip, gain):
        return UF._upfirdn2d_ref(x, f, up=[upx, upy], down=[dnx, dny], padding=[px0, px1, py0, py1], flip_filter=flip, gain=gain).detach()
    return types.SimpleNamespace(upfirdn2d=upfirdn2d)


@pytest.mark.parametrize('cfg', [dict(up=2, padding=[2, 1, 2, 1], gain=4), dict(down=2, padding=[1, 1, 1, 1]), dict(up=[2, 3], down=[3, 2], padding=[2, -1, 0, 3], flip_filter=True, gain=1.5),
                                 dict(padding=[1, 1, 1, 1], gain=4)])
@pytest.mark.parametrize('separable', [False, True])
def test_upfirdn2d_function_gradients_are_the_adjoint(cfg, separable, monkeypatch):
    monkeypatch.setattr(UF, '_plugin', _fake_upfirdn2d_plugin())
    f = torch.tensor([1., 3., 3., 1.]) / 8 if separable else rnd(6, 3, 4).abs()
    x0 = rnd(7, 2, 3, 9, 8)
    x, xr = x0.clone().requires_grad_(True), x0.clone().requires_grad_(True)
    plan = UF._FirPlan.parse(cfg.get('up', 1), cfg.get('down', 1), cfg.get('padding', 0), cfg.get('flip_filter', False), cfg.get('gain', 1))
    y = UF._UpFirDn.apply(x, f, plan)
    yr = UF._upfirdn2d_ref(xr, f, **cfg)
    assert y.shape == yr.shape and torch.allclose(y, yr, atol=1e-6)
    w = rnd(8, *y.shape)
    g, = torch.autograd.grad((y * w).sum(), x, create_graph=True)
    gr, = torch.autograd.grad((yr * w).sum(), xr, create_graph=True)
    assert g.shape == x.shape and torch.allclose(g, gr, atol=1e-5)
    # second order: the gradient is itself an upfirdn2d of w, differentiable w.r.t. w through the same Function
    w2 = w.clone().requires_grad_(True)
    g2, = torch.autograd.grad((UF._UpFirDn.apply(x0.clone().requires_grad_(True), f, plan) * w2).sum(), w2)
    assert torch.allclose(g2, y.detach(), atol=1e-6)
