"""CPU-side checks of the C-ABI boundary: the library builds/loads without a GPU and exports
exactly the entry points include/ia_hip.h declares.  No compute calls here."""
import ctypes

import pytest
import torch

from invertavatar_amd import _lib, build


def test_library_builds_and_exports_header_symbols():
    path = build.build()
    lib = ctypes.CDLL(path)
    declared = _lib.declared_symbols()
    assert len(declared) >= 5
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in include/ia_hip.h but not exported'
    # every typed signature refers to a declared symbol
    assert set(_lib._SIGNATURES) <= set(declared), sorted(set(_lib._SIGNATURES) - set(declared))
    assert set(declared) <= set(_lib._SIGNATURES), sorted(set(declared) - set(_lib._SIGNATURES))


def test_version_and_error_text():
    lib = _lib.load()
    assert lib.ia_version() == _lib.ABI_VERSION == 7
    # invalid argument is reported through the status code + ia_last_error, never an exception
    st = lib.ia_bias_act(None, None, None, None, None, None, 0, 16, 0, 1, 0, 3, 0.2, 1.0, -1.0, None)
    assert st == -1
    assert 'device pointers' in _lib.last_error()


def test_device_tensor_ops_never_fall_back(monkeypatch):
    """A CUDA tensor must go to the HIP kernel; a CPU tensor takes the reference's own 'ref' path."""
    from invertavatar_amd.torch_utils.ops import bias_act
    x = torch.randn(2, 3, 4, 4)
    y = bias_act.bias_act(x, torch.randn(3), act='lrelu')      # CPU tensor -> ref path, like the reference
    assert y.shape == x.shape
    with pytest.raises(RuntimeError):
        from invertavatar_amd.torch_utils.ops import _plugins
        _plugins.bias_act(x, torch.randn(3), None, None, None, 0, 1, 3, 0.2, 1.0, -1.0)  # CPU tensor at the plugin level


def _plan(b, i, o, h, w, ksize=3, transposed=0, form=0):
    lib = _lib.load()
    workers, nbytes = ctypes.c_int(-1), ctypes.c_size_t(0)
    st = lib.ia_conv2d_plan(b, i, o, h, w, ksize, transposed, form, ctypes.byref(workers), ctypes.byref(nbytes))
    assert st == 0, _lib.last_error()
    return workers.value, nbytes.value


def test_conv_planner_host_logic():
    """ia_conv2d_plan is host-only arithmetic: worker counts of the stream-K split (include/ia_hip.h).  Layers smaller than the machine
    up to 64^2 are capped at a quarter of the 256 CUs (one worker per tile if they have more tiles); layers that fill whole rounds of
    tiles have no stream-K part; the per-element share shrinks with the batch."""
    for res in (4, 8, 16):
        workers, nbytes = _plan(1, 512, 512, res, res)
        assert workers == 64 and nbytes > 0, (res, workers)
    assert _plan(1, 512, 512, 64, 64, form=2)[0] == 64
    assert _plan(1, 256, 256, 128, 128, form=2)[0] == 256           # 128^2 keeps the whole machine
    # split-DMA form: stride-1 layers smaller than the machine whose 32-channel x 256-point tiles give every CU one run whole tiles
    assert _plan(1, 512, 512, 64, 64, form=3) == (0, 0) and _plan(1, 256, 256, 128, 128, form=3) == (0, 0)
    assert _plan(1, 512, 512, 32, 32, form=3)[0] == 64              # 64 narrow tiles would leave 3/4 of the CUs idle: stream-K stays
    assert _plan(8, 512, 512, 64, 64, form=3) == (0, 0)             # (a batch of 8 already fills whole rounds of wide tiles)
    for shape in ((1, 128, 128, 512, 512), (1, 256, 256, 256, 256), (1, 128, 128, 256, 256)):
        assert _plan(*shape, form=3) == (0, 0), shape               # whole rounds of whole tiles: no workers, no scratch
    assert _plan(8, 512, 512, 16, 16)[0] <= 64                      # 512 slots shared by 8 batch elements
    lib = _lib.load()
    assert lib.ia_conv2d_plan(1, 512, 512, 16, 16, 3, 0, 7, ctypes.byref(ctypes.c_int()), ctypes.byref(ctypes.c_size_t())) == -1
    assert 'form' in _lib.last_error()


def test_streaming_torgb_shape_rules():
    from invertavatar_amd import hipops
    assert hipops.torgb_supported(512, 96, 4, 4, True) and hipops.torgb_supported(128, 3, 512, 512) and hipops.torgb_supported(1024, 8, 6, 10, True)
    assert not hipops.torgb_supported(64, 32, 8, 8) and not hipops.torgb_supported(128, 128, 8, 8) and not hipops.torgb_supported(128, 32, 7, 8, True)
    lib = _lib.load()
    for args in ((512, 96, 4, 4, 1), (128, 3, 512, 512, 0), (64, 32, 8, 8, 0), (128, 128, 8, 8, 0), (128, 32, 7, 8, 1), (128, 32, 7, 8, 0), (384, 32, 8, 8, 0)):
        assert bool(lib.ia_torgb_supported(*args)) == hipops.torgb_supported(args[0], args[1], args[2], args[3], bool(args[4])), args
    assert hipops.conv1x1_supported(512, 96, 4, 4) and hipops.conv1x1_supported(128, 3, 512, 512)
    assert not hipops.conv1x1_supported(48, 8, 8, 8)                # C_in % 32
    assert not hipops.conv1x1_supported(64, 128, 8, 8)              # C_out > 96
    assert not hipops.conv1x1_supported(64, 32, 3, 3)               # H*W % 4
    assert hipops.conv_sx_rgb_supported(1, 128, 128, 512, 512) and hipops.conv_sx_rgb_supported(1, 128, 128, 256, 256)
    assert not hipops.conv_sx_rgb_supported(1, 256, 256, 256, 256)  # two channel tiles
    assert not hipops.conv_sx_rgb_supported(1, 128, 128, 128, 128)  # stream-K layer


def test_stage_inputs_host_tensors_take_the_copy_route():
    """hipops.stage_inputs: pairs the one-launch kernel cannot take (host tensors here; strided / broadcast sources on the device)
    are copied by Tensor.copy_ -- the captured-frame wrapper works unchanged for a CPU generator."""
    import torch
    from invertavatar_amd import hipops
    src = [torch.arange(12.).reshape(3, 4), torch.arange(6.).reshape(2, 3)[:, ::2]]
    dst = [torch.zeros(3, 4), torch.zeros(2, 2)]
    hipops.stage_inputs(list(zip(src, dst)))
    assert all(torch.equal(s, d) for s, d in zip(src, dst))


def test_compose_upfir_weight_equals_transposed_convolution_then_fir():
    """hipops.compose_upfir_weight (host arithmetic of ia_upconv2d_fir_sx): conv2d with the composed weight + depth-to-space =
    conv_transpose2d(stride 2) followed by upfirdn2d(filter, padding [1,1,1,1], gain 4), in fp64, for an asymmetric filter too."""
    import torch
    from invertavatar_amd import hipops
    from invertavatar_amd.torch_utils.ops import upfirdn2d
    torch.manual_seed(0)
    x, w = torch.randn(2, 8, 9, 11, dtype=torch.float64), torch.randn(32, 8, 3, 3, dtype=torch.float64)
    for f in (upfirdn2d.setup_filter([1, 3, 3, 1]).double(), torch.randn(4, 4, dtype=torch.float64)):
        t = torch.nn.functional.conv_transpose2d(x, w.transpose(0, 1), stride=2)
        # upfirdn2d(t, f, padding 1, gain 4) = correlation of the zero-padded image with the flipped filter
        ref = torch.nn.functional.conv2d(torch.nn.functional.pad(t, [1, 1, 1, 1]).reshape(-1, 1, t.shape[2] + 2, t.shape[3] + 2),
                                         (4 * f.flip([0, 1]))[None, None]).reshape(2, 32, 18, 22)
        wc = hipops.compose_upfir_weight(w, f).double()          # rows ((py * O/32 + o // 32) * 2 + px) * 32 + o % 32
        y = torch.nn.functional.conv2d(x, wc, padding=1).view(2, 2, 1, 2, 32, 9, 11)     # [b, py, block, px, o % 32, r, c]
        y = y.permute(0, 2, 4, 5, 1, 6, 3).reshape(2, 32, 18, 22)
        assert (y - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()      # (the composed weight is returned in fp32)
    chk = upfirdn2d.upfirdn2d(torch.nn.functional.conv_transpose2d(x, w.transpose(0, 1), stride=2).float(),
                              upfirdn2d.setup_filter([1, 3, 3, 1]), padding=[1, 1, 1, 1], gain=4)
    wc = hipops.compose_upfir_weight(w, upfirdn2d.setup_filter([1, 3, 3, 1])).double()
    y = torch.nn.functional.conv2d(x, wc, padding=1).view(2, 2, 1, 2, 32, 9, 11).permute(0, 2, 4, 5, 1, 6, 3).reshape(2, 32, 18, 22)
    assert (y - chk.double()).abs().max().item() <= 1e-5 * chk.abs().max().item()       # and against the mirror's own upfirdn2d


def test_stride_2_planner_and_tiny_rule_host_logic():
    """ia_conv2d_down_plan is host-only arithmetic (r05): the stride-2 form keeps the 128-channel tile where its input windows fit the
    LDS stages beside the weights and falls to the narrow whole-tile family otherwise; outputs below 8^2, fewer than 64 output channels
    and channel counts outside octets are refused.  ia_conv3x3_s2_tiny_supported: square images of 2 / 4 / 8 pixels that fit the LDS."""
    from invertavatar_amd import hipops
    lib = _lib.load()

    def down(b, i, o, h, w):
        ks, nb = ctypes.c_int(-1), ctypes.c_size_t(0)
        st = lib.ia_conv2d_down_plan(b, i, o, h, w, ctypes.byref(ks), ctypes.byref(nb))
        return st, ks.value, nb.value
    for shape in ((1, 512, 512, 32, 32), (1, 512, 512, 64, 64), (1, 512, 512, 16, 16), (4, 256, 256, 64, 64), (1, 128, 128, 128, 128)):
        st, ks, nb = down(*shape)
        assert st == 0 and ks >= 0 and (ks == 0) == (nb == 0), (shape, st, ks, nb)
    assert down(1, 512, 512, 16, 16)[1] > 0                         # 8^2 outputs: one point tile, cut between stream-K workers
    for shape in ((4, 64, 64, 256, 256), (1, 128, 128, 256, 256)):  # 258-wide windows: narrow whole tiles, no workers, no scratch
        assert down(*shape) == (0, 0, 0), shape
    for shape in ((1, 64, 32, 64, 64), (1, 512, 512, 8, 8), (1, 60, 64, 64, 64), (1, 64, 60, 64, 64)):
        assert down(*shape)[0] != 0, shape
        assert not hipops.conv_down_supported(*shape)
    assert hipops.conv_down_supported(2, 64, 128, 33, 47)            # odd sizes: outputs (H - 1) // 2 + 1
    assert hipops.conv_tiny_supported(512, 512, 8, 8) and hipops.conv_tiny_supported(512, 512, 2, 2) and hipops.conv_tiny_supported(8, 5, 4, 4)
    assert not hipops.conv_tiny_supported(512, 512, 16, 16) and not hipops.conv_tiny_supported(512, 512, 8, 4) and not hipops.conv_tiny_supported(1024, 8, 8, 8)
    assert lib.ia_conv2d_down_plan(1, 512, 512, 32, 32, None, None) == -1 and 'null' in _lib.last_error()


def test_encoder_conv2d_keeps_torch_semantics_off_the_device_path():
    """layers.Conv2d is torch.nn.Conv2d for CPU tensors and under autograd (same parameters, same state-dict keys, same results)."""
    from invertavatar_amd.encoder_inversion.models import layers, trunk_hip
    torch.manual_seed(0)
    mine, ref = layers.Conv2d(6, 10, 3, 2, 1), torch.nn.Conv2d(6, 10, 3, 2, 1)
    ref.load_state_dict(mine.state_dict())
    assert list(mine.state_dict()) == list(ref.state_dict()) == ['weight', 'bias']
    x = torch.randn(2, 6, 9, 8, requires_grad=True)
    assert trunk_hip._conv_route(mine, x) is None and not trunk_hip.conv_covered(mine, x.detach())
    y = mine(x)
    assert torch.equal(y, ref(x)) and y.requires_grad
