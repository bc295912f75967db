"""Few-shot inversion flow of eval_seq.py (encode -> 2 x AR_eval_forward with carried ConvGRU state -> drive frame)
on the GPU backend against the reference fixture (BASELINE configs 3/5 pattern, reduced to 8 source frames + 1 drive frame)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL_FEATURES = 5e-5      # relative to max(1, max |ref|): features / GRU states after two train-mode-BatchNorm UNet passes (measured r03: <= 3.4e-6)
TOL_DRIVE_RGB = 1e-4     # BASELINE asks 1e-3 on rendered RGB; measured r03: 4.8e-6 at nrr 128, 2.4e-6 at nrr 32


def test_few_shot_inversion_matches_reference(golden):
    """H2: the harness' DEFAULT path = the script's own flow (interleaved groups, e4e features into every group,
    eval_seq.py:183-187) against the fixture the reference recorded with that flow; two drive frames (nrr 32 and nrr 128)."""
    from encoder_common import build_inversion_net, run_few_shot, fixture_deviations
    net = build_inversion_net('full').cuda()
    ws, res, r_list, image, image2 = run_few_shot(net, 'cuda')
    dev = fixture_deviations(golden('encoder_fewshot.npz'), ws, res, r_list, image, image2)
    print('few-shot inversion, deviation per recorded tensor:', {k: float(f'{v:.2e}') for k, v in dev.items()})
    assert image.shape == (1, 3, 512, 512) and image2.shape == (1, 3, 512, 512)
    bad = {k: v for k, v in dev.items() if v > (TOL_DRIVE_RGB if k.startswith('drive_image') else TOL_FEATURES)}
    assert not bad, bad
    # ... and through the harness' default for clips: >= 8 drive frames go out as captured calls of 8 (+ one-frame
    # calls), every frame with its own depth range.  Nine copies of the fixture's nrr-128 drive frame: each is the recorded frame.
    from encoder_common import drive_frame2
    from invertavatar_amd import eval_seq
    d2 = drive_frame2(128)
    rep = lambda t: t.cuda().expand(9, *t.shape[1:]).contiguous()
    clip, _ = eval_seq.drive_sequence(net, ws, res, rep(d2['c']), rep(d2['uvcoords']), jitter=rep(d2['jitter'].squeeze(-1)), neural_rendering_resolution=128)
    from invertavatar_amd import _runtime
    assert sorted(k[0] for k in _runtime.state(net).drive_graphs) == [1, 8]          # the captured calls really ran
    for k in (0, 5, 8):
        dev_k = fixture_deviations(golden('encoder_fewshot.npz'), ws, res, r_list, image, clip[k:k + 1])
        assert dev_k['drive_image_nrr128'] <= TOL_DRIVE_RGB and dev_k['drive_image_nrr128_crop'] <= TOL_DRIVE_RGB, (k, dev_k)
    net.generator.neural_rendering_resolution = 32
    # the ConvGRU state shapes a rank derives from the trunk features (inversion_parallel's header-free state broadcast) are the states' own
    from encoder_common import source_batch
    src = source_batch('cuda')
    with torch.no_grad():
        tf = net.trunk_features(src['image'][:1], src['uv'][:1], image2)
    for u, (unet, key) in enumerate(((net.unet_encoder.texture_unet, 'texture'), (net.unet_encoder.triplane_unet, 'triplane'))):
        assert unet.gru_state_shapes(tf[key]) == [tuple(h.shape) for h in r_list[u]], key


def test_renders_batched_across_groups_equal_the_per_group_calls():
    """eval_seq.group_renders (every group's source frames from the e4e features in one call of 8, each frame with its group's depth
    range) against the per-group calls AR_eval_forward makes itself, with the marcher's draws given explicitly (frame k sees the same
    numbers either way)."""
    from encoder_common import build_inversion_net
    from invertavatar_amd import eval_seq, synthetic
    net = build_inversion_net('full').cuda()
    g = net.generator
    nrr = g.neural_rendering_resolution = 64
    n = 8
    src = [int(round(k * 32 / n)) for k in range(n)]
    images = torch.cat([synthetic.source_frames(7 + k // 4, 4)[k % 4:k % 4 + 1] for k in range(n)]).cuda()
    cams, uvc = synthetic.camera_labels(src).cuda(), synthetic.uv_conditions(src).cuda()
    sels = [slice(k, None, 2) for k in range(2)]            # group 0 = frames 0, 2, 4, 6; group 1 = frames 1, 3, 5, 7
    gen = torch.Generator().manual_seed(11)
    jit = torch.rand(n, nrr * nrr, 48, generator=gen).cuda()                # group-major frame order
    u = torch.rand(n * nrr * nrr, 48, generator=gen).cuda()
    rays = nrr * nrr
    with torch.no_grad():
        ws = net.encode(images[:1])
        tex, sta = net._backbones(ws)
        e4e = {'w': ws, 'texture': tex, 'static': sta}
        want = [g.synthesis_withTexture(ws.expand(4, -1, -1), [f.expand(4, -1, -1, -1) for f in tex], cams[sel], {'uvcoords_image': uvc[sel]},
                                        static_feats=[f.expand(4, -1, -1, -1) for f in sta], noise_mode='const', jitter=jit[4 * k:4 * k + 4],
                                        u_importance=u[4 * k * rays:(4 * k + 4) * rays])['image'] for k, sel in enumerate(sels)]
        got = eval_seq.group_renders(net, ws, e4e, cams, uvc, sels, draws=(jit, u))
    for k in range(2):
        err = (got[k] - want[k]).abs().max().item()
        print(f'group {k}: batched render vs its own call: max |d| = {err:.2e}')
        assert got[k].shape == want[k].shape == (4, 3, 512, 512) and err <= 2e-5


def test_graphed_inversion_equals_the_eager_flow():
    """eval_seq.GraphedInversion (encode | renders | trunks | decoder chains as captured graphs over three streams, bench.py's encoder
    leg) against the eager sequential loop of the script, on the clip it was captured with and on another one: same calls, same
    arguments; the renderer's random draws pinned to device tensors made before the capture."""
    import contextlib
    from encoder_common import build_inversion_net
    from invertavatar_amd import eval_seq, synthetic
    net = build_inversion_net('full').cuda()
    net.generator.neural_rendering_resolution = 64
    n = 8
    draws = {}

    @contextlib.contextmanager
    def device_randomness():
        orig_like, orig_rand = torch.rand_like, torch.rand

        def draw(shape, dtype):
            if shape not in draws:
                draws[shape] = orig_rand(*shape, generator=torch.Generator().manual_seed(len(shape) + shape[-1])).cuda()
            return draws[shape].to(dtype)
        torch.rand_like = lambda t, *a, **k: draw(tuple(t.shape), t.dtype)
        torch.rand = lambda *size, **k: draw(tuple(size[0]) if len(size) == 1 and not isinstance(size[0], int) else tuple(size), torch.float32)
        try:
            yield
        finally:
            torch.rand_like, torch.rand = orig_like, orig_rand

    def clip(seed, first):
        src = [first + int(round(k * 32 / n)) for k in range(n)]
        images = torch.cat([synthetic.source_frames(seed + k // 4, 4)[k % 4:k % 4 + 1] for k in range(n)]).cuda()
        return images, synthetic.source_uv(seed + 10, src).cuda(), synthetic.camera_labels(src).cuda(), synthetic.uv_conditions(src).cuda()
    first, second = clip(7, 0), clip(3, 5)
    with device_randomness(), torch.no_grad():     # (the train-mode BatchNorms' running statistics move from call to call; no output reads them)
        cache = {'whole': True}
        eval_seq.few_shot_inversion(net, *first, graphed=cache)          # capture
        for inp in (first, second):
            ws_e, res_e, r_e = eval_seq.few_shot_inversion(net, *inp)
            ws_g, res_g, r_g = eval_seq.few_shot_inversion(net, *inp, graphed=cache)
            a_list = [ws_g] + list(res_g['texture']) + list(res_g['static']) + [t for r in r_g for t in r]
            b_list = [ws_e] + list(res_e['texture']) + list(res_e['static']) + [t for r in r_e for t in r]
            assert len(a_list) == len(b_list)
            worst = max((a - b).abs().max().item() / max(b.abs().max().item(), 1.0) for a, b in zip(a_list, b_list))
            print(f'graphed pipeline vs eager loop: worst relative deviation {worst:.2e} over {len(a_list)} tensors')
            assert worst <= 5e-6          # (library GEMMs may pick other kernels under capture; the fixture tolerance is TOL_FEATURES)


@pytest.mark.parametrize('b,c,h,w,oh,ow', [(1, 512, 16, 16, 32, 32), (1, 512, 32, 32, 64, 64), (2, 5, 7, 9, 20, 13), (1, 3, 4, 4, 1, 1)])
def test_bilinear_upsample_add_matches_interpolate(b, c, h, w, oh, ow):
    """ia_upsample_bilinear_add (the e4e pyramid's `_upsample_add`) against F.interpolate(align_corners=True) + y in fp64."""
    from invertavatar_amd import hipops
    torch.manual_seed(h + ow)
    x, y = torch.randn(b, c, h, w), torch.randn(b, c, oh, ow)
    want = torch.nn.functional.interpolate(x.double(), size=(oh, ow), mode='bilinear', align_corners=True) + y.double()
    got = hipops.upsample_bilinear_add(x.cuda(), y.cuda()).cpu()
    assert got.shape == want.shape and (got.double() - want).abs().max().item() <= 2e-6 * max(want.abs().max().item(), 1.0)


@pytest.mark.parametrize('spatial', [16, 32, 64])
def test_style_head_with_fused_epilogues_matches_the_module(spatial):
    """e4e.GradualStyleBlock on the device path (trunk_hip.style_head_forward: LeakyReLU in the epilogues of ia_conv2d_down_sx /
    ia_conv3x3_s2_tiny, split format handed from layer to layer, shared split of the input) against the module in fp64 on the CPU."""
    import copy
    from invertavatar_amd import hipops
    from invertavatar_amd.encoder_inversion.models import e4e, trunk_hip
    torch.manual_seed(spatial)
    head = e4e.GradualStyleBlock(512, 512, spatial).requires_grad_(False)
    for m in head.convs:
        if isinstance(m, torch.nn.Conv2d):
            m.weight.mul_(3.0)                         # keep the signal alive through 4 .. 6 stride-2 layers
    x = torch.randn(1, 512, spatial, spatial)
    want = copy.deepcopy(head).double()(x.double()).float()
    head = head.cuda()
    with torch.no_grad():
        assert trunk_hip.style_head_supported(head, x.cuda())
        got = head(x.cuda()).cpu()
        shared = head(x.cuda(), hipops.act_split(x.cuda())).cpu()
    err = (got - want).abs().max().item() / max(want.abs().max().item(), 1.0)
    print(f'style head @{spatial}^2: {err:.2e} of max |ref| = {want.abs().max().item():.3g}')
    assert err <= 2e-5 and torch.equal(got, shared)


@pytest.mark.parametrize('prelu', [False, True])
def test_convgru_cell_kernels_match_the_torch_cell(prelu):
    """ia_convgru_gates / ia_convgru_update around the two library convolutions against the cell written with ATen ops
    (unet_encoders.py:8-49), over a 4-frame series with a carried state; the CPU result of the same module is the reference."""
    from invertavatar_amd.encoder_inversion.models.unet_encoders import ConvGRU
    torch.manual_seed(3)
    cell = ConvGRU(24, out_act_prelu=prelu).requires_grad_(False)
    x = torch.randn(2, 4, 24, 16, 20)
    h0 = torch.randn(2, 24, 16, 20)
    want, want_h = cell(x, h0.clone(), seq2seq=True)
    dev = cell.cuda()
    got, got_h = dev(x.cuda(), h0.cuda(), seq2seq=True)
    assert dev._fused(x.cuda())
    assert (got.cpu() - want).abs().max().item() <= 2e-5 and (got_h.cpu() - want_h).abs().max().item() <= 2e-5
    one, _ = dev(x[:, 0].cuda(), h0.cuda())
    assert (one.cpu() - want[:, 0]).abs().max().item() <= 2e-5


def _randomise_unit(unit):
    for m in unit.modules():                       # non-trivial BatchNorm statistics and PReLU slopes
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.5); m.running_var.uniform_(0.5, 2.0); m.weight.uniform_(0.5, 1.5); m.bias.normal_(0, 0.3)
        if isinstance(m, torch.nn.PReLU):
            m.weight.uniform_(0.05, 0.4)


@pytest.mark.parametrize('in_c,depth,stride,res', [(64, 64, 2, 64), (64, 128, 2, 48), (128, 128, 1, 32), (256, 256, 1, 32), (512, 512, 1, 16),
                                                  (256, 512, 2, 32)])
def test_residual_unit_on_hip_convolutions_matches_torch(in_c, depth, stride, res):
    """bottleneck_IR_SE: BatchNorm folded into the staging / epilogue of ia_conv2d_mfma_sx, PReLU in the epilogue, stride 2 by
    sub-sampling -- against the module's own torch.nn forward in fp64 on the CPU (helpers.py:102-124).  Eval mode, then TRAIN mode
    (batch statistics, as eval_seq.py runs the e4e trunk): result and running statistics."""
    import copy
    from invertavatar_amd.encoder_inversion.models import helpers, layers, trunk_hip
    torch.manual_seed(in_c + depth + stride)
    unit = helpers.bottleneck_IR_SE(in_c, depth, stride).requires_grad_(False).eval()
    _randomise_unit(unit)
    x = torch.randn(3, in_c, res, res)
    ref = copy.deepcopy(unit).double()
    want = ref(x.double()).float()
    unit = unit.cuda()
    with torch.no_grad():
        assert trunk_hip.unit_supported(unit, x.cuda())
        got = trunk_hip.unit_forward(unit, x.cuda()).cpu()
        layers.HIP_CONVS = False
        try:
            lib = unit(x.cuda()).cpu()                 # the library route, for scale
        finally:
            layers.HIP_CONVS = True
    assert got.shape == want.shape
    err = (got - want).abs().max().item() / max(want.abs().max().item(), 1.0)
    err_lib = (lib - want).abs().max().item() / max(want.abs().max().item(), 1.0)
    print(f'unit {in_c}->{depth} s{stride} @{res}: HIP {err:.2e}, library fp32 {err_lib:.2e} (relative to max |ref|, vs fp64)')
    assert err <= 2e-5
    unit.train(); ref.train()
    want = ref(x.double()).float()
    with torch.no_grad():
        assert trunk_hip.unit_supported(unit, x.cuda())
        got = trunk_hip.unit_forward(unit, x.cuda()).cpu()
    err = (got - want).abs().max().item() / max(want.abs().max().item(), 1.0)
    print(f'   train mode: HIP {err:.2e}')
    assert err <= 3e-5
    for k in (0, 4):
        assert (unit.res_layer[k].running_mean.cpu() - ref.res_layer[k].running_mean.float()).abs().max().item() <= 2e-5
        assert (unit.res_layer[k].running_var.cpu() - ref.res_layer[k].running_var.float()).abs().max().item() <= 2e-5
        assert int(unit.res_layer[k].num_batches_tracked) == int(ref.res_layer[k].num_batches_tracked) == 1


@pytest.mark.parametrize('train_second', [False, True])
def test_unit_chain_with_the_tail_writing_the_next_units_input(train_second):
    """helpers.run_trunk over three residual units (stride 2, stride 1, stride 1) with trunk_hip.SE_WRITES_NEXT_SPLIT: a unit's
    squeeze-and-excitation tail writes the next eval-mode unit's normalised split input (ia_se_gate_split) -- against the same chain
    with the stand-alone split, and against the modules in fp64; a train-mode unit in the middle takes its own batch statistics."""
    import copy
    from invertavatar_amd.encoder_inversion.models import helpers, trunk_hip
    torch.manual_seed(9)
    body = torch.nn.Sequential(helpers.bottleneck_IR_SE(64, 128, 2), helpers.bottleneck_IR_SE(128, 128, 1), helpers.bottleneck_IR_SE(128, 128, 1)).requires_grad_(False).eval()
    for unit in body:
        _randomise_unit(unit)
    if train_second:
        body[1].train()
    x = torch.randn(2, 64, 64, 64)
    ref = copy.deepcopy(body).double()
    want_taps = []
    t = x.double()
    for unit in ref:
        t = unit(t)
        want_taps.append(t.float())
    body = body.cuda()
    state = copy.deepcopy(body.state_dict())
    outs = {}
    for fused in (True, False):
        body.load_state_dict(state)
        trunk_hip.SE_WRITES_NEXT_SPLIT = fused
        try:
            with torch.no_grad():
                last, taps = helpers.run_trunk(body, x.cuda(), (0, 1, 2))
        finally:
            trunk_hip.SE_WRITES_NEXT_SPLIT = True
        outs[fused] = [t.cpu() for t in taps]
        for got, want in zip(outs[fused], want_taps):
            assert (got - want).abs().max().item() <= 3e-5 * max(want.abs().max().item(), 1.0)
    for a, b_ in zip(outs[True], outs[False]):
        assert (a - b_).abs().max().item() <= 2e-6 * max(b_.abs().max().item(), 1.0)


@pytest.mark.parametrize('i,o,k,s,p,b,h,w,route', [
    (256, 512, 1, 2, 0, 4, 32, 32, 'gemm'), (64, 128, 1, 2, 0, 1, 128, 128, 'gemm'), (384, 32, 1, 1, 0, 1, 32, 32, 'gemm'), (96, 256, 1, 1, 0, 1, 128, 128, 'gemm'),
    (7, 64, 3, 1, 1, 4, 256, 256, 'f32'), (3, 64, 3, 1, 1, 1, 256, 256, 'f32'), (24, 96, 3, 1, 1, 1, 256, 256, 'sx'), (96, 96, 3, 1, 1, 1, 256, 256, 'sx'),
    (512, 512, 3, 2, 1, 1, 16, 16, 'sx'), (512, 512, 3, 2, 1, 1, 64, 64, 'sx'), (1024, 512, 3, 1, 1, 1, 16, 16, 'sx'), (64, 64, 2, 2, 0, 2, 64, 64, 'patch'),
    (32, 64, 8, 8, 0, 1, 128, 128, 'patch'), (512, 512, 3, 2, 1, 1, 4, 4, 'tiny'), (512, 512, 3, 2, 1, 1, 2, 2, 'tiny'), (512, 512, 3, 2, 1, 1, 8, 8, 'tiny'), (40, 24, 3, 2, 1, 3, 8, 8, 'tiny'), (16, 16, 3, 1, 1, 1, 8, 8, None), (3, 64, 7, 4, 3, 1, 256, 256, None)])
def test_conv2d_layer_routes(i, o, k, s, p, b, h, w, route):
    """layers.Conv2d on a device tensor: every route of trunk_hip.conv_forward (and the library fall-through for the shapes it does
    not take) against the same torch.nn.Conv2d in fp64 on the CPU."""
    from invertavatar_amd.encoder_inversion.models import layers, trunk_hip
    torch.manual_seed(i + o + k)
    conv = layers.Conv2d(i, o, k, s, p, bias=(o != 64)).requires_grad_(False)
    x = torch.randn(b, i, h, w)
    want = torch.nn.functional.conv2d(x.double(), conv.weight.double(), None if conv.bias is None else conv.bias.double(), s, p).float()
    conv = conv.cuda()
    with torch.no_grad():
        assert trunk_hip._conv_route(conv, x.cuda()) == route
        got = conv(x.cuda()).cpu()
    assert got.shape == want.shape
    err = (got - want).abs().max().item() / max(want.abs().max().item(), 1.0)
    print(f'conv {i}->{o} k{k} s{s} @{h}x{w} B{b} [{route}]: {err:.2e} relative to max |ref| (vs fp64)')
    assert err <= 2e-5
    with torch.enable_grad():                         # autograd takes torch.nn.Conv2d.forward
        xg = x.cuda().requires_grad_(True)
        assert trunk_hip._conv_route(conv, xg) is None and conv(xg).requires_grad


@pytest.mark.parametrize('ch,h,w', [(64, 32, 40), (512, 16, 16)])
def test_convgru_cell_with_hip_convolutions_matches_the_torch_cell(ch, h, w):
    """The cell's two convolutions on ia_conv2d_mfma_sx (unet_encoders.ConvGRU._hip_convs; the 16^2 cell of the decoders on the
    stream-K plan): against the same module on the CPU (unet_encoders.py:8-49), 4-frame series, carried state."""
    from invertavatar_amd.encoder_inversion.models.unet_encoders import ConvGRU
    torch.manual_seed(5)
    cell = ConvGRU(ch).requires_grad_(False)
    x = torch.randn(1, 4, ch, h, w) * 0.5
    h0 = torch.randn(1, ch, h, w) * 0.5
    want, want_h = cell(x, h0.clone(), seq2seq=True)
    dev = cell.cuda()
    with torch.no_grad():
        assert dev._hip_convs(x[:, 0].cuda()) is not None
        got, got_h = dev(x.cuda(), h0.cuda(), seq2seq=True)
    assert (got.cpu() - want).abs().max().item() <= 2e-5 and (got_h.cpu() - want_h).abs().max().item() <= 2e-5


def test_decoder_double_conv_and_sft_heads_on_hip_convolutions():
    """DoubleConv in TRAIN mode (batch statistics over the 4 frames of a group, eval_seq.py:92) and a CS-SFT head pair, through
    ia_conv2d_mfma_sx, against the same modules on the CPU in fp64; the BatchNorm's running statistics move as torch's do."""
    from invertavatar_amd.encoder_inversion.models import trunk_hip
    from invertavatar_amd.encoder_inversion.models.unet_encoders import DoubleConv
    torch.manual_seed(11)
    dc = DoubleConv(72, 64).requires_grad_(False).train()
    for m in dc.modules():
        if isinstance(m, torch.nn.PReLU):
            m.weight.uniform_(-0.2, 0.5)                       # both signs: the composite slope of PReLU(PReLU(.)) has two cases
    x = torch.randn(4, 72, 32, 36) * 1.5 + 0.3
    ref_mod = __import__('copy').deepcopy(dc).double()
    want = ref_mod(x.double()).float()
    dev = dc.cuda()
    with torch.no_grad():
        assert trunk_hip.double_conv_supported(dev, x.cuda())
        got = dev(x.cuda()).cpu()
    assert (got - want).abs().max().item() <= 3e-5 * max(want.abs().max().item(), 1.0)
    bn_dev, bn_ref = dev.double_conv[0], ref_mod.double_conv[0]
    assert (bn_dev.running_mean.cpu() - bn_ref.running_mean.float()).abs().max().item() <= 1e-5
    assert (bn_dev.running_var.cpu() - bn_ref.running_var.float()).abs().max().item() <= 1e-5
    head = torch.nn.Sequential(torch.nn.Conv2d(96, 96, 3, 1, 1), torch.nn.LeakyReLU(0.2, True), torch.nn.Conv2d(96, 64, 3, 1, 1)).requires_grad_(False)
    t = torch.randn(1, 96, 64, 64)
    want = __import__('copy').deepcopy(head).double()(t.double()).float()
    head = head.cuda()
    with torch.no_grad():
        assert trunk_hip.conv_lrelu_conv_supported(head, t.cuda())
        from invertavatar_amd import hipops
        got = trunk_hip.conv_lrelu_conv_forward(head, hipops.act_split(t.cuda())).cpu()
    assert (got - want).abs().max().item() <= 3e-5 * max(want.abs().max().item(), 1.0)


@pytest.mark.parametrize('in_c,depth,stride', [(64, 64, 2), (64, 128, 2), (256, 256, 1)])
def test_se_gate_kernel_matches_the_module(in_c, depth, stride):
    """ia_se_gate (pool + gate + multiply + shortcut + add, strided views) against SEModule + the add of bottleneck_IR_SE in fp64."""
    from invertavatar_amd.encoder_inversion.models import helpers, trunk_hip
    torch.manual_seed(depth + stride)
    unit = helpers.bottleneck_IR_SE(in_c, depth, stride).requires_grad_(False).eval()
    x = torch.randn(2, in_c, 24, 20)
    full = torch.randn(2, depth, 24, 20)                        # stands for the stride-1 result of conv2 (+ BatchNorm)
    v = full[:, :, ::stride, ::stride]
    ref = unit.double()
    want = (ref.res_layer[5](v.double()) + ref.shortcut_layer(x.double())).float()
    unit = unit.float().cuda()
    with torch.no_grad():
        got = trunk_hip.se_tail(unit, full.cuda()[:, :, ::stride, ::stride], x.cuda()).cpu()
    assert got.shape == want.shape and (got - want).abs().max().item() <= 2e-6 * max(want.abs().max().item(), 1.0)


def _fixture_draws(device='cpu'):
    """draws(group) of inversion_parallel from the fixture's pinned randomness (encoder_common.fixed_randomness): the group's jitter
    and RandomState(99) uniforms of the importance pass."""
    import numpy as np
    from encoder_common import source_batch
    jit = source_batch(device)['jitter']
    n_it = jit.shape[0] // 4

    def draws(idx):
        j = jit[idx::n_it]
        u = torch.from_numpy(np.random.RandomState(99).rand(j.shape[0] * j.shape[1], 48).astype(np.float32))
        return j.reshape(j.shape[0], j.shape[1], 48), u
    return draws


def _sharded_worker(rank, world, port, tmp):
    import os
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    torch.distributed.init_process_group('gloo', rank=rank, world_size=world)        # both ranks on cuda:0, collectives over gloo
    from encoder_common import build_inversion_net, source_batch
    from invertavatar_amd import inversion_parallel
    net = build_inversion_net('full').cuda()
    net.generator.neural_rendering_resolution = 32
    src = source_batch('cuda')
    ws, res, r_list = inversion_parallel.few_shot_inversion_sharded(net, src['image'], src['uv'], src['c'], src['uvcoords'], rank=rank,
                                                                    world_size=world, draws=_fixture_draws())
    torch.save(dict(ws=ws.cpu(), texture=[t.cpu() for t in res['texture']], static=[t.cpu() for t in res['static']],
                    gru=[[h.cpu() for h in states] for states in r_list]), f'{tmp}.{rank}')
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(1500)
def test_sharded_few_shot_inversion_equals_the_one_process_flow(tmp_path, golden):
    """the inversion of BASELINE configs[4] over two ranks (source renders sharded by frame with their group's depth
    range and draws; texture chain on rank 0, tri-plane chain on rank 1; one broadcast per owner) reproduces the one-process features
    and ConvGRU states -- train-mode BatchNorm groups stay whole -- and, through them, the reference fixture."""
    import torch.multiprocessing as mp
    from encoder_common import build_inversion_net, fixed_randomness, source_batch
    from invertavatar_amd import eval_seq
    tmp = str(tmp_path / 'sharded')
    mp.spawn(_sharded_worker, args=(2, 29661, tmp), nprocs=2, join=True)
    got = [torch.load(f'{tmp}.{r}') for r in range(2)]
    net = build_inversion_net('full').cuda()
    net.generator.neural_rendering_resolution = 32
    src = source_batch('cuda')
    n_it = src['image'].shape[0] // 4
    ws, res, r_list = eval_seq.few_shot_inversion(net, src['image'], src['uv'], src['c'], src['uvcoords'],
                                                  hook=lambda idx: fixed_randomness(src['jitter'][idx::n_it]))
    ref = dict(ws=ws.cpu(), texture=[t.cpu() for t in res['texture']], static=[t.cpu() for t in res['static']],
               gru=[[h.cpu() for h in states] for states in r_list])

    def deviation(a, b):
        worst = {}
        worst['ws'] = (a['ws'] - b['ws']).abs().max().item()
        for key in ('texture', 'static'):
            for i, (x, y) in enumerate(zip(a[key], b[key])):
                worst[f'{key}{i}'] = (x - y).abs().max().item() / max(1.0, y.abs().max().item())
        for u, (sa, sb) in enumerate(zip(a['gru'], b['gru'])):
            assert len(sa) == len(sb)
            for k, (x, y) in enumerate(zip(sa, sb)):
                worst[f'gru{u}_{k}'] = (x - y).abs().max().item() / max(1.0, y.abs().max().item())
        return worst
    for rank in range(2):
        dev = deviation(got[rank], ref)
        print(f'sharded inversion, rank {rank}: worst deviation from the one-process flow {max(dev.values()):.2e}')
        assert max(dev.values()) <= TOL_FEATURES, {k: v for k, v in dev.items() if v > TOL_FEATURES}
    assert deviation(got[0], got[1]) == {k: 0.0 for k in deviation(got[0], got[1])}        # both ranks hold the same bits
