"""Few-shot inversion + reenactment harness: counterpart of the reference's eval_seq.py:77-219 (``run_video_animation``) for
BASELINE configs[2] and [4].

The flow of the script, restated:
  1. module modes (:91-97): the whole inversion network in train() mode (train-mode BatchNorm in the recurrent up-path,
     non-fused modulated convolutions in the generator), except the IR-SE50 trunks of both UNets (``input_layer``, ``body``);
  2. ``ws = G.encode(first source frame)``; texture / static features of that identity (:168-171);
  3. sources in groups of four through ``AR_eval_forward`` with the ConvGRU states carried from group to group; groups are
     interleaved (``[idx::num_iter]``, :183-186) unless `sequential_sampling`; <= 4 sources are one group, 1 or 2 sources are
     repeated to fill it (:141-150);
  4. drive loop (:206-219): ``synthesis_withTexture(ws, texture, c, v, static_feats=static, noise_mode='const',
     evaluation=True)`` per drive frame, mosaics [ground truth | rendered] through ``layout_grid``.

Inputs are tensors (the datasets, FaceVerse and video writer of the script are outside this build's scope)."""
import torch

from .output import layout_grid


def set_eval_seq_modes(net):
    """Module modes exactly as eval_seq.py:91-97 leaves them."""
    net.train()
    for unet in (net.unet_encoder.triplane_unet, net.unet_encoder.texture_unet):
        unet.input_layer.eval()
        unet.body.eval()
    return net


def fill_group(t, n_src):
    """1 or 2 sources are repeated to a group of four (:141-150); 3 is not a case the script supports."""
    if n_src >= 4:
        return t
    assert n_src in (1, 2), n_src
    return torch.cat([t] * (4 // n_src), dim=0)


class GraphedGroupUpdate:
    """hipGraph of one ``AR_eval_forward`` call on a group of T source frames (uvnet.py:160-203): ~1 100 launches (two IR-SE50 UNets
    with ConvGRU decoders, two generator passes) that eager PyTorch issues in ~45 ms of host time for ~15 ms of GPU work.  The
    group's frames are staged into static buffers, the ConvGRU states live in static buffers that the captured call updates in
    place (a first group starts from zeros: ``h = zeros`` is what the cell does with ``r = None``, unet_encoders.py:44-46), and the
    renderer's random draws go through the captured generator state like any CUDA-graph-safe op.  Results are the graph's static
    output tensors: valid until the next replay."""

    def __init__(self, net, ws, e4e_results, group, warmup=2):
        self.net, self.ws, self.e4e = net, ws, e4e_results
        self.inputs = [t.clone() for t in group]                   # images, uvs, cams, uvcoords of one group
        images, uvs, cams, uvc = self.inputs
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                              # warm-up (library kernel selection, allocator) + state shapes
            for _ in range(warmup):
                _, r = net.AR_eval_forward({'image': images, 'uv': uvs}, cams, {'uvcoords_image': uvc}, ws, [None, None],
                                           e4e_results=e4e_results, return_fake=False)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.state = [[torch.zeros_like(h) for h in unet_states] for unet_states in r]
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            out, new = net.AR_eval_forward({'image': images, 'uv': uvs}, cams, {'uvcoords_image': uvc}, ws,
                                           [list(unet_states) for unet_states in self.state], e4e_results=e4e_results, return_fake=False)
            for old_states, new_states in zip(self.state, new):
                for h_old, h_new in zip(old_states, new_states):
                    h_old.copy_(h_new)
        self.out = out

    def reset(self):
        for unet_states in self.state:
            for h in unet_states:
                h.zero_()

    def __call__(self, group):
        for dst, src in zip(self.inputs, group):
            dst.copy_(src)
        self.graph.replay()
        return self.out, self.state


class GraphedEncode:
    """hipGraph of ``net.encode`` on one source frame (e4e: an IR-SE50 trunk at batch 1 + 14 style heads, ~500 launches that eager
    PyTorch issues in 9 ms for 6 ms of GPU work)."""

    def __init__(self, net, image, warmup=2):
        self.image = image.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                net.encode(self.image)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.ws = net.encode(self.image)

    def __call__(self, image):
        self.image.copy_(image)
        self.graph.replay()
        return self.ws.clone()


BATCH_GROUP_RENDERS = True     # device path: the renders of ALL groups' source frames from the e4e features in calls of up to 8 frames
RENDER_BATCH = 8


def group_renders(net, ws, e4e, cams, uvcoords, sels, draws=None):
    """The renders `y0` of every group's source frames from the e4e features, batched ACROSS groups.  Every group starts from the e4e
    features (eval_seq.py:187), so its render does not depend on the groups before it; rendered in its own call of T = 4 a frame costs
    1.67 ms, in a call of 8 frames 1.2 ms (r05: 13.4 -> 9.9 ms for 8 sources).  What a group's own call would share between its frames is
    the depth range, the batch mean of |camera origin| (renderer.py:311): every frame gets its GROUP's value as `ray_dist`
    (frame_parallel.global_ray_dist over the group's cameras), as inversion_parallel does when it deals the frames to ranks.  The
    stochastic draws of the ray marcher come from the device generator per call, as in the sequential loop (other numbers of the same
    distribution; the fixtures pin them through `hook`, which takes the sequential path) unless `draws` = (jitter [N, rays, samples],
    u_importance [N * rays, samples]) gives them for the N frames in group-major order.  Returns one [T, 3, H, W] image per group."""
    from .frame_parallel import global_ray_dist
    g = net.generator
    n = cams.shape[0]
    order = torch.cat([torch.arange(n, device=cams.device)[sel] for sel in sels])
    dist = torch.cat([global_ray_dist(cams[sel]).to(cams.device).expand(cams[sel].shape[0]) for sel in sels]).contiguous()
    cams_o, uv_o = cams[order], uvcoords[order]
    chunks = []
    for lo in range(0, n, RENDER_BATCH):
        t = min(RENDER_BATCH, n - lo)
        kw = {}
        if draws is not None:
            rays = draws[1].shape[0] // n
            kw = dict(jitter=draws[0][lo:lo + t], u_importance=draws[1][lo * rays:(lo + t) * rays])
        chunks.append(g.synthesis_withTexture(ws.expand(t, -1, -1), [f.expand(t, -1, -1, -1) for f in e4e['texture']], cams_o[lo:lo + t],
                                              {'uvcoords_image': uv_o[lo:lo + t]}, static_feats=[f.expand(t, -1, -1, -1) for f in e4e['static']],
                                              noise_mode='const', ray_dist=dist[lo:lo + t], **kw)['image'])
    y = torch.cat(chunks) if len(chunks) > 1 else chunks[0]
    out, at = [], 0
    for sel in sels:
        t = cams[sel].shape[0]
        out.append(y[at:at + t])
        at += t
    return out


class GraphedInversion:
    """The few-shot inversion of one clip shape (S source frames) as captured hipGraphs, one per stage, handing over through the
    tensors they were captured with and replayed in order on the caller's stream:
        E        e4e encode + the two backbones of the identity
        R        renders of every group's source frames from the e4e features (group_renders: calls of up to 8 frames)
        T_k      inversionNet.trunk_features of group k: IR-SE50 trunks of both UNets
        D_k      AR_eval_forward(y0_image, trunk_feats) of group k: decoder chains (ConvGRU states carried) + conditioned static backbone
    The same calls with the same arguments as the sequential loop; a replay on new inputs is the eager call on them (the train-mode
    BatchNorms move their running statistics once per replay).  Nothing is issued from the host (~5 000 launches otherwise), and the
    stage graphs replayed alone are the GPU times of the stages (tools/profile_inversion_graph_stages.py: the table of DESIGN.md 7).
    Only D_k depends on the previous group (every group starts from the e4e features, eval_seq.py:187), so T_k+1 could run
    beside D_k -- but not on this runtime (r05, profiles/r05_inversion_pipeline.txt): graphs replayed on different streams run one
    after the other (T_0 and T_1 on two streams: 9.5 ms for 4.8 + 4.8), one graph over the whole flow cannot hold the pipeline
    because a captured stream that forks a stream which forks another one crashes hipStreamEndCapture
    (tools/probes/nested_fork_capture.py), and issued eagerly the three-stream pipeline is host-bound (42.3 vs 41.9 ms).
    Results are copies.  The graphs hold the packed weights of the moment of capture: capture again after changing parameters."""

    def __init__(self, net, images, uvs, cams, uvcoords, sequential_sampling=False, warmup=2):
        self.net = net
        self.inputs = [t.clone() for t in (images, uvs, cams, uvcoords)]
        images, uvs, cams, uvcoords = self.inputs
        dev = images.device
        n = max(images.shape[0] // 4, 1)
        sels = [slice(4 * k, 4 * (k + 1)) if sequential_sampling else slice(k, None, n) for k in range(n)]
        group_in = [tuple(t[sel].clone() for t in self.inputs) for sel in sels]        # (images, uvs, cams, uvcoords) of each group
        self.group_in, self.sels = group_in, sels
        g = net.generator
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                _few_shot_inversion(net, images, uvs, cams, uvcoords, sequential_sampling)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)

        def capture(fn):
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = fn()
            return graph, out

        def encode():
            ws = net.encode(images[:1])
            tex = g.texture_backbone.synthesis(ws, cond_list=None, return_list=True, update_emas=False, noise_mode='const')
            sta = g.backbone.synthesis(ws, cond_list=None, return_list=True, update_emas=False, noise_mode='const')
            return ws, {'w': ws, 'texture': tex, 'static': sta}
        self.g_encode, (self.ws, self.e4e) = capture(encode)
        ws, e4e = self.ws, self.e4e
        self.g_render, self.g_trunks, self.g_decode = [], [], []
        r_list = [None, None]
        gr, y_groups = capture(lambda: group_renders(net, ws, e4e, cams, uvcoords, sels))
        self.g_render.append(gr)
        for k, (im, uv, cm, uc) in enumerate(group_in):
            y0 = y_groups[k]
            gt, feats = capture(lambda: net.trunk_features(im, uv, y0))
            gd, (updated, r_list) = capture(lambda: net.AR_eval_forward({'image': im, 'uv': uv}, cm, {'uvcoords_image': uc}, ws, r_list, e4e_results=e4e,
                                                                      return_fake=False, y0_image=y0, trunk_feats=feats))
            self.g_trunks.append(gt); self.g_decode.append(gd)
            self._keep = getattr(self, '_keep', []) + [y0, feats]
        self.updated, self.r_list = updated, r_list

    def __call__(self, images, uvs, cams, uvcoords):
        for dst, src in zip(self.inputs, (images, uvs, cams, uvcoords)):
            dst.copy_(src)
        for bufs, sel in zip(self.group_in, self.sels):
            for dst, src in zip(bufs, self.inputs):
                dst.copy_(src[sel])
        self.g_encode.replay()
        self.g_render[0].replay()
        for gt, gd in zip(self.g_trunks, self.g_decode):
            gt.replay()
            gd.replay()
        ws = self.ws.clone()
        res = {'w': ws, 'texture': [t.clone() for t in self.updated['texture']], 'static': [t.clone() for t in self.updated['static']]}
        return ws, res, [[h.clone() for h in states] for states in self.r_list]


@torch.no_grad()
def few_shot_inversion(net, images, uvs, cams, uvcoords, sequential_sampling=False, hook=None, graphed=None):
    """See _few_shot_inversion; on the device the always-on range watch of the fp16 hi / lo split brackets the call.
    ``graphed['whole'] = True``: the whole inversion as one hipGraph per clip shape (GraphedInversion, kept in the cache)."""
    from .reenact_avatar_next3d import _check_split_range
    _check_split_range(net, start=True)
    if graphed is not None and graphed.get('whole', False) and hook is None and images.is_cuda:
        key = ('inversion', tuple(images.shape), bool(sequential_sampling))
        if key not in graphed:
            graphed[key] = GraphedInversion(net, images, uvs, cams, uvcoords, sequential_sampling)
        out = graphed[key](images, uvs, cams, uvcoords)
        _check_split_range(net)
        return out
    out = _few_shot_inversion(net, images, uvs, cams, uvcoords, sequential_sampling, hook, graphed)
    _check_split_range(net)
    return out


def _few_shot_inversion(net, images, uvs, cams, uvcoords, sequential_sampling=False, hook=None, graphed=None):
    """images [S,3,512,512] in [-1,1], uvs [S,6,256,256] (x['uv']), cams [S,25], uvcoords [S,256,256,3].  S in {1, 2, 4} or a
    multiple of 4.  `hook(group_index)` may return a context manager entered around each AR_eval_forward (tests pin the
    renderer's random draws with it).  `graphed`: a dict the caller keeps as a cache of captured graphs (device tensors, no hook):
    the e4e encode is then replayed as one hipGraph (host-bound eagerly: 9 -> 6 ms), and with ``graphed['group_graph'] = True`` the
    groups too (GraphedGroupUpdate; they are GPU-bound, measured 68 ms eager vs 73 ms replayed for the whole inversion, so off by
    default; their results are the graph's static tensors, valid until the cache is used again).  Returns (ws, {'w','texture','static'} of the LAST group, r_list)."""
    s = images.shape[0]
    assert s in (1, 2, 4) or s % 4 == 0, f'{s} source frames: the script pads to a multiple of 4 first (:135-136)'
    images, uvs, cams, uvcoords = (fill_group(t, s) for t in (images, uvs, cams, uvcoords))
    g = net.generator
    if graphed is not None and hook is None and images.is_cuda:
        if 'encode' not in graphed:
            graphed['encode'] = GraphedEncode(net, images[:1])
        ws = graphed['encode'](images[:1])
    else:
        ws = net.encode(images[:1])
    tex = g.texture_backbone.synthesis(ws, cond_list=None, return_list=True, update_emas=False, noise_mode='const')
    sta = g.backbone.synthesis(ws, cond_list=None, return_list=True, update_emas=False, noise_mode='const')
    results, r_list = {'w': ws, 'texture': tex, 'static': sta}, [None, None]
    num_iter = max(images.shape[0] // 4, 1)
    updated = results
    if graphed is not None and graphed.get('group_graph', False) and hook is None and images.is_cuda:
        first = slice(0, 4) if sequential_sampling else slice(0, None, num_iter)
        step = graphed.get('group')
        if step is None:
            step = graphed['group'] = GraphedGroupUpdate(net, ws, results, (images[first], uvs[first], cams[first], uvcoords[first]))
        else:      # same network, new identity: the graph reads ws / the e4e features from the tensors it was captured with
            step.ws.copy_(ws)
            for dst, src in zip(step.e4e['texture'] + step.e4e['static'], tex + sta):
                dst.copy_(src)
        step.reset()
        for idx in range(num_iter):
            sel = slice(4 * idx, 4 * (idx + 1)) if sequential_sampling else slice(idx, None, num_iter)
            updated, r_list = step((images[sel], uvs[sel], cams[sel], uvcoords[sel]))
        return ws, updated, r_list
    sels = [slice(4 * idx, 4 * (idx + 1)) if sequential_sampling else slice(idx, None, num_iter) for idx in range(num_iter)]
    y0 = group_renders(net, ws, results, cams, uvcoords, sels) if (BATCH_GROUP_RENDERS and num_iter > 1 and hook is None and images.is_cuda) else None
    for idx in range(num_iter):
        sel = sels[idx]
        ctx = hook(idx) if hook is not None else None
        if ctx is not None:
            ctx.__enter__()
        try:
            # (every group starts from the e4e features: the script passes e4e_results=e4e_results each time, :187)
            updated, r_list = net.AR_eval_forward({'image': images[sel], 'uv': uvs[sel]}, cams[sel], {'uvcoords_image': uvcoords[sel]},
                                                  ws, r_list, e4e_results=results, return_fake=False, y0_image=None if y0 is None else y0[idx])
        finally:
            if ctx is not None:
                ctx.__exit__(None, None, None)
    return ws, updated, r_list


# Drive loop on the device: captured calls (graphed.GraphedDrive) instead of ~150 eager launches per frame.  The script renders one frame
# per call (eval_seq.py:206-219); here the frames go out in calls of DRIVE_GRAPH_BATCH with every frame's OWN depth range (per-frame
# `ray_dist`, frame_parallel.per_frame_ray_dist: the value its one-frame call computes, renderer.py:311), the remainder in one-frame calls.
# r05 measured the same frame at 445 frames/s eager, 516 replayed one per call, 813 in calls of 8.  The graphs are kept per network
# (`_runtime.state(net).drive_graphs`) and per (frames per call, nrr, precision switches); they hold the packed weights of the moment of
# capture -- `clear_drive_graphs(net)` after changing parameters.  Clips shorter than DRIVE_GRAPH_MIN_FRAMES are not worth a capture.
DRIVE_GRAPHS = True
DRIVE_GRAPH_BATCH = 8
DRIVE_GRAPH_MIN_FRAMES = 8


def clear_drive_graphs(net):
    from . import _runtime
    _runtime.state(net).drive_graphs = {}


def _drive_graph(net, ws, results, batch, nrr):
    from . import _runtime
    from .graphed import GraphedDrive
    from .training import networks_stylegan2 as sg2
    g = net.generator
    st = _runtime.state(net)
    cache = getattr(st, 'drive_graphs', None)
    if cache is None:
        cache = st.drive_graphs = {}
    key = (batch, nrr, bool(sg2.FP16_BLOCKS_COMPUTE_FP32), bool(sg2.SPLIT_FP16_PRODUCTS), bool(g.training), ws.device,
           tuple(tuple(t.shape[1:]) for t in list(results['texture']) + list(results['static'])))
    gd = cache.get(key)
    if gd is None:
        gd = cache[key] = GraphedDrive(g, ws, results['texture'], results['static'], batch=batch, neural_rendering_resolution=nrr,
                                       ray_dist_elems=batch if batch > 1 else 0)
        gd.identity = None
    ident = tuple((t.data_ptr(), t._version) for t in [ws] + list(results['texture']) + list(results['static']))
    if gd.identity != ident:
        gd.set_identity(ws, results['texture'], results['static'])
        gd.identity = ident
    return gd


@torch.no_grad()
def drive_sequence(net, ws, results, cams, uvcoords, jitter=None, batch=1, gt=None, neural_rendering_resolution=None, graphed=None):
    """Drive loop (:206-219) over cams [F,25] / uvcoords [F,256,256,3] in calls of `batch` frames (the script uses 1; with
    batch > 1 every frame keeps the depth range of its own call: per-frame `ray_dist`, frame_parallel.per_frame_ray_dist).
    `graphed`: None = the default (DRIVE_GRAPHS: device tensors, batch 1, >= DRIVE_GRAPH_MIN_FRAMES frames -> captured calls, see above),
    True / False force it.  Returns (images [F,3,H,W], mosaics or None): mosaics are the uint8 [gt | rendered] pictures when `gt`
    [F,3,H,W] is given."""
    from .frame_parallel import per_frame_ray_dist
    from .reenact_avatar_next3d import _check_split_range
    g = net.generator
    n = cams.shape[0]
    imgs, mosaics = [], ([] if gt is not None else None)

    def mosaic(images, lo):
        if gt is not None:
            for k in range(images.shape[0]):
                mosaics.append(layout_grid(torch.cat([gt[lo + k:lo + k + 1, :3], images[k:k + 1]], dim=0), grid_w=2, grid_h=1))
    _check_split_range(net, start=True)
    use_graphs = graphed if graphed is not None else (DRIVE_GRAPHS and n >= DRIVE_GRAPH_MIN_FRAMES)
    if use_graphs and batch == 1 and cams.is_cuda and not torch.is_grad_enabled():
        nrr = neural_rendering_resolution or g.neural_rendering_resolution
        lo = 0
        while lo < n:
            b = DRIVE_GRAPH_BATCH if n - lo >= DRIVE_GRAPH_BATCH else 1
            call = _drive_graph(net, ws, results, b, nrr)
            # the stratified noise the renderer draws per call (torch.rand_like, renderer.py:406) when the caller pins none
            jit = jitter[lo:lo + b] if jitter is not None else torch.rand(b, nrr * nrr, 48, device=cams.device)
            out = call(cams[lo:lo + b], uvcoords[lo:lo + b], jit, per_frame_ray_dist(cams[lo:lo + b]) if b > 1 else None)
            imgs.append(out['image'].clone())             # (the call's static output: the next replay overwrites it)
            mosaic(imgs[-1], lo)
            lo += b
        _check_split_range(net)
        return torch.cat(imgs, 0), mosaics
    for lo in range(0, n, batch):
        hi = min(lo + batch, n)
        b = hi - lo
        ws_b = ws.expand(b, -1, -1)
        tex = [t.expand(b, -1, -1, -1) for t in results['texture']]
        sta = [t.expand(b, -1, -1, -1) for t in results['static']]
        kw = {}
        if jitter is not None:
            kw['jitter'] = jitter[lo:hi]
        if neural_rendering_resolution is not None:
            kw['neural_rendering_resolution'] = neural_rendering_resolution
        if b > 1:
            kw['ray_dist'] = per_frame_ray_dist(cams[lo:hi])
        out = g.synthesis_withTexture(ws_b, tex, cams[lo:hi], {'uvcoords_image': uvcoords[lo:hi]}, noise_mode='const', static_feats=sta,
                                      evaluation=True, **kw)
        imgs.append(out['image'])
        mosaic(out['image'], lo)
    _check_split_range(net)      # (one device -> host read per drive sequence: see hipops.split_saturation_poll)
    return torch.cat(imgs, 0), mosaics
