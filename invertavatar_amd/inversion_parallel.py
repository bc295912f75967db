"""Few-shot inversion (eval_seq.py:136-203, uvnet.py:160-203) over several ranks -- the identity half of BASELINE configs[4].

What `eval_seq.few_shot_inversion` does on one GPU, in dependency order:

  A  ws = encode(first source); texture / static features of that identity (e4e features)            one frame: replicated
  B  for every group of four sources: y0 = synthesis_withTexture(e4e features, the group's cameras)   frames independent
     (every group starts from the e4e features, eval_seq.py:187, so the renders of ALL groups depend on A only)
  C  per group, in order (the ConvGRU states chain the groups; train-mode BatchNorm spans a group's four frames, SURVEY 8e iii):
       texture chain :  UV-space residual -> texture UNet -> texture feature offsets
       tri-plane chain: image residual -> tri-plane UNet -> CS-SFT conditions -> conditioned static backbone
     the two chains never read each other.

Sharding (r05: C split into C1 trunks / C2 decoders -- the IR-SE50 trunks of both UNets run in eval mode, so they are frame-parallel too;
they go to the ranks with the renders, and the owners of the chains run the recurrent decoders only).  B is frame-parallel: the S source frames are dealt to the ranks in contiguous blocks, each frame rendered with the depth
range (`ray_dist`) and the random draws of ITS group's four-frame call, and the renders meet in one all-gather.  C is two-way model
parallel: rank 0 runs the texture chain of all groups, rank 1 (when there is one) the tri-plane chain; each owner then broadcasts its
six feature maps and its ConvGRU states (one flat buffer per owner).  Groups stay whole on their owner (BatchNorm statistics), the
chains stay sequential (GRU recurrence).  Ranks >= 2 only help in B; they receive the features for the drive loop.

Per-clip time (r04, one MI355X: A 8 ms, B 2 x 8 ms, C 2 x (7 + 7 + 1.5) ms -> 55 ms):  N = 1: 55;  N = 2: A 8 + B 8 + C 2 x 8.5 + 1 = 34;
N >= 8: A 8 + B 2 + C 17 + 1 = 28 ms -- see DESIGN.md 7 for the clip arithmetic.

Random draws.  The inversion renders are stochastic (evaluation=False: stratified jitter + uniform importance draws).  A rank must give
frame t of group g the numbers it gets in the one-process call, so the draws are an INPUT here: `draws(group_index) -> (jitter [T,R,48],
u_importance [T*R,48])`; the default derives them from a seed and the group index on the CPU (identical on every rank)."""
import torch

from . import frame_parallel
from .eval_seq import fill_group


def seeded_draws(seed, n_rays, n_frames=4, n_samples=48):
    """draws(group_index) from torch.Generator(seed, group_index): the same numbers on every rank, whatever the sharding."""
    def draws(group_index):
        gen = torch.Generator().manual_seed(int(seed) * 1000003 + int(group_index))
        jitter = torch.rand(n_frames, n_rays, n_samples, generator=gen)
        u = torch.rand(n_frames * n_rays, n_samples, generator=gen)
        return jitter, u
    return draws


def _flat(tensors, device=None):
    """One buffer on the tensors' own device in their own dtype (torch.cat would silently promote a mixed list: refused instead)."""
    if not tensors:
        return torch.zeros(0, device=device)
    dtypes = {t.dtype for t in tensors}
    if len(dtypes) != 1:
        raise TypeError(f'one flat broadcast carries one dtype, got {sorted(str(d) for d in dtypes)}')
    return torch.cat([t.reshape(-1) for t in tensors])


def _unflat(flat, like):
    out, at = [], 0
    for t in like:
        out.append(flat[at:at + t.numel()].reshape(t.shape))
        at += t.numel()
    return out


def _broadcast_list(tensors, src, group=None, device=None):
    """One broadcast of a list of same-dtype tensors whose shapes every rank knows (flat buffer: one collective per owner)."""
    flat = _flat(tensors, device).contiguous()
    if flat.numel():
        torch.distributed.broadcast(flat, src=src, group=group)
    return _unflat(flat, tensors)


@torch.no_grad()
def few_shot_inversion_sharded(net, images, uvs, cams, uvcoords, rank=0, world_size=1, draws=None, sequential_sampling=False,
                               neural_rendering_resolution=None, group=None, shard_trunks=True):
    """`eval_seq.few_shot_inversion` with the source renders sharded by frame and the two UNet chains on ranks 0 and 1.
    Every rank passes the same inputs and gets (ws, {'w', 'texture', 'static'} of the last group, r_list).
    `draws`: see the module docstring (None: seeded_draws(0, R)); with world_size == 1 this is the one-process flow with the same
    draws, which is what the tests compare the sharded result with."""
    from .reenact_avatar_next3d import _check_split_range
    s = images.shape[0]
    assert s in (1, 2, 4) or s % 4 == 0, f'{s} source frames: the script pads to a multiple of 4 first (:135-136)'
    images, uvs, cams, uvcoords = (fill_group(t, s) for t in (images, uvs, cams, uvcoords))
    n = images.shape[0]
    g = net.generator
    _check_split_range(net, start=True)      # (range watch of the fp16 hi / lo split: see eval_seq.few_shot_inversion)
    nrr = neural_rendering_resolution or g.neural_rendering_resolution
    if draws is None:
        draws = seeded_draws(0, nrr * nrr)
    # ---- A: identity (replicated: one frame, deterministic kernels)
    ws = net.encode(images[:1])
    if world_size > 1:      # (the e4e trunk runs library convolutions whose algorithm choice may differ between processes: 1e-6 on ws;
        ws = ws.contiguous()      #  everything downstream is this package's deterministic kernels, so one 28 KB broadcast makes the ranks bit-equal)
        torch.distributed.broadcast(ws, src=0, group=group)
    tex = g.texture_backbone.synthesis(ws, cond_list=None, return_list=True, update_emas=False, noise_mode='const')
    sta = g.backbone.synthesis(ws, cond_list=None, return_list=True, update_emas=False, noise_mode='const')
    e4e = {'w': ws, 'texture': tex, 'static': sta}
    num_iter = max(n // 4, 1)
    sels = [slice(4 * i, 4 * (i + 1)) if sequential_sampling else slice(i, None, num_iter) for i in range(num_iter)]
    frames_of = [list(range(n))[sel] for sel in sels]                 # group -> its four source frames, in call order
    # ---- B: y0 of every (group, frame), sharded in contiguous blocks of the group-major list
    items = [(gi, t) for gi in range(num_iter) for t in range(len(frames_of[gi]))]
    lo, hi = frame_parallel.shard_range(len(items), rank, world_size)
    dev = images.device
    drawn = {}
    mine = []
    for gi, t in items[lo:hi]:
        if gi not in drawn:
            jit, u = draws(gi)
            drawn[gi] = (jit.to(dev), u.to(dev), frame_parallel.global_ray_dist(cams[sels[gi]]).to(dev))
        jit, u, dist = drawn[gi]
        f = frames_of[gi][t]
        r = jit.shape[1]
        y0 = g.synthesis_withTexture(ws, tex, cams[f:f + 1], {'uvcoords_image': uvcoords[f:f + 1]}, static_feats=sta, noise_mode='const',
                                     neural_rendering_resolution=nrr, jitter=jit[t:t + 1], u_importance=u[t * r:(t + 1) * r], ray_dist=dist)
        mine.append(y0['image'])
    shape = (3, g.img_resolution, g.img_resolution)
    block = torch.cat(mine, 0) if mine else torch.zeros((0,) + shape, device=dev)
    counts = [b_ - a_ for a_, b_ in (frame_parallel.shard_range(len(items), r_, world_size) for r_ in range(world_size))]
    y0_all = frame_parallel.all_gather_blocks(block, counts, group) if world_size > 1 else block
    # ---- C1: the IR-SE50 trunks of both UNets, frame-parallel like B (eval-mode BatchNorm: a frame's trunk features depend on that
    # frame's render only, SURVEY 8e iii names train-mode BatchNorm in the DECODERS).  Each rank runs the two trunks on its frames of the
    # group-major list; one all-gather of the flattened features (7.9 MB per frame and UNet at 256^2 inputs) hands them to the owners.
    trunk_all = None
    if shard_trunks and world_size > 1 and not net.trunks_in_eval_mode():
        #  train-mode trunk BatchNorms take batch statistics over the group's frames: a per-frame trunk pass would differ
        # from the one-process result and update the running statistics differently on every rank.  The groups then stay whole.
        shard_trunks = False
    if shard_trunks and world_size > 1:
        mine_feats, like = [], None
        for j, (gi, t) in enumerate(items[lo:hi]):
            f = frames_of[gi][t]
            tf = net.trunk_features(images[f:f + 1], uvs[f:f + 1], block[j:j + 1])
            like = like or tf
            mine_feats.append(torch.cat([v.reshape(-1) for key in ('texture', 'triplane') for v in tf[key]]))
        if like is None:         # (a rank without frames: shapes from one trunk pass on a zero frame -- every rank must know the row length)
            like = net.trunk_features(torch.zeros_like(images[:1]), uvs[:1], torch.zeros((1,) + shape, device=dev))
        row = sum(v.numel() for key in ('texture', 'triplane') for v in like[key])
        rows = torch.stack(mine_feats) if mine_feats else torch.zeros((0, row), device=dev)
        trunk_all = (frame_parallel.all_gather_blocks(rows, counts, group), like)

    def group_trunk_feats(at, k):
        """{'texture': [...], 'triplane': [...]} of frames at .. at + k - 1 of the group-major list, from the gathered rows."""
        rows, like = trunk_all
        out, col = {}, 0
        for key in ('texture', 'triplane'):
            out[key] = []
            for v in like[key]:
                n = v.numel()
                out[key].append(rows[at:at + k, col:col + n].reshape((k,) + tuple(v.shape[1:])))
                col += n
        return out
    # ---- C2: the two chains (recurrent decoders; with C1 off: trunks + decoders), groups in order, on their owners
    tex_owner, tri_owner = 0, (1 if world_size > 1 else 0)
    parts = tuple(p for p, owner in (('texture', tex_owner), ('triplane', tri_owner)) if owner == rank)
    r_list = [None, None]
    updated = e4e
    at = 0
    for gi in range(num_iter):
        sel, k = sels[gi], len(frames_of[gi])
        if parts:
            updated, r_list = net.AR_eval_forward({'image': images[sel], 'uv': uvs[sel]}, cams[sel], {'uvcoords_image': uvcoords[sel]}, ws, r_list,
                                                  e4e_results=e4e, return_fake=False, y0_image=y0_all[at:at + k], parts=parts,
                                                  trunk_feats=None if trunk_all is None else group_trunk_feats(at, k))
        at += k
    if world_size > 1:
        # shapes of the results are those of the e4e features (offsets are added in place of them); the GRU states' shapes are
        # known on the owner only, so they travel behind a small header of element counts
        texture = _broadcast_list([t.clone() for t in (updated['texture'] if rank == tex_owner else tex)], tex_owner, group)
        static = _broadcast_list([t.clone() for t in (updated['static'] if rank == tri_owner else sta)], tri_owner, group)
        updated = {'w': ws, 'texture': texture, 'static': static}
        unets = (net.unet_encoder.texture_unet, net.unet_encoder.triplane_unet)
        if trunk_all is not None and all(hasattr(u, 'gru_state_shapes') for u in unets):
            # every rank holds the trunk features' shapes (`like`) and the UNet derives its ConvGRU states' shapes from them
            # (_UNetBase.gru_state_shapes): ONE payload per owner, no header, no host round trip
            like = trunk_all[1]
            shapes = [unets[0].gru_state_shapes(like['texture']), unets[1].gru_state_shapes(like['triplane'])]
            r_list = [_broadcast_list([h.contiguous() for h in r_list[u]] if rank == owner else [torch.empty(sh, device=dev) for sh in shapes[u]],
                                      owner, group, dev) for u, owner in enumerate((tex_owner, tri_owner))]
        else:
            r_list = [_broadcast_states(r_list[0], tex_owner, rank, dev, group), _broadcast_states(r_list[1], tri_owner, rank, dev, group)]
    _check_split_range(net)
    return ws, updated, r_list


def _broadcast_states(states, src, rank, device, group=None):
    """ConvGRU states of one UNet (a list of tensors known on `src` only) when the other ranks cannot derive their shapes (trunks not
    dealt by frame): shapes first (int64 header), then one flat buffer."""
    if rank == src:
        shapes = [list(h.shape) for h in states]
        dtype_code = _STATE_DTYPES.index(states[0].dtype) if states else 0
        header = torch.tensor([len(shapes), dtype_code] + [v for sh in shapes for v in [len(sh)] + sh], dtype=torch.int64, device=device)
        count = torch.tensor([header.numel()], dtype=torch.int64, device=device)
    else:
        count = torch.zeros(1, dtype=torch.int64, device=device)
    torch.distributed.broadcast(count, src=src, group=group)
    if rank != src:
        header = torch.zeros(int(count.item()), dtype=torch.int64, device=device)
    torch.distributed.broadcast(header, src=src, group=group)
    if rank != src:
        vals, shapes, at = header.tolist(), [], 2
        for _ in range(vals[0]):
            nd = vals[at]
            shapes.append(vals[at + 1:at + 1 + nd])
            at += 1 + nd
        states = [torch.empty(sh, dtype=_STATE_DTYPES[vals[1]], device=device) for sh in shapes]
    return _broadcast_list([h.contiguous() for h in states], src, group, device)


_STATE_DTYPES = (torch.float32, torch.float16, torch.bfloat16, torch.float64)      # header code of the ConvGRU states' dtype
