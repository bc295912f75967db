"""Frame-sharded data parallelism for batched reenactment.

Frames are independent given (ws, texture/static features), so a batch of frames is split into contiguous
blocks, one per rank (one process per GPU), and the rendered frames are collected with ONE all-gather over
RCCL/xGMI.  Two quantities of the reference are batch-global and must not change when the batch is sharded:

  * ``dist`` -- the mean ray-origin norm that sets the depth range (volumetric_rendering/renderer.py:311):
    computed here from the FULL camera batch on every rank (cameras are 25 floats per frame), no collective;
  * the depth clamp bounds min/max over all sample depths (ray_marcher.py:50): they only touch ``image_depth`` on
    rays whose weights vanish; ranks use their local bounds (RGB is unaffected).
"""
import torch


def shard_range(n_frames, rank, world_size):
    """Contiguous block [lo, hi) of `n_frames` owned by `rank` (sizes differ by at most one)."""
    base, extra = divmod(n_frames, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def global_ray_dist(c):
    """Batch mean of |camera origin| from the labels c [N, >=25]: what torch.norm(ray_origins).mean() evaluates to
    for the whole batch, since all rays of a frame share one origin."""
    cam2world = c[:, -25:][:, :16].reshape(-1, 4, 4)
    return torch.norm(cam2world[:, :3, 3].float(), dim=-1).mean().reshape(1)


def per_frame_ray_dist(c):
    """|camera origin| of every frame, [N]: the `dist` each frame gets when the script renders it in a call of its own
    (eval_seq.py:206-212, B = 1: the batch mean of renderer.py:311 is then the frame's own value).  Passed as `ray_dist` it lets a
    rank render its block of drive frames in one batched call with the per-call results."""
    cam2world = c[:, -25:][:, :16].reshape(-1, 4, 4)
    return torch.norm(cam2world[:, :3, 3].float(), dim=-1).contiguous()


def all_gather_blocks(block, counts, group=None):
    """Blocks of `counts[r]` leading entries per rank -> the concatenation on every rank, with ONE all_gather_into_tensor.
    Unequal blocks are padded to the largest (neither gloo nor RCCL gathers tensors of different sizes)."""
    world, most = len(counts), max(counts)
    if block.shape[0] < most:
        block = torch.cat([block, block.new_zeros((most - block.shape[0],) + tuple(block.shape[1:]))], 0)
    full = torch.empty((world * most,) + tuple(block.shape[1:]), dtype=block.dtype, device=block.device)
    if torch.distributed.get_backend(group) == 'gloo':      # (rehearsal backend: no _allgather_base for every device; list form, same bytes)
        torch.distributed.all_gather(list(full.chunk(world)), block.contiguous(), group=group)
    else:
        torch.distributed.all_gather_into_tensor(full, block.contiguous(), group=group)
    if all(c == most for c in counts):
        return full
    return torch.cat([full[r * most:r * most + counts[r]] for r in range(world)], 0)


def render_sharded(generator, ws, c, mesh_condition, rank=0, world_size=1, jitter=None, gather=True, **synthesis_kwargs):
    """Render frames [lo, hi) of the batch on this rank and all-gather the images.

    ws [N or 1, num_ws, w_dim], c [N, 25+], mesh_condition['uvcoords_image'] [N,256,256,3], jitter [N,R,48] or None.
    Returns the full [N,3,H,W] batch on every rank (gather=True) or this rank's block."""
    n = c.shape[0]
    if n < world_size:      # every rank sees the same n: all of them raise, none is left waiting in the collective
        raise ValueError(f'render_sharded: {n} frames cannot be sharded over {world_size} ranks (every rank needs >= 1 frame)')
    lo, hi = shard_range(n, rank, world_size)
    dist = global_ray_dist(c)
    ws_local = ws[lo:hi] if ws.shape[0] == n else ws.expand(hi - lo, -1, -1)
    out = generator.synthesis(ws_local, c[lo:hi], {'uvcoords_image': mesh_condition['uvcoords_image'][lo:hi]},
                              jitter=None if jitter is None else jitter[lo:hi], ray_dist=dist, **synthesis_kwargs)
    img = out['image'].contiguous()
    if not gather or world_size == 1:
        return img
    return all_gather_blocks(img, [b_ - a_ for a_, b_ in (shard_range(n, r, world_size) for r in range(world_size))])
