"""GAN reenactment harness: counterpart of the reference's reenact_avatar_next3d.py:146-219 (``run_video_animation``) for
BASELINE configs[1], with its helpers ``parse_range`` (:86-98), ``parse_tuple`` (:103-112) and ``layout_grid`` (:117-131).

What the reference's command does per drive frame -- for every seed: ``G.synthesis(w_seed, camera, {'uvcoords_image'},
noise_mode='const', evaluation=True)['image'][0]``, then a one-row picture grid [target | seed 0 | seed 1 ...] as uint8 HWC --
is reproduced here on the MI355X backend: the latent draw (``RandomState(seed).randn``), the conditioning camera, the mapping
call with truncation and the per-frame loop are the script's; the frame conversion runs in ``ia_layout_grid_u8``.

Differences forced by the environment (SURVEY.md 8c "missing data"): there are no pickles, datasets or FaceVerse model on the
box, so the generator comes from constructor kwargs + a state dict (or name-seeded synthetic weights) and the drive sequence is
either a directory in the reference's on-disk layout (``dataset_realcam.json`` labels + ``orthRender256x256_face_eye/*.npy`` UV
renders, reenact_avatar_next3d.py:29-37,70-80) or the synthetic orbit of ``invertavatar_amd.synthetic``.  Frames are written
as ``.npy`` / ``.ppm`` (imageio / libx264 are not installed)."""
import json
import os
import re
from typing import List, Tuple, Union

import numpy as np
import torch

from . import synthetic
from .output import layout_grid  # noqa: F401  (same name and signature as the script's helper)
from .training_avatar_texture.camera_utils import FOV_to_intrinsics, LookAtPoseSampler
from .training_avatar_texture.triplane_v20 import TriPlaneGenerator


def parse_range(s: Union[str, List[int]]) -> List[int]:
    """'1,2,5-10' -> [1, 2, 5, 6, 7, 8, 9, 10]; lists pass through."""
    if isinstance(s, list):
        return s
    out = []
    for part in s.split(','):
        m = re.match(r'^(\d+)-(\d+)$', part)
        out.extend(range(int(m.group(1)), int(m.group(2)) + 1) if m else [int(part)])
    return out


def parse_tuple(s: Union[str, Tuple[int, int]]) -> Tuple[int, int]:
    """'4x2' or '4,2' -> (4, 2); tuples pass through."""
    if isinstance(s, tuple):
        return s
    m = re.match(r'^(\d+)[x,](\d+)$', s)
    if not m:
        raise ValueError(f'cannot parse tuple {s}')
    return int(m.group(1)), int(m.group(2))


def build_generator(network=None, width='full', device='cuda'):
    """TriPlaneGenerator on `device`.  `network`: None -> synthetic name-seeded weights; a path to a torch-saved state dict
    (or {'G_ema': state_dict, 'init_kwargs': {...}}) -> those weights, loaded by name (misc.copy_params_and_buffers contract)."""
    kwargs = synthetic.generator_kwargs(width)
    state = None
    if network is not None:
        blob = torch.load(network, map_location='cpu', weights_only=False)
        if isinstance(blob, dict) and 'G_ema' in blob:
            kwargs = blob.get('init_kwargs', kwargs)
            state = blob['G_ema'] if isinstance(blob['G_ema'], dict) else blob['G_ema'].state_dict()
        else:
            state = blob if isinstance(blob, dict) else blob.state_dict()
    g = TriPlaneGenerator(**kwargs).eval().requires_grad_(False)
    if state is None:
        synthetic.fill_parameters(g)
    else:
        g.load_state_dict(state)
    return g.to(device)


def seed_latents(G, seeds, truncation_psi=1.0, truncation_cutoff=14, fov_deg=18.837):
    """One w per seed exactly as :169-178: z = RandomState(seed).randn(1, z_dim); frontal conditioning camera at the average
    radius / pivot; mapping with truncation.  Returns (list of ws, conditioning_params [1,25])."""
    device = next(G.parameters()).device
    intr = FOV_to_intrinsics(fov_deg, device=device)
    pivot = torch.tensor(G.rendering_kwargs.get('avg_camera_pivot', [0, 0, 0]), device=device)
    radius = G.rendering_kwargs.get('avg_camera_radius', 2.7)
    pose = LookAtPoseSampler.sample(np.pi / 2, np.pi / 2, pivot, radius=radius, device=device)
    cond = torch.cat([pose.reshape(-1, 16), intr.reshape(-1, 9)], 1)
    ws = []
    for seed in seeds:
        z = torch.from_numpy(np.random.RandomState(seed).randn(1, G.z_dim)).to(device)
        ws.append(G.mapping(z, cond, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff))
    return ws, cond


class SyntheticDrive:
    """Drive sequence without a dataset: frame k of the 240-frame camera orbit, the synthetic UV render and target image."""

    def __init__(self, n_frames, nrr=None, with_jitter=False):
        self.n, self.nrr, self.with_jitter = n_frames, nrr, with_jitter

    def __len__(self):
        return self.n

    def __getitem__(self, k):
        item = dict(image=synthetic.source_frames(1000 + k, 1), label=synthetic.camera_labels([k]),
                    vert={'uvcoords_image': synthetic.uv_conditions([k])})
        if self.with_jitter:
            item['jitter'] = synthetic.jitter([k], self.nrr * self.nrr)
        return item


class FolderDrive:
    """Drive sequence in the reference's on-disk layout (reenact_avatar_next3d.py:24-83): `root`/dataset_realcam.json with
    {'labels': [[fname, 25 floats], ...]} and ../orthRender256x256_face_eye/<fname>.npy UV renders ([H,W,>=3], mask binarised
    as the reference does at :78)."""

    def __init__(self, root, mesh_path=None, label_file='dataset_realcam.json'):
        self.root = root
        self.mesh_path = mesh_path or os.path.join(os.path.dirname(root), 'orthRender256x256_face_eye')
        with open(os.path.join(root, label_file), 'rb') as fh:
            self.labels = json.load(fh)['labels']

    def __len__(self):
        return len(self.labels)

    def __getitem__(self, k):
        fname, label = self.labels[k]
        stem = os.path.splitext(fname)[0]
        uv = np.load(os.path.join(self.mesh_path, stem + '.npy')).astype(np.float32)[..., :3]
        uv[..., -1] = (uv[..., -1] >= 0.5).astype(np.float32)
        img_path = os.path.join(self.root, stem + '.npy')
        image = torch.from_numpy(np.load(img_path)).float()[None] if os.path.exists(img_path) else torch.zeros(1, 3, 512, 512)
        return dict(image=image, label=torch.tensor(label, dtype=torch.float32)[None], vert={'uvcoords_image': torch.from_numpy(uv)[None]})


@torch.no_grad()
def run_video_animation(G, drive, seeds, grid_dims=(None, 1), truncation_psi=1.0, truncation_cutoff=14, fov_deg=18.837,
                        fixed_camera=False, max_frames=51, outdir=None, fname='reenact', neural_rendering_resolution=None, to_numpy=True):
    """The frame loop of :191-217.  Returns the list of uint8 HWC mosaics ([target | one image per seed] per drive frame);
    with `outdir` they are also written as <outdir>/<fname>_%04d.npy.  `max_frames` = 51 is the script's `if k > 50: break`."""
    device = next(G.parameters()).device
    ws, cond = seed_latents(G, seeds, truncation_psi, truncation_cutoff, fov_deg)
    grid_w, grid_h = grid_dims
    frames = []
    if outdir is not None:
        os.makedirs(outdir, exist_ok=True)
    _check_split_range(device, start=True)
    for k in range(min(len(drive), max_frames)):
        item = drive[k]
        target = item['image'].to(device).float()
        cam = cond.expand(item['label'].shape[0], -1) if fixed_camera else item['label'].to(device).float()
        vert = {key: v.to(device).float() for key, v in item['vert'].items()}
        kw = {}
        if 'jitter' in item:
            kw['jitter'] = item['jitter'].to(device)
        if neural_rendering_resolution is not None:
            kw['neural_rendering_resolution'] = neural_rendering_resolution
        imgs = [target[0]]
        for w in ws:
            imgs.append(G.synthesis(w, cam, vert, noise_mode='const', evaluation=True, **kw)['image'][0])
        mosaic = layout_grid(torch.stack(imgs), grid_w=grid_w, grid_h=grid_h, to_numpy=to_numpy)
        if outdir is not None:
            np.save(os.path.join(outdir, f'{fname}_{k:04d}.npy'), mosaic if to_numpy else mosaic.cpu().numpy())
        frames.append(mosaic)
    _check_split_range(device)
    return frames


def _check_split_range(device, start=False):
    """The fp16 hi / lo split of the large convolutions clamps at +-65504: the library's always-on range watch says whether any
    activation of the clip hit the clamp (random-init weights stay 5 orders of magnitude below it; a real checkpoint is checked here).
    `device`: where the NETWORK lives (a module, a parameter device or a tensor's device).  `start`: clear the flag at the top of a
    clip -- stream-ordered, no host synchronisation (it is sticky: whatever ran before in this process must not be blamed on this
    clip).  The check at the end is the clip's one device -> host read."""
    if isinstance(device, torch.nn.Module):
        first = next(device.parameters(), None)
        if first is None:
            return
        device = first.device
    if torch.device(device).type != 'cuda':
        return
    from . import hipops
    if start:
        hipops.split_saturation_clear(device)
        return
    if hipops.split_saturation_poll(device):
        raise OverflowError('activations outside the fp16 range (+-65504) were clamped by the hi / lo split of the fp16-pair convolutions: '
                            'set training.networks_stylegan2.SPLIT_FP16_PRODUCTS = False (fp32 MFMA path) for this checkpoint')


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description='GAN reenactment on the MI355X backend (reference: reenact_avatar_next3d.py)')
    ap.add_argument('--network', default=None, help='torch-saved state dict; default: synthetic weights')
    ap.add_argument('--drive_root', default=None, help='drive sequence directory (reference layout); default: synthetic orbit')
    ap.add_argument('--frames', type=int, default=51)
    ap.add_argument('--seeds', type=parse_range, required=True)
    ap.add_argument('--grid', type=parse_tuple, default=None)
    ap.add_argument('--outdir', required=True)
    ap.add_argument('--fname', default='reenact')
    ap.add_argument('--fov-deg', type=float, default=18.837)
    ap.add_argument('--trunc', type=float, default=1.0)
    ap.add_argument('--trunc-cutoff', type=int, default=14)
    ap.add_argument('--fixed_camera', action='store_true')
    ap.add_argument('--width', default='full', choices=['full', 'small'])
    args = ap.parse_args(argv)
    G = build_generator(args.network, args.width)
    drive = FolderDrive(args.drive_root) if args.drive_root else SyntheticDrive(args.frames)
    grid = args.grid or (1 + len(args.seeds), 1)
    frames = run_video_animation(G, drive, args.seeds, grid, args.trunc, args.trunc_cutoff, args.fov_deg, args.fixed_camera,
                                 max_frames=args.frames, outdir=args.outdir, fname=args.fname)
    print(f'wrote {len(frames)} frames of {frames[0].shape} to {args.outdir}')


if __name__ == '__main__':
    main()
