"""VERDICT r5 item 4, "measure first": the distribution of the compositing weights of the 95 merged intervals per ray on the bench workload
(full-width generator, synthetic parameters, nrr 128, frame 1 of the orbit), computed by the CPU oracle (test infrastructure, not product).
An interval's colour contributes weight * mean(colour) to the pixel: intervals below a bound could skip the colour half of the decoder."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch

from invertavatar_amd import synthetic
from invertavatar_amd.training_avatar_texture.triplane_v20 import TriPlaneGenerator
from oracle import generator as OG, renderer as ORR

torch.set_num_threads(8)
nrr = int(os.environ.get('NRR', 128))
gen = TriPlaneGenerator(**synthetic.generator_kwargs('full')).eval().requires_grad_(False)
synthetic.fill_parameters(gen)
sd = {k: v.detach() for k, v in gen.state_dict().items()}
ws = gen.mapping(synthetic.latent(0, 1), synthetic.conditioning_camera(), truncation_psi=0.7, truncation_cutoff=14)
seen = []
orig = ORR.ray_march


def spy(colors, densities, depths, white_back=False):
    out = orig(colors, densities, depths, white_back)
    seen.append(out[2])
    return out


ORR.ray_march = spy
for frame in (1, 60):
    seen.clear()
    OG.synthesis(sd, ws, synthetic.camera_labels([frame]), synthetic.uv_conditions([frame]), synthetic.jitter([frame], nrr * nrr), nrr=nrr)
    w = seen[-1].reshape(-1, seen[-1].shape[-2])          # final pass: [rays, 95]
    print(f'frame {frame}: {w.shape[0]} rays x {w.shape[1]} intervals; sum of weights per ray: mean {w.sum(1).mean():.4f}')
    for bound in (1e-8, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3):
        below = (w < bound).float().mean().item()
        # a SAMPLE's colour is needed when either interval it ends / starts carries weight
        need = torch.zeros(w.shape[0], w.shape[1] + 1, dtype=torch.bool)
        need[:, :-1] |= w >= bound
        need[:, 1:] |= w >= bound
        groups = need.reshape(w.shape[0], 6, 16).any(-1).float().mean().item()
        print(f'  weight < {bound:.0e}: {100 * below:5.1f} % of intervals; samples whose colour is needed {100 * need.float().mean().item():5.1f} %; '
              f'groups of 16 merged samples with any needed colour {100 * groups:5.1f} %; dropped weight per ray (max) {w[w < bound].sum().item() / w.shape[0]:.2e} (mean)')
