"""Three few-shot inversions (8 sources, eval_seq.py flow) for rocprofv3 --kernel-trace --stats: the kernel table of the encoder side."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

from invertavatar_amd import eval_seq, synthetic
from invertavatar_amd.encoder_inversion.models.uvnet import inversionNet
from invertavatar_amd.training_avatar_texture.triplane_v20 import TriPlaneGenerator

gen = TriPlaneGenerator(**synthetic.generator_kwargs('full')).eval().requires_grad_(False)
synthetic.fill_parameters(gen)
net = inversionNet(generator=gen, encoding_triplane=True, encoding_texture=True).requires_grad_(False)
synthetic.fill_encoder_parameters(net)
net = eval_seq.set_eval_seq_modes(net.cuda())
gen.neural_rendering_resolution = 128
n = 8
src = [int(round(k * 32 / n)) for k in range(n)]
images = torch.cat([synthetic.source_frames(7 + k // 4, 4)[k % 4:k % 4 + 1] for k in range(n)]).cuda()
uvs, cams, uvc = synthetic.source_uv(17, src).cuda(), synthetic.camera_labels(src).cuda(), synthetic.uv_conditions(src).cuda()
with torch.no_grad():
    for _ in range(int(os.environ.get('REPS', 3))):
        eval_seq.few_shot_inversion(net, images, uvs, cams, uvc)
    torch.cuda.synchronize()
print('done')
