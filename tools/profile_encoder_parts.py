"""Few-shot inversion by MODULE: forward hooks with HIP events on the sub-modules of the three encoders (input layer, trunk stages, decoder
stages, heads, e4e style blocks): where the 53 ms go.  python tools/profile_encoder_parts.py"""
import collections
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

records = collections.OrderedDict()


def watch(name, module):
    def pre(m, inp):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        m._t0 = e

    def post(m, inp, out):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        records.setdefault(name, []).append((m._t0, e))
    module.register_forward_pre_hook(pre)
    module.register_forward_hook(post)


enc = net.encoder
watch('e4e.input_layer', enc.input_layer)
for i, u in enumerate(enc.body):
    watch(f'e4e.body[{"0-2" if i < 3 else "3-6" if i < 7 else "7-20" if i < 21 else "21-23"}]', u)
for i, s in enumerate(enc.styles):
    watch('e4e.styles (14 GradualStyleBlocks)', s)
watch('e4e.latlayers', enc.latlayer1); watch('e4e.latlayers', enc.latlayer2)
for tag, un in (('texture_unet', net.unet_encoder.texture_unet), ('triplane_unet', net.unet_encoder.triplane_unet)):
    watch(f'{tag}.input_layer', un.input_layer)
    for i, u in enumerate(un.body):
        watch(f'{tag}.body[{"0-2 @128" if i < 3 else "3-6 @64" if i < 7 else "7-20 @32" if i < 21 else "21-23 @16"}]', u)
    for k in (1, 2, 3, 4):
        watch(f'{tag}.up{k}', getattr(un, f'up{k}'))
    for nm, m in un.named_children():
        if nm.startswith('outconv') or nm.startswith('sft') or 'scale' in nm or 'shift' in nm:
            watch(f'{tag}.heads', m)
with torch.no_grad():
    for rep in range(3):
        records.clear()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        eval_seq.few_shot_inversion(net, images, uvs, cams, uvc)
        e1.record()
        torch.cuda.synchronize()
print(f'few_shot_inversion (eager, hooks on): {e0.elapsed_time(e1):.2f} ms')
tot = 0.0
for name, evs in records.items():
    ms = sum(a.elapsed_time(b) for a, b in evs)
    tot += ms
    print(f'  {ms:7.2f} ms  {len(evs):4d} calls  {name}')
print(f'  {tot:7.2f} ms  watched modules in total (the rest: generator passes, residual images, glue)')
