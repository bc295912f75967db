"""How many launches of each kind one inversionNet.trunk_features call (both UNet trunks, 4 frames) makes, counted at the hipops wrappers
(torch.profiler under ROCTracer drops events in some runs).  python tools/probes/count_trunk_launches.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch
from invertavatar_amd import eval_seq, synthetic, hipops
from invertavatar_amd.encoder_inversion.models.uvnet import inversionNet
from invertavatar_amd.training_avatar_texture.triplane_v20 import TriPlaneGenerator
gen = TriPlaneGenerator(**synthetic.generator_kwargs('full')).eval().requires_grad_(False)
synthetic.fill_parameters(gen)
net = inversionNet(generator=gen, encoding_triplane=True, encoding_texture=True).requires_grad_(False)
synthetic.fill_encoder_parameters(net)
net = eval_seq.set_eval_seq_modes(net.cuda())
counts = {}
for name in ('se_gate', 'se_gate_split', 'act_split', 'conv2d_mfma_sx', 'conv2d_down_sx'):
    fn = getattr(hipops, name)
    def wrap(fn=fn, name=name):
        def call(*a, **k):
            counts[name] = counts.get(name, 0) + 1
            return fn(*a, **k)
        return call
    setattr(hipops, name, wrap())
im = synthetic.source_frames(7, 4).cuda(); uv = synthetic.source_uv(17, [0, 8, 16, 24]).cuda()
y = torch.randn(4, 3, 512, 512, device='cuda')
with torch.no_grad():
    net.trunk_features(im, uv, y)
print(counts)
