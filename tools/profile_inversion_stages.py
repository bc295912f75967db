"""Stage times of the few-shot inversion on ONE GPU, in the decomposition inversion_parallel shards:
  A   e4e encode of the first source (captured) + texture / static backbones of that identity          (one frame: replicated)
  B   render of one source frame from the e4e features (synthesis_withTexture, B = 1)                  (frame-parallel)
  C1  IR-SE50 trunks of both UNets on 1 / 2 / 4 frames (inversionNet.trunk_features)                   (frame-parallel, r05)
  C2  per group of four: texture decoder chain, tri-plane decoder chain + conditioned static backbone   (owners; groups in order)
Prints GPU milliseconds (HIP events, best of 3 after one warm-up).  python tools/profile_inversion_stages.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

from invertavatar_amd import eval_seq, frame_parallel, synthetic
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


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = None
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None else min(best, ms)
    return best, out


with torch.no_grad():
    enc = eval_seq.GraphedEncode(net, images[:1])
    t_enc, ws = timed(lambda: enc(images[:1]))
    t_bb, (tex, sta) = timed(lambda: net._backbones(ws))
    print(f'A   e4e encode (captured)                         {t_enc:7.2f} ms')
    print(f'A   texture + static backbone of the identity      {t_bb:7.2f} ms')
    dist = frame_parallel.global_ray_dist(cams[:4]).cuda()
    t_b, y0 = timed(lambda: gen.synthesis_withTexture(ws, tex, cams[:1], {'uvcoords_image': uvc[:1]}, static_feats=sta, noise_mode='const', ray_dist=dist)['image'])
    print(f'B   render of one source frame                     {t_b:7.2f} ms')
    y4 = torch.cat([gen.synthesis_withTexture(ws, tex, cams[k:k + 1], {'uvcoords_image': uvc[k:k + 1]}, static_feats=sta, noise_mode='const', ray_dist=dist)['image']
                    for k in range(4)])
    for k in (1, 2, 4):
        t_c1, feats = timed(lambda: net.trunk_features(images[:k], uvs[:k], y4[:k]))
        print(f'C1  trunks of both UNets, {k} frame(s)                {t_c1:7.2f} ms')
    e4e = {'w': ws, 'texture': tex, 'static': sta}
    x = {'image': images[:4], 'uv': uvs[:4]}
    for parts, label in ((('texture',), 'texture chain'), (('triplane',), 'tri-plane chain + conditioned static backbone'),
                         (('texture', 'triplane'), 'both chains (two streams)')):
        t_dec, _ = timed(lambda: net.AR_eval_forward(x, cams[:4], {'uvcoords_image': uvc[:4]}, ws, [None, None], e4e_results=e4e, y0_image=y4, parts=parts,
                                                    trunk_feats=feats))
        t_all, _ = timed(lambda: net.AR_eval_forward(x, cams[:4], {'uvcoords_image': uvc[:4]}, ws, [None, None], e4e_results=e4e, y0_image=y4, parts=parts))
        print(f'C2  group of 4, decoders only: {label:46s} {t_dec:7.2f} ms   (trunks + decoders, the r04 split: {t_all:7.2f} ms)')
    t_inv, _ = timed(lambda: eval_seq.few_shot_inversion(net, images, uvs, cams, uvc, graphed={'encode': enc}))
    print(f'few_shot_inversion of 8 sources on one GPU (captured encode, chains on two streams): {t_inv:7.2f} ms')
