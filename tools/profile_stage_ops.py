"""Kernel table (name, launches, GPU us) of the stages of the few-shot inversion under torch.profiler: which launches make up the
dependent chains of the trunks (T), the decoder chains (D) and the e4e encode (E).  python tools/profile_stage_ops.py"""
import collections
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from torch.profiler import ProfilerActivity, profile

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


def table(label, fn, top=28):
    fn(); fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        fn()
        torch.cuda.synchronize()
    rows = collections.OrderedDict()
    for ev in prof.events():
        if ev.device_type is not None and 'cuda' in str(ev.device_type).lower():
            r = rows.setdefault(ev.name[:90], [0, 0.0])
            r[0] += 1
            r[1] += ev.device_time if hasattr(ev, 'device_time') else ev.cuda_time
    total_n, total_us = sum(r[0] for r in rows.values()), sum(r[1] for r in rows.values())
    print(f'== {label}: {total_n} launches, {total_us / 1e3:.2f} ms of kernel time')
    for name, (cnt, us) in sorted(rows.items(), key=lambda kv: -kv[1][0])[:top]:
        print(f'   {cnt:5d} x {us / cnt:7.1f} us = {us / 1e3:6.2f} ms   {name}')


with torch.no_grad():
    ws = net.encode(images[:1])
    tex, sta = net._backbones(ws)
    e4e = {'w': ws, 'texture': tex, 'static': sta}
    sel = slice(0, None, 2)
    im, uv, cm, uc = images[sel], uvs[sel], cams[sel], uvc[sel]
    y = gen.synthesis_withTexture(ws.expand(4, -1, -1), [f.expand(4, -1, -1, -1) for f in tex], cm, {'uvcoords_image': uc},
                                  static_feats=[f.expand(4, -1, -1, -1) for f in sta], noise_mode='const')['image']
    feats = net.trunk_features(im, uv, y)
    table('E  e4e encode', lambda: net.encode(images[:1]))
    table('T  trunks of both UNets, 4 frames', lambda: net.trunk_features(im, uv, y))
    table('D  decoder chains, group of 4', lambda: net.AR_eval_forward({'image': im, 'uv': uv}, cm, {'uvcoords_image': uc}, ws, [None, None], e4e_results=e4e,
                                                                      return_fake=False, y0_image=y, trunk_feats=feats))
