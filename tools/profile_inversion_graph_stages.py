"""GPU time of every stage graph of eval_seq.GraphedInversion replayed ALONE (no host issue time in it), their sum, the replay of
the whole inversion, and whether two stage graphs replayed on two streams overlap.  python tools/profile_inversion_graph_stages.py"""
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


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


with torch.no_grad():
    gi = eval_seq.GraphedInversion(net, images, uvs, cams, uvc)
    gi(images, uvs, cams, uvc)
    total = 0.0
    rows = [('E   e4e encode + backbones of the identity', gi.g_encode), (f'R   renders of all {n} source frames from the e4e features (one call of 8)', gi.g_render[0])]
    for k in range(len(gi.g_trunks)):
        rows += [(f'T{k}  IR-SE50 trunks of both UNets, group {k}', gi.g_trunks[k]), (f'D{k}  decoder chains + conditioned static backbone, group {k}', gi.g_decode[k])]
    for label, graph in rows:
        t = timed(graph.replay)
        total += t
        print(f'{label:66s} {t:7.2f} ms')
    print(f'{"sum of the stages replayed alone":66s} {total:7.2f} ms')
    print(f'{"all stage graphs replayed in order (GraphedInversion call)":66s} {timed(lambda: gi(images, uvs, cams, uvc)):7.2f} ms')
    cache = {}
    print(f'{"eager loop (few_shot_inversion, captured encode)":66s} {timed(lambda: eval_seq.few_shot_inversion(net, images, uvs, cams, uvc, graphed=cache)):7.2f} ms')

    # do two stage graphs replayed on two streams overlap?
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

    def both(ga, gb):
        main = torch.cuda.current_stream()
        sa.wait_stream(main); sb.wait_stream(main)
        with torch.cuda.stream(sa):
            ga.replay()
        with torch.cuda.stream(sb):
            gb.replay()
        main.wait_stream(sa); main.wait_stream(sb)
    for la, ga, lb, gb in (('T0', gi.g_trunks[0], 'T1', gi.g_trunks[1]), ('R', gi.g_render[0], 'T1', gi.g_trunks[1]), ('D0', gi.g_decode[0], 'T1', gi.g_trunks[1])):
        print(f'{la} and {lb} on two streams: {timed(lambda: both(ga, gb)):6.2f} ms   (alone {timed(ga.replay):.2f} + {timed(gb.replay):.2f})')

    # the pieces inversion_parallel shards: one frame's render and trunks (frame-parallel stages at N = 8), and the two
    # decoder chains of a group alone (their owners run them side by side)
    def graph_of(fn):
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fn(); fn()
        torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            out = fn()
        return gr, out
    ws, e4e = gi.ws, gi.e4e
    im, uv, cm, uc = gi.group_in[0]
    for t in (1, 2, 4):
        gr, y = graph_of(lambda: gen.synthesis_withTexture(ws.expand(t, -1, -1), [f.expand(t, -1, -1, -1) for f in e4e['texture']], cm[:t], {'uvcoords_image': uc[:t]},
                                                           static_feats=[f.expand(t, -1, -1, -1) for f in e4e['static']], noise_mode='const')['image'])
        print(f'render of {t} frame(s) from the e4e features                        {timed(gr.replay):7.2f} ms')
        gt, _ = graph_of(lambda: net.trunk_features(im[:t], uv[:t], y))
        print(f'IR-SE50 trunks of both UNets, {t} frame(s)                          {timed(gt.replay):7.2f} ms')
    y4, feats = gi._keep[0], gi._keep[1]
    for parts in (('texture',), ('triplane',)):
        gd, _ = graph_of(lambda: net.AR_eval_forward({'image': im, 'uv': uv}, cm, {'uvcoords_image': uc}, ws, [None, None], e4e_results=e4e, return_fake=False,
                                                     y0_image=y4, trunk_feats=feats, parts=parts))
        print(f'decoder chain of group 0 alone: {parts[0]:10s}                          {timed(gd.replay):7.2f} ms')
    ge, _ = graph_of(lambda: net.encode(gi.inputs[0][:1]))
    print(f'e4e encode alone                                                   {timed(ge.replay):7.2f} ms')
