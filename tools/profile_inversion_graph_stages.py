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
    rows = [('E   e4e encode + backbones of the identity', gi.g_encode)]
    for k in range(len(gi.g_render)):
        rows += [(f'R{k}  render of group {k} (4 frames) from the e4e features', gi.g_render[k]), (f'T{k}  IR-SE50 trunks of both UNets, group {k}', gi.g_trunks[k]),
                 (f'D{k}  decoder chains + conditioned static backbone, group {k}', gi.g_decode[k])]
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
    for la, ga, lb, gb in (('T0', gi.g_trunks[0], 'T1', gi.g_trunks[1]), ('R0', gi.g_render[0], 'T1', gi.g_trunks[1]), ('R0', gi.g_render[0], 'R1', gi.g_render[1])):
        print(f'{la} and {lb} on two streams: {timed(lambda: both(ga, gb)):6.2f} ms   (alone {timed(ga.replay):.2f} + {timed(gb.replay):.2f})')
