"""Few-shot inversion (BASELINE configs[2] / [4], 8 sources, eval_seq.py flow) by stage: HIP-event times of encode, the e4e feature
passes and each AR_eval_forward group, eager and as a captured graph.  Under rocprofv3 --kernel-trace --stats the same run gives the
kernel table (profiles/r03_encoder_kernel_stats.csv)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

from invertavatar_amd import eval_seq, synthetic
from invertavatar_amd.encoder_inversion.models.uvnet import inversionNet
from invertavatar_amd.training_avatar_texture.triplane_v20 import TriPlaneGenerator


def main():
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
    g = net.generator
    stages = {}

    def timed(name, fn):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        out = fn()
        e1.record()
        torch.cuda.synchronize()
        stages.setdefault(name, []).append((e0.elapsed_time(e1), (time.perf_counter() - t0) * 1e3))
        return out

    with torch.no_grad():
        for rep in range(3):
            ws = timed('encode (e4e, 1 frame)', lambda: net.encode(images[:1]))
            tex = timed('texture backbone', lambda: g.texture_backbone.synthesis(ws, cond_list=None, return_list=True, noise_mode='const'))
            sta = timed('static backbone', lambda: g.backbone.synthesis(ws, cond_list=None, return_list=True, noise_mode='const'))
            res, r_list = {'w': ws, 'texture': tex, 'static': sta}, [None, None]
            for idx in range(2):
                sel = slice(idx, None, 2)
                x = {'image': images[sel], 'uv': uvs[sel]}
                # the pieces of AR_eval_forward, timed one by one (same calls, uvnet.py:160-203)
                T = 4
                over = lambda feats: [f.expand(T, -1, -1, -1) for f in feats]   # noqa: E731
                y0 = timed('group: synthesis_withTexture (4 frames)', lambda: g.synthesis_withTexture(ws.expand(T, -1, -1), over(tex), cams[sel],
                           {'uvcoords_image': uvc[sel]}, static_feats=over(sta), noise_mode='const'))
                delta = y0['image'] - x['image'][:, :3]
                uv_in = net.get_unet_uvinput(x['uv'], delta)
                tri_in = torch.cat([x['image'][:, :3], delta], dim=-3)
                off, r_list[0] = timed('group: texture UNet (IR-SE50 + ConvGRU decoder)', lambda: net.unet_encoder.texture_unet(uv_in.unsqueeze(0), r_list=r_list[0], return_list=True))
                sft, r_list[1] = timed('group: tri-plane UNet (IR-SE50 + ConvGRU decoder + SFT heads)', lambda: net.unet_encoder.triplane_unet(tri_in.unsqueeze(0), r_list=r_list[1]))
                timed('group: static backbone with CS-SFT conditions', lambda: g.backbone.synthesis(ws, cond_list=None, return_list=True, feat_conditions=sft, noise_mode='const'))
        if '--graph-stages' in sys.argv:      # pure GPU time of each stage: captured once, replayed
            def gtime(name, fn):
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    fn()
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    fn()
                gr.replay()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    gr.replay()
                torch.cuda.synchronize()
                print(f'  graph replay  {name:58s} {(time.perf_counter() - t0) / 5 * 1e3:7.2f} ms', flush=True)
            sel = slice(0, None, 2)
            x = {'image': images[sel], 'uv': uvs[sel]}
            T = 4
            over = lambda feats: [f.expand(T, -1, -1, -1) for f in feats]   # noqa: E731
            gtime('encode (e4e, 1 frame)', lambda: net.encode(images[:1]))
            gtime('synthesis_withTexture (4 frames)', lambda: g.synthesis_withTexture(ws.expand(T, -1, -1), over(tex), cams[sel], {'uvcoords_image': uvc[sel]},
                                                                                       static_feats=over(sta), noise_mode='const'))
            y0 = g.synthesis_withTexture(ws.expand(T, -1, -1), over(tex), cams[sel], {'uvcoords_image': uvc[sel]}, static_feats=over(sta), noise_mode='const')
            delta = y0['image'] - x['image'][:, :3]
            uv_in = net.get_unet_uvinput(x['uv'], delta)
            tri_in = torch.cat([x['image'][:, :3], delta], dim=-3)
            tu, pu = net.unet_encoder.texture_unet, net.unet_encoder.triplane_unet
            gtime('texture UNet: trunk', lambda: tu._encode(uv_in))
            feats_t = tu._encode(uv_in)
            gtime('texture UNet: decoder + heads', lambda: tu._heads(feats_t, T, [None] * 4))
            gtime('tri-plane UNet: whole', lambda: pu(tri_in.unsqueeze(0), r_list=None))
            sft_, _ = pu(tri_in.unsqueeze(0), r_list=None)
            gtime('static backbone with CS-SFT', lambda: g.backbone.synthesis(ws, cond_list=None, return_list=True, feat_conditions=sft_, noise_mode='const'))
        print('stage                                                            GPU ms   host ms   (last of 3 repetitions)')
        tot = [0.0, 0.0]
        for name, vals in stages.items():
            per_rep = len(vals) // 3
            gpu = sum(v[0] for v in vals[-per_rep:])
            host = sum(v[1] for v in vals[-per_rep:])
            tot[0] += gpu; tot[1] += host
            print(f'{name:64s} {gpu:7.2f}  {host:7.2f}   x{per_rep}')
        print(f'{"sum":64s} {tot[0]:7.2f}  {tot[1]:7.2f}')
        for label, cache in (('eager', None), ('graphed encode', {}), ('graphed encode + groups', {'group_graph': True})):
            for rep in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ws2, res2, _ = eval_seq.few_shot_inversion(net, images, uvs, cams, uvc, graphed=cache)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) * 1e3
            print(f'few_shot_inversion, {label}: {ms:.2f} ms (third run)')
            if cache is None:
                ref = [t.clone() for t in res2['texture'] + res2['static']]
            else:
                err = max((a - b).abs().max().item() / max(b.abs().max().item(), 1.0) for a, b in zip(res2['texture'] + res2['static'], ref))
                print(f'graphed vs eager features: max relative deviation {err:.2e} (the renderer draws fresh importance samples in each run)')


if __name__ == '__main__':
    main()
