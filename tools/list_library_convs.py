"""Which convolutions of the few-shot inversion still reach the ATen / MIOpen convolution on the device, by signature.
Wraps torch.nn.functional.conv2d / conv_transpose2d for one inversion of 8 sources (after a warm-up inversion, so that the library's
find step is not in the times) and prints, per signature, calls and GPU microseconds (HIP events around each call: serialised, so the
sum is an upper bound of what the two-stream flow pays).  python tools/list_library_convs.py"""
import collections
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import torch.nn.functional as F

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

log = collections.OrderedDict()
recording = [False]


def wrap(name, fn):
    def call(x, w, bias=None, stride=1, padding=0, *a, **k):
        if not (recording[0] and x.is_cuda):
            return fn(x, w, bias, stride, padding, *a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        y = fn(x, w, bias, stride, padding, *a, **k)
        e1.record()
        torch.cuda.synchronize()
        groups = k.get('groups', a[1] if len(a) > 1 else 1)
        key = (name, tuple(x.shape), tuple(w.shape), str(stride), str(padding), groups, bias is not None, x.is_contiguous())
        c = log.setdefault(key, [0, 0.0])
        c[0] += 1
        c[1] += e0.elapsed_time(e1) * 1e3
        return y
    return call


F.conv2d = wrap('conv2d', F.conv2d)
F.conv_transpose2d = wrap('conv_transpose2d', F.conv_transpose2d)

if '--oneshot' in sys.argv:      # the one-shot flow (eval_updated_os.py: uvnet_new with the transformer-refined decoders)
    from invertavatar_amd import eval_updated_os
    from invertavatar_amd.encoder_inversion.models.uvnet_new import inversionNet as OneShotNet
    net = OneShotNet(generator=gen, encoding_triplane=True, encoding_texture=True).eval().requires_grad_(False)
    synthetic.fill_encoder_parameters(net)
    net = net.cuda()
    one = (synthetic.source_frames(9, 1).cuda(), synthetic.source_uv(19, [12]).cuda(), synthetic.camera_labels([12]).cuda(), synthetic.uv_conditions([12]).cuda())
    flow = lambda: eval_updated_os.one_shot_inversion(net, *one)      # noqa: E731
else:
    flow = lambda: eval_seq.few_shot_inversion(net, images, uvs, cams, uvc)      # noqa: E731
with torch.no_grad():
    flow()
    torch.cuda.synchronize()
    recording[0] = True
    flow()
    torch.cuda.synchronize()

total = sum(v[1] for v in log.values())
print(f'{len(log)} signatures, {sum(v[0] for v in log.values())} calls, {total / 1e3:.2f} ms serialised')
print(f'{"op":17s} {"input":24s} {"weight":22s} {"stride":8s} {"pad":8s} {"g":>4s} bias contig calls      us   us/call   GFLOP/s')
for key, (calls, us) in sorted(log.items(), key=lambda kv: -kv[1][1]):
    name, xs, ws, st, pd, g, hb, ct = key
    s = int(st.strip('()[]').split(',')[0]) if st[0] in '([' else int(st)
    if name == 'conv2d':
        flops = 2 * xs[0] * (xs[2] // s) * (xs[3] // s) * ws[0] * ws[1] * ws[2] * ws[3]
    else:
        flops = 2 * xs[0] * xs[2] * xs[3] * ws[0] * ws[1] * ws[2] * ws[3]
    print(f'{name:17s} {str(xs):24s} {str(ws):22s} {st:8s} {pd:8s} {g:4d} {int(hb):4d} {int(ct):6d} {calls:5d} {us:8.0f} {us / calls:8.1f} {flops * calls / us / 1e3:9.0f}')
