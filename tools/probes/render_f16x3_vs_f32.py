"""ia_render_rays with the decoder on fp16 hi / lo pairs (in-tree build) against the exact-fp32 decoder of r02 - r05 (a -DIA_RENDER_F16X3=0 build
given as argv[1]) on the micro-benchmark's inputs: max |d| per output, differing index entries.  Each library in its own process."""
import os
import subprocess
import sys

if len(sys.argv) > 2:      # child: dump outputs
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
    import torch
    from invertavatar_amd import hipops, synthetic
    from oracle import renderer as OR
    torch.manual_seed(0)
    nrr = 128
    planes = hipops.planes_channels_last(torch.randn(1, 3, 32, 256, 256, device='cuda') * 0.5)
    cams = synthetic.camera_labels([0])
    ro, rd = OR.ray_sampler_zxc(cams[:, :16].view(-1, 4, 4), cams[:, 16:25].view(-1, 3, 3), nrr)
    ro, rd = ro.cuda().contiguous(), rd.cuda().contiguous()
    jit = synthetic.jitter([0], nrr * nrr).squeeze(-1).cuda().contiguous()
    dist = torch.norm(ro, dim=-1).mean().reshape(1)
    w0, b0, w1, b1 = torch.randn(64, 32, device='cuda'), torch.randn(64, device='cuda') * .1, torch.randn(33, 64, device='cuda'), torch.randn(33, device='cuda') * .1
    rgb, depth, wsum, aux = hipops.render_rays(planes, ro, rd, jit, dist, w0, b0, w1, b1, debug=True)
    torch.save(dict(rgb=rgb.cpu(), depth=depth.cpu(), wsum=wsum.cpu(), **{k: v.cpu() for k, v in aux.items()}), sys.argv[2])
    sys.exit(0)
import torch
outs = []
for k, lib in enumerate((None, sys.argv[1])):
    env = dict(os.environ)
    if lib:
        env['IA_HIP_LIB'] = os.path.abspath(lib)
    path = f'/tmp/render_cmp_{k}.pt'
    subprocess.run([sys.executable, __file__, 'child', path], env=env, check=True, stderr=subprocess.DEVNULL)
    outs.append(torch.load(path))
a, b = outs
for k in a:
    if a[k].dtype in (torch.int32, torch.int64):
        print(f'{k:14s} differing entries {(a[k] != b[k]).float().mean().item() * 100:.4f} %')
    else:
        fin = torch.isfinite(a[k]) & torch.isfinite(b[k])
        print(f'{k:14s} max |d| {(a[k][fin] - b[k][fin]).abs().max().item():.3e}   (max |value| {b[k][fin].abs().max().item():.3e})')
