#!/usr/bin/env python3
"""Headline benchmark: frames/s of TriPlaneGenerator.synthesis (512^2 output, 128^2 neural render) on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W      (N > 1: launched by torch.distributed.run,
one rank per GPU, RCCL over xGMI).  A step = one pass of the generator hot path over one batch of synthetic
frames per rank (BASELINE configs[1]: reenact_avatar_next3d single-seed 512^2 render, nrr=128, 1 frame per call);
with N ranks the frames of a step are sharded one per rank and collected with a single all-gather
(configs[3] pattern).  Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

Extra legs on rank 0 at N == 1 (outside the timed region):
  roofline     -- per-launch HIP-event timing of the dominant kernel family (the fp32 MFMA convolution) over extra
                  frames; algorithmic FLOPs per launch / measured duration vs the 157.3 TFLOP/s fp32 MFMA peak
  cpu_baseline -- the CPU oracle (the reference's pure-PyTorch op path restated) timed on the host cores on a
                  bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from invertavatar_amd import hipops, synthetic  # noqa: E402
from invertavatar_amd.training_avatar_texture.triplane_v20 import TriPlaneGenerator  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md
PEAK_FP16_MFMA_TFLOPS = 2500.0  # dense fp16 MFMA
PEAK_HBM_GBS = 8000.0
NRR = 128
FRAMES_PER_RANK = 1


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--width', default='full', choices=['full', 'small'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-sr-fp16', action='store_true', help='skip the extra leg with the SR head on the fp16 MFMA')
    ap.add_argument('--cpu-frames', type=int, default=3)
    ap.add_argument('--eager', action='store_true', help='issue every launch from Python instead of replaying a HIP graph')
    ap.add_argument('--in-flight', type=int, default=1, help='frames in flight per rank (captured graphs on separate streams)')
    return ap.parse_args()


def setup_distributed(n):
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if n > 1 or world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        torch.cuda.set_device(local)
        torch.distributed.init_process_group('nccl', rank=rank, world_size=world)   # nccl == RCCL on ROCm
    else:
        torch.cuda.set_device(0)
    return rank, world, local


def make_step(gen, ws, cams, uvs, jits, world, rank, graphed=None):
    """Returns step(k): render frame k of this rank and all-gather the fp32 images of the step."""
    gathered = torch.empty(world * FRAMES_PER_RANK, 3, 512, 512, device='cuda') if world > 1 else None
    n_frames = cams.shape[0]

    def step(k):
        i = k % n_frames
        if graphed is not None:
            out = graphed(ws, cams[i:i + 1], uvs[i:i + 1], jits[i:i + 1])
        else:
            out = gen.synthesis(ws, cams[i:i + 1], {'uvcoords_image': uvs[i:i + 1]}, neural_rendering_resolution=NRR,
                                noise_mode='const', evaluation=True, jitter=jits[i:i + 1])
        img = out['image']
        if world > 1:
            torch.distributed.all_gather_into_tensor(gathered, img.contiguous())
            return gathered
        return img
    return step


def pmc_traffic(family):
    """HBM bytes per launch of the dominant kernel family from the committed PMC collection (rocprofv3 --pmc FETCH_SIZE /
    --pmc WRITE_SIZE, separate passes over this same command with --eager; tools/pmc_frame.sh).  Counters cannot be read
    from inside this process, so the figure comes from profiles/; None when that file is absent or names another family."""
    path = os.path.join(REPO, 'profiles', 'r01_pmc_frame_hbm_traffic.json')
    if family != 'conv2d_mfma' or not os.path.exists(path):
        return None
    try:
        with open(path) as fh:
            return round(json.load(fh)['_summary']['conv_family_per_logical_launch_mb'] * 1e6)
    except (KeyError, ValueError, OSError):
        return None


def _profiled_frames(step, frames, single_stream):
    from invertavatar_amd.training_avatar_texture import triplane_v20
    saved, triplane_v20.SINGLE_STREAM = triplane_v20.SINGLE_STREAM, single_stream
    hipops.PROFILE = []
    try:
        for k in range(frames):
            step(1000 + k)
        torch.cuda.synchronize()
        return hipops.PROFILE
    finally:
        hipops.PROFILE = None
        triplane_v20.SINGLE_STREAM = saved


def roofline_leg(step, frames=3):
    """Per-launch HIP-event timing of every fused stage over `frames` extra eager frames; dominant family by total time.
    The kernels are timed with the frame's launches in program order on ONE stream (triplane_v20.SINGLE_STREAM), i.e. without
    neighbours from the other streams of a frame stretching them -- the same condition rocprofv3 imposes on the committed
    kernel stats; `in_frame_avg_launch_us` is the same average with the five streams of a normal frame running."""
    _profiled_frames(step, 1, True)                     # (the first single-stream frame allocates that path's buffers)
    recs = _profiled_frames(step, frames, True)
    in_frame = [e0.elapsed_time(e1) for name, _, _, e0, e1, desc in _profiled_frames(step, frames, False)
                if name.startswith('conv2d_mfma') and desc.endswith('f16x3')]
    fam = {}
    split = dict(ms=0.0, flops=0.0, launches=0)     # conv launches whose products are fp16 hi/lo pairs (3 MFMAs per k-step)
    for name, flops, nbytes, e0, e1, desc in recs:
        key = 'conv2d_mfma' if name.startswith('conv2d_mfma') else name
        f = fam.setdefault(key, dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
        ms = e0.elapsed_time(e1)
        f['ms'] += ms; f['flops'] += flops; f['bytes'] += nbytes; f['launches'] += 1
        if key == 'conv2d_mfma' and desc.endswith('f16x3'):
            split['ms'] += ms; split['flops'] += flops; split['launches'] += 1
    dom = max(fam, key=lambda k: fam[k]['ms'])
    d = fam[dom]
    achieved = d['flops'] / (d['ms'] * 1e-3) / 1e12
    if dom == 'conv2d_mfma' and split['ms'] > 0.5 * d['ms']:
        # Most of the family's time is in the split form: price it against the matrix pipe it runs on.  Executed fp16 MFMA
        # FLOPs = 3 x the algorithmic fp32 FLOPs of those launches; peak = dense fp16 MFMA (MI355X_MICROARCH.md).
        executed = 3.0 * split['flops'] / (split['ms'] * 1e-3) / 1e12
        out = dict(bound='mfma', kernel='conv2d_mfma (3x3 layers >= 32^2: fp32 products from fp16 hi/lo pairs, 3 x v_mfma_f32_32x32x16_f16)',
                   achieved=round(executed, 2), peak=PEAK_FP16_MFMA_TFLOPS, unit='TFLOP/s', frac=round(executed / PEAK_FP16_MFMA_TFLOPS, 4),
                   traffic=pmc_traffic(dom), launches_per_frame=split['launches'] // frames,
                   avg_launch_us=round(split['ms'] * 1e3 / split['launches'], 2),
                   in_frame_avg_launch_us=round(sum(in_frame) * 1e3 / max(len(in_frame), 1), 2),
                   algorithmic_gflop_per_frame=round(split['flops'] / frames / 1e9, 1),
                   algorithmic_f32_tflops=round(split['flops'] / (split['ms'] * 1e-3) / 1e12, 2),
                   whole_conv_family=dict(algorithmic_f32_tflops=round(achieved, 2), launches_per_frame=d['launches'] // frames,
                                          ms_per_frame=round(d['ms'] / frames, 3),
                                          frac_of_f32_mfma_peak=round(achieved / PEAK_FP32_MFMA_TFLOPS, 4)))
    else:
        out = dict(bound='mfma', kernel=dom, achieved=round(achieved, 2), peak=PEAK_FP32_MFMA_TFLOPS, unit='TFLOP/s',
                   frac=round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), traffic=pmc_traffic(dom), launches_per_frame=d['launches'] // frames,
                   avg_launch_us=round(d['ms'] * 1e3 / d['launches'], 2), algorithmic_gflop_per_frame=round(d['flops'] / frames / 1e9, 1))
    others = {}
    for k, f in fam.items():
        others[k] = dict(ms_per_frame=round(f['ms'] / frames, 4), tflops=round(f['flops'] / (f['ms'] * 1e-3) / 1e12, 2),
                         algorithmic_gbs=round(f['bytes'] / (f['ms'] * 1e-3) / 1e9, 1), launches_per_frame=f['launches'] // frames)
    if 'render_rays' in fam:
        r = fam['render_rays']
        others['render_rays']['frac_fp32_peak'] = round(r['flops'] / (r['ms'] * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)
        others['render_rays']['frac_hbm_peak'] = round(r['bytes'] / (r['ms'] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)
    return out, others


def cpu_baseline_leg(gen, ws, cams, uvs, jits, frames):
    """The oracle (test infrastructure, CPU restatement of the reference's torch op path) on the host cores."""
    from oracle import generator as OG
    sd = {k: v.detach().cpu() for k, v in gen.state_dict().items()}
    ws_c, cams_c, uvs_c, jits_c = ws.cpu(), cams.cpu(), uvs.cpu(), jits.cpu()
    cores = min(torch.get_num_threads(), 32)     # the torch-CPU path stops scaling (and oversubscribes) beyond ~32 threads
    torch.set_num_threads(cores)
    with torch.no_grad():
        OG.synthesis(sd, ws_c, cams_c[:1], uvs_c[:1], jits_c[:1].unsqueeze(-1), nrr=NRR)   # warm
        t0 = time.perf_counter()
        for i in range(frames):
            j = (i + 1) % cams_c.shape[0]
            OG.synthesis(sd, ws_c, cams_c[j:j + 1], uvs_c[j:j + 1], jits_c[j:j + 1].unsqueeze(-1), nrr=NRR)
        dt = time.perf_counter() - t0
    return dict(value=round(frames / dt, 4), unit='frames/s', cores=cores, kind='port',
                sample=f'{frames} frames of the same workload (B=1, nrr={NRR}, 512^2 out, fp32) after 1 warm-up frame')


def f32_mfma_only_leg(gen, ws, cams, uvs, jits, args, eager_step):
    """Same workload with every convolution on v_mfma_f32_32x32x2_f32 (SPLIT_FP16_PRODUCTS = False), for comparison with the
    headline, whose large 3x3 layers form their fp32 products from fp16 hi/lo pairs."""
    from invertavatar_amd.graphed import GraphedSynthesis
    from invertavatar_amd.training import networks_stylegan2 as sg2
    saved, sg2.SPLIT_FP16_PRODUCTS = sg2.SPLIT_FP16_PRODUCTS, False
    try:
        graphed = GraphedSynthesis(gen, batch=FRAMES_PER_RANK, neural_rendering_resolution=NRR)
        img = graphed(ws, cams[:1], uvs[:1], jits[:1])['image'].clone()
        n = cams.shape[0]
        for k in range(args.warmup):
            graphed(ws, cams[k % n:k % n + 1], uvs[k % n:k % n + 1], jits[k % n:k % n + 1])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.steps):
            i = k % n
            graphed(ws, cams[i:i + 1], uvs[i:i + 1], jits[i:i + 1])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        sg2.SPLIT_FP16_PRODUCTS = saved
    err = (img - eager_step(0)).abs().max().item()
    return dict(value=round(args.steps / dt, 3), unit='frames/s', ms_per_step=round(dt / args.steps * 1e3, 3),
                max_abs_rgb_vs_headline_run=float(f'{err:.3e}'))


def drive_loop_leg(gen, ws, cams, uvs, jits, args):
    """The drive loop of eval_seq.py:212 / BASELINE configs[2]: the texture and static features of the identity are computed
    once (inversion result) and every drive frame is `synthesis_withTexture` = rasterize + face backbone + renderer + SR."""
    with torch.no_grad():
        tex = gen.texture_backbone.synthesis(ws, cond_list=None, return_list=True, noise_mode='const')
        sta = gen.backbone.synthesis(ws, cond_list=None, return_list=True, noise_mode='const')
        c_s, uv_s, jit_s = cams[:1].clone(), uvs[:1].clone(), jits[:1].clone()

        def call():
            return gen.synthesis_withTexture(ws, tex, c_s, {'uvcoords_image': uv_s}, static_feats=sta, neural_rendering_resolution=NRR,
                                             noise_mode='const', evaluation=True, jitter=jit_s)['image']
        ref = gen.synthesis(ws, cams[:1], {'uvcoords_image': uvs[:1]}, neural_rendering_resolution=NRR, noise_mode='const',
                            evaluation=True, jitter=jits[:1])['image']
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                img = call()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        err = (img - ref).abs().max().item()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = call()
        n = cams.shape[0]

        def step(k):
            i = k % n
            c_s.copy_(cams[i:i + 1]); uv_s.copy_(uvs[i:i + 1]); jit_s.copy_(jits[i:i + 1])
            graph.replay()
            return out
        for k in range(args.warmup):
            step(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.steps):
            step(k)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return dict(value=round(args.steps / dt, 3), unit='frames/s', ms_per_step=round(dt / args.steps * 1e3, 3),
                workload='synthesis_withTexture per drive frame, texture + static backbone features cached (eval_seq.py:212)',
                max_abs_rgb_vs_full_synthesis=float(f'{err:.3e}'))


def sr_fp16_leg(gen, ws, cams, uvs, jits, args, eager_step):
    """Same workload with the SR head as the reference deploys it (sr_num_fp16_res = 4, train_avatar_texture.py:215,365):
    its six 3x3 convolutions run with fp16 operands / fp32 accumulation on the fp16 MFMA (ia_conv2d_mfma_h); backbones and
    renderer stay fp32.  Reported beside the fp32 headline, never as `value`."""
    from invertavatar_amd.graphed import GraphedSynthesis
    from invertavatar_amd.training import networks_stylegan2 as sg2
    gen16 = TriPlaneGenerator(**synthetic.generator_kwargs(args.width, sr_num_fp16_res=4)).eval().requires_grad_(False)
    gen16.load_state_dict(gen.state_dict())
    gen16 = gen16.cuda()
    saved, sg2.FP16_BLOCKS_COMPUTE_FP32 = sg2.FP16_BLOCKS_COMPUTE_FP32, False
    try:
        graphed = GraphedSynthesis(gen16, batch=FRAMES_PER_RANK, neural_rendering_resolution=NRR)
        img16 = graphed(ws, cams[:1], uvs[:1], jits[:1])['image'].clone()
        err = (img16 - eager_step(0)).abs().max().item()
        n = cams.shape[0]
        for k in range(args.warmup):
            graphed(ws, cams[k % n:k % n + 1], uvs[k % n:k % n + 1], jits[k % n:k % n + 1])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.steps):
            i = k % n
            graphed(ws, cams[i:i + 1], uvs[i:i + 1], jits[i:i + 1])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        sg2.FP16_BLOCKS_COMPUTE_FP32 = saved
    return dict(value=round(args.steps / dt, 3), unit='frames/s', ms_per_step=round(dt / args.steps * 1e3, 3),
                dtype='f32 backbones + renderer, SR head: fp16 operands / f32 accumulate / f32 storage',
                max_abs_rgb_vs_f32_run=float(f'{err:.3e}'))


def main():
    args = parse()
    rank, world, _ = setup_distributed(args.gpus)
    torch.backends.cudnn.benchmark = False
    gen = TriPlaneGenerator(**synthetic.generator_kwargs(args.width)).eval().requires_grad_(False)
    synthetic.fill_parameters(gen)
    gen = gen.cuda()
    n_frames = 16
    # every rank renders its own slice of the orbit: frame index = step * world + rank
    frames = [(k * world + rank) % 240 for k in range(n_frames)]
    with torch.no_grad():
        ws = gen.mapping(synthetic.latent(0, 1).cuda(), synthetic.conditioning_camera().cuda(), truncation_psi=0.7, truncation_cutoff=14)
        cams = synthetic.camera_labels(frames).cuda()
        uvs = synthetic.uv_conditions(frames).cuda()
        jits = synthetic.jitter(frames, NRR * NRR).squeeze(-1).cuda()
        eager_step = make_step(gen, ws, cams, uvs, jits, world, rank)
        graphed, launch_mode = None, 'eager'
        if not args.eager:
            from invertavatar_amd.graphed import GraphedSynthesis
            try:
                graphed = GraphedSynthesis(gen, batch=FRAMES_PER_RANK, neural_rendering_resolution=NRR)
                img_g = graphed(ws, cams[:1], uvs[:1], jits[:1])['image'].clone()
                img_e = eager_step(0)[rank * FRAMES_PER_RANK:(rank + 1) * FRAMES_PER_RANK] if world > 1 else eager_step(0)
                err = (img_g - img_e).abs().max().item()
                if not err <= 1e-5:
                    raise RuntimeError(f'graph replay differs from eager by {err}')
                launch_mode = 'hipGraph replay (validated against eager: max |d| = %.1e)' % err
            except Exception as exc:   # fall back to eager launches, and say so in the JSON line
                graphed, launch_mode = None, f'eager (graph capture unavailable: {exc})'
        step = make_step(gen, ws, cams, uvs, jits, world, rank, graphed)
        pipeline = None
        if graphed is not None and args.in_flight > 1 and world == 1:
            from invertavatar_amd.graphed import FramePipeline
            pipeline = FramePipeline(gen, depth=args.in_flight, batch=FRAMES_PER_RANK, neural_rendering_resolution=NRR)
            pipeline.capture(ws, cams[:1], uvs[:1], jits[:1])
            out_p, ev, _ = pipeline.submit(ws, cams[1:2], uvs[1:2], jits[1:2])
            pipeline.drain(); torch.cuda.synchronize()
            err = (out_p['image'] - eager_step(1)).abs().max().item()
            if not err <= 1e-5:
                raise RuntimeError(f'pipelined replay differs from eager by {err}')
            launch_mode += f'; {args.in_flight} frames in flight on separate streams'
            n_frames_ = cams.shape[0]

            def step(k, _p=pipeline):   # noqa: F811  (same work per step; consecutive steps overlap on the GPU)
                i = k % n_frames_
                return _p.submit(ws, cams[i:i + 1], uvs[i:i + 1], jits[i:i + 1])[0]['image']

        for k in range(args.warmup):
            step(k)
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.steps):
            step(k)
        if pipeline is not None:
            pipeline.drain()
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device='cuda', dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = t.item()

        result = {
            'metric': 'frames/sec (512^2 out, 128^2 neural render)', 'value': round(world * FRAMES_PER_RANK * args.steps / dt, 3),
            'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32 (3x3 convolutions >= 32^2 form their f32 products from fp16 hi/lo pairs on the f16 MFMA, f32 accumulate; '
                     'all other arithmetic f32)', 'data': 'synthetic',
            'config': {'workload': 'TriPlaneGenerator.synthesis, reenact_avatar_next3d single-seed render (BASELINE configs[1]): '
                                   '512^2 out, neural_rendering_resolution=128, 1 frame per rank per step, all three backbones + '
                                   'rasterize + fused renderer + SR 8XDC recomputed every frame',
                       'width': args.width, 'frames_per_rank_per_step': FRAMES_PER_RANK, 'parallelism': f'frame-sharded dp{world}',
                       'collective': 'one all_gather of the step\'s [N,3,512,512] fp32 frames' if world > 1 else 'none',
                       'launch': launch_mode},
        }
        if rank == 0 and world == 1:
            if not args.no_sr_fp16:
                result['f32_mfma_only'] = f32_mfma_only_leg(gen, ws, cams, uvs, jits, args, eager_step)
                result['sr_fp16'] = sr_fp16_leg(gen, ws, cams, uvs, jits, args, eager_step)
                result['drive_loop'] = drive_loop_leg(gen, ws, cams, uvs, jits, args)
            if not args.no_roofline:
                result['roofline'], result['kernels'] = roofline_leg(eager_step)
            if not args.no_cpu_baseline:
                result['cpu_baseline'] = cpu_baseline_leg(gen, ws, cams, uvs, jits, args.cpu_frames)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
