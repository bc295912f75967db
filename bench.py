#!/usr/bin/env python3
"""Headline benchmark: frames/s of TriPlaneGenerator.synthesis (512^2 output, 128^2 neural render) on MI355X, and the
max |dRGB| of those frames against the CPU oracle.

Contract (driver):  python bench.py --gpus N --steps K --warmup W      (N > 1: launched by torch.distributed.run, one rank
per GPU, RCCL over xGMI; run WITHOUT a launcher -- WORLD_SIZE unset -- it starts the N ranks itself the same way).  A step = one pass of the generator hot path over one batch of synthetic frames:

  N = 1  BASELINE configs[1]: reenact_avatar_next3d single-seed 512^2 render, nrr = 128, ONE frame per synthesis call.
  N > 1  BASELINE configs[3]: batched reenactment, 8 frames per rank per step (B = 8 N; B = 64 at N = 8), sharded by
         frame_parallel (contiguous blocks, batch-global `dist` computed on every rank from the full camera batch) and
         collected with ONE all-gather per step.  Per-GPU work is fixed as N grows: "scaling": "weak".  The single-GPU
         rate of the same per-rank work (8 frames per call) is printed at N = 1 as `batch8`.

Inputs are resident in HBM before the timed region; EXACTLY K steps are timed between barrier + synchronize pairs, MAX
over ranks; rank 0 prints ONE JSON line.  Other workloads: --workload drive (BASELINE configs[4]: few-shot ConvGRU inversion of
8 source frames -> identity features, then 256 drive frames sharded over the ranks in calls of 8, SR head in its deployed fp16
precision; the per-rank call is a captured hipGraph).

Plumbing switches for the test-suite / one-GPU rehearsals of the N > 1 path (never used by the driver): --dist-backend gloo
(collectives over gloo instead of RCCL: several ranks can then share ONE GPU), --device cpu (the product's CPU formulation, eager),
--nrr (neural rendering resolution, 128 = BASELINE).

Extra legs on rank 0 at N == 1 (all outside the timed region):
  sustained    -- the same step repeated until the GPU has been busy for >= 2 s (an external sampler can see it)
  roofline     -- per-launch HIP-event timing of the dominant kernel family over extra eager frames
  cpu_baseline -- the CPU oracle (restatement of the reference's pure-PyTorch op path) timed on the host cores on a bounded
                  sample of the same workload; the SAME frames give `max_abs_rgb_vs_oracle` (second half of the metric)
  f32_mfma_only / sr_fp16 / drive_loop / batch8 / encoder / oneshot -- variants reported beside the headline, never as `value`
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC (what this driver supports): RCCL between the ranks of a node needs it

import torch  # noqa: E402

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from invertavatar_amd import frame_parallel, hipops, synthetic  # noqa: E402
from invertavatar_amd.torch_utils import custom_ops  # noqa: E402
from invertavatar_amd.training_avatar_texture.triplane_v20 import TriPlaneGenerator  # noqa: E402

custom_ops.verbosity = 'none'      # the reference's "Setting up PyTorch plugin ..." lines would share stdout with the JSON line

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md
PEAK_FP16_MFMA_TFLOPS = 2500.0  # dense fp16 MFMA
PEAK_HBM_GBS = 8000.0
NRR = 128
FRAMES_PER_RANK_SHARDED = 8     # configs[3]: B = 64 over 8 GPUs
PMC_FILE = next((p for p in (os.path.join(REPO, 'profiles', f'r{r:02d}_pmc_frame_hbm_traffic.json') for r in (6, 5, 4)) if os.path.exists(p)),
                os.path.join(REPO, 'profiles', 'r05_pmc_frame_hbm_traffic.json'))      # newest committed PMC collection
DEV = torch.device('cuda')      # set by setup_distributed
DIST = False                    # a process group is up (N > 1, or --force-dist at N = 1): steps end in the all-gather, timing brackets in barriers


def sync():
    if DEV.type == 'cuda':
        torch.cuda.synchronize()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--width', default='full', choices=['full', 'small'])
    ap.add_argument('--workload', default='reenact', choices=['reenact', 'drive'])
    ap.add_argument('--frames-per-rank', type=int, default=0, help='frames per rank per step (default: 1 at N = 1, 8 at N > 1)')
    ap.add_argument('--gather', default='f32', choices=['f32', 'u8'], help='dtype of the all-gathered frames (N > 1)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-extra', action='store_true', help='skip the variant legs (f32_mfma_only, sr_fp16, drive_loop, batch8, encoder)')
    ap.add_argument('--cpu-frames', type=int, default=3)
    ap.add_argument('--eager', action='store_true', help='issue every launch from Python instead of replaying a HIP graph')
    ap.add_argument('--dist-backend', default='nccl', choices=['nccl', 'gloo'], help='nccl = RCCL over xGMI (the product); gloo: rehearsal plumbing')
    ap.add_argument('--force-dist', action='store_true',
                    help='N = 1 only: bring up the process group anyway (RCCL at world size 1, communicator bound to the GPU) and end every step '
                         'in the same all_gather_into_tensor as N > 1 -- the collective path executed on a one-GPU box')
    ap.add_argument('--device', default='cuda', choices=['cuda', 'cpu'], help='cpu: CPU formulation of the API mirror (test-suite plumbing)')
    ap.add_argument('--nrr', type=int, default=128, help='neural rendering resolution (BASELINE: 128)')
    ap.add_argument('--features', default='encoder', choices=['encoder', 'backbone'],
                    help='--workload drive: identity features from the few-shot inversion (configs[4]) or straight from the backbones')
    ap.add_argument('--set', action='append', default=[], metavar='MODULE.ATTR=VALUE',
                    help='tuning A/Bs (tools/ab_frame.py): set a module-level switch of invertavatar_amd before the model is built')
    ap.add_argument('--drive-frames', type=int, default=256, help='--workload drive: length of the drive sequence (BASELINE: 256)')
    args = ap.parse_args()
    for item in getattr(args, 'set'):
        import ast
        import importlib
        target, value = item.split('=', 1)
        mod, attr = target.rsplit('.', 1)
        module = importlib.import_module('invertavatar_amd.' + mod)
        if not hasattr(module, attr):
            raise SystemExit(f'--set: invertavatar_amd.{mod} has no attribute {attr}')
        setattr(module, attr, ast.literal_eval(value))
    return args


def setup_distributed(args):
    global DEV, NRR, DIST
    NRR = args.nrr
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if args.device == 'cuda':
        idx = local % torch.cuda.device_count()        # (gloo rehearsal: more ranks than GPUs share them)
        torch.cuda.set_device(idx)
        DEV = torch.device('cuda', idx)
    else:
        DEV = torch.device('cpu')
    if args.gpus > 1 or world > 1 or args.force_dist:
        DIST = True
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        kw = {'device_id': DEV} if args.dist_backend == 'nccl' and DEV.type == 'cuda' else {}      # bind the communicator to this rank's GPU at once
        torch.distributed.init_process_group(args.dist_backend, rank=rank, world_size=world, **kw)   # nccl == RCCL on ROCm
    return rank, world, local


def all_gather_frames(gathered, part):
    """ONE collective per step: every rank's [per, ...] block into the [world * per, ...] batch."""
    if torch.distributed.get_backend() == 'gloo':      # (gloo has no _allgather_base for every dtype / device: list form, same bytes)
        torch.distributed.all_gather(list(gathered.chunk(torch.distributed.get_world_size())), part)
    else:
        torch.distributed.all_gather_into_tensor(gathered, part)
    return gathered


class Workload:
    """Device-resident inputs of `n_sets` different steps; step k uses set k % n_sets.  A step's batch is B = per_rank * world
    consecutive frames of the 240-frame orbit; this rank renders frames [rank*per_rank, (rank+1)*per_rank) of it."""

    def __init__(self, gen, per_rank, rank, world, n_sets):
        self.per_rank, self.rank, self.world, self.n_sets = per_rank, rank, world, n_sets
        batch = per_rank * world
        self.ws = gen.mapping(synthetic.latent(0, 1).to(DEV), synthetic.conditioning_camera().to(DEV), truncation_psi=0.7, truncation_cutoff=14)
        self.cams, self.uvs, self.jits, self.dists, self.frames, self.frame_dists = [], [], [], [], [], []
        for s in range(n_sets):
            frames_all = [(s * batch + j) % 240 for j in range(batch)]
            mine = frames_all[rank * per_rank:(rank + 1) * per_rank]
            cams_all = synthetic.camera_labels(frames_all).to(DEV)
            self.frames.append(mine)
            self.cams.append(cams_all[rank * per_rank:(rank + 1) * per_rank].contiguous())
            self.uvs.append(synthetic.uv_conditions(mine).to(DEV))
            self.jits.append(synthetic.jitter(mine, NRR * NRR).squeeze(-1).to(DEV))
            self.dists.append(frame_parallel.global_ray_dist(cams_all))          # batch-global, from the FULL camera batch
            self.frame_dists.append(frame_parallel.per_frame_ray_dist(self.cams[-1]))   # (drive workload: one value per frame)

    def eager(self, gen, s):
        return gen.synthesis(self.ws.expand(self.per_rank, -1, -1), self.cams[s], {'uvcoords_image': self.uvs[s]}, neural_rendering_resolution=NRR, noise_mode='const',
                             evaluation=True, jitter=self.jits[s], ray_dist=self.dists[s] if self.world > 1 else None)

    def replay(self, graphed, s):
        return graphed(self.ws, self.cams[s], self.uvs[s], self.jits[s], self.dists[s] if self.world > 1 else None)


def make_step(gen, wl, graphed, gather_dtype):
    """Returns step(k): render this rank's frames of step k and (N > 1) all-gather the frames of the step."""
    from invertavatar_amd.output import to_uint8_hwc
    world, per = wl.world, wl.per_rank
    if DIST:
        shape = (world * per, 512, 512, 3) if gather_dtype == 'u8' else (world * per, 3, 512, 512)
        gathered = torch.empty(shape, device=DEV, dtype=torch.uint8 if gather_dtype == 'u8' else torch.float32)

    def step(k):
        s = k % wl.n_sets
        out = wl.replay(graphed, s) if graphed is not None else wl.eager(gen, s)
        img = out['image']
        if DIST:
            part = to_uint8_hwc(img) if gather_dtype == 'u8' else img.contiguous()
            return all_gather_frames(gathered, part)
        return img
    return step


def timed(step, steps, warmup, world):
    for k in range(warmup):
        step(k)
    if DIST:
        torch.distributed.barrier()
    sync()
    t0 = time.perf_counter()
    for k in range(steps):
        step(k)
    sync()
    if DIST:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if DIST:
        t = torch.tensor([dt], device=DEV, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
    return dt


def capture(gen, wl, batch, **kw):
    """hipGraph of one synthesis call at this batch size, validated against the eager call (same bits expected)."""
    from invertavatar_amd.graphed import GraphedSynthesis
    graphed = GraphedSynthesis(gen, batch=batch, neural_rendering_resolution=NRR, with_ray_dist=wl.world > 1, **kw)
    img_g = wl.replay(graphed, 0)['image'].clone()
    err = (img_g - wl.eager(gen, 0)['image']).abs().max().item()
    if not err <= 1e-5:
        raise RuntimeError(f'graph replay differs from eager by {err}')
    return graphed, 'hipGraph replay (validated against eager: max |d| = %.1e)' % err


def pmc_traffic(family):
    """HBM-side bytes per logical launch of the dominant kernel family from the committed PMC collection (rocprofv3 --pmc FETCH_SIZE /
    --pmc WRITE_SIZE, separate passes over this same command with --eager; tools/profile_round.sh + tools/pmc_traffic_summary.py, which
    applies the guide's gfx950 FETCH_SIZE correction and sums over the SAME launches as `algorithmic_bytes_per_launch`).  Counters
    cannot be read from inside this process, so the figure comes from profiles/ (file named per round); it is used only when the file
    was measured on THIS tree's kernels (digest of csrc/ + include/).  Returns (bytes or None, note)."""
    if family != 'conv2d_mfma' or not os.path.exists(PMC_FILE):
        return None, 'no PMC collection for this family under profiles/'
    try:
        from invertavatar_amd import build
        with open(PMC_FILE) as fh:
            summ = json.load(fh)['_summary']
        if summ.get('csrc_digest') != build.source_digest():
            return None, f'{os.path.basename(PMC_FILE)} was measured on other kernel sources (digest {summ.get("csrc_digest")}): not used'
        fam = summ['fp16_pair_family']
        return int(fam['traffic_bytes_per_logical_launch']), (
            f'{os.path.basename(PMC_FILE)}: (2 x FETCH_SIZE + WRITE_SIZE) of conv_split_kernel / up_rows_kernel + their fix-ups / {fam["logical_launches_per_frame"]} '
            f'launches per frame; uncorrected {fam["traffic_bytes_per_logical_launch_uncorrected"]}')
    except (KeyError, ValueError, OSError) as exc:
        return None, f'unreadable PMC summary: {exc}'


def _profiled_frames(gen, wl, frames, single_stream):
    from invertavatar_amd.training_avatar_texture import triplane_v20
    saved, triplane_v20.SINGLE_STREAM = triplane_v20.SINGLE_STREAM, single_stream
    hipops.PROFILE = []
    try:
        for k in range(frames):
            wl.eager(gen, (7 + k) % wl.n_sets)
        torch.cuda.synchronize()
        return hipops.PROFILE
    finally:
        hipops.PROFILE = None
        triplane_v20.SINGLE_STREAM = saved


def roofline_leg(gen, wl, frames=3):
    """Per-launch HIP-event timing of every fused stage over `frames` extra eager frames; dominant family by total time.
    The kernels are timed with the frame's launches in program order on ONE stream (triplane_v20.SINGLE_STREAM), i.e. without
    neighbours from the other streams of a frame stretching them -- the condition rocprofv3 imposes on the committed kernel
    stats.  For the fp16-pair convolution family two fractions are printed: `frac_algorithmic` = algorithmic
    fp32 FLOPs / time / peak of the pipe the kernel occupies, and `mfma_util` = 3 x that (three fp16 products per fp32
    product are executed); `frac` = mfma_util, the utilisation of the matrix pipe."""
    _profiled_frames(gen, wl, 1, True)                     # (the first single-stream frame allocates that path's buffers)
    recs = _profiled_frames(gen, wl, frames, True)
    fam = {}
    # conv launches whose products are fp16 hi/lo pairs (3 MFMAs per k-step).  `split`: the layers from 32^2 up -- the 25 launches per
    # frame the figure has been quoted on since r01; `split_all`: those plus the 8^2 / 16^2 layers that joined the fp16-pair tiles in r03
    # (stream-K fragments of a tile, latency-bound: they are in the PMC family, whose kernels cannot be told apart by layer).
    split = dict(ms=0.0, flops=0.0, bytes=0.0, launches=0)
    split_all = dict(ms=0.0, flops=0.0, bytes=0.0, launches=0)
    for name, flops, nbytes, e0, e1, desc in recs:
        key = 'conv2d_mfma' if name.startswith('conv2d_mfma') else name
        f = fam.setdefault(key, dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
        ms = e0.elapsed_time(e1)
        f['ms'] += ms; f['flops'] += flops; f['bytes'] += nbytes; f['launches'] += 1
        if key == 'conv2d_mfma' and 'f16x3' in desc:
            split_all['ms'] += ms; split_all['flops'] += flops; split_all['bytes'] += nbytes; split_all['launches'] += 1
            hw = [t for t in desc.split() if 'x' in t and t.replace('x', '').isdigit()]
            if hw and min(int(v) for v in hw[0].split('x')) >= 32:
                split['ms'] += ms; split['flops'] += flops; split['bytes'] += nbytes; split['launches'] += 1
    dom = max(fam, key=lambda k: fam[k]['ms'])
    d = fam[dom]
    achieved = d['flops'] / (d['ms'] * 1e-3) / 1e12
    if dom == 'conv2d_mfma' and split['ms'] > 0.5 * d['ms']:
        alg = split['flops'] / (split['ms'] * 1e-3) / 1e12
        traffic, traffic_note = pmc_traffic(dom)
        alg_all = round(split_all['bytes'] / split_all['launches'])      # the launch set `traffic` is counted over
        out = dict(bound='mfma', kernel='conv2d_mfma (3x3 layers >= 32^2: fp32 products from fp16 hi/lo pairs, 3 x v_mfma_f32_32x32x16_f16)',
                   achieved=round(3 * alg, 2), peak=PEAK_FP16_MFMA_TFLOPS, unit='TFLOP/s', frac=round(3 * alg / PEAK_FP16_MFMA_TFLOPS, 4),
                   frac_algorithmic=round(alg / PEAK_FP16_MFMA_TFLOPS, 4), mfma_util=round(3 * alg / PEAK_FP16_MFMA_TFLOPS, 4),
                   algorithmic_f32_tflops=round(alg, 2), traffic=traffic, traffic_source=traffic_note,
                   traffic_launch_set='all_fp16_pair_launches (PMC kernel names do not separate layers)',
                   traffic_algorithmic_bytes_per_launch=alg_all,
                   traffic_over_algorithmic=None if traffic is None else round(traffic / alg_all, 3),
                   algorithmic_bytes_per_launch=round(split['bytes'] / split['launches']),
                   launches_per_frame=split['launches'] // frames, avg_launch_us=round(split['ms'] * 1e3 / split['launches'], 2),
                   algorithmic_gflop_per_frame=round(split['flops'] / frames / 1e9, 1),
                   all_fp16_pair_launches=dict(
                       note='with the 8^2 / 16^2 layers (r03): the launch set of `traffic` (PMC kernel names do not separate layers)',
                       launches_per_frame=split_all['launches'] // frames, avg_launch_us=round(split_all['ms'] * 1e3 / split_all['launches'], 2),
                       mfma_util=round(3 * split_all['flops'] / (split_all['ms'] * 1e-3) / 1e12 / PEAK_FP16_MFMA_TFLOPS, 4),
                       algorithmic_gflop_per_frame=round(split_all['flops'] / frames / 1e9, 1),
                       algorithmic_bytes_per_launch=round(split_all['bytes'] / split_all['launches'])),
                   whole_conv_family=dict(algorithmic_f32_tflops=round(achieved, 2), launches_per_frame=d['launches'] // frames,
                                          ms_per_frame=round(d['ms'] / frames, 3),
                                          frac_of_f32_mfma_peak=round(achieved / PEAK_FP32_MFMA_TFLOPS, 4)))
    else:
        out = dict(bound='mfma', kernel=dom, achieved=round(achieved, 2), peak=PEAK_FP32_MFMA_TFLOPS, unit='TFLOP/s',
                   frac=round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), traffic=pmc_traffic(dom)[0], traffic_source=pmc_traffic(dom)[1], launches_per_frame=d['launches'] // frames,
                   avg_launch_us=round(d['ms'] * 1e3 / d['launches'], 2), algorithmic_gflop_per_frame=round(d['flops'] / frames / 1e9, 1))
    others = {}
    for k, f in fam.items():
        others[k] = dict(ms_per_frame=round(f['ms'] / frames, 4), tflops=round(f['flops'] / (f['ms'] * 1e-3) / 1e12, 2),
                         algorithmic_gbs=round(f['bytes'] / (f['ms'] * 1e-3) / 1e9, 1), launches_per_frame=f['launches'] // frames)
    if 'render_rays' in fam:   # 13.09 GFLOP, 27.7 MB per frame; the FLOP fraction is the binding one
        r = fam['render_rays']
        others['render_rays']['frac_fp32_peak'] = round(r['flops'] / (r['ms'] * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)
        others['render_rays']['frac_hbm_peak'] = round(r['bytes'] / (r['ms'] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)
        others['render_rays']['avg_launch_us'] = round(r['ms'] * 1e3 / r['launches'], 1)
        others['render_rays']['note'] = ('r06: the decoder layers form their fp32 products from fp16 hi / lo pairs on v_mfma_f32_16x16x32_f16 (three products '
                                         'per k-step), so frac_fp32_peak = algorithmic fp32 FLOPs / time / the FP32 peak is a statement about the work, not '
                                         'about the fp32 pipe; on the fp16 pipe the same launch is 3 x that / 2500 TF')
        others['render_rays']['frac_f16_pipe'] = round(3 * r['flops'] / (r['ms'] * 1e-3) / 1e12 / PEAK_FP16_MFMA_TFLOPS, 4)
    out['note'] = ('`frac` / `kernels.*` are ONE-STREAM per-launch times (the frame\'s launches in program order on one stream, the condition '
                   'rocprofv3 imposes on the committed kernel stats): an upper bound on each kernel\'s share, not a decomposition of the timed '
                   'step, which replays four concurrent branches; `frame_mfma_util` (added by main) is the whole timed step')
    return out, others


def host_physical_cores():
    """Physical cores this process may run on: distinct (package, core) pairs of the CPUs in the affinity mask, capped by the
    cgroup CPU quota.  SMT siblings and CPUs outside the mask / quota only oversubscribe the torch-CPU path (r03: 256 logical
    CPUs = 147 s per frame)."""
    cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else list(range(os.cpu_count() or 1))
    cores = set()
    for c in cpus:
        try:
            base = f'/sys/devices/system/cpu/cpu{c}/topology/'
            with open(base + 'physical_package_id') as f, open(base + 'core_id') as g:
                cores.add((f.read().strip(), g.read().strip()))
        except OSError:
            cores.add(('cpu', str(c)))
    n = max(1, len(cores))
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            quota, period = f.read().split()
        if quota != 'max':
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline_leg(gen, wl, frames):
    """The oracle (test infrastructure: CPU restatement of the reference's torch op path) on the host cores, on a bounded
    sample of the workload; the same frames rendered by the device path give max |dRGB| (BASELINE metric, second half).
    Beside it: the product's own CPU formulation (API mirror on CPU tensors = the route the reference itself takes without a
    GPU: conv2d_resample / _ref ops), which BASELINE.md timed at 0.34 frames/s on 8 cores of the survey container."""
    from oracle import generator as OG
    sd = {k: v.detach().cpu() for k, v in gen.state_dict().items()}
    ws_c = wl.ws.cpu()
    cores = host_physical_cores()                # all physical cores the process may run on, stated in the line
    sets = [(i + 1) % wl.n_sets for i in range(frames)]

    def oracle(s):
        return OG.synthesis(sd, ws_c, wl.cams[s].cpu(), wl.uvs[s].cpu(), wl.jits[s].cpu().unsqueeze(-1), nrr=NRR)

    def oracle_rate(n_thr, n_frames):
        """(frames/s, results) at n_thr threads: one warm-up frame, then n_frames timed; a warm-up frame slower than 12 s ends the
        point there (the torch-CPU path stops scaling beyond ~32 threads: the bounded sample must stay bounded)."""
        torch.set_num_threads(n_thr)
        t0 = time.perf_counter()
        oracle(0)
        warm = time.perf_counter() - t0
        if warm > 12.0:
            return 1.0 / warm, None
        t0 = time.perf_counter()
        refs = [oracle(s) for s in sets[:n_frames]]
        return n_frames / (time.perf_counter() - t0), refs
    rate, refs = oracle_rate(cores, frames)
    by_threads = {str(cores): round(rate, 4)}
    for n_thr in sorted({8, 32} - {cores}):
        if n_thr < cores:
            r_n, refs_n = oracle_rate(n_thr, 1 if refs is not None else frames)
            by_threads[str(n_thr)] = round(r_n, 4)
            refs = refs if refs is not None else refs_n
    torch.set_num_threads(cores)
    if refs is None:       # every point's warm-up frame was over the bound: the rate is the warm-up's; parity still gets ONE reference frame
        refs = [oracle(sets[0])]
    err_rgb = err_raw = 0.0
    for s, ref in zip(sets, refs):
        out = wl.eager(gen, s)
        err_rgb = max(err_rgb, (out['image'].cpu() - ref['image']).abs().max().item())
        err_raw = max(err_raw, (out['image_raw'].cpu() - ref['image_raw']).abs().max().item())
    base = dict(value=round(rate, 4), unit='frames/s', cores=cores, kind='port',
                sample=f'{len(refs)} frames of the same workload (B=1, nrr={NRR}, 512^2 out, fp32) after 1 warm-up frame',
                frames_per_s_by_threads=by_threads, logical_cores_of_this_box=os.cpu_count(),
                note='cores = physical cores this process may run on; oracle/ restates the reference op by op for checking, not for '
                     'speed; see product_cpu_route for the reference\'s own CPU route')
    try:   # the product's CPU formulation = the reference's pure-PyTorch route (not the measured product path)
        gen_c = TriPlaneGenerator(**gen.init_kwargs).eval().requires_grad_(False)
        gen_c.load_state_dict(sd)
        call = lambda s: gen_c.synthesis(ws_c, wl.cams[s].cpu(), {'uvcoords_image': wl.uvs[s].cpu()}, neural_rendering_resolution=NRR,   # noqa: E731
                                         noise_mode='const', evaluation=True, jitter=wl.jits[s].cpu())
        by_threads = {}
        for n_thr in sorted({8, 32, cores}):     # n = 8 is the survey container's core count (0.34 frames/s there)
            if n_thr > cores:
                continue
            warm, n_timed = (3, 10) if n_thr == cores else (1, 2)      # 3 warm + 10 timed at all physical cores
            torch.set_num_threads(n_thr)
            for k in range(warm):
                call(sets[k % len(sets)])
            t0 = time.perf_counter()
            for k in range(n_timed):
                call(sets[k % len(sets)])
            by_threads[str(n_thr)] = round(n_timed / (time.perf_counter() - t0), 4)
        torch.set_num_threads(cores)
        base['product_cpu_route'] = dict(value=by_threads[str(cores)], unit='frames/s', cores=cores,
                                         sample='10 frames after 3 warm-up frames at `cores` threads (SURVEY 8d); 2 after 1 at the other points',
                                         survey_container_8_cores=0.34, frames_per_s_by_threads=by_threads)
    except Exception as exc:   # noqa: BLE001
        base['product_cpu_route'] = f'failed: {exc}'
    return base, dict(max_abs_rgb_vs_oracle=float(f'{err_rgb:.3e}'), max_abs_raw_rgb_vs_oracle=float(f'{err_raw:.3e}'),
                      frames_compared=len(refs), tolerance=1e-3)


def variant_leg(gen, wl, args, batch=1, **flags):
    """The headline step under a module-level switch (captured graph, same timing loop)."""
    from invertavatar_amd.training import networks_stylegan2 as sg2
    saved = {k: getattr(sg2, k) for k in flags}
    for k, v in flags.items():
        setattr(sg2, k, v)
    try:
        graphed, _ = capture(gen, wl, batch)
        img = wl.replay(graphed, 0)['image'].clone()
        step = make_step(gen, wl, graphed, 'f32')
        dt = timed(step, args.steps, args.warmup, 1)
    finally:
        for k, v in saved.items():
            setattr(sg2, k, v)
    return dict(value=round(batch * args.steps / dt, 3), unit='frames/s', ms_per_step=round(dt / args.steps * 1e3, 3)), img


def drive_loop_leg(gen, wl, args):
    """The drive loop of eval_seq.py:212 / BASELINE configs[2]: the texture and static features of the identity are computed
    once (inversion result) and every drive frame is `synthesis_withTexture` = rasterize + face backbone + renderer + SR."""
    from invertavatar_amd.graphed import GraphedDrive
    ws = wl.ws
    tex = gen.texture_backbone.synthesis(ws, cond_list=None, return_list=True, noise_mode='const')
    sta = gen.backbone.synthesis(ws, cond_list=None, return_list=True, noise_mode='const')
    graphed = GraphedDrive(gen, ws, tex, sta, batch=1, neural_rendering_resolution=NRR)
    err = (graphed(wl.cams[0], wl.uvs[0], wl.jits[0])['image'] - wl.eager(gen, 0)['image']).abs().max().item()
    dt = timed(lambda k: graphed(wl.cams[k % wl.n_sets], wl.uvs[k % wl.n_sets], wl.jits[k % wl.n_sets])['image'], args.steps, args.warmup, 1)
    return dict(value=round(args.steps / dt, 3), unit='frames/s', ms_per_step=round(dt / args.steps * 1e3, 3),
                workload='synthesis_withTexture per drive frame, texture + static backbone features cached (eval_seq.py:212)',
                max_abs_rgb_vs_full_synthesis=float(f'{err:.3e}'))


def extra_legs(result, gen, wl, args):
    """Variants reported beside the headline (never as `value`); a failing variant is recorded, it does not lose the headline."""
    def guarded(name, fn):
        try:
            result[name] = fn()
        except Exception as exc:   # noqa: BLE001
            result[name] = f'failed: {type(exc).__name__}: {exc}'
    headline_img = wl.eager(gen, 0)['image'].clone()

    def f32_only():
        r, img = variant_leg(gen, wl, args, SPLIT_FP16_PRODUCTS=False)
        r['max_abs_rgb_vs_headline_run'] = float(f'{(img - headline_img).abs().max().item():.3e}')
        return r

    def sr_fp16():
        gen16 = sr_fp16_generator(gen, args.width)
        r, img = variant_leg(gen16, wl, args, FP16_BLOCKS_COMPUTE_FP32=False)
        r.update(dtype='f32 backbones + renderer, SR head: fp16 operands and fp16 activation storage / f32 accumulate',
                 max_abs_rgb_vs_f32_run=float(f'{(img - headline_img).abs().max().item():.3e}'),
                 note='the reference\'s own fp16 SR path (fp16 storage + conv_clamp 256, networks_stylegan2.py:417-437), restated in oracle/'
                      ' (the reference forces fp32 on CPU, so it is unpinned), is 3.4e-3 from its fp32 output: the 1e-3 RGB bound of'
                      ' BASELINE is a statement about the fp32 head (the headline), not about this mode')
        return r

    def batch8():
        wl8 = Workload(gen, FRAMES_PER_RANK_SHARDED, 0, 1, n_sets=2)
        r, _ = variant_leg(gen, wl8, args, batch=FRAMES_PER_RANK_SHARDED)
        r['workload'] = 'one rank\'s share of BASELINE configs[3]: 8 frames per synthesis call on one GPU'
        return r

    def encoder():
        from invertavatar_amd.encoder_bench import encoder_leg
        return encoder_leg(gen)
    guarded('f32_mfma_only', f32_only)
    guarded('sr_fp16', sr_fp16)
    guarded('drive_loop', lambda: drive_loop_leg(gen, wl, args))
    guarded('batch8', batch8)
    guarded('encoder', encoder)

    def oneshot():
        from invertavatar_amd.encoder_bench import oneshot_leg
        return oneshot_leg(gen)
    guarded('oneshot', oneshot)


def sr_fp16_generator(gen, width):
    gen16 = TriPlaneGenerator(**synthetic.generator_kwargs(width, sr_num_fp16_res=4)).eval().requires_grad_(False)
    gen16.load_state_dict(gen.state_dict())
    return gen16.to(DEV)


def inversion_features(gen, n_sources=8, repeats=2, rank=0, world=1):
    """BASELINE configs[4], first half: few-shot ConvGRU inversion of 8 source frames through the script's own flow
    (invertavatar_amd.eval_seq.few_shot_inversion: encode + 2 interleaved AR_eval_forward groups, module modes of eval_seq.py:91-97).
    N = 1: every step on the one GPU.  N > 1: inversion_parallel.few_shot_inversion_sharded -- the source renders sharded by frame, the
    texture / tri-plane UNet chains on ranks 0 / 1, one all-gather + one broadcast per chain owner.
    Returns (ws, results, milliseconds of the last run: max over ranks)."""
    from invertavatar_amd import eval_seq
    from invertavatar_amd.encoder_inversion.models.uvnet import inversionNet
    net = inversionNet(generator=gen, encoding_triplane=True, encoding_texture=True).requires_grad_(False)
    synthetic.fill_encoder_parameters(net)
    net = net.to(DEV)
    eval_seq.set_eval_seq_modes(net)
    gen.neural_rendering_resolution = NRR
    src_frames = [int(round(k * 32 / n_sources)) for k in range(n_sources)]
    images = torch.cat([synthetic.source_frames(7 + k // 4, 4)[k % 4:k % 4 + 1] for k in range(n_sources)]).to(DEV)
    uvs = synthetic.source_uv(17, src_frames).to(DEV)
    cams, uvc = synthetic.camera_labels(src_frames).to(DEV), synthetic.uv_conditions(src_frames).to(DEV)
    cache = {} if DEV.type == 'cuda' else None          # captured e4e encode (eval_seq.GraphedEncode)
    for _ in range(repeats + (1 if cache is not None else 0)):      # (the first runs pay allocations, kernel selection, the capture)
        sync()
        t0 = time.perf_counter()
        if world > 1:
            from invertavatar_amd import inversion_parallel
            torch.distributed.barrier()
            ws, res, _ = inversion_parallel.few_shot_inversion_sharded(net, images, uvs, cams, uvc, rank=rank, world_size=world,
                                                                       draws=inversion_parallel.seeded_draws(0, NRR * NRR))
        else:
            ws, res, _ = eval_seq.few_shot_inversion(net, images, uvs, cams, uvc, graphed=cache)
        sync()
        ms = (time.perf_counter() - t0) * 1e3
    if world > 1:
        worst = torch.tensor([ms], dtype=torch.float64, device=DEV)
        torch.distributed.all_reduce(worst, op=torch.distributed.ReduceOp.MAX)
        ms = float(worst.item())
    return ws, res, ms


def drive_main(args, rank, world):
    """BASELINE configs[4]: few-shot ConvGRU incremental inversion of 8 source frames -> `--drive-frames` (256) drive frames, SR head
    in fp16.  The inversion runs once per identity on every rank (reported as `inversion_ms`, outside the timed steps); the drive
    sequence is sharded over the ranks in calls of `per` = 8 frames: a step = one call on every rank (8 N frames) + ONE all-gather.
    Every drive frame keeps the depth range of the script's one-frame call (per-frame `ray_dist`, eval_seq.py:206-212), so sharding
    and batching do not change the frames.  The per-rank call is a captured hipGraph (validated against the eager call)."""
    from invertavatar_amd.graphed import GraphedDrive
    from invertavatar_amd.training import networks_stylegan2 as sg2
    gen = TriPlaneGenerator(**synthetic.generator_kwargs(args.width, sr_num_fp16_res=4)).eval().requires_grad_(False)
    synthetic.fill_parameters(gen)
    gen = gen.to(DEV)
    sg2.FP16_BLOCKS_COMPUTE_FP32 = False
    per = args.frames_per_rank or FRAMES_PER_RANK_SHARDED
    features = args.features if args.width == 'full' else 'backbone'      # (the UNet heads are sized for the full-width pyramid)
    with torch.no_grad():
        wl = Workload(gen, per, rank, world, n_sets=max(1, args.drive_frames // (per * world)))
        inversion_ms = None
        if features == 'encoder':
            ws, res, inversion_ms = inversion_features(gen, rank=rank, world=world)
            tex, sta = res['texture'], res['static']
        else:
            ws = wl.ws
            tex = gen.texture_backbone.synthesis(ws, cond_list=None, return_list=True, noise_mode='const')
            sta = gen.backbone.synthesis(ws, cond_list=None, return_list=True, noise_mode='const')
        ws8 = ws.expand(per, -1, -1).contiguous()
        tex8 = [t.expand(per, -1, -1, -1).contiguous() for t in tex]
        sta8 = [t.expand(per, -1, -1, -1).contiguous() for t in sta]
        gathered = torch.empty(world * per, 3, 512, 512, device=DEV) if DIST else None

        def eager(s):
            return gen.synthesis_withTexture(ws8, tex8, wl.cams[s], {'uvcoords_image': wl.uvs[s]}, static_feats=sta8, neural_rendering_resolution=NRR,
                                             noise_mode='const', evaluation=True, jitter=wl.jits[s],
                                             ray_dist=wl.frame_dists[s] if per > 1 else None)['image']
        graphed, launch = None, 'eager'
        if DEV.type == 'cuda' and not args.eager:
            try:
                graphed = GraphedDrive(gen, ws, tex, sta, batch=per, neural_rendering_resolution=NRR, ray_dist_elems=per if per > 1 else 0)
                img_g = graphed(wl.cams[0], wl.uvs[0], wl.jits[0], wl.frame_dists[0] if per > 1 else None)['image'].clone()
                err = (img_g - eager(0)).abs().max().item()
                if not err <= 1e-5:
                    raise RuntimeError(f'graph replay differs from eager by {err}')
                launch = 'hipGraph replay (validated against eager: max |d| = %.1e)' % err
            except Exception as exc:   # noqa: BLE001
                graphed, launch = None, f'eager (graph capture unavailable: {exc})'

        def step(k):
            s = k % wl.n_sets
            img = graphed(wl.cams[s], wl.uvs[s], wl.jits[s], wl.frame_dists[s] if per > 1 else None)['image'] if graphed is not None else eager(s)
            if DIST:
                return all_gather_frames(gathered, img.contiguous())
            return img
        dt = timed(step, args.steps, args.warmup, world)
    fps = world * per * args.steps / dt
    out = {'metric': 'frames/sec (512^2 out, 128^2 neural render)', 'value': round(fps, 3), 'unit': 'frames/s',
           'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
           'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
           'dtype': 'f32 backbones + renderer; SR head fp16 operands and fp16 activation storage / f32 accumulate', 'data': 'synthetic',
           'config': {'workload': 'BASELINE configs[4]: few-shot ConvGRU inversion (8 sources, eval_seq.py flow) -> drive loop '
                                  f'synthesis_withTexture, {per} frames per rank per call, SR head fp16',
                      'width': args.width, 'nrr': NRR, 'frames_per_rank_per_step': per, 'global_batch': per * world,
                      'identity_features': features, 'parallelism': f'frame-sharded dp{world}', 'launch': launch,
                      'collective': f'one all_gather of the step\'s [{per * world},3,512,512] f32 frames' if DIST else 'none'}}
    if inversion_ms is not None:
        n = args.drive_frames
        out['inversion_ms'] = round(inversion_ms, 2)
        out['clip'] = dict(frames=n, seconds=round(inversion_ms * 1e-3 + n / fps, 4), frames_per_s=round(n / (inversion_ms * 1e-3 + n / fps), 2),
                           note='inversion (N > 1: source renders and UNet trunks sharded by frame, the recurrent decoder chains on ranks 0 / 1) + the whole drive sequence at the measured rate')
    return out


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): start the N ranks ourselves through torch.distributed.run,
    one per GPU, rendezvous on 127.0.0.1 at a free port, and pass the command line through unchanged.  The ranks' stdout is ours, so
    rank 0's ONE JSON line is this command's JSON line; `n_gpus` in it is the world size the ranks saw."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC: RCCL across processes needs it on this driver
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args))
    rank, world, _ = setup_distributed(args)
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but the launcher started {world} rank(s)')
    torch.backends.cudnn.benchmark = False
    if args.workload == 'drive':
        result = drive_main(args, rank, world)
        if rank == 0:
            print(json.dumps(result), flush=True)
        if DIST:
            torch.distributed.destroy_process_group()
        return
    gen = TriPlaneGenerator(**synthetic.generator_kwargs(args.width)).eval().requires_grad_(False)
    synthetic.fill_parameters(gen)
    gen = gen.to(DEV)
    per = args.frames_per_rank or (1 if world == 1 else FRAMES_PER_RANK_SHARDED)
    with torch.no_grad():
        wl = Workload(gen, per, rank, world, n_sets=16 if per == 1 else 4)
        graphed, launch_mode = None, 'eager'
        if not args.eager and DEV.type == 'cuda':
            try:
                graphed, launch_mode = capture(gen, wl, per)
            except Exception as exc:   # noqa: BLE001  fall back to eager launches, and say so in the JSON line
                graphed, launch_mode = None, f'eager (graph capture unavailable: {exc})'
        step = make_step(gen, wl, graphed, args.gather)
        dt = timed(step, args.steps, args.warmup, world)
        cfg = 'configs[1]: reenact_avatar_next3d single-seed render, 1 frame per synthesis call' if per == 1 and world == 1 else \
              f'configs[3]: batched reenactment, batch {per * world} = {per} frames per rank per step, frame-sharded'
        result = {
            'metric': 'frames/sec (512^2 out, 128^2 neural render); max |dRGB|', 'value': round(world * per * args.steps / dt, 3),
            'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32 (3x3 convolutions >= 8^2 and the two layers of the ray decoder form their f32 products from fp16 hi/lo pairs on the f16 '
                     'MFMA -- three products per term, lo x lo ~ 2^-22 dropped -- with f32 accumulation; all other arithmetic f32)', 'data': 'synthetic',
            'config': {'workload': f'TriPlaneGenerator.synthesis, BASELINE {cfg}: 512^2 out, neural_rendering_resolution={NRR}, all three '
                                   'backbones + rasterize + fused renderer + SR 8XDC recomputed every frame',
                       'width': args.width, 'frames_per_rank_per_step': per, 'global_batch': per * world,
                       'parallelism': f'frame-sharded dp{world}',
                       'collective': f'one all_gather of the step\'s [{per * world},3,512,512] frames as {args.gather}' if DIST else 'none',
                       'launch': launch_mode},
        }
        if args.dist_backend != 'nccl' or args.device != 'cuda':
            result['config']['rehearsal'] = f'backend {args.dist_backend}, device {args.device}: plumbing run, not a measurement'
        if DIST and world == 1:
            result['config']['force_dist'] = f'process group up at world size 1 (backend {torch.distributed.get_backend()}): every timed step ends in all_gather_into_tensor'
        if rank == 0 and world == 1 and per == 1 and DEV.type == 'cuda' and not DIST:
            # keep the GPU busy for >= 2 s with the same step (the timed K steps alone can be shorter than a sampler's period)
            n, t_s = 0, time.perf_counter()
            while time.perf_counter() - t_s < 2.0:
                for k in range(50):
                    step(n + k)
                torch.cuda.synchronize()
                n += 50
            el = time.perf_counter() - t_s
            result['sustained'] = dict(value=round(n / el, 3), unit='frames/s', steps=n, seconds=round(el, 2))
            result['split_range_watch'] = dict(clamped=bool(hipops.split_saturation_poll(DEV)),
                                               note='always-on device flag of the fp16 hi / lo split (ia_split_saturation_poll), over every frame so far')
            if not args.no_extra:
                extra_legs(result, gen, wl, args)
            if not args.no_roofline:
                result['roofline'], result['kernels'] = roofline_leg(gen, wl)
                rf = result['roofline']
                if 'all_fp16_pair_launches' in rf:
                    # the whole timed step: 3 x the fp16-pair family's algorithmic FLOPs of a frame / ms_per_step / the fp16 pipe's peak
                    a = rf['all_fp16_pair_launches']
                    gf = a.get('algorithmic_gflop_per_frame')
                    if gf:
                        rf['frame_mfma_util'] = round(3 * gf * 1e9 / (result['ms_per_step'] * 1e-3) / 1e12 / PEAK_FP16_MFMA_TFLOPS, 4)
                    rf['frac_all_fp16_pair_launches'] = a['mfma_util']
            if not args.no_cpu_baseline:
                try:
                    result['cpu_baseline'], parity = cpu_baseline_leg(gen, wl, args.cpu_frames)
                    result.update(parity)
                except Exception as exc:   # noqa: BLE001  a failure of the reported baseline must not lose the measured line
                    result['cpu_baseline'] = dict(value=None, unit='frames/s', cores=host_physical_cores(), kind='port', sample=f'failed: {exc!r}')
    if rank == 0:
        print(json.dumps(result), flush=True)
    if DIST:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
