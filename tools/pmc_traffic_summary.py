"""profiles/<tag>_pmc_summary.json (tools/profile_round.sh) -> profiles/<tag>_pmc_frame_hbm_traffic.json: HBM-side bytes per frame and
per logical convolution launch of the conv family (FETCH_SIZE + WRITE_SIZE, reported in KB by rocprofv3), the figure bench.py prints as
roofline.traffic.   python tools/pmc_traffic_summary.py r02 <frames in the PMC pass> <logical conv launches per frame>"""
import json
import os
import sys

tag, frames, logical = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'profiles')
d = json.load(open(os.path.join(root, f'{tag}_pmc_summary.json')))
fam = {k: e for k, e in d.items() if any(s in k for s in ('conv_split_kernel', 'conv_mfma_kernel', 'conv_fixup_kernel'))}
fetch = sum(e.get('FETCH_SIZE', 0) for e in fam.values()) * 1e3 / frames
write = sum(e.get('WRITE_SIZE', 0) for e in fam.values()) * 1e3 / frames
per_kernel = {k.replace('_ZN12_GLOBAL__N_117', '')[:60]: dict(dispatches_per_frame=round(e['dispatches'] / frames, 1),
                                                             fetch_mb_per_frame=round(e.get('FETCH_SIZE', 0) / frames / 1e3, 1),
                                                             write_mb_per_frame=round(e.get('WRITE_SIZE', 0) / frames / 1e3, 1)) for k, e in fam.items()}
out = dict(_summary=dict(
    how='rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 2 --warmup 1 --eager --no-cpu-baseline '
        '--no-roofline --no-extra` (tools/profile_round.sh); counters are KB summed over dispatches; raw values (the 2x gfx950 FETCH_SIZE '
        'correction of MI355X_MICROARCH.md applies to 16-byte-per-lane streaming reads, so the fetch figure is a lower bound)',
    frames=frames, logical_conv_launches_per_frame=logical,
    conv_family_per_frame_gb=dict(fetch=round(fetch / 1e9, 3), write=round(write / 1e9, 3)),
    conv_family_per_logical_launch_mb=round((fetch + write) / logical / 1e6, 1),
    reading='stream-K accumulator slabs (written by the convolution kernels of layers smaller than the machine, read back by '
            'conv_fixup_kernel) and the per-XCD re-fetch of weight slabs are the traffic above the algorithmic bytes'),
    per_kernel=per_kernel)
path = os.path.join(root, f'{tag}_pmc_frame_hbm_traffic.json')
json.dump(out, open(path, 'w'), indent=1)
print(path, out['_summary']['conv_family_per_frame_gb'], out['_summary']['conv_family_per_logical_launch_mb'])
