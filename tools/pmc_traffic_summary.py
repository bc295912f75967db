"""profiles/<tag>_pmc_summary.json (tools/profile_round.sh) -> profiles/<tag>_pmc_frame_hbm_traffic.json: HBM-side bytes per frame of the
convolution kernels from the FETCH_SIZE / WRITE_SIZE passes (KB, summed over dispatches), corrected as MI355X_MICROARCH.md (HBM section)
prescribes, and the figure bench.py prints as roofline.traffic: bytes per logical launch of the fp16-pair family -- the SAME launches
`roofline.algorithmic_bytes_per_launch` averages over (conv_split_kernel + the fix-up launches that finish its stream-K tiles).

    python tools/pmc_traffic_summary.py <tag> <frames in the PMC pass> <fp16-pair launches per frame (roofline.launches_per_frame)>

The file is stamped with the digest of csrc/ + include/ (invertavatar_amd.build.source_digest): bench.py ignores it when the tree differs."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from invertavatar_amd import build  # noqa: E402

tag, frames, logical = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'profiles')
d = json.load(open(os.path.join(root, f'{tag}_pmc_summary.json')))


def fetch_factor(kernel):
    """MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE reports half of the bytes of a 16-byte-per-lane coalesced read (`global_load_dwordx4`
    and `buffer_load_dwordx4 ... lds` alike).  conv_split_kernel fetches everything that way (LDS-DMA); conv_fixup_kernel reads its slabs
    as float4.  The register-staged kernels read dwords (uncalibrated: factor 1, flagged)."""
    if any(n in kernel for n in ('conv_split_kernel', 'conv_small_kernel', 'conv_fixup_kernel', 'up_rows_kernel', 'up_edge_fixup_kernel')):
        return 2.0
    return 1.0


def in_pair_family(kernel):
    # conv_split_kernel with two operand planes (template argument NP = 2), and the fix-ups of its tile families (2 x 2 / 2 x 4 waves)
    if 'conv_split_kernel' in kernel:
        return 'conv_split_kernelILi2E' in kernel or 'conv_split_kernel<2,' in kernel
    if 'up_rows_kernel' in kernel or 'up_edge_fixup_kernel' in kernel:      # r04: the row-phase form of the large transposed layers
        return True
    if 'conv_small_kernel' in kernel:      # r06: the low-resolution layers (K split inside the workgroup), two operand planes
        return 'conv_small_kernelILi2E' in kernel or 'conv_small_kernel<2,' in kernel
    if 'conv_fixup_kernel' in kernel:
        t = kernel.split('<')[-1].split('>')[0].replace(' ', '').split(',')
        return len(t) >= 5 and t[3] == '2'
    return False


# bench.py's sustained loop is time-based, so the passes run different numbers of frames (the 9-counter SQ pass fewer): each counter is
# divided by the frames of ITS pass, counted on a once-per-frame kernel (profile_round.sh records dispatches per pass since r05)
once = next((e for k, e in d.items() if 'render_rays_kernel' in k), {})
frames_fetch, frames_write = once.get('dispatches_fetch', frames), once.get('dispatches_write', frames)
conv = {k: e for k, e in d.items() if any(s in k for s in ('conv_split_kernel', 'conv_small_kernel', 'conv_mfma_kernel', 'conv_fixup_kernel', 'up_rows_kernel', 'up_edge_fixup_kernel', 'conv1x1_kernel'))}
per_kernel, fam = {}, dict(fetch_raw=0.0, fetch_corrected=0.0, write=0.0, dispatches=0)
for k, e in conv.items():
    f_raw, w = e.get('FETCH_SIZE', 0.0) * 1e3 / frames_fetch, e.get('WRITE_SIZE', 0.0) * 1e3 / frames_write
    fac = fetch_factor(k)
    per_kernel[k.replace('_ZN12_GLOBAL__N_117', '')[:72]] = dict(
        dispatches_per_frame=round(e.get('dispatches_fetch', e['dispatches']) / frames_fetch, 1), fetch_raw_mb_per_frame=round(f_raw / 1e6, 1),
        fetch_corrected_mb_per_frame=round(f_raw * fac / 1e6, 1), fetch_factor=fac, write_mb_per_frame=round(w / 1e6, 1),
        fp16_pair_family=in_pair_family(k))
    if in_pair_family(k):
        fam['fetch_raw'] += f_raw; fam['fetch_corrected'] += f_raw * fac; fam['write'] += w; fam['dispatches'] += e.get('dispatches_fetch', e['dispatches']) / frames_fetch
out = dict(_summary=dict(
    how='rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (--kernel-trace only) over `bench.py --steps 2 --warmup 1 --eager '
        '--no-cpu-baseline --no-roofline --no-extra` (tools/profile_round.sh); counters are KB summed over dispatches',
    correction='FETCH_SIZE x 2 for the kernels whose reads are 16 bytes per lane (conv_split_kernel / up_rows_kernel: buffer_load_dwordx4 ... lds; '
               'conv_fixup_kernel: float4 slab reads), as MI355X_MICROARCH.md (HBM) prescribes for gfx950; WRITE_SIZE and dword reads are '
               'uncalibrated there and taken as reported; Infinity-Cache hits are counted by these counters, so this is fabric-side traffic',
    frames=frames_fetch, frames_of_the_write_pass=frames_write, csrc_digest=build.source_digest(),
    fp16_pair_family=dict(
        kernels='conv_split_kernel<NP = 2, ...> + conv_fixup_kernel of its tile families + up_rows_kernel / up_edge_fixup_kernel: the launches roofline.algorithmic_bytes_per_launch averages over',
        logical_launches_per_frame=logical, kernel_dispatches_per_frame=round(fam['dispatches'], 1),
        fetch_raw_gb_per_frame=round(fam['fetch_raw'] / 1e9, 3), fetch_corrected_gb_per_frame=round(fam['fetch_corrected'] / 1e9, 3),
        write_gb_per_frame=round(fam['write'] / 1e9, 3),
        traffic_bytes_per_logical_launch=round((fam['fetch_corrected'] + fam['write']) / logical),
        traffic_bytes_per_logical_launch_uncorrected=round((fam['fetch_raw'] + fam['write']) / logical))),
    per_kernel=per_kernel)
path = os.path.join(root, f'{tag}_pmc_frame_hbm_traffic.json')
json.dump(out, open(path, 'w'), indent=1)
print(path, json.dumps(out['_summary']['fp16_pair_family'], indent=1))
