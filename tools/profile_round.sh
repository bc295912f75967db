#!/bin/bash
# Round profile on the GPU box (run through gpurun from the repo root):  tools/profile_round.sh r02
#   1. rocprofv3 --kernel-trace --stats of the default `python bench.py`      -> gpurun_out/<tag>_bench_kernel_stats.csv (+ the bench line)
#   2. PMC passes (one counter set per pass, --kernel-trace only, as MI355X_MICROARCH.md prescribes) over 3 eager frames:
#      FETCH_SIZE | WRITE_SIZE | SQ matrix/VALU/LDS activity            -> gpurun_out/<tag>_pmc_<set>.csv + <tag>_pmc_summary.json
tag=${1:-r02}
repo=$(pwd)
out=$repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*\|SQ_LDS_[A-Z_]*\|SQ_BUSY_CYCLES\|SQ_WAVE_CYCLES\|SQ_ACTIVE_INST_[A-Z]*\|GRBM_GUI_ACTIVE" | sort -u > $out/${tag}_counters_available.txt
rm -rf /tmp/prof_stats
[ -n "$PMC_ONLY" ] || rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o bench -- python $repo/bench.py > $out/${tag}_bench_under_rocprof.json 2> /tmp/prof_stats.err
[ -n "$PMC_ONLY" ] || cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $out/${tag}_bench_kernel_stats.csv 2>/dev/null || tail -5 /tmp/prof_stats.err
declare -A SETS=( [fetch]="FETCH_SIZE" [write]="WRITE_SIZE" [sq]="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE" )
for s in ${PMC_PASSES:-fetch write sq}; do
  rm -rf /tmp/pmc_$s
  timeout 400 rocprofv3 --pmc ${SETS[$s]} --kernel-trace --output-format csv -d /tmp/pmc_$s -o p -- \
      python $repo/bench.py --steps 2 --warmup 1 --eager --no-cpu-baseline --no-roofline --no-extra > /tmp/pmc_$s.log 2>&1
  f=$(find /tmp/pmc_$s -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ]; then echo "pmc $s failed"; tail -3 /tmp/pmc_$s.log; else cp $f /tmp/${tag}_pmc_$s.csv; fi     # (raw CSVs are tens of MB: only the summary travels)
done
python3 - "$out" "$tag" <<'PY'
import csv, json, collections, os, sys
out, tag = sys.argv[1], sys.argv[2]
def fam(name):
    n = name.replace('(anonymous namespace)::', '').replace('void ', '')
    base = n.split('<')[0].split('(')[0]
    if base == 'conv_mfma_kernel':
        return 'conv_mfma_kernel<HM=%s>' % n.split('>')[0].split(',')[-1].strip()
    if base == 'conv_fixup_kernel':      # keep the tile family (TR, FO, FP, WO, WP): WO = 2 are the fix-ups of the fp16-pair kernels
        return 'conv_fixup_kernel<%s>' % n.split('<')[1].split('>')[0].replace(' ', '')
    return base
summary = {}
for s in ('fetch', 'write', 'sq'):
    p = f'/tmp/{tag}_pmc_{s}.csv'
    if not os.path.exists(p):
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    for r in csv.DictReader(open(p)):
        k = fam(r['Kernel_Name'])
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[k].add(r['Dispatch_Id'])
    for k, d in agg.items():
        e = summary.setdefault(k, {})
        e['dispatches'] = len(cnt[k])
        e['dispatches_' + s] = len(cnt[k])      # (bench.py's sustained loop is time-based: the passes do not run the same number of frames)
        e.update({c: v for c, v in d.items()})
# durations of the SAME dispatches (kernel trace of the sq pass): matrix-pipe utilisation = busy cycles / (SIMDs x kernel cycles)
import glob
durs = collections.defaultdict(list)
for kt in glob.glob('/tmp/pmc_sq/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(kt)):
        durs[fam(r['Kernel_Name'])].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
# a dispatch whose timestamps span a stall of the profiled process (r06: one family read 137 us per dispatch where GRBM_GUI_ACTIVE of the same
# dispatches and every other measurement say 74) counts at most three times its family's median
dur = {}
for k, v in durs.items():
    med = sorted(v)[len(v) // 2]
    dur[k] = sum(min(d, 3.0 * med) for d in v)
for k, e in summary.items():
    if dur.get(k) and e.get('SQ_INSTS_MFMA'):
        e['kernel_ns_in_sq_pass'] = dur[k]
        # SQ_VALU_MFMA_BUSY_CYCLES / SQ_INSTS_MFMA = 32 (fp16 32x32x16) or 64 (fp32 32x32x2): per-SIMD pipe cycles of the whole chip;
        # 1024 SIMDs, peak clock 2.4 GHz (the sustained clock under MFMA load is lower: this is utilisation against the PEAK)
        e['mfma_pipe_util_vs_peak_clock'] = round(e['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * dur[k] * 2.4), 4)
for k, e in summary.items():
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in e and e.get('SQ_BUSY_CYCLES'):
        # Ratios of counters of the SAME block are independent of how many XCDs / SEs the tool samples:
        #   MFMA pipe busy per SQ-busy cycle (SQ_BUSY_CYCLES counts, per SE, cycles with any wave resident; 4 SIMDs per CU share
        #   nothing here: SQ_VALU_MFMA_BUSY_CYCLES is already per-SIMD-summed the same way) and LDS conflict cycles per LDS cycle
        e['mfma_busy_per_sq_busy'] = round(e['SQ_VALU_MFMA_BUSY_CYCLES'] / e['SQ_BUSY_CYCLES'], 4)
        e['mfma_busy_cycles_per_mfma_inst'] = round(e['SQ_VALU_MFMA_BUSY_CYCLES'] / max(e.get('SQ_INSTS_MFMA', 0), 1), 2)
        e['valu_inst_per_mfma_inst'] = round(e.get('SQ_ACTIVE_INST_VALU', 0) / max(e.get('SQ_INSTS_MFMA', 0), 1), 2)
        e['lds_bank_conflict_over_lds_active'] = round(e.get('SQ_LDS_BANK_CONFLICT', 0) / max(e.get('SQ_LDS_IDX_ACTIVE', 1), 1), 4)
json.dump(summary, open(f'{out}/{tag}_pmc_summary.json', 'w'), indent=1, sort_keys=True)
top = sorted(summary.items(), key=lambda kv: -kv[1].get('GRBM_GUI_ACTIVE', kv[1].get('FETCH_SIZE', 0)))[:10]
for k, e in top:
    print(k, {c: e[c] for c in e if c in ('dispatches', 'FETCH_SIZE', 'WRITE_SIZE', 'mfma_pipe_util_vs_peak_clock', 'mfma_busy_cycles_per_mfma_inst', 'lds_bank_conflict_over_lds_active', 'SQ_INSTS_MFMA')})
PY
