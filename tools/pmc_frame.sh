#!/bin/bash
# HBM traffic of the frame's kernels from PMC counters, one counter per pass (MI355X_MICROARCH.md, HBM / rocprofv3 PMC slots):
#   tools/pmc_frame.sh            -> gpurun_out/pmc_frame/{fetch,write}_counter_collection.csv + summary.json
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
mkdir -p $repo/gpurun_out/pmc_frame
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 280 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- \
      python $repo/bench.py --steps 2 --warmup 1 --eager --no-cpu-baseline --no-roofline --no-sr-fp16 > /tmp/pmc_$c.log 2>&1
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  if [ -z "$f" ]; then tail -5 /tmp/pmc_$c.log; else cp $f $repo/gpurun_out/pmc_frame/${c}_counter_collection.csv; fi
done
python3 - <<'PY'
import csv, json, collections, os
out = {}
root = os.environ.get('GRAFT_REPO_ROOT', '/root/repo') + '/gpurun_out/pmc_frame'
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    p = f'{root}/{c}_counter_collection.csv'
    if not os.path.exists(p):
        continue
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(p)):
        if r['Counter_Name'] != c:
            continue
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        fam = 'conv_mfma_kernel' if name.startswith('conv_mfma_kernel') else ('conv_fixup_kernel' if name.startswith('conv_fixup') else name.split('<')[0])
        agg[fam][0] += float(r['Counter_Value']); agg[fam][1] += 1
    out[c] = {k: {'sum_kb': round(v[0], 1), 'launches': v[1], 'kb_per_launch': round(v[0] / v[1], 1)} for k, v in agg.items()}
json.dump(out, open(f'{root}/summary.json', 'w'), indent=1)
for c, d in out.items():
    for k, v in sorted(d.items(), key=lambda kv: -kv[1]['sum_kb'])[:8]:
        print(c, k, v)
PY
