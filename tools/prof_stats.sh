#!/bin/bash
# Kernel-trace stats of a command on the GPU box, top kernels printed:  tools/prof_stats.sh <out-name> <cmd...>
name="$1"; shift
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/$name
timeout 280 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/$name -o $name -- "$@" > /tmp/$name.log 2>&1
tail -12 /tmp/$name.log
f=$(find /tmp/$name -name "*kernel_stats.csv" | head -1)
mkdir -p $repo/gpurun_out/$name && cp $f $repo/gpurun_out/$name/ 2>/dev/null
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:28]:
    print(f"{r['Name'][:100]:100s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:8.1f}  {r['Percentage']:>6s}%")
PY
