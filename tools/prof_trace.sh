#!/bin/bash
# Kernel trace + stats of a command on the GPU box; copies the trace and stats into gpurun_out/<name>/:  tools/prof_trace.sh <name> <cmd...>
name="$1"; shift
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/$name
timeout 280 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/$name -o $name -- "$@" > /tmp/$name.log 2>&1
grep -v "^[WEI]2026" /tmp/$name.log | tail -5
mkdir -p $repo/gpurun_out/$name
for f in $(find /tmp/$name -name "*kernel_stats.csv" -o -name "*kernel_trace.csv"); do cp $f $repo/gpurun_out/$name/; done
ls -la $repo/gpurun_out/$name
