#!/bin/bash
# PMC counters of the conv kernel on given shapes (run on the GPU box): tools/pmc_conv.sh "I O res tr" ...
cd /tmp && export TMPDIR=/tmp
for sh in "$@"; do
  rm -rf /tmp/pmc
  timeout 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES ${PMC_EXTRA:-SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY} \
     --kernel-trace --output-format csv -d /tmp/pmc -o p -- python $GRAFT_REPO_ROOT/tools/bench_conv_one.py $sh > /tmp/pmc.log 2>&1
  python - "$sh" <<'PY'
import csv, glob, sys
from collections import defaultdict
fs = glob.glob("/tmp/pmc/**/*counter_collection.csv", recursive=True)
if not fs:
    print(open('/tmp/pmc.log').read()[-2000:]); sys.exit()
d = defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if "conv_mfma" in r["Kernel_Name"]:
        d[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[1], {k: round(sum(v[-2:]) / 2) for k, v in d.items()})
ks = glob.glob("/tmp/pmc/**/*kernel_trace.csv", recursive=True)
if ks:
    rows = [r for r in csv.DictReader(open(ks[0])) if "conv_mfma" in r["Kernel_Name"]]
    print('  dur_us', [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1) for r in rows[-3:]])
PY
done
