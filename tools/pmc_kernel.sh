#!/bin/bash
# PMC counters for kernels matching $1 while running "$2..." (on the GPU box). Bounded by timeout.
pat="$1"; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc
timeout 150 rocprofv3 --pmc ${PMC_LIST:-SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY} \
   --kernel-trace --output-format csv -d /tmp/pmc -o p -- "$@" > /tmp/pmc.log 2>&1
python - "$pat" <<'PY'
import csv, glob, sys
from collections import defaultdict
fs = glob.glob("/tmp/pmc/**/*counter_collection.csv", recursive=True)
if not fs:
    print(open('/tmp/pmc.log').read()[-1500:]); sys.exit()
d = defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if sys.argv[1] in r["Kernel_Name"]:
        d[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[1], {k: round(sum(v[-3:]) / len(v[-3:])) for k, v in d.items()})
PY
