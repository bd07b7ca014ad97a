#!/bin/bash
# Write side of the stage linearisation in the cold solve of 4096 x 100: bytes and request counts
# per launch (one rocprofv3 --pmc pass), for the library named by NMPC_AMD_LIB.
set -o pipefail
R=$GRAFT_REPO_ROOT
tag=${1:-x}
out=$R/gpurun_out/pmc_writes_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp PROF_STEPS=2
rm -rf /tmp/pw_$tag
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum --output-format csv -d /tmp/pw_$tag -o t -- python3 $R/tools/prof_run.py > $out/run.log 2>&1 || { echo "pmc failed"; tail -n 5 $out/run.log; exit 1; }
cp $(find /tmp/pw_$tag -name "*counter_collection.csv" | head -1) $out/writes.csv
python3 - $out/writes.csv <<'PY'
import csv, sys
from collections import defaultdict
per = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    for k in ("k_linearise", "k_qp_begin", "k_newton_step", "k_newton_solve"):
        if k in r["Kernel_Name"]:
            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in per.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
