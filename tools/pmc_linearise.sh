#!/bin/bash
# Issue and stall counters of the stage linearisation on 2^20 samples (three rocprofv3 --pmc passes).
set -o pipefail
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/pmc_lin_${1:-x}
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp PROF_STEPS=2
pass() {
  local name=$1 ctr=$2
  rm -rf /tmp/pm_$name
  timeout -s KILL 200 rocprofv3 --pmc $ctr --output-format csv -d /tmp/pm_$name -o t -- python3 $R/tools/${PROG:-prof_samples.py} > $out/$name.log 2>&1 || { echo "pmc $name failed"; tail -n 5 $out/$name.log; return 1; }
  cp $(find /tmp/pm_$name -name "*counter_collection.csv" | head -1) $out/$name.csv
}
pass valu "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAVE_CYCLES SQ_WAVES" &&
pass busy "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS" &&
pass misc "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS" || exit 1
python3 - $out <<'PY'
import csv, re, sys
from collections import defaultdict
out = sys.argv[1]
for name in ("valu", "busy", "misc"):
    per = defaultdict(list)
    for r in csv.DictReader(open("%s/%s.csv" % (out, name))):
        if "k_linearise" in r["Kernel_Name"]:
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(name, {c: round(sum(v) / len(v)) for c, v in per.items()})
PY
