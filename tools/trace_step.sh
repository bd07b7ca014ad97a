#!/bin/bash
# Kernel trace of tools/prof_run.py (cold solves) on the GPU box: per-dispatch durations in launch order.
# usage: tools/trace_step.sh <tag> [B] [N]
set -o pipefail
tag=${1:-trace}; export PROF_B=${2:-4096}; export PROF_N=${3:-100}; export PROF_STEPS=3
out=gpurun_out/${tag}
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out -o t -- python3 $GRAFT_REPO_ROOT/tools/prof_run.py > $GRAFT_REPO_ROOT/$out/run.log 2>&1 || { tail -n 20 $GRAFT_REPO_ROOT/$out/run.log; exit 1; }
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, re
f = glob.glob("$out/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    m = re.search(r"nmpc::(?:\w+::)*(k_\w+)", n); return m.group(1) if m else n[:30]
# last step only
names = [short(r["Kernel_Name"]) for r in rows]
idx = [i for i, n in enumerate(names) if n == "k_qp_begin"]
start = idx[-1]
# walk back to the linearise before it
while start > 0 and names[start] != "k_linearise": start -= 1
t0 = int(rows[start]["Start_Timestamp"])
for r in rows[start:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-28s start %9.1f us  dur %8.1f us  vgpr %s" % (short(r["Kernel_Name"]), (s - t0) / 1e3, (e - s) / 1e3, r.get("VGPR_Count", r.get("Arch_VGPR_Count", "?"))))
PY
