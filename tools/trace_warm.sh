#!/bin/bash
# Kernel trace of warm control steps (tools/warm_profile.py) on the GPU box: per-dispatch durations of the last step.
set -o pipefail
tag=${1:-warmtrace}; export WARM=${2:-6}
out=gpurun_out/${tag}
rm -rf $out; mkdir -p $out
timeout -k 10 120 python3 tools/warm_profile.py > $out/plain.log 2>&1 || { tail -n 20 $out/plain.log; exit 1; }
cat $out/plain.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out -o t -- python3 $GRAFT_REPO_ROOT/tools/warm_profile.py > $GRAFT_REPO_ROOT/$out/run.log 2>&1 || { tail -n 20 $GRAFT_REPO_ROOT/$out/run.log; exit 1; }
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, re
f = glob.glob("$out/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    m = re.search(r"nmpc::(?:\w+::)*(k_\w+)", n); return m.group(1) if m else n[:30]
names = [short(r["Kernel_Name"]) for r in rows]
idx = [i for i, n in enumerate(names) if n == "k_qp_begin"]
start = idx[-1]
t0 = int(rows[start]["Start_Timestamp"])
for r in rows[start:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if (s - t0) > 2e7: break
    print("%-28s start %9.1f us  dur %8.1f us  grid %s" % (short(r["Kernel_Name"]), (s - t0) / 1e3, (e - s) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
PY
