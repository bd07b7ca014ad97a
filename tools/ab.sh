#!/bin/bash
# A/B of library builds on the GPU box: tools/ab.sh "<lib1> <lib2> ..." "<bench args>"
# prints ms/step and the per-kernel averages of each build
set -o pipefail
for lib in $1; do
  for args in "${@:2}"; do
    NMPC_AMD_LIB=$PWD/$lib timeout -k 10 300 python bench.py --no-cpu-baseline --no-peaks $args > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "$lib $args FAILED"; tail -n 5 gpurun_out/ab.err; exit 1; }
    python - "$lib" "$args" <<'PY'
import json, sys
j = json.load(open("gpurun_out/ab.json"))
k = {n: v["avg_ms"] for n, v in j.get("kernels", {}).items()}
w = j.get("warm_step") or {}
print("%-32s %-40s ms/step %8.4f  warm p50 %s  %s" % (sys.argv[1], sys.argv[2], j["ms_per_step"], w.get("p50_ms"), k))
PY
  done
done
