#!/bin/bash
# Runs bench.py for every configuration on the GPU box; one JSON line each under gpurun_out/<tag>_*.json
# usage: tools/bench_all.sh <tag>
set -o pipefail
tag=${1:-run}
mkdir -p gpurun_out
timeout -k 10 240 python bench.py > gpurun_out/${tag}_cfg2.json 2> gpurun_out/${tag}_cfg2.err || { echo "cfg2 failed"; tail -n 5 gpurun_out/${tag}_cfg2.err; exit 1; }
timeout -k 10 240 python bench.py --config 4 > gpurun_out/${tag}_cfg4.json 2> gpurun_out/${tag}_cfg4.err || { echo "cfg4 failed"; tail -n 5 gpurun_out/${tag}_cfg4.err; exit 1; }
timeout -k 10 240 python bench.py --config 5 > gpurun_out/${tag}_cfg5.json 2> gpurun_out/${tag}_cfg5.err || { echo "cfg5 failed"; tail -n 5 gpurun_out/${tag}_cfg5.err; exit 1; }
timeout -k 10 240 python bench.py --config 1 > gpurun_out/${tag}_cfg1.json 2> gpurun_out/${tag}_cfg1.err || { echo "cfg1 failed"; tail -n 5 gpurun_out/${tag}_cfg1.err; exit 1; }
timeout -k 10 300 python bench.py --instances 32768 --steps 5 --warm-steps 5 > gpurun_out/${tag}_b32768.json 2> gpurun_out/${tag}_b32768.err || { echo "b32768 failed"; tail -n 5 gpurun_out/${tag}_b32768.err; exit 1; }
python - <<PY
import json
for c in ("cfg2","cfg4","cfg5","cfg1","b32768"):
    j=json.load(open("gpurun_out/${tag}_%s.json"%c))
    r=j["roofline"]; w=j.get("warm_step") or {}
    print(c, "value", j["value"], j["unit"], "ms/step", j["ms_per_step"], "p50", j.get("p50_step_ms"), "p99", j.get("p99_step_ms"))
    print("   roofline", r.get("kernel"), r.get("bound"), "frac", r.get("frac"), "avg_ms", r.get("avg_launch_ms"), "fp64frac", r.get("fp64_vector_frac"), "of_meas_noFMA", r.get("frac_of_measured_fp64_no_fma"))
    if "linearise" in r: print("   linearise", r["linearise"]["avg_launch_ms"], r["linearise"]["fp64_vector_TFLOPs"])
    print("   peaks", r.get("peak_measured"))
    print("   cpu", j.get("cpu_baseline"))
    if w: print("   warm", {k: w[k] for k in ("p50_ms","p99_ms","steps_per_s","ok_frac")})
    if "kernels" in j: print("   kernels", {k:(v["avg_ms"], v["launches"]) for k,v in j["kernels"].items()})
PY
