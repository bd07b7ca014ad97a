#!/bin/bash
# Everything profiles/ holds for this round, made on the GPU box under gpurun_out/prof_<tag>/:
#   kernel statistics (rocprofv3 --kernel-trace --stats) of the bench command per configuration
#   and of warm control steps; HBM traffic per launch (separate --pmc passes); the VALU
#   instruction mix of the stage linearisation.
set -o pipefail
tag=${1:-run}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp

stats() {   # name, then the program and its arguments
  local name=$1; shift
  rm -rf /tmp/ks_$name
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$name -o t -- "$@" > $out/${name}_run.log 2>&1 || { echo "stats $name failed"; tail -n 5 $out/${name}_run.log; return 1; }
  cp $(find /tmp/ks_$name -name "*kernel_stats.csv" | head -1) $out/${tag}_${name}_kernel_stats.csv
  grep -E '^\{"metric"' $out/${name}_run.log > $out/${tag}_${name}_bench.json || true
}
pmc() {     # name, counters, then the program
  local name=$1 ctr=$2; shift 2
  rm -rf /tmp/pm_$name
  timeout -s KILL 200 rocprofv3 --pmc $ctr --output-format csv -d /tmp/pm_$name -o t -- "$@" > $out/pmc_${name}.log 2>&1 || { echo "pmc $name failed"; tail -n 5 $out/pmc_${name}.log; return 1; }
  cp $(find /tmp/pm_$name -name "*counter_collection.csv" | head -1) $out/pmc_${name}.csv
}

stats cfg2 python3 $R/bench.py --no-cpu-baseline --no-peaks --warm-steps 0 --steps 10 --warmup 2 &&
stats cfg4 python3 $R/bench.py --config 4 --no-cpu-baseline --no-peaks --warm-steps 0 --steps 5 --warmup 2 &&
stats cfg5 python3 $R/bench.py --config 5 --no-cpu-baseline --no-peaks --steps 20 --warmup 2 &&
stats b32768 python3 $R/bench.py --instances 32768 --no-cpu-baseline --no-peaks --warm-steps 0 --steps 3 --warmup 1 &&
WARM=12 stats warm python3 $R/tools/warm_profile.py || exit 1

export PROF_STEPS=2
pmc cfg2_fetch FETCH_SIZE python3 $R/tools/prof_run.py &&
pmc cfg2_write WRITE_SIZE python3 $R/tools/prof_run.py &&
PROF_B=1024 PROF_N=500 pmc cfg4_fetch FETCH_SIZE python3 $R/tools/prof_run.py &&
PROF_B=1024 PROF_N=500 pmc cfg4_write WRITE_SIZE python3 $R/tools/prof_run.py &&
PROF_B=32768 PROF_STEPS=1 pmc b32768_fetch FETCH_SIZE python3 $R/tools/prof_run.py &&
PROF_B=32768 PROF_STEPS=1 pmc b32768_write WRITE_SIZE python3 $R/tools/prof_run.py &&
pmc cfg5_fetch FETCH_SIZE python3 $R/tools/prof_samples.py &&
pmc cfg5_write WRITE_SIZE python3 $R/tools/prof_samples.py &&
pmc cfg5_valu "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAVE_CYCLES SQ_WAVES" python3 $R/tools/prof_samples.py &&
pmc cfg5_busy "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_SMEM" python3 $R/tools/prof_samples.py &&
pmc cfg2_valu "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAVE_CYCLES SQ_WAVES" python3 $R/tools/prof_run.py || exit 1

cd $R
python3 tools/pmc_traffic.py $out/pmc_cfg2_fetch.csv $out/pmc_cfg2_write.csv cfg2 "4096 instances, horizon 100, cold solve" &&
python3 tools/pmc_traffic.py $out/pmc_cfg4_fetch.csv $out/pmc_cfg4_write.csv cfg4 "1024 instances, horizon 500, cold solve" &&
python3 tools/pmc_traffic.py $out/pmc_b32768_fetch.csv $out/pmc_b32768_write.csv b32768_n100 "32768 instances, horizon 100, cold solve" &&
python3 tools/pmc_traffic.py $out/pmc_cfg5_fetch.csv $out/pmc_cfg5_write.csv cfg5 "2^20 samples, stage linearisation" || exit 1
python3 - $out $tag <<'PY'
import csv, json, re, sys
from collections import defaultdict
out, tag = sys.argv[1], sys.argv[2]
def mix(path):
    per = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(path)):
        m = re.search(r"nmpc::(?:\w+::)*(k_\w+)", r["Kernel_Name"])
        if m: per[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in per.items()}
res = {"source": "rocprofv3 --pmc, one pass per file, mean per launch; SQ instruction counters count wavefront instructions"}
for name in ("cfg5_valu", "cfg5_busy", "cfg2_valu"):
    res[name] = mix("%s/pmc_%s.csv" % (out, name))
json.dump(res, open("gpurun_out/%s_instruction_mix.json" % tag, "w"), indent=1)
for k, d in res["cfg5_valu"].items(): print(k, {c: round(v) for c, v in d.items()})
PY
ls $out
