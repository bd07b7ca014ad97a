#!/bin/bash
# Soaks of the current build on the GPU box against the CPU port (tests/soak_warm.py), one line per batch.
# usage: tools/soak_all.sh [seed offset]
set -o pipefail
o=${1:-0}
export SOAK_DUMPS=64
run() { timeout -k 10 600 python -u tests/soak_warm.py "$@" || exit 1; }
run 4096 100 6 $((9100+o)) 8192 100 3 $((9101+o)) 1024 500 3 $((9102+o)) 2000 230 3 $((9103+o)) 1500 100 4 $((9104+o)) &&
SOAK_TIGHT_BOUNDS=1 run 4096 100 5 $((9110+o)) 1024 500 2 $((9111+o)) &&
SOAK_NOISE=6 run 4096 100 5 $((9120+o)) 600 230 3 $((9121+o)) &&
SOAK_HOSTILE=0.2 run 4096 100 4 $((9130+o)) 1024 500 2 $((9131+o)) &&
SOAK_RTI=1 run 4096 100 5 $((9140+o)) 700 230 3 $((9141+o)) &&
SOAK_PRECISION=float NMPC_AMD_LIB=nmpc_amd/libcnmpc_f32.so run 4096 100 4 $((9150+o)) 1024 500 2 $((9151+o))
