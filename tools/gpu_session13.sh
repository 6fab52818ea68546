#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
run() { echo "== $*"; env "$@" timeout -k 10 300 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --overlap 1 > $O/r02_h_tmp.log 2>&1 || { tail -5 $O/r02_h_tmp.log; exit 1; }; tail -1 $O/r02_h_tmp.log | cut -c1-140; }
run A=0
run PC_PIPELINE=0
run PC_PIPELINE_DEC=0
run PC_DUAL_STREAM=0
run PC_LANES=1
run PC_LANES=2
run PC_LANES=2 PC_PIPELINE=0
run A=0
