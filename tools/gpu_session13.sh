#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
run() { echo "== $*"; env "$@" timeout -k 10 300 python bench.py --steps 60 --warmup 3 --no-cpu-baseline > $O/r02_h_tmp.log 2>&1 || { tail -5 $O/r02_h_tmp.log; exit 1; }; tail -1 $O/r02_h_tmp.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['serial_step_ms'], j['roofline']['achieved'])"; }
run A=0
run PC_PIPELINE=0
run PC_PIPELINE_DEC=0
run PC_DUAL_STREAM=0
run PC_LANES=1
run PC_LANES=2
run PC_LANES=4
run PC_GROUPED=0
run A=0
