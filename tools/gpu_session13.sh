#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
run() { echo "== $*"; env "$@" timeout -k 10 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline > $O/r02_h_tmp.log 2>&1 || { tail -5 $O/r02_h_tmp.log; exit 1; }; tail -1 $O/r02_h_tmp.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['serial_step_ms'], j['roofline']['achieved'], j['roofline']['kernel_ms_per_step'])"; }
for rep in 1 2; do
run PC_CONV_POLICY=1
run PC_CONV_POLICY=3
run PC_CONV_POLICY=3 PC_CONV_SMALL_THR=256
run PC_CONV_POLICY=3 PC_CONV_SMALL_THR=1024
done
PC_CONV_POLICY=3 PC_PROFILE_CSV=$O/r02_o_conv_launches_pol3.csv timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
