#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
echo "== ops tests (all grids permuted)"; PC_CONV_ROWPERM_MIN=0 timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu > $O/r02_q_ops.log 2>&1; rc=$?; tail -2 $O/r02_q_ops.log; [ $rc -eq 0 ] || exit 1
run() { echo "== $*"; env "$@" timeout -k 10 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline > $O/r02_h_tmp.log 2>&1 || { tail -5 $O/r02_h_tmp.log; exit 1; }; tail -1 $O/r02_h_tmp.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['serial_step_ms'], j['roofline']['achieved'], j['roofline']['kernel_ms_per_step'])"; }
for rep in 1 2; do
run PC_CONV_ROWPERM=1
run PC_CONV_ROWPERM=1 PC_CONV_ROWPERM_MIN=0
run PC_CONV_ROWPERM=0
done
