#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
echo "== ops tests (all grids permuted)"; PC_CONV_ROWPERM_MIN=0 timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu > $O/r02_q_ops.log 2>&1; rc=$?; tail -2 $O/r02_q_ops.log; [ $rc -eq 0 ] || exit 1
echo "== codec tests"; timeout -k 10 900 python -m pytest tests/test_gpu_codec.py -x -q -m gpu > $O/r02_q_codec.log 2>&1; rc=$?; tail -2 $O/r02_q_codec.log; [ $rc -eq 0 ] || exit 1
PC_PROFILE_CSV=$O/r02_q_launches.csv timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['serial_step_ms'], j['roofline']['achieved'], j['roofline']['kernel_ms_per_step'])"
