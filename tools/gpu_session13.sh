#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
echo "== ops tests BK=16"; PC_CONV_BK=16 timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv or deconv" > $O/r02_i_ops_bk16.log 2>&1; rc=$?; tail -3 $O/r02_i_ops_bk16.log; [ $rc -eq 0 ] || exit 1
run() { echo "== $*"; env "$@" timeout -k 10 300 python bench.py --steps 12 --warmup 2 --no-cpu-baseline > $O/r02_h_tmp.log 2>&1 || { tail -5 $O/r02_h_tmp.log; exit 1; }; tail -1 $O/r02_h_tmp.log | cut -c1-140; }
run A=0
run PC_CONV_BK=16
run A=0
run PC_CONV_BK=16
