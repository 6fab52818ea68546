#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
echo "== gpu codec tests"; timeout -k 10 900 python -m pytest tests/test_gpu_codec.py -x -q -m gpu -s > $O/r02_gpu_tests_d.log 2>&1; echo "rc $?"; tail -3 $O/r02_gpu_tests_d.log; grep "Config 4" $O/r02_gpu_tests_d.log
echo "== bench"; PC_TIMING=1 timeout -k 10 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/r02_d_bench.log 2>&1; grep "decompress:" $O/r02_d_bench.log | tail -3; tail -1 $O/r02_d_bench.log | cut -c1-330
