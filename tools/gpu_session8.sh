#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
echo "== gpu tests"; timeout -k 10 900 python -m pytest tests/ -x -q -m gpu -s > $O/r02_gpu_tests_c.log 2>&1; echo "rc $?"; tail -4 $O/r02_gpu_tests_c.log
echo "== bench default"; timeout -k 10 900 python bench.py > $O/r02_c_bench_default.log 2>&1; echo "rc $?"; tail -1 $O/r02_c_bench_default.log | cut -c1-1500
echo "== bench --gpus 2 on a 1-GPU box"; python bench.py --gpus 2 > $O/r02_c_bench_gpus2.log 2>&1; echo "rc $?"; tail -2 $O/r02_c_bench_gpus2.log
echo "== profile round"; bash tools/profile_round.sh r02_c 2>&1 | tail -12
