#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
bash tools/env_matrix.sh "ROWPERM|POLICY=0|TM=2|KERN=0" 2>&1 | tee $O/r02_z_env_matrix_subset.log | grep -E "^==|passed|failed|matrix"
echo "== full gpu tests"; timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/r02_z_gpu_tests.log 2>&1; rc=$?; tail -2 $O/r02_z_gpu_tests.log; [ $rc -eq 0 ] || exit 1
echo "== smoke"; timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r02_z_smoke.log 2>&1; rc=$?; tail -1 $O/r02_z_smoke.log; [ $rc -eq 0 ] || exit 1
bash tools/gpu_final2.sh r02_z
