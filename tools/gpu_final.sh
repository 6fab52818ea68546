#!/bin/bash
# final GPU pass of a round, in two gpurun calls (one call may run 1200 s at most):
#   gpurun -- 'bash tools/env_matrix.sh > gpurun_out/TAG_env_config_matrix_gpu_tests.log'      (23 configurations of the GPU suite, ~17 min)
#   gpurun -- 'bash tools/gpu_final.sh TAG'                                                   (this file: suite, smoke, profile round, bench lines)
set -o pipefail
TAG=${1:-r02_z}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
echo "== full gpu tests"; timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/${TAG}_gpu_tests.log 2>&1; rc=$?; tail -2 $O/${TAG}_gpu_tests.log; [ $rc -eq 0 ] || exit 1
echo "== smoke"; timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/${TAG}_smoke.log 2>&1; rc=$?; tail -1 $O/${TAG}_smoke.log; [ $rc -eq 0 ] || exit 1
bash tools/gpu_final2.sh $TAG
