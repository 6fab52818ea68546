#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
echo "== full gpu tests"; timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/r02_m_gpu_tests.log 2>&1; rc=$?; tail -5 $O/r02_m_gpu_tests.log; [ $rc -eq 0 ] || exit 1
echo "== smoke"; timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r02_m_smoke.log 2>&1; rc=$?; tail -3 $O/r02_m_smoke.log; [ $rc -eq 0 ] || exit 1
