#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
echo "== threads tests"; timeout -k 10 300 python -m pytest tests/test_gpu_codec.py -x -q -m gpu -s -k "side_by_side or two_threads" > $O/r02_p_sbs.log 2>&1; rc=$?; tail -4 $O/r02_p_sbs.log; grep "same object" $O/r02_p_sbs.log; [ $rc -eq 0 ] || exit 1
echo "== full gpu tests"; timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/r02_z_gpu_tests.log 2>&1; rc=$?; tail -4 $O/r02_z_gpu_tests.log; [ $rc -eq 0 ] || exit 1
echo "== smoke"; timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r02_z_smoke.log 2>&1; rc=$?; tail -2 $O/r02_z_smoke.log; [ $rc -eq 0 ] || exit 1
bash tools/profile_round.sh r02_z > $O/profile_round_r02_z.log 2>&1 || { tail -5 $O/profile_round_r02_z.log; exit 1; }
echo "== default bench"; timeout -k 10 600 python bench.py > $O/r02_z_bench_default.log 2>&1; rc=$?; tail -1 $O/r02_z_bench_default.log | cut -c1-700; [ $rc -eq 0 ] || exit 1
