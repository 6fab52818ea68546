#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
echo "== config 5 test"; timeout -k 10 600 python -m pytest tests/test_gpu_codec.py -x -q -m gpu -s -k "config5 or 4k_frame" > $O/r02_g_cfg5.log 2>&1; rc=$?; tail -4 $O/r02_g_cfg5.log; grep "Config 5" $O/r02_g_cfg5.log; [ $rc -eq 0 ] || exit 1
echo "== configs bench"; timeout -k 10 600 python tools/configs_bench.py 2 > $O/r02_g_configs_bench.jsonl 2>$O/r02_g_configs_bench.err; rc=$?; cat $O/r02_g_configs_bench.jsonl; tail -3 $O/r02_g_configs_bench.err; [ $rc -eq 0 ] || exit 1
