#!/bin/bash
# profile round + bench lines of a build (the matrix and the test suite are tools/env_matrix.sh and pytest): everything under gpurun_out/
set -o pipefail
TAG=${1:-r02_z}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
echo "== profile round"; bash tools/profile_round.sh $TAG > $O/profile_round_$TAG.log 2>&1 || { tail -5 $O/profile_round_$TAG.log; exit 1; }
mkdir -p $R/profiles && cp $O/profiles_$TAG/* $R/profiles/      # on the box only: lets the bench below find this build's traffic profile
echo "== default bench"; timeout -k 10 600 python bench.py > $O/${TAG}_bench_default.log 2>&1; rc=$?; tail -1 $O/${TAG}_bench_default.log | cut -c1-260; [ $rc -eq 0 ] || exit 1
echo "== long bench"; timeout -k 10 300 python bench.py --steps 60 --warmup 3 --no-cpu-baseline > $O/${TAG}_bench_60steps.log 2>&1; tail -1 $O/${TAG}_bench_60steps.log | cut -c1-200
echo "== configs"; timeout -k 10 300 python tools/configs_bench.py 2 > $O/${TAG}_configs_3_4_5_bench.jsonl 2>/dev/null; cut -c1-160 $O/${TAG}_configs_3_4_5_bench.jsonl
