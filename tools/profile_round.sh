#!/bin/bash
# The rocprofv3 passes behind profiles/rNN_*: run on the GPU box through gpurun (`gpurun -- 'bash tools/profile_round.sh'`), then
# fold the rocpd databases here with tools/rocpd_stats.py (kernel summary) and tools/pmc_traffic.py (HBM bytes per launch).
# Counters are collected in their own passes with --kernel-trace only (no sys/hip/hsa traces next to --pmc).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_stats_2lanes $O/prof_stats_1lane $O/pmc_FETCH $O/pmc_WRITE
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/prof_stats_2lanes -o run -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_stats_2lanes.log 2>&1 || exit 1
PC_LANES=1 PC_DUAL_STREAM=0 PC_PIPELINE=0 timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/prof_stats_1lane -o run -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/prof_stats_1lane.log 2>&1 || exit 1
PC_LANES=1 PC_DUAL_STREAM=0 PC_PIPELINE=0 timeout -k 10 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_FETCH -o runc -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_FETCH.log 2>&1 || exit 1
PC_LANES=1 PC_DUAL_STREAM=0 PC_PIPELINE=0 timeout -k 10 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_WRITE -o runc -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_WRITE.log 2>&1 || exit 1
echo "then: python tools/rocpd_stats.py gpurun_out/prof_stats_1lane > profiles/rNN_kernel_stats_bench_b32_1lane.csv"
echo "      python tools/pmc_traffic.py gpurun_out/pmc_FETCH gpurun_out/pmc_WRITE > profiles/rNN_hbm_traffic.json"
