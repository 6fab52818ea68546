#!/bin/bash
# The rocprofv3 passes behind profiles/rNN_*: run on the GPU box through gpurun (`gpurun -- 'bash tools/profile_round.sh TAG'`); it writes the
# folded summaries to gpurun_out/profiles_TAG/ -- copy them into profiles/ and commit.  Counters are collected in their own passes
# with --kernel-trace only (no sys/hip/hsa traces next to --pmc).  bench.py ties a summary to the build through its source hash.
set -o pipefail
TAG=${1:-r02_x}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; P=$O/profiles_$TAG; mkdir -p $P
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_stats_default $O/prof_stats_1lane $O/pmc_FETCH $O/pmc_WRITE $O/prof_stage
SER="PC_LANES=1 PC_DUAL_STREAM=0 PC_PIPELINE=0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_stats_default -o run -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_stats_default.log 2>&1 || exit 1
echo "default stats done"
env $SER timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_stats_1lane -o run -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/prof_stats_1lane.log 2>&1 || exit 1
echo "1lane stats done"
env $SER timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_FETCH -o runc -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_FETCH.log 2>&1 || exit 1
env $SER timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_WRITE -o runc -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_WRITE.log 2>&1 || exit 1
echo "pmc done"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_stage -o run -- python3 $R/tools/stage_bench.py 256 > $O/prof_stage.log 2>&1 || exit 1
cd $R
python3 tools/rocpd_stats.py $O/prof_stats_default > $P/${TAG}_kernel_stats_bench_b32_default.csv
python3 tools/rocpd_stats.py $O/prof_stats_1lane > $P/${TAG}_kernel_stats_bench_b32_1lane.csv
python3 tools/pmc_traffic.py $O/pmc_FETCH $O/pmc_WRITE > $P/${TAG}_hbm_traffic.json
python3 tools/stage_rocprof.py $O/prof_stage 256 > $P/${TAG}_stage_kernels_rocprof.json
grep -h '^{' $O/prof_stats_1lane.log | tail -1 > $P/${TAG}_bench_under_rocprof_1lane.json
grep -h '^{' $O/prof_stage.log > $P/${TAG}_stage_hbm_roofline_b256_events.jsonl
ls -la $P
