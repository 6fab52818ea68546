#!/bin/bash
# The rocprofv3 passes behind profiles/rNN_*: run on the GPU box through gpurun (`gpurun -- 'bash tools/profile_round.sh TAG'`); it writes the
# folded summaries to gpurun_out/profiles_TAG/ -- copy them into profiles/ and commit.  Counters are collected in their own passes
# with --kernel-trace only (no sys/hip/hsa traces next to --pmc).  bench.py ties a summary to the build through its source hash.
set -o pipefail
TAG=${1:-r03_x}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; P=$O/profiles_$TAG; mkdir -p $P
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_stats_default $O/prof_stats_1lane $O/pmc_FETCH $O/pmc_WRITE $O/prof_stage $O/pmc_stage_FETCH $O/pmc_stage_WRITE $O/pmc_clk_serial
# serial form: one lane, one stream, no pipelining inside the codec (pc_codec_set_option "serial_schedule": bench.py --serial-schedule) and no
# encoder / decoder overlap -- a launch's duration is its own
SER="PC_UNUSED=0"
SS="--serial-schedule"
# (--lean: the timed steps, the enc/dec split and the roofline leg only -- no CPU baseline, no rANS leg, no second sequential run)
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_stats_default -o run -- python3 $R/bench.py --steps 30 --warmup 2 --lean > $O/prof_stats_default.log 2>&1 || exit 1
echo "default stats done"
env $SER timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_stats_1lane -o run -- python3 $R/bench.py --steps 5 --warmup 1 --lean --overlap 0 $SS > $O/prof_stats_1lane.log 2>&1 || exit 1
echo "1lane stats done"
env $SER PC_PROFILE_CSV=$O/pmc_FETCH_launches.csv timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_FETCH -o runc -- python3 $R/bench.py --steps 1 --warmup 1 --lean --overlap 0 $SS > $O/pmc_FETCH.log 2>&1 || exit 1
env $SER timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_WRITE -o runc -- python3 $R/bench.py --steps 1 --warmup 1 --lean --overlap 0 $SS > $O/pmc_WRITE.log 2>&1 || exit 1
env $SER timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pmc_clk_serial -o run -- python3 $R/bench.py --steps 2 --warmup 1 --lean --overlap 0 $SS > $O/pmc_clk_serial.log 2>&1 || exit 1
echo "pmc done"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_stage -o run -- python3 $R/tools/stage_bench.py 256 > $O/prof_stage.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_stage_FETCH -o runc -- python3 $R/tools/stage_bench.py 256 > $O/pmc_stage_FETCH.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_stage_WRITE -o runc -- python3 $R/tools/stage_bench.py 256 > $O/pmc_stage_WRITE.log 2>&1 || exit 1
echo "stage passes done"
cd $R
# per-launch shapes and HIP-event times of the serial profile step (the per-shape table of DESIGN.md section 6), no profiler attached
PC_PROFILE_CSV=$P/${TAG}_conv_launches_bench_b32.csv timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --lean > $O/prof_launches.log 2>&1 || exit 1
grep -h '^{' $O/prof_launches.log | tail -1 > $P/${TAG}_bench_lean_no_profiler.json
python3 tools/rocpd_stats.py $O/prof_stats_default > $P/${TAG}_kernel_stats_bench_b32_default.csv
python3 tools/rocpd_stats.py $O/prof_stats_1lane > $P/${TAG}_kernel_stats_bench_b32_1lane.csv
python3 tools/pmc_clock.py $O/pmc_clk_serial > $P/${TAG}_clock_mfma_util_serial.json
grep -h '^{' $O/prof_stats_default.log | tail -1 > $P/${TAG}_bench_under_rocprof_default.json
python3 tools/overlap_mfma.py $O/prof_stats_default $P/${TAG}_bench_under_rocprof_default.json > $P/${TAG}_overlap_schedule_mfma.json
python3 tools/pmc_traffic.py $O/pmc_FETCH $O/pmc_WRITE > $P/${TAG}_hbm_traffic.json
python3 tools/traffic_by_shape.py $O/pmc_FETCH $O/pmc_WRITE $O/pmc_FETCH_launches.csv $P/${TAG}_conv_launches_bench_b32.csv > $P/${TAG}_traffic_by_shape.json
python3 tools/stage_rocprof.py $O/prof_stage 256 $O/pmc_stage_FETCH $O/pmc_stage_WRITE > $P/${TAG}_stage_kernels_rocprof.json
grep -h '^{' $O/prof_stats_1lane.log | tail -1 > $P/${TAG}_bench_under_rocprof_1lane.json
grep -h '^{' $O/prof_stage.log > $P/${TAG}_stage_hbm_roofline_b256_events.jsonl
ls -la $P
