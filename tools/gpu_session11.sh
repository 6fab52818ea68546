#!/bin/bash
# ops tests (new quantile / prep kernels) + stage bench under rocprof, with the scalar prep kernels as the same-box A/B
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
echo "== ops tests"; timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu > $O/r02_e_ops.log 2>&1; rc=$?; tail -5 $O/r02_e_ops.log; [ $rc -eq 0 ] || exit 1
echo "== stage bench (events)"; timeout -k 10 300 python tools/stage_bench.py 256 > $O/r02_e_stage_events.jsonl 2>&1 || exit 1; cut -c1-200 $O/r02_e_stage_events.jsonl
echo "== stage bench scalar prep (events)"; PC_PREP_SCALAR=1 timeout -k 10 300 python tools/stage_bench.py 256 > $O/r02_e_stage_events_scalar.jsonl 2>&1 || exit 1; cut -c1-200 $O/r02_e_stage_events_scalar.jsonl
cd /tmp && export TMPDIR=/tmp; rm -rf $O/prof_stage
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof_stage -o run -- python3 $R/tools/stage_bench.py 256 > $O/prof_stage.log 2>&1 || exit 1
cd $R; python3 tools/stage_rocprof.py $O/prof_stage 256 > $O/r02_e_stage_kernels_rocprof.json && cat $O/r02_e_stage_kernels_rocprof.json
echo "== codec tests"; timeout -k 10 900 python -m pytest tests/test_gpu_codec.py -x -q -m gpu > $O/r02_e_codec.log 2>&1; rc=$?; tail -3 $O/r02_e_codec.log; [ $rc -eq 0 ] || exit 1
