#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trace_ov1 $O/trace_ov0
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_ov1 -o run -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/trace_ov1.log 2>&1 || { tail -5 $O/trace_ov1.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_ov0 -o run -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --overlap 0 > $O/trace_ov0.log 2>&1 || { tail -5 $O/trace_ov0.log; exit 1; }
cd $R
tail -1 $O/trace_ov1.log | cut -c1-120; python3 tools/gpu_busy.py $O/trace_ov1 0.45 0.75
tail -1 $O/trace_ov0.log | cut -c1-120; python3 tools/gpu_busy.py $O/trace_ov0 0.45 0.75
