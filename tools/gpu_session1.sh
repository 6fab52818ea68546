#!/bin/bash
# round-2 GPU session 1: issue-cost probe, parity of the unified conv kernel, per-shape sweep of kernel variants, bench
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
echo "== probe"; timeout -k 10 120 tools/bin/mfma_issue_probe > $O/r02_probe.log 2>&1 || { echo probe failed; tail -5 $O/r02_probe.log; }
echo "== ops parity (uni 1x1 default)"
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu > $O/r02_ops_uni11.log 2>&1; echo "rc $?"; tail -3 $O/r02_ops_uni11.log
for cfg in "2 1" "1 2" "2 2"; do set -- $cfg
  echo "== ops parity uni TM=$1 TN=$2"
  PC_CONV_TM=$1 PC_CONV_TN=$2 timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu > $O/r02_ops_uni$1$2.log 2>&1; echo "rc $?"; tail -3 $O/r02_ops_uni$1$2.log
done
SH="stackg_L1 stackg_L2 stackg_L3 stackg_L4 stackg_L5 stack_L1 ga_conv2 ga_conv3 ga_conv4 ru_3x3 ru_1x1 ru_1x1a qkv gdn_like wam16_3x3 wam16_1x1 ha_0 gs_d6 gs_d3"
echo "== sweep"; : > $O/r02_sweep1.log
echo "## KERN 0 (round-1 wave-specialised)" >> $O/r02_sweep1.log
PC_CONV_KERN=0 PC_CONV_DBG=256 timeout -k 10 300 python tools/conv_tune.py $SH >> $O/r02_sweep1.log 2>&1
for cfg in "1 1 3" "1 1 2" "2 1 3" "2 1 2" "1 2 3" "1 2 2" "2 2 3" "2 2 2"; do set -- $cfg
  echo "## KERN 1 uni TM=$1 TN=$2 S=$3" >> $O/r02_sweep1.log
  PC_CONV_KERN=1 PC_CONV_TM=$1 PC_CONV_TN=$2 PC_CONV_S=$3 PC_CONV_DBG=256 timeout -k 10 300 python tools/conv_tune.py $SH >> $O/r02_sweep1.log 2>&1
done
echo "== bench uni default"; timeout -k 10 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/r02_bench_uni11.log 2>&1; tail -1 $O/r02_bench_uni11.log | cut -c1-400
echo "== bench KERN 0"; PC_CONV_KERN=0 timeout -k 10 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/r02_bench_kern0.log 2>&1; tail -1 $O/r02_bench_kern0.log | cut -c1-400
