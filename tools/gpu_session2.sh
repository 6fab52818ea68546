#!/bin/bash
# round-2 GPU session 2: MFMA rate probe; ablations of the unified and the wave-specialised conv kernels
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
echo "== rate probe"; timeout -k 10 120 tools/bin/mfma_rate_probe > $O/r02_rate_probe.log 2>&1 || { echo probe failed; }
cat $O/r02_rate_probe.log
SH="stackg_L1 ga_conv2 ru_3x3 stackg_L4"
: > $O/r02_ablate.log
for d in 0 1 2 4 8 10 3; do
  echo "## uni 1x1 S=3 DBG=$d" >> $O/r02_ablate.log
  PC_CONV_KERN=1 PC_CONV_S=3 PC_CONV_DBG=$d timeout -k 10 300 python tools/conv_tune.py $SH 2>&1 | grep TFLOP >> $O/r02_ablate.log
done
for d in 0 1 2 4; do
  echo "## spec S=3 DBG=$d" >> $O/r02_ablate.log
  PC_CONV_KERN=0 PC_CONV_S=3 PC_CONV_DBG=$d timeout -k 10 300 python tools/conv_tune.py $SH 2>&1 | grep TFLOP >> $O/r02_ablate.log
done
cat $O/r02_ablate.log
