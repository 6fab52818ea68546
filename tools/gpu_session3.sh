#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
PC_CONV_KERN=1 PC_CONV_S=3 PC_CONV_DBG=64 timeout -k 10 300 python tools/conv_timeline.py stackg_L1 stackg_L4 ga_conv3 > $O/r02_timeline.log 2>&1
PC_CONV_KERN=1 PC_CONV_S=3 PC_CONV_DBG=72 timeout -k 10 300 python tools/conv_timeline.py stackg_L1 > $O/r02_timeline_nomem.log 2>&1
cut -c1-220 $O/r02_timeline.log | head -150
echo; echo "#### no DMA, no reads"; cut -c1-220 $O/r02_timeline_nomem.log | head -60
