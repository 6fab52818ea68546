#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
SH="stackg_L4 stackg_L5 stack_L4 stack_L5 hs_8x8 hs_4x4 stackg_L3 wam16_1x1"
t() { echo "== $*"; env "$@" timeout -k 10 300 python tools/conv_tune.py $SH 2>&1 | grep TFLOP; }
t PC_CONV_POLICY=1
t PC_CONV_POLICY=0
t PC_CONV_POLICY=0 PC_CONV_S=4
t PC_CONV_BK=16 PC_CONV_S=4 PC_CONV_TM_THR=100000000
t PC_CONV_BK=16 PC_CONV_S=6 PC_CONV_TM_THR=100000000
t PC_CONV_POLICY=0 PC_CONV_S=2
