#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
for pol in 0 1; do
echo "== tune policy $pol"; PC_CONV_POLICY=$pol timeout -k 10 300 python tools/conv_tune.py > $O/r02_l_tune_pol$pol.log 2>&1 || { tail -5 $O/r02_l_tune_pol$pol.log; exit 1; }; grep TFLOP $O/r02_l_tune_pol$pol.log
done
echo "== tune 16,4"; PC_CONV_BK=16 PC_CONV_S=4 PC_CONV_TM_THR=100000000 timeout -k 10 300 python tools/conv_tune.py > $O/r02_l_tune_164.log 2>&1; grep TFLOP $O/r02_l_tune_164.log
