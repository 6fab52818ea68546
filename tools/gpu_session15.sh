#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
BASE=$R/tools/bin/libpcodec_base.so
echo "== ops tests"; timeout -k 10 240 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv or deconv or gdn" > $O/r02_k_ops.log 2>&1; rc=$?; tail -3 $O/r02_k_ops.log; [ $rc -eq 0 ] || exit 1
echo "== ops tests BK=16"; PC_CONV_BK=16 timeout -k 10 240 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv or deconv" > $O/r02_k_ops16.log 2>&1; rc=$?; tail -3 $O/r02_k_ops16.log; [ $rc -eq 0 ] || exit 1
SH="stackg_L1 stackg_L2 stackg_L3 stackg_L4 stackg_L5 ga_conv2 ru_3x3 ru_1x1 gdn_like gs_d6 wam16_3x3"
echo "== tune new"; timeout -k 10 300 python tools/conv_tune.py $SH > $O/r02_k_tune_new.log 2>&1 || { tail -5 $O/r02_k_tune_new.log; exit 1; }; grep TFLOP $O/r02_k_tune_new.log
echo "== tune base"; PC_LIB=$BASE timeout -k 10 300 python tools/conv_tune.py $SH > $O/r02_k_tune_base.log 2>&1 || { tail -5 $O/r02_k_tune_base.log; exit 1; }; grep TFLOP $O/r02_k_tune_base.log
for rep in 1 2; do for arm in "A=0" "PC_LIB=$BASE"; do
  v=$(env $arm timeout -k 10 300 python bench.py --steps 12 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['enc_ms'], j['dec_ms'], j['roofline']['achieved'])")
  echo "[$arm] MP/s ms/step enc dec convTF: $v"
done; done
