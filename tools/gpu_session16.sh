#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
BASE=$R/tools/bin/libpcodec_base.so
echo "== ops tests"; timeout -k 10 240 python -m pytest tests/test_gpu_ops.py -x -q -m gpu > $O/r02_k_ops.log 2>&1; rc=$?; tail -3 $O/r02_k_ops.log; [ $rc -eq 0 ] || exit 1
echo "== codec tests"; timeout -k 10 600 python -m pytest tests/test_gpu_codec.py -x -q -m gpu > $O/r02_k_codec.log 2>&1; rc=$?; tail -12 $O/r02_k_codec.log; [ $rc -eq 0 ] || exit 1
for rep in 1 2; do for arm in "A=0" "PC_LIB=$BASE"; do
  v=$(env $arm timeout -k 10 300 python bench.py --steps 12 --warmup 2 --no-cpu-baseline 2>$O/r02_k_bench_err.log | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['enc_ms'], j['dec_ms'], j['roofline']['achieved'])")
  echo "[$arm] MP/s ms/step enc dec convTF: $v"
done; done
