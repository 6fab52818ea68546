#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
run() { echo "== $*"; env "$@" timeout -k 10 300 python bench.py --steps 60 --warmup 3 --no-cpu-baseline > $O/r02_h_tmp.log 2>&1 || { tail -5 $O/r02_h_tmp.log; exit 1; }; tail -1 $O/r02_h_tmp.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['serial_step_ms'], j['roofline']['achieved'])"; }
for rep in 1 2; do
run GPU_MAX_HW_QUEUES=12
run GPU_MAX_HW_QUEUES=16
run GPU_MAX_HW_QUEUES=24
run GPU_MAX_HW_QUEUES=32
run GPU_MAX_HW_QUEUES=16 PC_CONV_POLICY=0
echo "== 16 overlap 0"; GPU_MAX_HW_QUEUES=16 timeout -k 10 300 python bench.py --steps 60 --warmup 3 --no-cpu-baseline --overlap 0 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"
done
