#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; cd $R
for cfg in "PC_CONV_S=2" "PC_CONV_POLICY=1" "A=0"; do
 for rep in 1 2 3; do
  echo "== $cfg rep $rep"; env $cfg timeout -k 10 200 python -m pytest tests/test_gpu_codec.py -x -q -m gpu -k "side_by_side or two_threads" 2>&1 | grep -E "^E  |passed|failed" | head -4
 done
done
bash tools/env_matrix.sh "POLICY=[012]|S=2|S=4|TM=2|GROUPED=0" 2>&1 | tail -20
