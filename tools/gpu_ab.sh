#!/bin/bash
# A/B of bench.py under environment switches on ONE box (the switches live in the tuning build: export PC_LIB=.../libpcodec_tuning.so first): usage  bash tools/gpu_ab.sh "NAME=VAL ..." "NAME=VAL ..." ...   (each arm run twice, interleaved)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
for rep in 1 2; do
  for arm in "$@"; do
    v=$(env $arm timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['enc_ms'], j['dec_ms'], j['roofline']['achieved'])")
    echo "[$arm] MP/s ms/step enc dec convTF: $v"
  done
done
