#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
echo "== ops parity spec2"; PC_CONV_KERN=0 timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu > $O/r02_ops_spec2.log 2>&1; echo "rc $?"; tail -1 $O/r02_ops_spec2.log
SH="stackg_L1 stackg_L2 stackg_L3 stackg_L4 stackg_L5 stack_L1 ga_conv2 ga_conv3 ga_conv4 ru_3x3 ru_1x1 ru_1x1a qkv gdn_like wam16_3x3 wam16_1x1 ha_0 gs_d6 gs_d3"
: > $O/r02_sweep4.log
run() { echo "## $1" >> $O/r02_sweep4.log; shift; env "$@" timeout -k 10 300 python tools/conv_tune.py $SH 2>&1 | grep TFLOP >> $O/r02_sweep4.log; }
run "spec2 S3" PC_CONV_KERN=0 PC_CONV_S=3
run "spec2 S2" PC_CONV_KERN=0 PC_CONV_S=2
run "spec2 S4" PC_CONV_KERN=0 PC_CONV_S=4
run "uni3 auto" PC_CONV_KERN=1
run "uni3 1x1 S3" PC_CONV_KERN=1 PC_CONV_S=3 PC_CONV_TM=1
python - <<'PY'
import re,collections
rows=collections.OrderedDict(); cfgs=[]
for l in open('gpurun_out/r02_sweep4.log'):
    if l.startswith('##'): cfg=l[3:].strip(); cfgs.append(cfg); continue
    m=re.match(r'(\S+)\s+M=\s*(\d+) N=\s*(\d+) K=\s*(\d+) cfg 0:\s+([\d.]+) us\s+([\d.]+) TFLOP',l)
    if m: rows.setdefault((m.group(1),m.group(2),m.group(3),m.group(4)),{})[cfg]=float(m.group(6))
print('%-10s %7s %4s %5s | '%('shape','M','N','K')+' | '.join(cfgs))
for k,v in rows.items(): print('%-10s %7s %4s %5s | '%k+' '.join('%9.1f'%v.get(c,0) for c in cfgs))
PY
echo "== bench spec2"; PC_CONV_KERN=0 timeout -k 10 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/r02_bench_spec2.log 2>&1; tail -1 $O/r02_bench_spec2.log | cut -c1-200
echo "== bench uni3 auto"; timeout -k 10 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/r02_bench_uni3.log 2>&1; tail -1 $O/r02_bench_uni3.log | cut -c1-200
