#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
SH="stackg_L1 stackg_L2 stackg_L3 stackg_L4 ga_conv2 ga_conv3 ru_3x3 ru_1x1 gdn_like wam16_3x3 gs_d6"
: > $O/r02_sweep2.log
run() { echo "## $1" >> $O/r02_sweep2.log; shift; env "$@" timeout -k 10 300 python tools/conv_tune.py $SH 2>&1 | grep TFLOP >> $O/r02_sweep2.log; }
run "uni S3" PC_CONV_KERN=1 PC_CONV_S=3
run "uni S3 prio" PC_CONV_KERN=1 PC_CONV_S=3 PC_CONV_DBG=512
run "uni S3 front" PC_CONV_KERN=1 PC_CONV_S=3 PC_CONV_DBG=16
run "uni S3 front prio" PC_CONV_KERN=1 PC_CONV_S=3 PC_CONV_DBG=528
run "uni S4" PC_CONV_KERN=1 PC_CONV_S=4
run "uni S2 prio" PC_CONV_KERN=1 PC_CONV_S=2 PC_CONV_DBG=512
run "uni 2x1 S2 prio" PC_CONV_KERN=1 PC_CONV_S=2 PC_CONV_TM=2 PC_CONV_DBG=512
run "spec S3" PC_CONV_KERN=0 PC_CONV_S=3
run "spec S3 prio" PC_CONV_KERN=0 PC_CONV_S=3 PC_CONV_DBG=512
python - <<'PY'
import re,collections
rows=collections.OrderedDict(); cfgs=[]
for l in open('gpurun_out/r02_sweep2.log'):
    if l.startswith('##'): cfg=l[3:].strip(); cfgs.append(cfg); continue
    m=re.match(r'(\S+)\s+M=\s*(\d+) N=\s*(\d+) K=\s*(\d+) cfg 0:\s+([\d.]+) us\s+([\d.]+) TFLOP',l)
    if m: rows.setdefault((m.group(1),m.group(2),m.group(3),m.group(4)),{})[cfg]=float(m.group(6))
print('%-10s %7s %4s %5s | '%('shape','M','N','K')+' | '.join(cfgs))
for k,v in rows.items(): print('%-10s %7s %4s %5s | '%k+' '.join('%9.1f'%v.get(c,0) for c in cfgs))
PY
