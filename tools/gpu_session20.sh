#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc_lds
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_lds -o run -- python3 $R/tools/conv_tune.py stackg_L1 stackg_L3 ga_conv2 ru_3x3 > $O/pmc_lds.log 2>&1 || { tail -5 $O/pmc_lds.log; exit 1; }
cd $R
python3 - <<'PY'
import csv,glob,collections
d='gpurun_out/pmc_lds'
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(d+'/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'conv_igemm' not in r['Kernel_Name']: continue
        key=(r['Kernel_Name'][:60], r.get('Grid_Size',''))
        agg[key][r['Counter_Name']]+=float(r['Counter_Value']); agg[key]['_n']+=1.0/8
for k,v in agg.items():
    print(k, {a: round(b) for a,b in v.items()})
    if v.get('SQ_LDS_IDX_ACTIVE'): print("   bank conflict / idx active = %.3f"%(v['SQ_LDS_BANK_CONFLICT']/v['SQ_LDS_IDX_ACTIVE']), " wait_any/wave_cycles = %.3f"%(v['SQ_WAIT_ANY']/v['SQ_WAVE_CYCLES']), " wait_inst_any = %.3f"%(v['SQ_WAIT_INST_ANY']/v['SQ_WAVE_CYCLES']), " wait_inst_lds = %.3f"%(v['SQ_WAIT_INST_LDS']/v['SQ_WAVE_CYCLES']))
PY
