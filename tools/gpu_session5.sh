#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/r02_counters_list.txt 2>&1
grep -c . $O/r02_counters_list.txt
for K in 1; do
rm -rf $O/pmc_sqA$K $O/pmc_tcc$K $O/pmc_sqB$K
PC_CONV_KERN=$K PC_CONV_S=3 PC_TUNE_ITERS=3 timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_sqA$K -o run -- python3 $R/tools/conv_tune.py ga_conv2 stackg_L1 ru_3x3 > $O/pmc_sqA$K.log 2>&1 || echo "sqA$K failed"
PC_CONV_KERN=$K PC_CONV_S=3 PC_TUNE_ITERS=3 timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_INSTS_MFMA --output-format csv -d $O/pmc_sqB$K -o run -- python3 $R/tools/conv_tune.py ga_conv2 stackg_L1 ru_3x3 > $O/pmc_sqB$K.log 2>&1 || echo "sqB$K failed"
PC_CONV_KERN=$K PC_CONV_S=3 PC_TUNE_ITERS=3 timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_tcc$K -o run -- python3 $R/tools/conv_tune.py ga_conv2 stackg_L1 ru_3x3 > $O/pmc_tcc$K.log 2>&1 || echo "tcc$K failed"
done
ls -R $O/pmc_sqA1 | head; tail -3 $O/pmc_sqA1.log
