#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc_clk_serial $O/pmc_clk_default
SER="PC_LANES=1 PC_DUAL_STREAM=0 PC_PIPELINE=0"
env $SER timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pmc_clk_serial -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --overlap 0 > $O/pmc_clk_serial.log 2>&1 || { tail -5 $O/pmc_clk_serial.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pmc_clk_default -o run -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/pmc_clk_default.log 2>&1 || { tail -5 $O/pmc_clk_default.log; exit 1; }
cd $R
python3 tools/pmc_clock.py $O/pmc_clk_serial > $O/r02_j_clock_mfma_util_serial.json && cat $O/r02_j_clock_mfma_util_serial.json
python3 tools/pmc_clock.py $O/pmc_clk_default > $O/r02_j_clock_mfma_util_default.json && cat $O/r02_j_clock_mfma_util_default.json
ls $O/pmc_clk_serial | head
