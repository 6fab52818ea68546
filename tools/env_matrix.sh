#!/bin/bash
# The GPU test suite (minus the full-size Config 4 / 5 cases) under the run-time switches that select kernels, tiles, stages, lanes,
# pipelining and host threads: every configuration must give the same bits.  usage (through gpurun): bash tools/env_matrix.sh > log
# Since round 4 the switches exist only in the TUNING build of the library (csrc/Makefile `tuning`: -DPC_TUNING -> libpcodec_tuning.so, the
# same sources); the matrix runs the suite against it (PC_LIB).  The product library ignores these variables.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
export PC_LIB=$R/progressivecodec_amd/libpcodec_tuning.so
[ -f "$PC_LIB" ] || { echo "build the tuning library first: make -C progressivecodec_amd/csrc tuning"; exit 2; }
fail=0
ONLY=${1:-}
run() {
  if [ -n "$ONLY" ] && ! echo "$*" | grep -qE "$ONLY"; then return; fi
  echo "== $*"
  env "$@" timeout -k 10 400 python -m pytest tests -x -q -m gpu -k "not config4 and not config5 and not 4k_frame and not large_tiles" > /tmp/env_matrix_one.log 2>&1
  rc=$?
  tail -1 /tmp/env_matrix_one.log
  if [ $rc -ne 0 ]; then fail=1; grep -nE "^E |^(FAILED|ERROR)|Error" /tmp/env_matrix_one.log | head -40; fi
}
run PC_CONV_POLICY=0
run PC_CONV_POLICY=1
run PC_CONV_POLICY=2
run PC_CONV_POLICY=3 PC_CONV_SMALL_THR=100000
run PC_CONV_BK=16 PC_CONV_S=2 PC_CONV_TM_THR=100000000
run PC_CONV_BK=16 PC_CONV_S=4 PC_CONV_TM_THR=100000000
run PC_CONV_S=2
run PC_CONV_S=3 PC_CONV_TM=2
run PC_CONV_TM=2 PC_CONV_TN=2
run PC_CONV_ROWPERM=0
run PC_CONV_ROWPERM_MIN=0
run PC_CONV_GROUP_XCD=0
run PC_PREP_SCALAR=1
run PC_DEC_FAST=0
run PC_LANES=1
run PC_LANES=3
run PC_GROUPED=0 PC_DUAL_STREAM=0 PC_LANES=1
run PC_PIPELINE=0
run PC_PIPELINE_DEC=0
run PC_HOST_THREADS=1
run PC_HYPER_PARALLEL=1 PC_NO_STREAMED_ENCODE=0
run GPU_MAX_HW_QUEUES=16
echo "matrix failures: $fail"
exit $fail
