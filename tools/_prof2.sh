set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_r1n1
PC_LANES=1 PC_DUAL_STREAM=0 timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/prof_r1n1 -o run -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/prof_r1n1.log 2>&1 || exit 1
grep -m1 '"metric"' $O/prof_r1n1.log | cut -c1-200
