set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_r1l $O/pmc_l_FETCH $O/pmc_l_WRITE
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/prof_r1l -o run -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_r1l.log 2>&1 || exit 1
tail -1 $O/prof_r1l.log | cut -c1-300
PC_LANES=1 PC_DUAL_STREAM=0 timeout -k 10 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_l_FETCH -o runc -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_l_FETCH.log 2>&1 || exit 1
PC_LANES=1 PC_DUAL_STREAM=0 timeout -k 10 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_l_WRITE -o runc -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_l_WRITE.log 2>&1 || exit 1
find $O/prof_r1l -name "*kernel_trace.csv" -delete
find $O/pmc_l_FETCH $O/pmc_l_WRITE -name "*kernel_trace.csv" -delete
ls -la $O/prof_r1l/* | head
