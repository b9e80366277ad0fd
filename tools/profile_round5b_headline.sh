# the headline under rocprofv3 --kernel-trace --stats with the second session's final library: three proofs in flight (the default) and one at a time
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5bprof
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/b -o b --output-format csv -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-guest-flow --no-aggregate > $O/b_bench.json 2> $O/b_err.txt
rocprofv3 --kernel-trace --stats -d $O/a -o a --output-format csv -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-guest-flow --no-aggregate --inflight 1 > $O/a_bench.json 2> $O/a_err.txt
cp $(find $O/a -name "*kernel_stats.csv" | head -1) $O/a_kernel_stats.csv
cp $(find $O/b -name "*kernel_stats.csv" | head -1) $O/b_kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
head -4 $O/b_kernel_stats.csv | cut -c1-200; cut -c1-300 $O/b_bench.json
