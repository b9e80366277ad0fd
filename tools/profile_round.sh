set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r2prof
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/a -o a --output-format csv -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --inflight 1 > $O/a_bench.json 2> $O/a_err.txt
rocprofv3 --kernel-trace --stats -d $O/b -o b --output-format csv -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline > $O/b_bench.json 2> $O/b_err.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f -o f --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --inflight 1 > $O/f_bench.json 2> $O/f_err.txt
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/w -o w --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --inflight 1 > $O/w_bench.json 2> $O/w_err.txt
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES -d $O/v -o v --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --inflight 1 > $O/v_bench.json 2> $O/v_err.txt
find $O -name "*.csv" | head -30
FC=$(find $O/f -name "*counter_collection.csv" | head -1); WC=$(find $O/w -name "*counter_collection.csv" | head -1); VC=$(find $O/v -name "*counter_collection.csv" | head -1)
python3 tools/pmc_traffic.py $FC $WC 4 $O/pmc_traffic.json
VT=$(find $O/v -name "*kernel_trace.csv" | head -1)
python3 tools/pmc_valu.py $VC $VT 4 $O/pmc_valu.json 2>&1 | tail -12
cp $(find $O/a -name "*kernel_stats.csv" | head -1) $O/a_kernel_stats.csv
cp $(find $O/b -name "*kernel_stats.csv" | head -1) $O/b_kernel_stats.csv
# keep the merge small: drop the raw traces
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O
head -12 $O/a_kernel_stats.csv; head -6 $O/b_kernel_stats.csv; cat $O/b_bench.json | cut -c1-400
