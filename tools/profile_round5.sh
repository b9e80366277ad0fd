# Round-5 profiles of the headline bench (2^22 x 300, reference FRI parameters): kernel stats single stream and three in flight,
# HBM traffic (FETCH_SIZE / WRITE_SIZE) and the VALU counters, every --pmc in a run of its own.  Writes gpurun_out/r5prof/.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r5prof
mkdir -p $O
B="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --inflight 1"
rocprofv3 --kernel-trace --stats -d $O/a -o a --output-format csv -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --inflight 1 > $O/a_bench.json 2> $O/a_err.txt
rocprofv3 --kernel-trace --stats -d $O/b -o b --output-format csv -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline > $O/b_bench.json 2> $O/b_err.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f -o f --output-format csv -- $B > $O/f_bench.json 2> $O/f_err.txt
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/w -o w --output-format csv -- $B > $O/w_bench.json 2> $O/w_err.txt
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES -d $O/v/v1 -o v --output-format csv -- $B > $O/v_bench.json 2> $O/v_err.txt
rocprofv3 --pmc SQ_ACTIVE_INST_VALU -d $O/v/v2 -o v --output-format csv -- $B > /dev/null 2> $O/v2_err.txt
rocprofv3 --pmc SQ_BUSY_CYCLES -d $O/v/v3 -o v --output-format csv -- $B > /dev/null 2> $O/v3_err.txt
rocprofv3 --pmc GRBM_GUI_ACTIVE -d $O/v/v4 -o v --output-format csv -- $B > /dev/null 2> $O/v4_err.txt
FC=$(find $O/f -name "*counter_collection.csv" | head -1); WC=$(find $O/w -name "*counter_collection.csv" | head -1)
python3 tools/pmc_traffic.py $FC $WC 4 $O/pmc_traffic.json
python3 tools/pmc_valu3.py $O/v 4 $O/pmc_valu.json profiles/round05_isa_mix_hash_rows.json 2>&1 | tail -8
cp $(find $O/a -name "*kernel_stats.csv" | head -1) $O/a_kernel_stats.csv
cp $(find $O/b -name "*kernel_stats.csv" | head -1) $O/b_kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O
head -5 $O/a_kernel_stats.csv; cat $O/b_bench.json | cut -c1-300; tail -3 $O/v2_err.txt $O/v3_err.txt $O/v4_err.txt
