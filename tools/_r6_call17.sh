cd ${GRAFT_REPO_ROOT:-.}
bash tools/profile_round5_fib_trace.sh > gpurun_out/r6_fib_trace.log 2>&1
cd gpurun_out/r5fib19_trace && python3 ../../tools/trace_split_proofs.py 6 > ../r6_fib_1lane_per_proof.txt 2>&1; cd ../..
head -14 gpurun_out/r6_fib_1lane_per_proof.txt | cut -c1-230
rm -f gpurun_out/r5fib19_trace/trace_compact.csv.gz
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
