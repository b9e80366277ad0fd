cd ${GRAFT_REPO_ROOT:-.}
FRAME=19 bash tools/profile_round5_mixed_trace.sh > gpurun_out/r6_mixed_trace.log 2>&1
cd gpurun_out/r5mixed19_trace && python3 ../../tools/trace_split_proofs.py 6 > ../r6_mixed_1lane_per_proof.txt 2>&1; cd ../..
head -60 gpurun_out/r6_mixed_1lane_per_proof.txt | cut -c1-260
python3 -c "
import json
g=json.loads(open('gpurun_out/r5mixed19_trace/profiled.json').read().strip().splitlines()[-1]); print({k:g.get(k) for k in ('segments','chips_per_shape','segments_per_shape','levels')})"
rm -f gpurun_out/r5mixed19_trace/trace_compact.csv.gz
