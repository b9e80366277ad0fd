cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
python -m pytest tests -q -m gpu > gpurun_out/r6_final_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r6_final_pytest.txt
tail -4 gpurun_out/r6_final_pytest.txt
python3 bench.py > gpurun_out/r6_final_bench.json 2> gpurun_out/r6_final_bench.err
python3 -c "
import json
for l in open('gpurun_out/r6_final_bench.json'):
    if l.startswith('{'):
        d=json.loads(l); print('bench', d['value'], d['ms_per_step'], d['cpu_baseline'].get('proof_bytes_equal_gpu'), [ (g, (d.get(g) or {}).get('value')) for g in ('guest_flow','guest_flow_chunk_config','guest_flow_mixed','guest_flow_mixed_frame19','guest_flow_memory_bound')])
"
for n in 2 4; do python3 bench.py --no-guest-flow --no-cpu-baseline --inflight $n --steps 12 --warmup 3 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('inflight $n', d['value'], d['ms_per_step'])
"; done
FLOW_STRESS_TAG=r6_final FLOW_STRESS_ENV=ZKHIP_SELF_CHECK=1 bash tools/flow_stress.sh 300 14
