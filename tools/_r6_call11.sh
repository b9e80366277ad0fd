cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
(time python3 bench.py) > gpurun_out/r6_bench_default.json 2> gpurun_out/r6_bench_default.err
tail -3 gpurun_out/r6_bench_default.err
python3 - <<'PY'
import json
for l in open('gpurun_out/r6_bench_default.json'):
    if l.startswith('{'):
        d=json.loads(l)
        print({k: d[k] for k in ('metric','value','ms_per_step','n_gpus','steps')})
        print('roofline', {k: d['roofline'][k] for k in ('bound','achieved','peak','frac','traffic')})
        print('valu whole proof', d.get('roofline_valu_whole_proof'))
        print('cpu_baseline', {k: d['cpu_baseline'].get(k) for k in ('value','unit','cores','kind','proof_bytes_equal_gpu')})
        for g in ('guest_flow','guest_flow_chunk_config','guest_flow_mixed','guest_flow_mixed_frame19','guest_flow_memory_bound'):
            b=d.get(g) or {}
            print(g, b.get('value'), b.get('segments_per_shape'), 'retried', b.get('segments_retried_in_all_four_runs'), 'exec_ms', b.get('execution_ms'), b.get('note'))
PY
echo "== one task over a device list on one GPU"
ZKHIP_DEVICES=0,0 ZKHIP_LANES=3 python3 tools/guest_bench2.py 5600000 20 | python3 -c "
import sys,json
g=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:g.get(k) for k in ('total_cycles','segments','segments_per_lane','devices','tree_nodes_per_device_slot','execution_ms','segment_tracegen_and_proving_ms','executor_record_threads','executor_metered_pass_busy_ms','executor_record_passes_busy_ms_sum','segments_retried','verified')})"
