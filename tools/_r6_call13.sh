cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_vm2.py tests/test_gpu_tracegen_tables.py tests/test_gpu_int256.py tests/test_gpu_modular.py tests/test_gpu_ecc.py tests/test_gpu_fp2.py tests/test_gpu_vm_slice.py tests/test_gpu_chipset.py -q -m gpu -x > gpurun_out/r6_call13_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r6_call13_pytest.txt
tail -4 gpurun_out/r6_call13_pytest.txt
for g in "2800000 20" "8192 20 mixed" "512 20 mem"; do
  ZKHIP_LANES=3 python3 tools/guest_bench2.py $g > /dev/null 2>&1
  for i in 1 2; do ZKHIP_LANES=3 python3 tools/guest_bench2.py $g | python3 -c "
import sys,json
g=json.loads(sys.stdin.read().strip().splitlines()[-1]); secs=(g['segment_tracegen_and_proving_ms']+g['aggregation_setup_wait_ms']+g['aggregation_ms'])/1e3
print('$g  %.2f M instr/s seg_ms %d agg_ms %d' % (g['total_cycles']/secs/1e6, g['segment_tracegen_and_proving_ms'], g['aggregation_ms']))"; done
done
