cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r6_call6.txt
: > $O
# the shared-rows constraint kernel forced for every compiled chip of >= 2^12 LDE rows: parity suites that cover VAR / PERM / PREP / CHAL leaves
ZKHIP_JIT_SHARED=1 ZKHIP_FORCE_JIT=1 timeout 1200 python -m pytest tests/test_gpu_stark.py tests/test_gpu_chipset.py tests/test_gpu_logup.py tests/test_gpu_prep.py tests/test_gpu_cached_main.py tests/test_gpu_refshape.py -x -q -m gpu > gpurun_out/r6_call6_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r6_call6_pytest.txt
tail -6 gpurun_out/r6_call6_pytest.txt
for v in "ZKHIP_X=1" "ZKHIP_JIT_SHARED=0"; do
  echo "== bench $v" >> $O
  env $v python3 bench.py --no-guest-flow --no-cpu-baseline --steps 9 --warmup 3 2>>$O | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); s=d.get('stage_ms_single_stream',{})
        print(json.dumps({'value':d['value'],'ms_per_step':d['ms_per_step'],'quot_jit':s.get('quotient_eval_jit'),'stages':dict(list(s.items())[:8])}))
" >> $O
done
for lanes in 3 4 5 6; do
  for g in "2800000 20" "512 20 mem" "8192 20 mixed"; do
    echo "== flow $g lanes $lanes" >> $O
    ZKHIP_LANES=$lanes python3 tools/guest_bench2.py $g > /dev/null 2>&1
    ZKHIP_LANES=$lanes python3 tools/guest_bench2.py $g >> $O 2>&1
  done
done
grep -v amdgpu.ids $O | python3 -c "
import sys,json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('=='): print(l)
    elif l.startswith('{\"total'):
        g=json.loads(l); secs=(g['segment_tracegen_and_proving_ms']+g['aggregation_setup_wait_ms']+g['aggregation_ms'])/1e3
        print('   %.2f M instr/s  seg_ms %d agg_ms %d exec_wait %d' % (g['total_cycles']/secs/1e6, g['segment_tracegen_and_proving_ms'], g['aggregation_ms'], g['execution_ms']))
    elif l.startswith('{'): print('  ', l[:400])
"
