cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r6_call3.txt
: > $O
for u in 4 1 2 8; do
  echo "== unroll $u" >> $O
  ZKHIP_JIT_UNROLL=$u python3 bench.py --no-guest-flow --no-cpu-baseline --steps 9 --warmup 3 2>>$O | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); s=d.get('stage_ms_single_stream',{})
        print(json.dumps({'value':d['value'],'ms_per_step':d['ms_per_step'],'verified':d.get('verified'),'quot_jit':s.get('quotient_eval_jit'),'stages':dict(list(s.items())[:12])}))
" >> $O
done
python -m pytest tests/test_gpu_config_forms.py tests/test_gpu_agg_cache.py tests/test_gpu_merkle_stress.py tests/test_gpu_stark.py -x -q -m gpu > gpurun_out/r6_call3_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r6_call3_pytest.txt
tail -15 gpurun_out/r6_call3_pytest.txt
cat $O
