cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r6_call8.txt
: > $O
ZKHIP_JIT_SHARED=1 ZKHIP_FORCE_JIT=1 timeout 900 python -m pytest tests/test_gpu_chipset.py tests/test_gpu_logup.py tests/test_gpu_prep.py tests/test_gpu_cached_main.py -x -q -m gpu > gpurun_out/r6_call8_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r6_call8_pytest.txt
ZKHIP_JIT_SHARED=1 ZKHIP_JIT_SHARED_WAVES=8 ZKHIP_FORCE_JIT=1 timeout 900 python -m pytest tests/test_gpu_chipset.py tests/test_gpu_stark.py -x -q -m gpu >> gpurun_out/r6_call8_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r6_call8_pytest.txt
timeout 600 python -m pytest tests/test_gpu_config_forms.py -x -q -m gpu -k shared >> gpurun_out/r6_call8_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r6_call8_pytest.txt
grep "passed\|failed\|rc " gpurun_out/r6_call8_pytest.txt
for v in "ZKHIP_JIT_SHARED=1" "ZKHIP_JIT_SHARED=1 ZKHIP_JIT_SHARED_WAVES=8" "ZKHIP_JIT_SHARED=1 ZKHIP_JIT_SHARED_WAVES=4" "ZKHIP_JIT_SHARED=0"; do
  echo "== bench $v" >> $O
  env $v python3 bench.py --no-guest-flow --no-cpu-baseline --steps 9 --warmup 3 2>>$O | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); s=d.get('stage_ms_single_stream',{})
        print(json.dumps({'value':d['value'],'ms_per_step':d['ms_per_step'],'quot_jit':s.get('quotient_eval_jit'),'stages':dict(list(s.items())[:6])}))
" >> $O
done
grep -v amdgpu.ids $O
