cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r6_call4.txt
: > $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "coset_lde or lde_properties" > gpurun_out/r6_call4_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r6_call4_pytest.txt
tail -30 gpurun_out/r6_call4_pytest.txt
for v in "ZKHIP_X=1" "ZKHIP_LDE_FUSED_WAVES=4" "ZKHIP_NO_LDE_FUSED=1"; do
  echo "== $v" >> $O
  env $v python3 bench.py --no-guest-flow --no-cpu-baseline --steps 9 --warmup 3 2>>$O | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); s=d.get('stage_ms_single_stream',{})
        print(json.dumps({'value':d['value'],'ms_per_step':d['ms_per_step'],'verified':d.get('verified'),'stages':dict(list(s.items())[:14])}))
" >> $O
done
cat $O | grep -v amdgpu.ids
