cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c24
( time python bench.py --steps 20 --warmup 3 ) > gpurun_out/c24/bench.txt 2>&1
python3 - <<'PY'
import json
for l in open('gpurun_out/c24/bench.txt'):
    if l.startswith('{"metric"'):
        g=json.loads(l)
        print('value',g['value'],'ms',g['ms_per_step'])
        for k in g:
            if k.startswith('guest_flow'):
                f=g[k]; print(k, {q:f.get(q) for q in ('value','instr_per_s_of_the_three_measured_runs','execution_ms','segments_ms','tree_tail_ms','process_wall_s','segments','leaf_circuits_at_setup','leaf_circuits_on_demand')})
PY
tail -4 gpurun_out/c24/bench.txt | cut -c1-200
( time python -m pytest tests -m gpu -q -x ) > gpurun_out/c24/tests.txt 2>&1; tail -4 gpurun_out/c24/tests.txt
