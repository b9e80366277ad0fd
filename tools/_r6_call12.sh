cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
O=gpurun_out/r6_call12.txt
: > $O
for v in "ZKHIP_X=1" "ZKHIP_NO_HOST_SPONGE=1" "ZKHIP_JIT_MIN_LOG_WORK=22" "ZKHIP_FORCE_JIT=1" "ZKHIP_HOST_SPONGE_MIN_WORDS=65536"; do
  for g in "8192 20 mixed" "2800000 20"; do
    echo "== flow $g | $v" >> $O
    env $v ZKHIP_LANES=3 python3 tools/guest_bench2.py $g > /dev/null 2>&1
    env $v ZKHIP_LANES=3 python3 tools/guest_bench2.py $g >> $O 2>&1
    env $v ZKHIP_LANES=3 python3 tools/guest_bench2.py $g >> $O 2>&1
  done
done
python3 - <<'PY'
import json
for l in open('gpurun_out/r6_call12.txt'):
    l=l.strip()
    if l.startswith('=='): print(l)
    elif l.startswith('{"total'):
        g=json.loads(l); secs=(g['segment_tracegen_and_proving_ms']+g['aggregation_setup_wait_ms']+g['aggregation_ms'])/1e3
        print('   %.2f M instr/s  seg_ms %d agg_ms %d wall %.2f' % (g['total_cycles']/secs/1e6, g['segment_tracegen_and_proving_ms'], g['aggregation_ms'], g['wall_s']))
    elif l: print('   ', l[:200])
PY
