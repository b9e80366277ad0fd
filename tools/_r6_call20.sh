cd ${GRAFT_REPO_ROOT:-.}
for v in "ZKHIP_ROWS_COOP_MAX_LOG=15" "ZKHIP_ROWS_COOP_MAX_LOG=17" "ZKHIP_ROWS_COOP_MAX_LOG=19" "ZKHIP_ROWS_COOP_MAX_LOG=13" "ZKHIP_HASH_BLOCK=128"; do
  env $v ZKHIP_LANES=3 python3 tools/guest_bench2.py 8192 20 mixed > /dev/null 2>&1
  for i in 1 2; do env $v ZKHIP_LANES=3 python3 tools/guest_bench2.py 8192 20 mixed | python3 -c "
import sys,json
g=json.loads(sys.stdin.read().strip().splitlines()[-1]); secs=(g['segment_tracegen_and_proving_ms']+g['aggregation_setup_wait_ms']+g['aggregation_ms'])/1e3
print('$v  %.2f M instr/s seg_ms %d agg_ms %d' % (g['total_cycles']/secs/1e6, g['segment_tracegen_and_proving_ms'], g['aggregation_ms']))"; done
done
