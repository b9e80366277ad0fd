cd $GRAFT_REPO_ROOT
run() { python3 tools/guest_bench2.py 700000 17 | python3 -c "
import json,sys; g=json.loads(sys.stdin.read().strip().splitlines()[-1]); secs=(g['segment_tracegen_and_proving_ms']+g['aggregation_setup_wait_ms']+g['aggregation_ms'])/1e3; print('$1', round(g['total_cycles']/secs), 'segments_ms', g['segment_tracegen_and_proving_ms'], 'tree_ms', g['aggregation_ms'], 'wall', g['wall_s'], 'sum_prove', g['sum_over_segments_prove_ms'], 'sum_tracegen', g['sum_over_segments_tracegen_ms'])"; }
for l in 3 4 5 6; do ZKHIP_LANES=$l run "lanes=$l"; done
for s in 4 5; do ZKHIP_LANES=3 ZKHIP_AGG_SLOTS=$s run "lanes=3 agg_slots=$s"; done
for k in 24 22 20; do ZKHIP_LANES=3 ZKHIP_JIT_MIN_LOG_WORK=$k run "jit_min_log_work=$k (first: compiles)"; ZKHIP_LANES=3 ZKHIP_JIT_MIN_LOG_WORK=$k run "jit_min_log_work=$k"; done
bash tools/refshape_valu_floor.sh 2>&1 | tail -6
