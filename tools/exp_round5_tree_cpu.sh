# Is the aggregation tree behind the segment lanes because of the HOST (witness generation: ~20 ms of all 16 cores per node) when more lanes run?
# Fibonacci flow, lanes x witness threads x node pipelines; prints instructions per second ELF -> verified root, segments / tree tail ms.
cd $GRAFT_REPO_ROOT
nproc
run() { python3 tools/guest_bench2.py 700000 17 | python3 -c "
import json,sys; g=json.loads(sys.stdin.read().strip().splitlines()[-1]); secs=(g['segment_tracegen_and_proving_ms']+g['aggregation_setup_wait_ms']+g['aggregation_ms'])/1e3; print('$1', round(g['total_cycles']/secs), 'segments_ms', g['segment_tracegen_and_proving_ms'], 'tree_ms', g['aggregation_ms'], 'exec_ms', g['execution_ms'])"; }
ZKHIP_LANES=3 run "warm"
for l in 3 4 5; do for w in 4 8 16; do ZKHIP_LANES=$l ZKHIP_WITNESS_THREADS=$w run "lanes=$l witness_threads=$w"; done; done
for l in 4 5; do ZKHIP_LANES=$l ZKHIP_WITNESS_THREADS=8 ZKHIP_AGG_SLOTS=2 run "lanes=$l witness_threads=8 agg_slots=2"; done
