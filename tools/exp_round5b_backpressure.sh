# lanes 2 / 3 with and without the lanes' look at the tree's backlog (FlowOptions::tree_backpressure: no new segment proof while ZKHIP_TREE_BACKLOG or more
# segment proofs wait for their leaf node) -- frames of 2^20, three node pipelines
run() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d.get('prove_speed_mhz'), d.get('segment_tracegen_and_proving_ms'), d.get('aggregation_ms'))" "$1"; }
for l in 2 3; do
for b in 4 off; do
if [ $b = off ]; then export ZKHIP_NO_TREE_BACKPRESSURE=1; else unset ZKHIP_NO_TREE_BACKPRESSURE; export ZKHIP_TREE_BACKLOG=$b; fi
for i in 1 2 3; do ZKHIP_LANES=$l python tools/guest_bench2.py 8192 20 mixed | run "mixed lanes$l backlog=$b"; done
for i in 1 2 3; do ZKHIP_LANES=$l python tools/guest_bench2.py 2800000 20 | run "fib lanes$l backlog=$b"; done
ZKHIP_LANES=$l python tools/guest_bench2.py 512 20 mem | run "mem lanes$l backlog=$b"
ZKHIP_LANES=$l python tools/guest_bench2.py 2800000 20 chunk | run "fib chunkcfg lanes$l backlog=$b"
done
done
