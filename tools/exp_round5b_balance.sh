# lanes x node pipelines at frames of 2^20: the tree must keep up with the segment proofs (a tree that lags is finished alone on an under-filled GPU)
run() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d.get('prove_speed_mhz'), d.get('segment_tracegen_and_proving_ms'), d.get('aggregation_ms'))" "$1"; }
for combo in "3 3" "2 3" "2 4" "3 5" "3 6" "4 6" "4 8"; do
set -- $combo
for i in 1 2 3; do ZKHIP_LANES=$1 ZKHIP_AGG_SLOTS=$2 python tools/guest_bench2.py 8192 20 mixed | run "mixed lanes$1 slots$2"; done
for i in 1 2; do ZKHIP_LANES=$1 ZKHIP_AGG_SLOTS=$2 python tools/guest_bench2.py 2800000 20 | run "fib lanes$1 slots$2"; done
done
