# at most ZKHIP_WIDE_IN_FLIGHT segment proofs of wrapped shapes at a time (FlowOptions::wide_in_flight) -- frames of 2^20, three lanes, three node pipelines
run() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d.get('prove_speed_mhz'), d.get('segment_tracegen_and_proving_ms'), d.get('aggregation_ms'))" "$1"; }
for w in 2 0 1; do
export ZKHIP_WIDE_IN_FLIGHT=$w ZKHIP_LANES=3
for i in 1 2 3; do python tools/guest_bench2.py 8192 20 mixed | run "mixed wide_in_flight=$w"; done
python tools/guest_bench2.py 16384 20 mixed | run "mixed27M wide_in_flight=$w"
python tools/guest_bench2.py 8192 19 mixed | run "mixed frame19 wide_in_flight=$w"
for i in 1 2; do python tools/guest_bench2.py 2800000 20 | run "fib wide_in_flight=$w"; done
python tools/guest_bench2.py 2800000 20 chunk | run "fib chunkcfg wide_in_flight=$w"
done
