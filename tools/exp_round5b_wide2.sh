# interleaved A/B at frames of 2^20 after warm_lanes (the other lanes' segment keys built at setup for the shapes the key cache knows): lanes 2 / 3 / 4;
# ZKHIP_WIDE_IN_FLIGHT=2 with three lanes beside them
run() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d.get('prove_speed_mhz'), d.get('segment_tracegen_and_proving_ms'), d.get('aggregation_ms'), d.get('wall_s'))" "$1"; }
ZKHIP_LANES=3 python tools/guest_bench2.py 8192 20 mixed > /dev/null
for i in 1 2 3; do
for l in 2 3 4; do ZKHIP_LANES=$l python tools/guest_bench2.py 8192 20 mixed | run "mixed lanes$l"; done
ZKHIP_WIDE_IN_FLIGHT=2 ZKHIP_LANES=3 python tools/guest_bench2.py 8192 20 mixed | run "mixed lanes3 wide2"
done
for i in 1 2 3; do for l in 2 3; do ZKHIP_LANES=$l python tools/guest_bench2.py 2800000 20 chunk | run "fib chunkcfg lanes$l"; ZKHIP_LANES=$l python tools/guest_bench2.py 2800000 20 | run "fib lanes$l"; done; done
for i in 1 2; do ZKHIP_LANES=3 python tools/guest_bench2.py 512 20 mem | run "mem lanes3"; done
ZKHIP_LANES=3 python tools/guest_bench2.py 16384 20 mixed | run "mixed27M lanes3"
