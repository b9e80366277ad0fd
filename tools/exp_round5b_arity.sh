run() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], {k:d.get(k) for k in ('prove_speed_mhz','segments','levels','execution_ms','segment_tracegen_and_proving_ms','aggregation_ms','wall_s','node_log_heights','tree_nodes_per_device_slot')})" "$1"; }
for a in 3 4; do
export ZKHIP_INTERNAL_ARITY=$a
for i in 1 2; do ZKHIP_LANES=3 python tools/guest_bench2.py 8192 19 mixed | run "mixed arity$a"; done
for i in 1 2; do ZKHIP_LANES=3 python tools/guest_bench2.py 2800000 19 | run "fib arity$a"; done
done
