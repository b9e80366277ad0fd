# mixed guest (and Fibonacci) at 3 .. 6 lanes / 3 .. 4 node pipelines after the second session's changes (rows in bulk, Horner rows, 2^20 / 2^17 nodes)
run() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], {k:d.get(k) for k in ('prove_speed_mhz','segments','execution_ms','segment_tracegen_and_proving_ms','aggregation_ms','wall_s','sum_prove_ms_per_shape')})" "$1"; }
for l in 3 4 5 6; do
for s in 3 4; do
ZKHIP_LANES=$l ZKHIP_AGG_SLOTS=$s python tools/guest_bench2.py 8192 19 mixed | run "mixed lanes$l slots$s"
done
done
for l in 3 4; do ZKHIP_LANES=$l python tools/guest_bench2.py 2800000 19 | run "fib lanes$l"; done
ZKHIP_NO_HOST_SPONGE=1 ZKHIP_LANES=3 python tools/guest_bench2.py 8192 19 mixed | run "mixed lanes3 device sponge"
