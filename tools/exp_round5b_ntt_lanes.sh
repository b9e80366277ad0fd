# transform passes of at most 2^10 rows in workgroups of 1024 / 512 / 256 lanes (zkhip_config.ntt_log_lanes): parity tests under each setting, then the flows
run() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], {k:d.get(k) for k in ('prove_speed_mhz','segments','execution_ms','segment_tracegen_and_proving_ms','aggregation_ms','wall_s')})" "$1"; }
for t in 9 8; do
ZKHIP_NTT_LOG_LANES=$t python -m pytest tests/test_gpu_kernels.py tests/test_gpu_stark.py tests/test_gpu_chipset.py -x -q 2>&1 | tail -2
done
for t in 10 9 8; do
export ZKHIP_NTT_LOG_LANES=$t
for i in 1 2; do ZKHIP_LANES=3 python tools/guest_bench2.py 8192 19 mixed | run "mixed ntt_log_lanes=$t"; done
for i in 1 2; do ZKHIP_LANES=3 python tools/guest_bench2.py 2800000 19 | run "fib ntt_log_lanes=$t"; done
done
