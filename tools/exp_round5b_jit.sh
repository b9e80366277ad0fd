# more chips through compiled constraint kernels (zkhip_config.jit_min_log_work): first run compiles into the on-disk cache, the later ones show the steady state
run() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], {k:d.get(k) for k in ('prove_speed_mhz','segments','execution_ms','segment_tracegen_and_proving_ms','aggregation_ms','wall_s','sum_prove_ms_per_shape')})" "$1"; }
for w in 26 22 18; do
export ZKHIP_JIT_MIN_LOG_WORK=$w
for i in 1 2 3; do ZKHIP_LANES=3 python tools/guest_bench2.py 8192 19 mixed | run "mixed jit_min_log_work=$w run$i"; done
done
