# headline with the transform passes at wave priority 0 / 3, row-sponge workgroups of 256 / 768 lanes, 3 and 4 proofs in flight
for hb in 256 768; do for pr in 0 3; do for inf in 3 4; do
ZKHIP_HASH_BLOCK=$hb ZKHIP_NTT_PRIO=$pr python bench.py --steps 12 --no-cpu-baseline --no-aggregate --no-guest-flow --inflight $inf 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hash_block $hb ntt_prio $pr inflight $inf', d['value'], d['ms_per_step'], d['roofline']['valu'].get('measured_ms_per_launch'))"
done; done; done
