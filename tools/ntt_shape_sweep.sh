cd $GRAFT_REPO_ROOT
# per-pass stage limit: 11 = two passes of [2^11 x 8] tiles (32-byte row segments); 10 / 9 = three passes with wider rows
for r in 11 10 9; do
  ZKHIP_NTT_MAX_LOG_R=$r python bench.py --inflight 1 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d['stage_ms_single_stream']; print('max_log_r $r', d['ms_per_step'], 'ms/proof; ntt fwd', s.get('ntt_pass_fwd'), 'inv', s.get('ntt_pass_inv'), 'bitrev', s.get('bitrev_scale'), 'verified', d['config']['verified'])"
done
ZKHIP_NTT_MAX_LOG_R=9 python bench.py --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('3-inflight max_log_r 9:', d['value'], d['ms_per_step'])"
