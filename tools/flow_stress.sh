# Runs the guest flow many times over (different lengths, both configurations, device lists) and counts what does not end with a verified
# root: the greedy fold, three node pipelines and six internal workers are threads racing for proofs -- a rare ordering bug shows up here.
# Usage: bash tools/flow_stress.sh [runs] [frame]   (writes gpurun_out/flow_stress.json)
# With a frame (e.g. 14: ~100 segment proofs per run) every run uses it, the host verifies every segment proof beside the proving
# (ZKHIP_VERIFY_SEGMENTS=1: a failure names the segment and the verifier's check) and extra variables pass through FLOW_STRESS_ENV
# (e.g. FLOW_STRESS_ENV=ZKHIP_SELF_CHECK=1) -- the form that found the stored tree node of DESIGN.md 15 (3 % of runs before the change).
# With a frame the runs also carry ZKHIP_NO_RETRY=1 (a failed segment proof ends the run instead of being made once more), and every run's
# JSON line is read: "retried" sums `segments_retried` over the runs and must stay 0 -- a retried wrong node cannot pass for a clean run.
# FLOW_STRESS_RETRY=1 leaves the retry on (to show that the counter sees what the retry would otherwise hide).
# FLOW_STRESS_LIB=test runs prove_cli over libzkhip_test.so (LD_PRELOAD; the A/B bodies of ZKHIP_TREE_STORE_EARLY exist only there).
cd ${GRAFT_REPO_ROOT:-.}
N=${1:-60}
FRAME=${2:-}
mkdir -p /tmp/fs gpurun_out
python3 tools/make_guest_files.py /tmp/fs 300000 > /dev/null
cp /tmp/fs/openvm.toml /tmp/fs_chunk.toml
ok=0; bad=0; retried=0; segs=0; t0=$(date +%s)
PRE=""; [ "$FLOW_STRESS_LIB" = test ] && PRE="LD_PRELOAD=$PWD/zkvm-prover_amd/libzkhip_test.so"
for i in $(seq 1 $N); do
  n=$((20000 + (i * 7919) % 400000))
  python3 -c "import sys; open('/tmp/fs/stdin.bin','wb').write(($n).to_bytes(4,'little'))"
  cfg=-; [ $((i % 5)) = 0 ] && cfg=/tmp/fs_chunk.toml
  lf=17; [ $((i % 3)) = 0 ] && lf=14
  env="ZKHIP_X=1"; [ $((i % 7)) = 0 ] && env="ZKHIP_DEVICES=0,0"
  if [ -n "$FRAME" ]; then lf=$FRAME; env="$env ZKHIP_VERIFY_SEGMENTS=1 $FLOW_STRESS_ENV"; [ -z "$FLOW_STRESS_RETRY" ] && env="$env ZKHIP_NO_RETRY=1"; fi
  out=/tmp/fs/o$i; mkdir -p $out
  if env $env $PRE timeout 300 ./zkvm-prover_amd/prove_cli prove-elf /tmp/fs/fib.elf /tmp/fs/stdin.bin $out $cfg $lf > $out/log.json 2> $out/err.txt \
     && ./zkvm-prover_amd/prove_cli verify-guest /tmp/fs/fib.elf $out/root.vk $([ $cfg = - ] && echo $out/openvm.toml || echo $cfg) $out/root.json > /dev/null 2>> $out/err.txt; then
    ok=$((ok+1))
    r=$(python3 -c "import json,sys; l=json.loads(open('$out/log.json').read().strip().splitlines()[-1]); print(l['segments_retried'], l['segments'])")
    retried=$((retried + ${r% *})); segs=$((segs + ${r#* }))
    [ "${r% *}" != 0 ] && { echo "run $i (n=$n cfg=$cfg lf=$lf $env) RETRIED ${r% *} segment(s)"; cp $out/err.txt gpurun_out/flow_stress_${FLOW_STRESS_TAG:-run}_retried_$i.txt; }
  else
    bad=$((bad+1)); echo "run $i (n=$n cfg=$cfg lf=$lf $env) FAILED: $(tail -c 300 $out/err.txt)"
    cp $out/err.txt gpurun_out/flow_stress_${FLOW_STRESS_TAG:-run}_fail_$i.txt   # (the self-check's whole diagnosis)
  fi
  rm -rf $out
done
echo "{\"runs\": $N, \"verified\": $ok, \"failed\": $bad, \"segments_retried\": $retried, \"segment_proofs\": $segs, \"library\": \"${FLOW_STRESS_LIB:-shipped}\", \"retry\": \"${FLOW_STRESS_RETRY:-off}\", \"seconds\": $(( $(date +%s) - t0 )), \"frame\": \"$FRAME\", \"env\": \"$FLOW_STRESS_ENV\"}" | tee gpurun_out/flow_stress${FLOW_STRESS_TAG:+_$FLOW_STRESS_TAG}.json
