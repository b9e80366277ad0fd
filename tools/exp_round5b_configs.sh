# the Fibonacci guest under the three reference configurations (and none) at frames of 2^20: common node heights and arities after the second session
cd $GRAFT_REPO_ROOT
mkdir -p /tmp/gc /tmp/oc
python3 - <<'PY'
import sys
sys.path.insert(0, "tests")
import rv32_model as rv
from test_vm_cpu import fib_program, chunk_circuit_toml, batch_circuit_toml
open("/tmp/gc/fib.elf", "wb").write(rv.elf_bytes(fib_program()))
open("/tmp/gc/fib.in", "wb").write((2800000).to_bytes(4, "little"))
open("/tmp/gc/chunk.toml", "w").write(chunk_circuit_toml((1, 0, 100, 16, 16)))
open("/tmp/gc/batch.toml", "w").write(batch_circuit_toml((1, 0, 100, 16, 16)))
PY
export ZKHIP_LANES=3
for cfg in - /tmp/gc/chunk.toml /tmp/gc/batch.toml; do
  ./zkvm-prover_amd/prove_cli prove-elf /tmp/gc/fib.elf /tmp/gc/fib.in /tmp/oc $cfg 20 > /dev/null 2>&1
  ./zkvm-prover_amd/prove_cli prove-elf /tmp/gc/fib.elf /tmp/gc/fib.in /tmp/oc $cfg 20 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', {k:d.get(k) for k in ('segments','levels','execution_ms','segment_tracegen_and_proving_ms','aggregation_ms','node_log_heights','tree_nodes_per_device_slot','chips_per_shape','verified')})"
done
