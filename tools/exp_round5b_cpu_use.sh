cd $GRAFT_REPO_ROOT
mkdir -p /tmp/gm /tmp/om
python3 - <<'PY'
import os, sys
sys.path.insert(0, "tests")
import rv32_model as rv
from test_vm_cpu import fib_program, chunk_circuit_toml, mixed_chunk_data, mixed_chunk_program
open("/tmp/gm/fib.elf", "wb").write(rv.elf_bytes(fib_program()))
open("/tmp/gm/fib.in", "wb").write((2800000).to_bytes(4, "little"))
open("/tmp/gm/mixed.elf", "wb").write(rv.elf_bytes(mixed_chunk_program(), data=mixed_chunk_data()))
open("/tmp/gm/mixed.in", "wb").write((8192).to_bytes(4, "little"))
open("/tmp/gm/openvm.toml", "w").write(chunk_circuit_toml((1, 0, 100, 16, 16)))
PY
export ZKHIP_LANES=3
./zkvm-prover_amd/prove_cli prove-elf /tmp/gm/fib.elf /tmp/gm/fib.in /tmp/om - 19 > /dev/null 2>&1
for w in 0 4 8; do
echo "witness_threads=$w"
ZKHIP_WITNESS_THREADS=$w python3 tools/cpu_seconds.py fib ./zkvm-prover_amd/prove_cli prove-elf /tmp/gm/fib.elf /tmp/gm/fib.in /tmp/om - 19 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print({k:d.get(k) for k in ('execution_ms','segment_tracegen_and_proving_ms','aggregation_ms','wall_s')})
    else: print(l.strip())"
done
./zkvm-prover_amd/prove_cli prove-elf /tmp/gm/mixed.elf /tmp/gm/mixed.in /tmp/om /tmp/gm/openvm.toml 19 > /dev/null 2>&1
python3 tools/cpu_seconds.py mixed ./zkvm-prover_amd/prove_cli prove-elf /tmp/gm/mixed.elf /tmp/gm/mixed.in /tmp/om /tmp/gm/openvm.toml 19 2>&1 | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print({k:d.get(k) for k in ('execution_ms','segment_tracegen_and_proving_ms','aggregation_ms','wall_s','aggregation_circuits_build_s')})
    else: print(l.strip())"
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
