# Cold-start cost of the one-statement flow on a box without compiled constraint kernels: clears the compiler cache, runs prove_cli
# prove-elf (4.2 M-instruction Fibonacci guest) with ZKHIP_KEYGEN_TIMING=1 once per ZKHIP_JIT_OPT level given, then once warm.
cd $GRAFT_REPO_ROOT
python3 - <<PY
import sys, os
sys.path.insert(0, "tests")
import rv32_model as rv
from test_vm_cpu import fib_program
open("/tmp/fib.elf","wb").write(rv.elf_bytes(fib_program()))
open("/tmp/stdin.bin","wb").write((700000).to_bytes(4,"little"))
PY
mkdir -p /tmp/o
for OPT in ${@:--O3}; do
  rm -rf ~/.cache/comgr*
  ( time ZKHIP_JIT_OPT=$OPT ZKHIP_KEYGEN_TIMING=1 ./zkvm-prover_amd/prove_cli prove-elf /tmp/fib.elf /tmp/stdin.bin /tmp/o - 17 ) > gpurun_out/cold_cli_$OPT.log 2>&1
  ( time ZKHIP_JIT_OPT=$OPT ./zkvm-prover_amd/prove_cli prove-elf /tmp/fib.elf /tmp/stdin.bin /tmp/o - 17 ) > gpurun_out/warm_cli_$OPT.log 2>&1
  ( time ZKHIP_JIT_OPT=$OPT ./zkvm-prover_amd/prove_cli prove-elf /tmp/fib.elf /tmp/stdin.bin /tmp/o - 17 ) > gpurun_out/warm2_cli_$OPT.log 2>&1
done
rm -rf ~/.cache/comgr*
