# rocprofv3 kernel trace of the one-statement flow (prove_cli prove-elf: 700 k-iteration Fibonacci guest = 4.2 M instructions,
# 43 segments over 2 lanes, aggregation tree): writes gpurun_out/prof_seg2/{seg_results.db, kernel_stats.csv, flow.json}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
python3 - <<PY
import sys, os
sys.path.insert(0, os.path.join("$R", "tests"))
import rv32_model as rv
from test_vm_cpu import fib_program
open("/tmp/fib.elf","wb").write(rv.elf_bytes(fib_program()))
open("/tmp/stdin.bin","wb").write((700000).to_bytes(4,"little"))
PY
mkdir -p /tmp/o $R/gpurun_out/prof_seg2
$R/zkvm-prover_amd/prove_cli prove-elf /tmp/fib.elf /tmp/stdin.bin /tmp/o - 17 > /dev/null 2>&1   # warm the JIT cache
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_seg2 -o seg -- $R/zkvm-prover_amd/prove_cli prove-elf /tmp/fib.elf /tmp/stdin.bin /tmp/o - 17 2>/dev/null | tail -1 > $R/gpurun_out/prof_seg2/flow.json
python3 $R/tools/rocpd_stats.py $R/gpurun_out/prof_seg2/seg_results.db $R/gpurun_out/prof_seg2/kernel_stats.csv
rm -f $R/gpurun_out/prof_seg2/seg_results.db
cat $R/gpurun_out/prof_seg2/flow.json; head -8 $R/gpurun_out/prof_seg2/kernel_stats.csv
