"""Guest in, proofs out at scale: a Fibonacci guest of ~1 M instructions through `prove_cli prove-guest` with continuation
segments (include/zkhip_vm_prover.hpp: execute, per-segment device trace generation for nineteen chips, prove, self-verify; the
execution streams under the proving) -- reports executed MHz and instructions per second from ELF to verified proofs.  Usage: python tools/guest_bench.py [n_iterations] [segment_instr] [inflight]"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rv32_model as rv  # noqa: E402  (the assembler; test infrastructure, not the oracle)
from test_vm_cpu import fib_program  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 150000
seg = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
inflight = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cli = os.path.join(ROOT, "zkvm-prover_amd", "prove_cli")
with tempfile.TemporaryDirectory() as d:
    exe, inp, out = os.path.join(d, "guest.elf"), os.path.join(d, "stdin.bin"), os.path.join(d, "out")
    os.mkdir(out)
    open(exe, "wb").write(rv.elf_bytes(fib_program()))
    open(inp, "wb").write(int(n).to_bytes(4, "little"))
    t0 = time.time()
    r = subprocess.run([cli, "prove-guest", exe, inp, out, "-", "0", str(seg), str(inflight)], capture_output=True, text=True)
    wall = time.time() - t0
    if r.returncode:
        print(r.stderr)
        sys.exit(1)
    js = json.loads(r.stdout.strip().splitlines()[-1])
    js["wall_s"] = round(wall, 2)
    js["executed_mhz"] = round(js["total_cycles"] / max(js["execution_ms"], 1) / 1e3, 1)
    js["proven_kinstr_per_s"] = round(js["total_cycles"] / max(js["proving_wall_ms"], 1), 1)
    js["proof_bytes_total"] = sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out))
    print(json.dumps(js))
