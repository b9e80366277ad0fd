"""Writes the files of a guest run into a directory (no GPU, no prover): fib.elf, stdin.bin, openvm.toml (the reference's chunk-circuit
sections at the reference's FRI parameters) -- so that a profiler can be put directly in front of `prove_cli prove-elf`.
Usage: python tools/make_guest_files.py <dir> [n_iterations]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rv32_model as rv  # noqa: E402
from test_vm_cpu import chunk_circuit_toml, fib_program  # noqa: E402

d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 350000
os.makedirs(d, exist_ok=True)
open(os.path.join(d, "fib.elf"), "wb").write(rv.elf_bytes(fib_program()))
open(os.path.join(d, "stdin.bin"), "wb").write(n.to_bytes(4, "little"))
open(os.path.join(d, "openvm.toml"), "w").write(chunk_circuit_toml((1, 0, 100, 16, 16)))
print(d)
