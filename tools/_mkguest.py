import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import rv32_model as rv
from test_vm_cpu import fib_program, mixed_program
d = sys.argv[1]
os.makedirs(d + "/out", exist_ok=True)
open(d + "/guest.elf", "wb").write(rv.elf_bytes(fib_program()))
open(d + "/stdin.bin", "wb").write(int(sys.argv[2]).to_bytes(4, "little"))
