"""Stage timings of the extension guests (keccak / sha256 / modmul) through `prove_cli prove-elf` at test parameters.
Usage: python tools/ext_guest_timing.py [log_frame]"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rv32_model as rv  # noqa: E402
import prover_mirror_util as pm  # noqa: E402
from test_vm_cpu import (SECP256K1_N, SECP256K1_P, keccak_data, keccak_program, modmul_data, modmul_program, sha256_data, sha256_program)  # noqa: E402

log_frame = sys.argv[1] if len(sys.argv) > 1 else "8"
PARAMS = (1, 0, 4, 3, 3)
sd, nb = sha256_data(bytes(range(150)))
cases = {
    "keccak": (keccak_program(2), keccak_data(b"abc"), "\n[app_vm_config.keccak]\n"),
    "sha256": (sha256_program(nb), sd, "\n[app_vm_config.sha2]\n"),
    "modmul": (modmul_program(), modmul_data(), "\n[app_vm_config.modular]\nsupported_moduli = [\n \"%d\",\n \"%d\"\n]\n" % (SECP256K1_P, SECP256K1_N)),
}
for name, (words, data, ext) in cases.items():
    tmp = tempfile.mkdtemp(prefix="zkhip_ext_")
    exe = os.path.join(tmp, "guest.elf")
    open(exe, "wb").write(rv.elf_bytes(words, data=data))
    open(os.path.join(tmp, "openvm.toml"), "w").write(pm.TOML.format(*PARAMS) + ext)
    for attempt in range(1):
        t0 = time.time()
        r = subprocess.run([pm.CLI, "prove-elf", exe, "-", tmp, os.path.join(tmp, "openvm.toml"), log_frame], capture_output=True, text=True)
        dt = time.time() - t0
        line = r.stdout.strip().splitlines()[-1] if r.returncode == 0 else r.stderr[-400:]
        print(name, "run", attempt, "wall %.1f s" % dt, line[:400], flush=True)
        for ln in r.stderr.splitlines():   # ZKHIP_KEYGEN_TIMING=1: the compiled constraint kernels that took long
            if "[zkhip keygen]" in ln and " 0.0" not in ln:
                print("   ", ln, flush=True)
