"""ONE FLOW at scale: a Fibonacci guest of ~n instructions through `prove_cli prove-elf` (include/zkhip_vm_flow.hpp): execution in
segments of fixed heights, every segment ONE statement (22 chips, adapters + execution bus + persistent memory), aggregation tree
to ONE root proof, self-verified, at the reference's FRI parameters.  Prints the CLI's JSON line + instructions per second.
Usage: python tools/guest_bench2.py [n_iterations] [log_frame] [chunk | mem | mixed]     (mem: a memory-bound guest, n = passes over a 64 KiB array;
       mixed: a CHUNK-LIKE guest under the reference's chunk-circuit openvm.toml -- tests/test_vm_cpu.py mixed_chunk_program: register loops, strided
       loads, Keccak-f, SHA-256, secp256k1 additions / doublings, modular and 256-bit arithmetic in phases, so that segments land in the 22-, 26- and
       51-chip shapes; n = iterations of ~1.6 k instructions)
       python tools/guest_bench2.py [n_iterations] [log_frame] [chunk]     (chunk: under the reference's chunk-circuit openvm.toml --
keccak, sha2, bigint, six moduli, three curves: 49 chips per segment instead of 22)"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rv32_model as rv  # noqa: E402
from test_vm_cpu import MIXED_MIX, MIXED_PHASE_ITERATIONS, chunk_circuit_toml, fib_program, mixed_chunk_data, mixed_chunk_program  # noqa: E402


def memsum_program(n_words=16384):
    """A MEMORY-bound guest beside the register-bound Fibonacci one: fills an array of n_words words (64 KiB: 4096 memory blocks), then sums
    it `passes` times (the input) with a load per element -- four instructions per element, one of them a load; every segment touches
    every block of the array, so its memory chips (blocks, path nodes, permutations) are full where Fibonacci's are empty."""
    A0, A1, A7, T0, T1, T2, T3, S0, S1 = 10, 11, 17, 5, 6, 7, 28, 8, 9
    p = [("addi", A7, 0, 2), ("ecall",), ("add", S1, A0, 0)]                       # s1 = passes
    p += rv.li(S0, 0x00400000) + rv.li(T3, 4 * n_words)
    p += [("add", T3, T3, S0), ("add", T0, S0, 0), ("addi", T1, 0, 1),
          ("label", "fill"), ("sw", T1, T0, 0), ("addi", T1, T1, 3), ("addi", T0, T0, 4), ("bne", T0, T3, "fill"),
          ("addi", T2, 0, 0),
          ("label", "pass"), ("beq", S1, 0, "done"), ("add", T0, S0, 0),
          ("label", "sum"), ("lw", T1, T0, 0), ("add", T2, T2, T1), ("addi", T0, T0, 4), ("bne", T0, T3, "sum"),
          ("addi", S1, S1, -1), ("jal", 0, "pass"),
          ("label", "done"), ("add", A0, T2, 0), ("addi", A1, 0, 0), ("addi", A7, 0, 1), ("ecall",),
          ("addi", A0, 0, 0), ("addi", A7, 0, 93), ("ecall",)]
    return rv.assemble(p)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    log_frame = sys.argv[2] if len(sys.argv) > 2 else "20"
    tmp = tempfile.mkdtemp(prefix="zkhip_guest2_")
    exe, inp = os.path.join(tmp, "fib.elf"), os.path.join(tmp, "stdin.bin")
    mem = len(sys.argv) > 3 and sys.argv[3] == "mem"     # n = the number of passes over the array (64: ~4.3 M instructions)
    mixed = len(sys.argv) > 3 and sys.argv[3] == "mixed"
    open(exe, "wb").write(rv.elf_bytes(mixed_chunk_program(), data=mixed_chunk_data()) if mixed else rv.elf_bytes(memsum_program() if mem else fib_program()))
    open(inp, "wb").write(n.to_bytes(4, "little"))
    cli = os.path.join(ROOT, "zkvm-prover_amd", "prove_cli")
    cfg = "-"
    if len(sys.argv) > 3 and sys.argv[3] in ("chunk", "mixed"):
        cfg = os.path.join(tmp, "openvm.toml")
        open(cfg, "w").write(chunk_circuit_toml((1, 0, 100, 16, 16)))
    r = subprocess.run([cli, "prove-elf", exe, inp, tmp, cfg, log_frame], capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-3000:])
        sys.exit(r.returncode)
    info = json.loads(r.stdout.strip().splitlines()[-1])
    info["instr_per_s_wall"] = round(info["total_cycles"] / info["wall_s"])
    # segments only: the wall time from the flow's start to the last segment proof -- the executor runs BENEATH the proving (its own busy
    # time, execution_ms, lies inside that interval).  Rounds 2 - 4 quoted instructions / (execution_ms + that wall): the executor counted
    # twice; kept beside it for continuity.
    info["instr_per_s_segments_only"] = round(info["total_cycles"] / max(1e-9, info["segment_tracegen_and_proving_ms"] / 1e3))
    info["instr_per_s_segments_plus_executor_busy_time"] = round(info["total_cycles"] / max(1e-9, (info["execution_ms"] + info["segment_tracegen_and_proving_ms"]) / 1e3))
    info["log_frame"] = int(log_frame)
    info["config"] = "chunk-circuit (49 chips)" if cfg != "-" else "base (22 chips)"
    info["guest"] = "memsum (a load every fourth instruction, 4096 blocks touched per segment)" if mem else "fibonacci (register-bound)"
    # the reference's own speed figure (crates/prover/src/prover/mod.rs:358-366): cycles / 1e6 / seconds of sdk.prove (segments + aggregation)
    info["prove_speed_mhz"] = round(info["total_cycles"] / 1e6 / max(1e-9, (info["segment_tracegen_and_proving_ms"] + info["aggregation_setup_wait_ms"] + info["aggregation_ms"]) / 1e3), 3)
    if mixed:
        info["guest"] = "mixed chunk-like guest: phases of %d iterations (plain, hash, hash, full); per iteration %s" % (MIXED_PHASE_ITERATIONS, json.dumps(MIXED_MIX))
        info["config"] = "chunk-circuit (51 chips in the full shape)"
        sps, pms, ips = info.get("segments_per_shape") or [], info.get("sum_prove_ms_per_shape") or [], info.get("instructions_per_shape") or []
        info["ms_per_segment_proof_per_shape"] = [round(m / n, 2) if n else None for m, n in zip(pms, sps)]
        info["instructions_per_segment_per_shape"] = [round(i / n) if n else None for i, n in zip(ips, sps)]
    if mem:
        expect = (sum(1 + 3 * i for i in range(16384)) * n) & 0xFFFFFFFF
        info["public_value_word0_expected"] = expect
    print(json.dumps(info))


if __name__ == "__main__":
    main()
