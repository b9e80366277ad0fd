"""chunk -> batch -> bundle at the reference's FRI parameters, timed (crates/integration/src/testers: chunk proofs, a batch over them, a
bundle over the batch -- every layer defers the verification of the one below).  N Fibonacci chunk guests of ~n
instructions each -> N root proofs under the chunk app's ONE aggregation key -> a batch guest that states N claims (prove-deferral: its
segments, aggregation tree, the deferral node(s) over the N chunk roots, the fold if there are more than one, the join) -> a bundle guest
over the batch proof (its deferral node opens the batch's claims in the batch's memory root in-circuit).  Prints one JSON line.
Usage: python tools/deferral_bench.py [n_chunks] [iterations_per_chunk] [log_frame]"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import prover_mirror_util as pm  # noqa: E402
import rv32_model as rv  # noqa: E402
from test_vm_cpu import deferral_guest_program, fib_program  # noqa: E402

PARAMS = (1, 0, 100, 16, 16)


def run(cmd, env=None):
    t0 = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, text=True, env=env)
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-3000:])
        sys.exit(r.returncode)
    return json.loads(r.stdout.strip().splitlines()[-1]) if r.stdout.strip().startswith("{") or "{" in r.stdout else {}, time.perf_counter() - t0


def pvs_of(path):
    upv = pm.un_b64_bincode(json.load(open(path))["user_pvs_proof"])
    n = (len(upv) - 32 - 4 * 2 * 8 * 28) // 4
    return upv[4 * n:4 * n + 32]


def main():
    n_chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    log_frame = sys.argv[3] if len(sys.argv) > 3 else "17"
    tmp = tempfile.mkdtemp(prefix="zkhip_deferral_")
    cfg = os.path.join(tmp, "openvm.toml")
    open(cfg, "w").write(pm.TOML.format(*PARAMS))
    chunk_elf = os.path.join(tmp, "chunk.elf")
    open(chunk_elf, "wb").write(rv.elf_bytes(fib_program()))
    chunks, chunk_s, cycles = [], [], 0
    for i in range(n_chunks):
        d = os.path.join(tmp, "c%d" % i)
        os.mkdir(d)
        open(os.path.join(d, "stdin.bin"), "wb").write((iters + i).to_bytes(4, "little"))
        info, dt = run([pm.CLI, "prove-elf", chunk_elf, os.path.join(d, "stdin.bin"), d, cfg, log_frame])
        chunks.append(os.path.join(d, "root.json"))
        chunk_s.append(round((info["segment_tracegen_and_proving_ms"] + info["aggregation_ms"]) / 1e3, 3))
        cycles += info["total_cycles"]
    chunk_vk = os.path.join(tmp, "c0", "root.vk")
    pc, _ = run([pm.CLI, "program-commit", chunk_elf, chunk_vk, cfg])
    batch_elf = os.path.join(tmp, "batch.elf")
    open(batch_elf, "wb").write(rv.elf_bytes(deferral_guest_program(), data=b"".join(int(x).to_bytes(4, "little") for x in pc["exe"] + pc["vm"])))
    env = dict(os.environ, ZKHIP_DEFERRAL_CHILDREN="8", ZKHIP_DEFERRAL_NODES=str(max(1, (n_chunks + 7) // 8)))
    bd = os.path.join(tmp, "batch")
    os.mkdir(bd)
    open(os.path.join(bd, "witness.bin"), "wb").write(b"".join(pvs_of(c) for c in chunks))
    binfo, batch_wall = run([pm.CLI, "prove-deferral", batch_elf, cfg, bd, "9", chunk_vk, cfg, os.path.join(bd, "witness.bin")] + chunks, env)
    batch_vk = os.path.join(bd, "root.vk")
    bpc, _ = run([pm.CLI, "program-commit", batch_elf, batch_vk, cfg])
    bundle_elf = os.path.join(tmp, "bundle.elf")
    open(bundle_elf, "wb").write(rv.elf_bytes(deferral_guest_program(), data=b"".join(int(x).to_bytes(4, "little") for x in bpc["exe"] + bpc["vm"])))
    ud = os.path.join(tmp, "bundle")
    os.mkdir(ud)
    open(os.path.join(ud, "witness.bin"), "wb").write(pvs_of(os.path.join(bd, "root.json")))
    uinfo, bundle_wall = run([pm.CLI, "prove-deferral", bundle_elf, cfg, ud, "9", batch_vk + "@" + batch_elf, cfg, os.path.join(ud, "witness.bin"), os.path.join(bd, "root.json")])
    v = subprocess.run([pm.CLI, "verify-guest", bundle_elf, os.path.join(ud, "root.vk"), cfg, os.path.join(ud, "root.json")], capture_output=True, text=True)
    print(json.dumps({"chunks": n_chunks, "chunk_instructions_total": cycles, "params": PARAMS, "chunk_prove_s_each": chunk_s,
                      "batch": {"children": binfo["children"], "setup_s": binfo["setup_s"], "prove_s": binfo["prove_s"], "process_wall_s": round(batch_wall, 2),
                                "deferral_nodes": int(env["ZKHIP_DEFERRAL_NODES"]), "root_proof_bytes": binfo["root_proof_bytes"]},
                      "bundle": {"children": uinfo["children"], "setup_s": uinfo["setup_s"], "prove_s": uinfo["prove_s"], "process_wall_s": round(bundle_wall, 2),
                                 "root_proof_bytes": uinfo["root_proof_bytes"]},
                      "bundle_verified_under_its_join_key": v.returncode == 0}))


if __name__ == "__main__":
    main()
