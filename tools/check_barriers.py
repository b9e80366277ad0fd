"""Every s_barrier of every kernel must be reached with NO LDS operation of the wave still in flight (docs/stale_node.md).

The stale tree node of round 4 was a miscompile: hipcc 7.2 dropped the `s_waitcnt lgkmcnt(0)` that __syncthreads()' release fence asks
for, in a loop whose LDS write sits at the END of the body and whose barrier sits at the TOP of the next iteration (the wait of the loop's
pre-header was kept, the one on the back edge was not).  gfx950's s_barrier does not wait for outstanding LDS traffic by itself; the wave
that wrote arrives at the barrier with its ds_write still queued, another wave leaves the barrier and reads the previous contents.

This script recomputes what the compiler's wait-count pass should have: per kernel, a forward data-flow over the basic blocks of the
gfx950 assembly (`hipcc -S --cuda-device-only`): state = "an LDS instruction has been issued since the last `s_waitcnt lgkmcnt(0)`",
joined over all predecessors (fall-through, s_branch, s_cbranch_*), iterated to the fixed point.  A barrier reached with an LDS WRITE
(ds_write*, LDS atomics) possibly in flight is an error -- another wave may read the old contents behind the barrier.  A barrier reached
with only LDS READS in flight is listed as a note: the value a read returns is sampled when the LDS executes it, so a write of another wave
behind the barrier could in principle overtake it (write-after-read); rocPRIM's block merge sort compiles to that shape and the compiler
emits it freely, it is not treated as an error.  ds_bpermute / ds_permute / ds_swizzle move data between lanes and touch no LDS memory.
Usage: python3 tools/check_barriers.py [file.s | file.hip ...]   (no arguments: every csrc/*.hip); exit code 1 on an error.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "zkvm-prover_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# (-DZKHIP_TEST_KERNELS: the checker sees the superset -- the kernels that ship AND the A/B bodies of libzkhip_test.so, which it must flag)
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-value", "-ffp-contract=off", "-w", "-DZKHIP_TEST_KERNELS"]

LABEL = re.compile(r"^([.\w$]+):")
KERNEL_END = re.compile(r"^\s*\.(section|amdhsa_kernel|size)\b|^\s*s_endpgm")


def assembly_of(path):
    path = os.path.abspath(path)
    if path.endswith(".s"):
        return open(path).read()
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.check_call([HIPCC, *FLAGS, "-S", "--cuda-device-only", path, "-o", out], cwd=os.path.dirname(path) or ".")
        return open(out).read()


def kernels(asm):
    """{name: [instruction lines]} for every function of the assembly text (from its label to the matching .Lfunc_end)"""
    out, name, body = {}, None, []
    for line in asm.split("\n"):
        s = line.split(";")[0].rstrip()
        m = LABEL.match(s)
        if m and not m.group(1).startswith("."):
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        if s.strip().startswith(".Lfunc_end"):
            out[name] = body
            name = None
            continue
        if s.strip():
            body.append(s.strip())
    return out


def lds_waits_to_zero(instr):
    if not instr.startswith("s_waitcnt"):
        return False
    if "lgkmcnt" not in instr:   # bare `s_waitcnt 0` style immediates are not emitted by this compiler; be conservative
        return instr.split()[-1] in ("0", "0x0")
    return re.search(r"lgkmcnt\(0\)", instr) is not None


def lds_kind(ins):
    """'w' for an instruction that writes LDS memory, 'r' for one that only reads it, None otherwise"""
    op = ins.split()[0]
    if op.startswith(("ds_bpermute", "ds_permute", "ds_swizzle", "ds_nop", "ds_gws", "ds_ordered", "ds_consume", "ds_append")):
        return None
    if op.startswith("ds_read"):
        return "r"
    if op.startswith("ds_") or op.endswith("_lds") or "_lds_" in op:   # ds_write*, LDS atomics, buffer/global loads with the LDS as destination
        return "w"
    return None


def check_kernel(lines):
    """[(block label, kind 'w' / 'r', the pending LDS instruction or "a predecessor block")] for one kernel's instruction list"""
    # basic blocks
    blocks, cur, label_of = [], {"label": None, "ins": []}, {}
    for ln in lines:
        m = LABEL.match(ln)
        if m:
            if cur["ins"] or cur["label"]:
                blocks.append(cur)
            cur = {"label": m.group(1), "ins": []}
            continue
        if ln.startswith("."):   # directives
            continue
        cur["ins"].append(ln)
        if ln.startswith("s_branch") or ln.startswith("s_cbranch") or ln.startswith("s_endpgm") or ln.startswith("s_setpc"):
            blocks.append(cur)
            cur = {"label": None, "ins": []}
    if cur["ins"] or cur["label"]:
        blocks.append(cur)
    for i, b in enumerate(blocks):
        if b["label"]:
            label_of[b["label"]] = i
    succ = []
    for i, b in enumerate(blocks):
        last = b["ins"][-1] if b["ins"] else ""
        s = []
        if last.startswith("s_branch"):
            s.append(label_of.get(last.split()[-1]))
        elif last.startswith("s_cbranch"):
            s.append(label_of.get(last.split()[-1]))
            s.append(i + 1)
        elif last.startswith("s_endpgm") or last.startswith("s_setpc"):
            pass
        else:
            s.append(i + 1)
        succ.append([x for x in s if x is not None and x < len(blocks)])
    # forward data-flow: pending[i] = (a write, a read) may be in flight at the ENTRY of block i
    pending_in = [[False, False] for _ in blocks]

    def step(st, ins):
        k = lds_kind(ins)
        if k == "w":
            return [True, st[1]]
        if k == "r":
            return [st[0], True]
        if lds_waits_to_zero(ins):
            return [False, False]
        return st

    changed = True
    while changed:
        changed = False
        for i, b in enumerate(blocks):
            st = list(pending_in[i])
            for ins in b["ins"]:
                st = step(st, ins)
            for j in succ[i]:
                for q in (0, 1):
                    if st[q] and not pending_in[j][q]:
                        pending_in[j][q] = True
                        changed = True
    found = []
    for i, b in enumerate(blocks):
        st = list(pending_in[i])
        src = ["a predecessor block" if st[0] else None, "a predecessor block" if st[1] else None]
        for ins in b["ins"]:
            k = lds_kind(ins)
            if k == "w":
                src[0] = ins
            elif k == "r":
                src[1] = ins
            if ins.startswith("s_barrier"):
                if st[0]:
                    found.append((b["label"] or "(fall-through block %d)" % i, "w", src[0]))
                elif st[1]:
                    found.append((b["label"] or "(fall-through block %d)" % i, "r", src[1]))
            st = step(st, ins)
    return found


def check_file(path):
    bad = []
    for name, lines in kernels(assembly_of(path)).items():
        for where, kind, src in check_kernel(lines):
            bad.append((os.path.basename(path), name, where, kind, src))
    return bad


def main(argv):
    files = argv or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    bad = []
    for f in files:
        found = check_file(f)
        print("%s: %d barrier(s) reached with an LDS write in flight, %d with only reads" % (os.path.basename(f), sum(1 for x in found if x[3] == "w"),
                                                                                            sum(1 for x in found if x[3] == "r")))
        bad += found
    for f, k, where, kind, src in bad:
        print("  %s %s: kernel %s, block %s: s_barrier after `%s` with no s_waitcnt lgkmcnt(0) on some path" % ("ERROR" if kind == "w" else "note ", f, k[:100], where, src))
    return 1 if any(x[3] == "w" for x in bad) else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
