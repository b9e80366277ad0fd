"""sha256 of ONE kernel's machine code inside libzkhip.so.

bench.py prices this run's time with VALU instruction counts that were collected by an earlier rocprofv3 counter pass
(profiles/roundNN_pmc_valu.json, tools/pmc_valu3.py).  Those counts belong to one compiled body of the kernel: the counter pass stores the
hash computed here next to them, bench.py recomputes it from the library it has loaded and flags `roofline.valu.stale` on a mismatch.

The library embeds one clang offload bundle per translation unit (magic `__CLANG_OFFLOAD_BUNDLE__`, then u64 n, then n x {u64 offset, u64 size,
u64 triple length, triple}); the `hipv4-amdgcn-amd-amdhsa--gfx950` entry is an ELF64 code object whose symbol table holds the kernel as a
FUNC symbol (value = address in .text, size = bytes).  The hash covers exactly those bytes, with ONE normalisation: the 32-bit literals
of `s_getpc_b64 ; s_add_u32 lo, lo, <rel32> ; s_addc_u32 hi, hi, <rel32>` (pc-relative addresses of the round-constant tables) are zeroed --
they move whenever another kernel is added to the translation unit although the kernel's instructions stay the same.  Pure Python: no
binutils on the path needed.
"""
import hashlib
import struct
import sys

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _bundle_entries(blob, at):
    (n,) = struct.unpack_from("<Q", blob, at + len(MAGIC))
    p = at + len(MAGIC) + 8
    for _ in range(n):
        off, size, tlen = struct.unpack_from("<QQQ", blob, p)
        triple = blob[p + 24:p + 24 + tlen].decode()
        p += 24 + tlen
        yield triple, at + off, size


def _elf_function_bytes(elf, name):
    """bytes of FUNC symbol `name` in the ELF64 little-endian image `elf`, or None"""
    if elf[:4] != b"\x7fELF" or elf[4] != 2:
        return None
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, _ = struct.unpack_from("<HHH", elf, 0x3A)
    secs = []
    for i in range(shnum):
        s = struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize)
        secs.append(dict(type=s[1], addr=s[3], off=s[4], size=s[5], link=s[6], entsize=s[9]))
    for sec in secs:
        if sec["type"] != 2:  # SHT_SYMTAB
            continue
        strtab = secs[sec["link"]]
        for k in range(sec["size"] // 24):
            st_name, st_info, _, st_shndx, st_value, st_size = struct.unpack_from("<IBBHQQ", elf, sec["off"] + 24 * k)
            if (st_info & 0xF) != 2 or st_shndx == 0 or st_shndx >= len(secs):  # STT_FUNC, defined
                continue
            end = elf.index(b"\0", strtab["off"] + st_name)
            if elf[strtab["off"] + st_name:end].decode() != name:
                continue
            text = secs[st_shndx]
            start = text["off"] + (st_value - text["addr"])
            return elf[start:start + st_size]
    return None


def normalise(code):
    """zero the pc-relative literals behind s_getpc_b64 (see the module text)"""
    w = list(struct.unpack("<%dI" % (len(code) // 4), code[:len(code) // 4 * 4]))
    for i, d in enumerate(w):
        if d & 0xFF80FFFF != 0xBE801C00:   # SOP1 s_getpc_b64 sdst
            continue
        j, seen = i + 1, 0
        while j + 1 < len(w) and j < i + 12 and seen < 2:
            op = w[j] & 0xFF800000
            if op in (0x80000000, 0x82000000) and (w[j] >> 8) & 0xFF == 0xFF:   # s_add_u32 / s_addc_u32 with a literal as src1
                w[j + 1] = 0
                seen += 1
                j += 2
            else:
                j += 1
    return struct.pack("<%dI" % len(w), *w) + code[len(code) // 4 * 4:]


def kernel_code_sha256(so_path, mangled_name, arch="gfx950"):
    """(sha256 hex, code bytes) of kernel `mangled_name` compiled for `arch` inside `so_path`; raises KeyError when absent"""
    with open(so_path, "rb") as f:
        blob = f.read()
    at = blob.find(MAGIC)
    while at >= 0:
        for triple, off, size in _bundle_entries(blob, at):
            if not triple.endswith(arch) or size == 0:
                continue
            code = _elf_function_bytes(blob[off:off + size], mangled_name)
            if code:
                return hashlib.sha256(normalise(code)).hexdigest(), len(code)
        at = blob.find(MAGIC, at + 1)
    raise KeyError("%s: no %s kernel %s" % (so_path, arch, mangled_name))


HASH_ROWS = "_ZN2zk11k_hash_rowsEPKPKjjmPj"   # zk::k_hash_rows(unsigned const* const*, unsigned, unsigned long, unsigned*)

if __name__ == "__main__":
    so = sys.argv[1] if len(sys.argv) > 1 else "zkvm-prover_amd/libzkhip.so"
    name = sys.argv[2] if len(sys.argv) > 2 else HASH_ROWS
    h, n = kernel_code_sha256(so, name)
    print('{"kernel": "%s", "code_bytes": %d, "code_sha256": "%s"}' % (name, n, h))
