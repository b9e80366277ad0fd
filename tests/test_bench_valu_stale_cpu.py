"""bench.py's VALU roofline is priced with instruction counts of an earlier counter pass (profiles/roundNN_pmc_valu.json): they are only
valid for the compiled body of zk::k_hash_rows they were counted on.  The pass stores that body's sha256 (tools/code_object_hash.py),
bench.valu_counts() recomputes it from the loaded library: one edited byte of the kernel must raise the `stale` flag."""
import importlib.util
import json
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
SO = os.path.join(ROOT, "zkvm-prover_amd", "libzkhip.so")


def _bench():
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.skipif(not os.path.exists(SO), reason="libzkhip.so not built")
def test_one_edited_byte_of_the_kernel_flags_the_counts_stale(tmp_path, monkeypatch):
    import code_object_hash as coh

    bench = _bench()
    have, n_bytes = coh.kernel_code_sha256(SO, coh.HASH_ROWS)
    assert n_bytes > 4096   # (a Poseidon2 row sponge is ~11 KB of code)
    monkeypatch.delenv("ZKHIP_LIBRARY_FOR_HASH", raising=False)
    name, pv, stale, note = bench.valu_counts()
    assert name and pv, note
    want = pv["kernels"]["zk::k_hash_rows"].get("code_sha256")
    assert want, "the newest committed counter pass must carry the hash of the kernel it counted"
    assert stale == (have != want), note
    # one byte of the kernel's code in a copy of the library: find the code through the same parser, flip a byte in its middle
    blob = open(SO, "rb").read()
    at, found = blob.find(coh.MAGIC), None
    while at >= 0 and found is None:
        for triple, off, size in coh._bundle_entries(blob, at):
            if triple.endswith("gfx950") and size:
                code = coh._elf_function_bytes(blob[off:off + size], coh.HASH_ROWS)
                if code:
                    found = blob.index(code, off)
                    break
        at = blob.find(coh.MAGIC, at + 1)
    assert found is not None
    edited = tmp_path / "libzkhip_edited.so"
    shutil.copy(SO, edited)
    with open(edited, "r+b") as f:
        f.seek(found + n_bytes // 2)
        b = f.read(1)
        f.seek(found + n_bytes // 2)
        f.write(bytes([b[0] ^ 0x10]))
    assert coh.kernel_code_sha256(str(edited), coh.HASH_ROWS)[0] != have
    monkeypatch.setenv("ZKHIP_LIBRARY_FOR_HASH", str(edited))
    _, _, stale2, note2 = bench.valu_counts()
    assert stale2 is True, note2
    assert "not the body" in note2


def test_the_hash_ignores_where_the_round_constants_lie():
    """the pc-relative literals behind s_getpc_b64 move with every kernel added to the translation unit: the hash masks them"""
    import struct

    import code_object_hash as coh

    body = [0xBE821C00, 0x8002FF02, 0x00001234, 0x8203FF03, 0x00000000, 0x7E000280]   # s_getpc ; s_add_u32 s2,s2,lit ; s_addc_u32 s3,s3,lit ; v_mov
    moved = list(body)
    moved[2], moved[4] = 0x00ABCDEF, 0x00000001
    other = list(body)
    other[5] ^= 1   # a real instruction differs
    pack = lambda w: struct.pack("<%dI" % len(w), *w)
    assert coh.normalise(pack(body)) == coh.normalise(pack(moved))
    assert coh.normalise(pack(body)) != coh.normalise(pack(other))
