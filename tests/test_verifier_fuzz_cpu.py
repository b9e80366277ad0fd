"""CPU: the product's host verifier under AddressSanitizer + UBSan, fed mutated proofs (sanitizers
run on the CPU build only).  The verifier is the piece that parses untrusted bytes."""
import os
import subprocess

import numpy as np
import pytest

from zkvm_prover_amd import air

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PARAMS = (1, 0, 6, 3, 3)


def _logup_airs():
    s, t = air.lookup_traces(5, 3, seed=2)
    mt, mpv = air.bus_mix_trace(4, seed=1)
    nopv = np.zeros(0, np.uint32)
    return [dict(program=air.lookup_sender_air().program(), log_height=5, width=3, n_pvs=0, trace=s, pvs=nopv),
            dict(program=air.bus_mix_air().program(), log_height=4, width=6, n_pvs=1, trace=mt, pvs=mpv),
            dict(program=air.lookup_table_air().program(), log_height=3, width=3, n_pvs=0, trace=t, pvs=nopv)]


def _prep_airs():
    u, m, prep = air.range_traces(5, 3, seed=2)
    nopv = np.zeros(0, np.uint32)
    ftr, fpv = air.fibonacci_trace(4)
    return [dict(program=air.range_user_air().program(), log_height=5, width=4, n_pvs=0, trace=u, pvs=nopv),
            dict(program=air.fibonacci_air().program(), log_height=4, width=2, n_pvs=3, trace=ftr, pvs=fpv),
            dict(program=air.range_table_air().program(), log_height=3, width=1, n_pvs=0, trace=m, pvs=nopv, prep=prep)]


def _cached_airs():
    """cached main partitions next to a bus and a preprocessed table"""
    airs = _prep_airs() + _logup_airs()[:1] + _logup_airs()[2:]
    sa = air.SyntheticAir(width=14, n_free=5, n_bool=2, n_boundary=2, seed=3)
    sa.builder.cached_width = 6
    tr, pv = sa.gen_trace(5, seed=4)
    lb = air.lookup_sender_air()
    lb.cached_width = 1
    airs[3] = dict(airs[3], program=lb.program())
    return [dict(program=sa.program(), log_height=5, width=14, n_pvs=len(pv), trace=tr, pvs=pv)] + airs


def _plain_airs():
    sa = air.SyntheticAir(width=14, n_free=5, n_bool=2, n_boundary=2, seed=3)
    tr, pv = sa.gen_trace(5, seed=4)
    fa = air.fibonacci_air()
    ftr, fpv = air.fibonacci_trace(4)
    return [dict(program=sa.program(), log_height=5, width=14, n_pvs=len(pv), trace=tr, pvs=pv),
            dict(program=fa.program(), log_height=4, width=2, n_pvs=3, trace=ftr, pvs=fpv)]


@pytest.fixture(scope="module")
def fuzz_exe(tmp_path_factory):
    exe = tmp_path_factory.mktemp("fuzz") / "fuzz_verify"
    csrc = os.path.join(ROOT, "zkvm-prover_amd", "csrc")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-DZK_NO_HOST_AVX512", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-I", os.path.join(ROOT, "include"), "-I", csrc, "-x", "c++", os.path.join(csrc, "verifier.hip"),
           os.path.join(ROOT, "tests", "fuzz_verify.cpp"), "-o", str(exe)]
    subprocess.check_call(cmd)
    return exe


@pytest.mark.parametrize("case", [_plain_airs, _logup_airs, _prep_airs, _cached_airs])
def test_verifier_asan_ubsan_fuzz(ora, tmp_path, fuzz_exe, case):
    exe = fuzz_exe
    airs = case()
    proof = ora.stark_prove(PARAMS, airs)
    words = list(PARAMS) + [len(airs)]
    for a in airs:
        prog = np.asarray(a["program"], dtype=np.uint32)
        words += [a["log_height"], a["width"], a["n_pvs"], prog.size] + prog.tolist() + np.asarray(a["pvs"]).tolist()
        words += [1] + ora.prep_commit(PARAMS, a).tolist() if a.get("prep") is not None else [0]
    np.array(words, dtype=np.uint32).tofile(tmp_path / "case.bin")
    proof.tofile(tmp_path / "proof.bin")
    r = subprocess.run([str(exe), str(tmp_path / "case.bin"), str(tmp_path / "proof.bin"), "2400"],
                       capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 mutated proofs accepted" in r.stdout


def test_codec_asan_ubsan_fuzz(tmp_path):
    """The v1 proof-container codec parses untrusted bytes too: the reference's stored proof and 3000 mutated variants under
    ASan + UBSan; whatever still parses must re-encode to exactly its input."""
    exe = tmp_path / "fuzz_codec"
    csrc = os.path.join(ROOT, "zkvm-prover_amd", "csrc")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-DZK_NO_HOST_AVX512", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-I", os.path.join(ROOT, "include"), "-I", csrc, "-x", "c++", os.path.join(csrc, "codec.hip"),
           os.path.join(csrc, "verifier.hip"), os.path.join(ROOT, "tests", "fuzz_codec.cpp"), "-o", str(exe)]
    subprocess.check_call(cmd)
    blob = os.path.join(ROOT, "tests", "golden", "ref_proofs", "chunk-proof-feynman.proofs.bin")
    r = subprocess.run([str(exe), blob, "3000"], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 round-trip mismatches" in r.stdout
