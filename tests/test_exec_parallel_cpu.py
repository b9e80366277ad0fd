"""CPU: the PARALLEL executor (a metered pass that cuts the run and keeps the persistent memory tree + record passes that replay the planned
segments side by side, include/zkhip_vm_exec.hpp ParallelSegmentExecutor) gives the serial executor's segments -- the same cuts, the same
records word for word, the same final tree root, public values and cycle count -- for register-bound, memory-bound and intrinsic-heavy
guests, the hint stream of phantom instructions and the input stream included, at frames from 2^7 to 2^17 and 1 .. 7 record threads
(tests/exec_parallel_cpp.cpp does the comparison).  The role of the reference's metered execution before its per-segment runs:
crates/prover/src/utils/vm.rs:19 (`execute_metered_cost`), host parallelism as in crates/integration/src/testers/chunk.rs:352-368."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import rv32_model as rv  # noqa: E402
import test_vm_cpu as t  # noqa: E402


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = tmp_path_factory.mktemp("exec_parallel") / "exec_parallel"
    lib_dir = os.path.join(ROOT, "zkvm-prover_amd")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "exec_parallel_cpp.cpp"), "-o", str(out),
                           "-L", lib_dir, "-lzkhip", "-Wl,-rpath," + lib_dir])
    return str(out)


def _compare(exe, tmp_path, elf, stdin, log_frame, threads, toml=None):
    (tmp_path / "g.elf").write_bytes(elf)
    (tmp_path / "in.bin").write_bytes(stdin)
    cmd = [exe, str(tmp_path / "g.elf"), str(tmp_path / "in.bin"), str(log_frame), str(threads)]
    if toml is not None:
        (tmp_path / "openvm.toml").write_text(toml)
        cmd.append(str(tmp_path / "openvm.toml"))
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    info = json.loads(r.stdout.strip().splitlines()[-1])
    assert r.returncode == 0 and info.get("equal") is True, (info, r.stderr[-2000:])
    return info


@pytest.mark.parametrize("log_frame,threads", [(10, 1), (12, 3), (14, 7), (17, 4)])
def test_register_bound_guest(exe, tmp_path, log_frame, threads):
    info = _compare(exe, tmp_path, rv.elf_bytes(t.fib_program()), (60000).to_bytes(4, "little"), log_frame, threads)
    assert info["segments"] >= (360017 >> log_frame) and info["instructions"] == 360017


@pytest.mark.parametrize("log_frame,threads", [(12, 2), (15, 5)])
def test_memory_bound_guest_thousands_of_touched_blocks_per_segment(exe, tmp_path, log_frame, threads):
    from guest_bench2 import memsum_program

    info = _compare(exe, tmp_path, rv.elf_bytes(memsum_program()), (6).to_bytes(4, "little"), log_frame, threads)
    assert info["segments"] > 8


@pytest.mark.parametrize("log_frame,threads", [(12, 3), (16, 6)])
def test_chunk_like_guest_every_intrinsic_under_the_chunk_circuits_configuration(exe, tmp_path, log_frame, threads):
    toml = t.chunk_circuit_toml((1, 0, 100, 16, 16))
    info = _compare(exe, tmp_path, rv.elf_bytes(t.mixed_chunk_program(), data=t.mixed_chunk_data()), (96).to_bytes(4, "little"), log_frame, threads, toml)
    assert info["segments"] >= 3


def test_hint_stream_of_phantom_instructions_crosses_segment_cuts(exe, tmp_path):
    """square roots / non-residues pushed by phantom instructions are read back word by word over many instructions: at frames of 2^7 the
    hints a phantom left are still queued when a segment ends -- the snapshot of a cut carries the queue"""
    toml = "[app_vm_config.modular]\nsupported_moduli = [\n" + ",\n".join('    "%d"' % m for m in t.PHANTOM_MODULI) + "\n]\n"
    info = _compare(exe, tmp_path, rv.elf_bytes(t.phantom_program(), data=t.phantom_data()), b"", 7, 3, toml)
    assert info["segments"] >= 4


def test_native_and_castf_calls(exe, tmp_path):
    toml = "[app_vm_config.native]\n[app_vm_config.castf]\n"
    _compare(exe, tmp_path, rv.elf_bytes(t.native_program(), data=t.native_data()), b"", 7, 2, toml)


def test_a_guest_failure_surfaces_with_the_serial_executors_message(exe, tmp_path):
    # exit(3) after a loop: the metered pass throws on its thread, the caller gets "guest exited with code 3" from run_segment -- after the
    # segments before the failing one -- and every thread ends
    A0, A7, T0 = 10, 17, 5
    words = rv.assemble(rv.li(T0, 3000) + [("label", "l"), ("addi", T0, T0, -1), ("bne", T0, 0, "l"), ("addi", A0, 0, 3), ("addi", A7, 0, 93), ("ecall",)])
    (tmp_path / "g.elf").write_bytes(rv.elf_bytes(words))
    for extra in ([], ["-", "parallel-only"]):
        r = subprocess.run([exe, str(tmp_path / "g.elf"), "-", "8", "3"] + extra, capture_output=True, text=True, timeout=120)
        assert r.returncode == 3 and "guest exited with code 3" in r.stdout, r.stdout + r.stderr


@pytest.fixture(scope="module")
def exe_tsan(tmp_path_factory):
    out = tmp_path_factory.mktemp("exec_parallel_tsan") / "exec_parallel_tsan"
    lib_dir = os.path.join(ROOT, "zkvm-prover_amd")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-pthread", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "exec_parallel_cpp.cpp"),
                           "-o", str(out), "-L", lib_dir, "-lzkhip", "-Wl,-rpath," + lib_dir])
    return str(out)


def test_under_thread_sanitizer(exe_tsan, tmp_path):
    """the three kinds of threads (machine, memory tree, record passes) and the caller under TSan: a full comparison run, a run the caller
    abandons after two segments (the destructor joins threads that are mid-segment), and a guest failure"""
    from guest_bench2 import memsum_program

    def run(*args):
        r = subprocess.run([exe_tsan] + [str(a) for a in args], capture_output=True, text=True, timeout=900)
        assert "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
        return r

    (tmp_path / "fib.elf").write_bytes(rv.elf_bytes(t.fib_program()))
    (tmp_path / "mem.elf").write_bytes(rv.elf_bytes(memsum_program(2048)))
    (tmp_path / "in.bin").write_bytes((3000).to_bytes(4, "little"))
    (tmp_path / "in2.bin").write_bytes((3).to_bytes(4, "little"))
    r = run(tmp_path / "fib.elf", tmp_path / "in.bin", 9, 4)
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["equal"] is True, r.stdout + r.stderr[-2000:]
    r = run(tmp_path / "mem.elf", tmp_path / "in2.bin", 11, 3)
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["equal"] is True, r.stdout + r.stderr[-2000:]
    r = run(tmp_path / "fib.elf", tmp_path / "in.bin", 9, 4, "-", "parallel-only", 2)
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["segments_taken"] == 2, r.stdout + r.stderr[-2000:]


def test_256_bit_branches_across_cuts(exe, tmp_path):
    """the bigint extension's branches read a2 in the ecall chip's row: the metered pass touches that register block as the record pass does"""
    _compare(exe, tmp_path, rv.elf_bytes(t.branch256_program(), data=t.branch256_data()), b"", 5, 3, "[app_vm_config.bigint]\n")
    _compare(exe, tmp_path, rv.elf_bytes(t.branch256_program(), data=t.branch256_data()), b"", 4, 2, "[app_vm_config.bigint]\n")


@pytest.mark.parametrize("seed", [3, 21])
def test_every_instruction_class(exe, tmp_path, seed):
    """tests/test_vm_cpu.py mixed_program: every RV32IM instruction class (ALU, shifts, comparisons, mul / mulh / div / rem, branches, jumps,
    loads and stores of every width and sign) -- the metered pass runs them without a record sink, the record passes with one"""
    _compare(exe, tmp_path, rv.elf_bytes(t.mixed_program()), int(seed).to_bytes(4, "little"), 8, 3)
