"""GPU: leaf circuits built on demand, their commitments cached on disk (include/zkhip_aggregation.hpp ensure_leaf / load_agg_cache; VERDICT
round 4 item 3: under the chunk-circuit configuration the 26- and 51-chip leaf circuits cost seconds of every process, used or not).

The first run of a key builds every shape's leaf circuit and writes the cache; later runs build shape 0 and the internal circuit only, take
the other shapes' commitments from the file -- the aggregation key (root.vk) must be byte-identical -- and build a shape when its first
segment proof arrives, insisting that it commits to what the key states.  The file (one per guest: the program's commitment is part of the
key) remembers which shapes the guest's flows used; those are built at setup again."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import prover_mirror_util as pm  # noqa: E402
import rv32_model as rv  # noqa: E402
from test_vm_cpu import CHUNK_CIRCUIT_CURVES, CHUNK_CIRCUIT_MODULI, MIXED_PHASE_ITERATIONS, chunk_circuit_toml, fib_program, mixed_chunk_data, mixed_chunk_program  # noqa: E402

pytestmark = pytest.mark.gpu
PARAMS = (1, 0, 100, 16, 16)


def prove(tmp, name, elf, stdin, frame, **env):
    out = tmp / name
    out.mkdir()
    r = subprocess.run([pm.CLI, "prove-elf", str(elf), str(stdin), str(out), str(tmp / "openvm.toml"), str(frame)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    info = json.loads(r.stdout.strip().splitlines()[-1])
    assert info["verified"]
    return info, (out / "root.vk").read_bytes(), out


def test_unused_shapes_are_not_built_and_the_key_does_not_change(tmp_path):
    cache = tmp_path / "cache"
    cache.mkdir()
    (tmp_path / "openvm.toml").write_text(chunk_circuit_toml(PARAMS))
    elf, stdin = tmp_path / "fib.elf", tmp_path / "stdin.bin"
    elf.write_bytes(rv.elf_bytes(fib_program()))
    stdin.write_bytes((60000).to_bytes(4, "little"))
    env = {"ZKHIP_AGG_CACHE_DIR": str(cache)}
    first, vk1, _ = prove(tmp_path, "first", elf, stdin, 14, **env)
    assert len(list(cache.glob("agg_*.key"))) == 1
    second, vk2, out2 = prove(tmp_path, "second", elf, stdin, 14, **env)
    eager, vk3, _ = prove(tmp_path, "eager", elf, stdin, 14, ZKHIP_AGG_NO_LAZY="1", **env)
    assert vk1 == vk2 == vk3                                                   # ONE aggregation key, cached or not
    assert first["segments_per_shape"][2:] == [0, 0]                           # a Fibonacci guest never needs the wide shapes ...
    assert first["leaf_circuits_at_setup"] == eager["leaf_circuits_at_setup"] == len(first["chips_per_shape"]) and first["leaf_circuits_on_demand"] == 0
    assert second["leaf_circuits_at_setup"] == 1 and second["leaf_circuits_on_demand"] == 0      # ... and with the cache they are never built
    print("setup (circuits + keys): first %.2f s, with the cache %.2f s; wall %.2f -> %.2f s" % (
        first["aggregation_circuits_build_s"] + first["aggregation_keygen_s"], second["aggregation_circuits_build_s"] + second["aggregation_keygen_s"], first["wall_s"], second["wall_s"]))
    assert pm.run_cli("verify-guest", str(elf), str(out2 / "root.vk"), str(tmp_path / "openvm.toml"), str(out2 / "root.json")).returncode == 0


def _body_digest(words):
    """the file's last eight words: the Poseidon2 sponge (overwrite 8, permute, no padding) of the body reduced mod p, through the library's host
    permutation (zkhip_poseidon2_permute_host: no GPU work)"""
    import ctypes as C

    import zkvm_prover_amd as z

    lib = z.load_library()
    st = (C.c_uint32 * 16)()
    for i in range(0, len(words), 8):
        for k, v in enumerate(words[i:i + 8]):
            st[k] = int(v) % 2013265921
        lib.zkhip_poseidon2_permute_host(st)
    return [st[k] for k in range(8)]


def _edit_cache(cache, forget=False, flip_last_commit=False, fix_digest=True):
    """the cache file: [magic, S, H0, H1, then per shape: natural heights (2), commitment (8), used (1)] + the body's digest (8 words);
    fix_digest: the edited body gets a matching digest (what a stale file of another build, or a careful forger, would carry)"""
    import numpy as np

    f = next(cache.glob("agg_*.key"))
    w = np.frombuffer(f.read_bytes(), dtype=np.uint32).copy()
    S = int(w[1])
    assert w.size == 4 + 11 * S + 8
    assert [int(x) for x in w[-8:]] == _body_digest(w[:-8])
    if forget:
        for sh in range(S):
            w[4 + 11 * sh + 10] = 0
    if flip_last_commit:
        w[4 + 11 * (S - 1) + 9] ^= 1
    if fix_digest:
        w[-8:] = _body_digest(w[:-8])
    f.write_bytes(w.tobytes())
    return [int(w[4 + 11 * sh + 10]) for sh in range(S)]


def test_a_cached_shape_is_built_when_its_first_segment_arrives(tmp_path):
    cache = tmp_path / "cache"
    cache.mkdir()
    (tmp_path / "openvm.toml").write_text(chunk_circuit_toml(PARAMS))
    words, data = mixed_chunk_program(), mixed_chunk_data()
    elf, stdin = tmp_path / "mixed.elf", tmp_path / "stdin.bin"
    elf.write_bytes(rv.elf_bytes(words, data=data))
    n = 3 * MIXED_PHASE_ITERATIONS + 5
    stdin.write_bytes(n.to_bytes(4, "little"))
    env = {"ZKHIP_AGG_CACHE_DIR": str(cache)}
    first, vk1, _ = prove(tmp_path, "first", elf, stdin, 15, **env)
    wide_used = sum(1 for k, n_ in enumerate(first["segments_per_shape"]) if k and n_)
    assert first["segments_per_shape"][2] > 0 and first["segments_per_shape"][3] > 0
    assert first["leaf_circuits_at_setup"] == len(first["chips_per_shape"])
    # the file is per guest (the program's commitment is part of the key) and remembers which shapes this guest put into the tree: those
    # are built at setup again -- beside the segment proving they would cost more than they save -- the others stay out
    assert _edit_cache(cache) == [1 if n_ else 0 for n_ in first["segments_per_shape"]]
    second, vk2, _ = prove(tmp_path, "second", elf, stdin, 15, **env)
    assert vk1 == vk2
    assert second["leaf_circuits_at_setup"] == 1 + wide_used and second["leaf_circuits_on_demand"] == 0
    # a file that does not know it (an earlier run of this key that never met the wide shapes): built when their first segment proof arrives
    _edit_cache(cache, forget=True)
    third, vk3, out3 = prove(tmp_path, "third", elf, stdin, 15, **env)
    assert vk1 == vk3
    assert third["leaf_circuits_at_setup"] == 1 and third["leaf_circuits_on_demand"] == wide_used
    model = rv.run(words, n.to_bytes(4, "little"), data=data, moduli=CHUNK_CIRCUIT_MODULI, curves=tuple((c[1], c[3]) for c in CHUNK_CIRCUIT_CURVES))
    assert third["total_cycles"] == model["instret"]
    assert pm.run_cli("verify-guest", str(elf), str(out3 / "root.vk"), str(tmp_path / "openvm.toml"), str(out3 / "root.json")).returncode == 0
    assert _edit_cache(cache) == [1 if n_ else 0 for n_ in first["segments_per_shape"]]          # ... and the file has learnt it again
    # a corrupted file (its body no longer hashes to its digest) is no cache at all: every shape is built, the key is the same, the file rewritten
    good = next(cache.glob("agg_*.key")).read_bytes()
    _edit_cache(cache, flip_last_commit=True, fix_digest=False)
    fourth, vk4, _ = prove(tmp_path, "corrupt", elf, stdin, 15, **env)
    assert vk4 == vk1 and fourth["leaf_circuits_at_setup"] == len(first["chips_per_shape"]) and next(cache.glob("agg_*.key")).read_bytes() == good
    # a cache file that states another commitment for a wide shape UNDER A VALID DIGEST is caught: at setup when the shape is built there ...
    _edit_cache(cache, flip_last_commit=True)
    out = tmp_path / "stale_setup"
    out.mkdir()
    r = subprocess.run([pm.CLI, "prove-elf", str(elf), str(stdin), str(out), str(tmp_path / "openvm.toml"), "15"], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "does not belong to this build" in r.stderr
    # ... and when it is built on demand
    _edit_cache(cache, forget=True)
    out = tmp_path / "stale"
    out.mkdir()
    r = subprocess.run([pm.CLI, "prove-elf", str(elf), str(stdin), str(out), str(tmp_path / "openvm.toml"), "15"], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "stale key cache" in r.stderr


def test_a_key_that_leaves_the_process_rests_on_no_unchecked_cache_entry(tmp_path):
    """`prove_cli agg-vk` (UniversalProver::get_agg_vk: the key handed to a verifier or into a parent guest's program commitment) builds every
    shape the cache would have left out and compares its commitment.  With a stale entry (valid digest, another commitment) for a shape the
    guest never uses, a flow still ends -- its root verifies under the key it derived, which is NOT the key of a cache-less process --, but
    the key that leaves the process is refused."""
    cache = tmp_path / "cache"
    cache.mkdir()
    (tmp_path / "openvm.toml").write_text(chunk_circuit_toml(PARAMS))
    elf, stdin = tmp_path / "fib.elf", tmp_path / "stdin.bin"
    elf.write_bytes(rv.elf_bytes(fib_program()))
    stdin.write_bytes((20000).to_bytes(4, "little"))
    env = dict(os.environ, ZKHIP_AGG_CACHE_DIR=str(cache))
    _, vk1, _ = prove(tmp_path, "first", elf, stdin, 14, ZKHIP_AGG_CACHE_DIR=str(cache))

    def agg_vk(name):
        return subprocess.run([pm.CLI, "agg-vk", str(elf), str(tmp_path / "openvm.toml"), str(tmp_path / name), "14"], env=env, capture_output=True, text=True, timeout=900)

    r = agg_vk("a.vk")
    assert r.returncode == 0 and (tmp_path / "a.vk").read_bytes() == vk1, r.stderr[-2000:]
    _edit_cache(cache, flip_last_commit=True)          # (the last shape: the 51-chip one, which a Fibonacci guest never uses)
    _, vk2, _ = prove(tmp_path, "second", elf, stdin, 14, ZKHIP_AGG_CACHE_DIR=str(cache))
    assert vk2 != vk1                                   # the flow's key follows the file: exactly why it must not be handed on unchecked
    r = agg_vk("b.vk")
    assert r.returncode != 0 and "stale key cache" in r.stderr, r.stderr[-2000:]
