"""CPU: include/zkhip.h zkhip_config -- every behaviour-changing switch of the library is a field; the ZKHIP_* environment variables are
overrides read in ONE place (zkhip_config_default)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = "import sys; sys.path.insert(0, %r); import zkvm_prover_amd as z; c = z.Config.default(); print(c.host_sponge, c.host_sponge_min_words, c.jit, c.jit_min_log_work, c.quot_slices, c.coop_max_log, c.witness_threads, c.pin_witness, c.parallel_queries, c.jit_cache_dir.decode())" % ROOT


def run(env):
    e = {k: v for k, v in os.environ.items() if not k.startswith("ZKHIP_")}
    e.update(env)
    r = subprocess.run([sys.executable, "-c", CODE], capture_output=True, text=True, env=e)
    assert r.returncode == 0, r.stderr
    return r.stdout.split()


def test_defaults_and_environment_overrides(tmp_path):
    d = run({})
    assert d[:9] == ["1", "8192", "1", "26", "1", "15", "0", "1", "1"]
    o = run({"ZKHIP_NO_HOST_SPONGE": "1", "ZKHIP_HOST_SPONGE_MIN_WORDS": "100", "ZKHIP_FORCE_JIT": "1", "ZKHIP_JIT_MIN_LOG_WORK": "20", "ZKHIP_NO_QUOT_SLICES": "1",
             "ZKHIP_COOP_MAX_LOG": "12", "ZKHIP_WITNESS_THREADS": "4", "ZKHIP_NO_PIN_WITNESS": "1", "ZKHIP_RECURSION_SERIAL_QUERIES": "1", "ZKHIP_JIT_CACHE_DIR": str(tmp_path)})
    assert o == ["0", "100", "2", "20", "0", "12", "4", "0", "0", str(tmp_path)]
    assert run({"ZKHIP_NO_JIT": "1"})[2] == "0"


def test_the_library_reads_its_overrides_in_one_place():
    """getenv("ZKHIP_...") in the library: the configuration reader, plus the listed measurement / A-B switches of single kernels."""
    allowed = {"ZKHIP_KEYGEN_TIMING", "ZKHIP_RECURSION_TIMING", "ZKHIP_JIT_TILE", "ZKHIP_JIT_FLAT", "ZKHIP_JIT_OPT", "ZKHIP_NTT_MAX_LOG_R", "ZKHIP_NTT_LOG_C", "ZKHIP_NTT_LEGACY", "ZKHIP_NTT_PRIO",
               # round 6: the measured-and-closed forms (docs/round6.md 3) stay selectable for their parity tests and for re-measurement
               "ZKHIP_LDE_FUSED", "ZKHIP_LDE_FUSED_WAVES", "ZKHIP_JIT_UNROLL", "ZKHIP_JIT_SHARED", "ZKHIP_JIT_SHARED_WAVES"}
    src = os.path.join(ROOT, "zkvm-prover_amd", "csrc")
    for f in sorted(os.listdir(src)):
        if not f.endswith((".hip", ".hpp", ".cpp")):
            continue
        text = open(os.path.join(src, f)).read()
        names = set(re.findall(r'getenv\("(ZKHIP_[A-Z0-9_]+)"\)', text))
        if f == "api.hip":
            continue
        assert names <= allowed, (f, names - allowed)
