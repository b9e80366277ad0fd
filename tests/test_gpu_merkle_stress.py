"""GPU: the tree kernels under concurrency and across their launch forms (tools/merkle_stress.cpp, through the C ABI).

Several threads, each with a context of its own on the one GPU, commit the same matrices again and again; every digest layer must equal
the first commit's -- what the rare stale node of the guest flow would have broken -- and every form of the top of the
tree (the one-workgroup top kernel from 2^3 .. 2^8 nodes, the fused cooperative layers on and off) must give one root."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def stress_exe(tmp_path_factory):
    exe = tmp_path_factory.mktemp("merkle_stress") / "merkle_stress"
    lib_dir = os.path.join(ROOT, "zkvm-prover_amd")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "merkle_stress.cpp"), "-o", str(exe),
                           "-L", lib_dir, "-lzkhip", "-Wl,-rpath," + lib_dir, "-lpthread"])
    return str(exe)


def _run(exe, args, **env):
    out = subprocess.run([exe] + [str(a) for a in args], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert re.search(r", 0 differing layers", out.stdout), out.stdout[-2000:]
    return re.search(r"root of thread 0's tree: (.*)", out.stdout).group(1)


def test_layers_stay_equal_under_concurrent_commits(stress_exe):
    _run(stress_exe, [3, 1500, 15, 24])
    _run(stress_exe, [6, 600, 14, 40])


def test_every_form_of_the_tree_top_gives_one_root(stress_exe):
    roots = {_run(stress_exe, [1, 2, 13, 9], **{name: value}) for name, value in
             [("ZKHIP_TOP_MAX_LOG", "6"), ("ZKHIP_TOP_MAX_LOG", "8"), ("ZKHIP_TOP_MAX_LOG", "7"), ("ZKHIP_TOP_MAX_LOG", "3"), ("ZKHIP_COOP_MAX_LOG", "3")]}
    assert len(roots) == 1, roots
