"""CPU: the Rust FFI of the shim crate (integration/hip-backend/src/ffi.rs, source only -- no Rust toolchain in the
image) declares exactly the entry points, error codes and struct fields of include/zkhip.h."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = open(os.path.join(ROOT, "include", "zkhip.h")).read()
FFI = open(os.path.join(ROOT, "integration", "hip-backend", "src", "ffi.rs")).read()


def _c_functions():
    body = re.sub(r"/\*.*?\*/", "", HDR, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(zkhip_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", body, flags=re.S):
        args = [a.strip() for a in m.group(2).split(",")]
        out[m.group(1)] = 0 if args == ["void"] else len(args)
    return out


def _rust_functions():
    out = {}
    for m in re.finditer(r"pub fn (zkhip_[a-z0-9_]+)\s*\((.*?)\)\s*(?:->[^;]+)?;", FFI, flags=re.S):
        args = [a for a in m.group(2).split(",") if a.strip()]
        out[m.group(1)] = len(args)
    return out


def test_every_entry_point_is_bound_with_the_same_arity():
    c, r = _c_functions(), _rust_functions()
    assert len(c) >= 45
    assert sorted(c) == sorted(r), (sorted(set(c) - set(r)), sorted(set(r) - set(c)))
    assert {k: v for k, v in c.items() if r[k] != v} == {}


def test_error_codes_and_constants_match():
    for name, val in re.findall(r"#define (ZKHIP_[A-Z0-9_]+) \(?(-?\d+)\)?", HDR):
        m = re.search(r"pub const %s: [a-z_0-9]+ = (-?\d+);" % name, FFI)
        assert m and int(m.group(1)) == int(val), name


def test_struct_fields_match_in_order():
    body = re.sub(r"/\*.*?\*/", "", HDR, flags=re.S)
    for cname in ("zkhip_matrix", "zkhip_params", "zkhip_air", "zkhip_proof_layout", "zkhip_kernel_stat", "zkhip_v1_summary"):
        cdef = re.search(r"typedef struct \{([^{}]*)\} %s;" % cname, body).group(1)
        cfields = [re.search(r"([a-z_0-9]+)(?:\[[A-Z0-9_]+\])?\s*$", part.strip()).group(1)
                   for f in cdef.split(";") if f.strip() for part in f.split(",")]  # `size_t a, b;` declares two fields
        rdef = re.search(r"pub struct %s \{(.*?)\n\}" % cname, FFI, flags=re.S).group(1)
        rfields = re.findall(r"pub ([a-z_0-9]+):", rdef)
        assert cfields == rfields, (cname, cfields, rfields)
