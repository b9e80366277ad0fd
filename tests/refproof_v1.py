"""Reader / writer of the proof container the reference's stored proofs use (TEST INFRASTRUCTURE).

`/root/reference/crates/{verifier/testdata/proofs,prover/testdata}/{chunk,batch}-proof*.json` hold
`VmInternalStarkProof { proofs: Vec<Proof<SC>>, public_values: Vec<BabyBear> }`
(`crates/types/src/proof.rs:69-74`), each field base64 of bincode-v1 (`crates/types/src/utils.rs:20-39`):
`Vec` = u64 LE length + items, fixed arrays inline, `Option` = u8 tag, every field element the u32 the
p3 Montgomery form keeps in memory (canonical = v * 2^-32 mod p).  `Proof<SC>` is OpenVM-v1's
(openvm-stark-backend 1.x `proof.rs`; the pipeline of BASELINE.json's north_star: quotient + FRI):

  commitments   { main_trace: Vec<[u32;8]>, after_challenge: Vec<[u32;8]>, quotient: [u32;8] }
  opening.proof { commit_phase_commits: Vec<[u32;8]>,
                  query_proofs: Vec<{ input_proof: Vec<{ opened_values: Vec<Vec<u32>>, opening_proof: Vec<[u32;8]> }>,
                                      commit_phase_openings: Vec<{ sibling_value: [u32;4], opening_proof: Vec<[u32;8]> }> }>,
                  final_poly: Vec<[u32;4]>, pow_witness: u32 }
  opening.values{ preprocessed: Vec<Adj>, main: Vec<Vec<Adj>>, after_challenge: Vec<Vec<Adj>>,
                  quotient: Vec<Vec<Vec<[u32;4]>>> }          Adj = { local: Vec<[u32;4]>, next: Vec<[u32;4]> }
  per_air       Vec<{ air_id: u64, degree: u64, exposed_values_after_challenge: Vec<Vec<[u32;4]>>, public_values: Vec<u32> }>
  rap_phase_seq_proof  Option<{ logup_pow_witness: u32 }>

The product-side codec of the same container is include/zkhip_codec.hpp; this Python twin exists so the fixture
generator and the tests do not depend on the thing they check.
"""
import struct

P = 2013265921
R_INV = pow(1 << 32, -1, P)
R = (1 << 32) % P


def from_monty(v):
    return v * R_INV % P


def to_monty(v):
    return v * R % P


class _Rd:
    def __init__(self, b):
        self.b, self.o = b, 0

    def u64(self):
        v = struct.unpack_from("<Q", self.b, self.o)[0]
        self.o += 8
        return v

    def u32(self):
        v = struct.unpack_from("<I", self.b, self.o)[0]
        self.o += 4
        return v

    def u8(self):
        v = self.b[self.o]
        self.o += 1
        return v

    def arr(self, n):
        v = list(struct.unpack_from("<%dI" % n, self.b, self.o))
        self.o += 4 * n
        return v

    def vec(self, f):
        n = self.u64()
        if n > len(self.b) - self.o:
            raise ValueError("vector length %d exceeds the remaining %d bytes" % (n, len(self.b) - self.o))
        return [f() for _ in range(n)]


def _read_proof(r):
    dig = lambda: r.arr(8)
    ext = lambda: r.arr(4)
    p = {"main_trace": r.vec(dig), "after_challenge": r.vec(dig), "quotient": dig()}

    def query():
        ip = r.vec(lambda: {"opened_values": r.vec(lambda: r.vec(r.u32)), "path": r.vec(dig)})
        co = r.vec(lambda: {"sibling": ext(), "path": r.vec(dig)})
        return {"input_proof": ip, "commit_phase_openings": co}

    p["fri"] = {"commit_phase_commits": r.vec(dig), "query_proofs": r.vec(query), "final_poly": r.vec(ext),
                "pow_witness": r.u32()}
    adj = lambda: {"local": r.vec(ext), "next": r.vec(ext)}
    p["opened"] = {"preprocessed": r.vec(adj), "main": r.vec(lambda: r.vec(adj)),
                   "after_challenge": r.vec(lambda: r.vec(adj)),
                   "quotient": r.vec(lambda: r.vec(lambda: r.vec(ext)))}
    p["per_air"] = r.vec(lambda: {"air_id": r.u64(), "degree": r.u64(), "exposed": r.vec(lambda: r.vec(ext)),
                                  "pvs": r.vec(r.u32)})
    p["logup_pow"] = r.u32() if r.u8() else None
    return p


def decode_proofs(blob):
    """bincode(Vec<Proof<SC>>) -> list of dicts (Montgomery words as stored)."""
    r = _Rd(blob)
    out = r.vec(lambda: _read_proof(r))
    if r.o != len(blob):
        raise ValueError("trailing bytes: %d of %d consumed" % (r.o, len(blob)))
    return out


def decode_public_values(blob):
    r = _Rd(blob)
    out = r.vec(r.u32)
    if r.o != len(blob):
        raise ValueError("trailing bytes")
    return out


class _Wr:
    def __init__(self):
        self.parts = []

    def u64(self, v):
        self.parts.append(struct.pack("<Q", v))

    def u32(self, v):
        self.parts.append(struct.pack("<I", v))

    def arr(self, a):
        self.parts.append(struct.pack("<%dI" % len(a), *a))

    def vec(self, xs, f):
        self.u64(len(xs))
        for x in xs:
            f(x)


def _write_proof(w, p):
    w.vec(p["main_trace"], w.arr)
    w.vec(p["after_challenge"], w.arr)
    w.arr(p["quotient"])
    f = p["fri"]
    w.vec(f["commit_phase_commits"], w.arr)

    def query(q):
        def batch(b):
            w.vec(b["opened_values"], lambda row: w.vec(row, w.u32))
            w.vec(b["path"], w.arr)

        def step(s):
            w.arr(s["sibling"])
            w.vec(s["path"], w.arr)

        w.vec(q["input_proof"], batch)
        w.vec(q["commit_phase_openings"], step)

    w.vec(f["query_proofs"], query)
    w.vec(f["final_poly"], w.arr)
    w.u32(f["pow_witness"])

    def adj(a):
        w.vec(a["local"], w.arr)
        w.vec(a["next"], w.arr)

    o = p["opened"]
    w.vec(o["preprocessed"], adj)
    w.vec(o["main"], lambda m: w.vec(m, adj))
    w.vec(o["after_challenge"], lambda m: w.vec(m, adj))
    w.vec(o["quotient"], lambda a: w.vec(a, lambda c: w.vec(c, w.arr)))

    def air(a):
        w.u64(a["air_id"])
        w.u64(a["degree"])
        w.vec(a["exposed"], lambda ph: w.vec(ph, w.arr))
        w.vec(a["pvs"], w.u32)

    w.vec(p["per_air"], air)
    if p["logup_pow"] is None:
        w.parts.append(b"\x00")
    else:
        w.parts.append(b"\x01")
        w.u32(p["logup_pow"])


def encode_proofs(proofs):
    w = _Wr()
    w.vec(proofs, lambda p: _write_proof(w, p))
    return b"".join(w.parts)


def shape_of(p):
    """The quantities that fix a proof's byte layout (what a verifying key + FRI parameters tell a decoder)."""
    q = p["fri"]["query_proofs"][0]
    return {
        "n_airs": len(p["per_air"]),
        "log_degrees": [a["degree"].bit_length() - 1 for a in p["per_air"]],
        "n_queries": len(p["fri"]["query_proofs"]),
        "n_fri_layers": len(p["fri"]["commit_phase_commits"]),
        "n_final_poly": len(p["fri"]["final_poly"]),
        "batches": [{"widths": [len(r) for r in b["opened_values"]], "log_height": len(b["path"])}
                    for b in q["input_proof"]],
    }
