"""Shared by the SHA-256 chip's CPU and GPU tests: padding into blocks, chaining records, reading digests out of a trace."""
import ctypes as C

import numpy as np

IV = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19]
WIDTH, ROWS = 433, 65


def padded_blocks(msg):
    """FIPS 180-4 5.1.1: message || 0x80 || zeros || 64-bit big-endian bit length, as rows of sixteen big-endian words"""
    data = msg + b"\x80" + b"\x00" * ((55 - len(msg)) % 64) + (8 * len(msg)).to_bytes(8, "big")
    assert len(data) % 64 == 0
    return np.frombuffer(data, dtype=">u4").astype(np.uint32).reshape(-1, 16)


def chained_records(msgs, compress):
    """one record (H_in[8] | M[16]) per block of every message, H_in chained through `compress`; returns (records, index of each message's last block)"""
    recs, last = [], []
    for m in msgs:
        h = np.array(IV, np.uint32)
        for blk in padded_blocks(m):
            recs.append(np.concatenate([h, blk]))
            h = compress(h, blk)
        last.append(len(recs) - 1)
    return np.stack(recs).astype(np.uint32), last


def digest_of_row(tr, row):
    """the eight state words of a row (bits -> big-endian bytes)"""
    words = [sum(int(tr[32 * w + j, row]) << j for j in range(32)) for w in range(8)]
    return b"".join(w.to_bytes(4, "big") for w in words)


def ora_sha256(ora, msg):
    l = ora.lib()
    l.ora_sha256.restype = None
    l.ora_sha256.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint8)]
    out = (C.c_uint8 * 32)()
    l.ora_sha256(msg, len(msg), out)
    return bytes(out).hex()


def ora_compress(ora, h, m):
    l = ora.lib()
    l.ora_sha256_compress.restype = None
    l.ora_sha256_compress.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    st = np.ascontiguousarray(h, dtype=np.uint32).copy()
    blk = np.ascontiguousarray(m, dtype=np.uint32)
    l.ora_sha256_compress(ora.p32(st), ora.p32(blk))
    return st


def ora_trace(ora, records, log_height):
    l = ora.lib()
    l.ora_sha256_trace.restype = None
    l.ora_sha256_trace.argtypes = [C.POINTER(C.c_uint32), C.c_size_t, C.c_uint, C.POINTER(C.c_uint32)]
    recs = np.ascontiguousarray(records, dtype=np.uint32).reshape(-1, 24)
    tr = np.zeros((WIDTH, 1 << log_height), np.uint32)
    l.ora_sha256_trace(ora.p32(recs), len(recs), log_height, ora.p32(tr))
    return tr
