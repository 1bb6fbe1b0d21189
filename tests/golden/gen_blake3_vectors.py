#!/usr/bin/env python3
"""Generate tests/golden/blake3_vectors.json.

Source of truth: the official BLAKE3 C implementation, version 1.8.2 (the same
version the reference pins in Cargo.toml:30), as vendored by LLVM and exported
from /opt/rocm/lib/llvm/lib/libclang-cpp.so (`llvm_blake3_hasher_*`).  This is
an independent implementation of the hash the reference calls at
src/zip/pcs/utils.rs:90,107-112; it is used ONLY here, to produce fixtures.

Vectors: the official test-vector input pattern (byte i = i % 251) for every
length 0..64, plus seeded random 32-byte (Merkle leaf) and 64-byte (Merkle node)
messages.  Run once in the authoring container; the JSON is committed.
"""
import ctypes
import json
import os
import random

LIB = "/opt/rocm/lib/llvm/lib/libclang-cpp.so"


def main():
    lib = ctypes.CDLL(LIB)
    lib.llvm_blake3_version.restype = ctypes.c_char_p
    version = lib.llvm_blake3_version().decode()

    def b3(data: bytes) -> bytes:
        hasher = ctypes.create_string_buffer(4096)  # sizeof(blake3_hasher) == 1912
        lib.llvm_blake3_hasher_init(hasher)
        lib.llvm_blake3_hasher_update(hasher, data, ctypes.c_size_t(len(data)))
        out = ctypes.create_string_buffer(32)
        lib.llvm_blake3_hasher_finalize(hasher, out, ctypes.c_size_t(32))
        return out.raw

    assert b3(b"").hex() == "af1349b9f5f9a1a6a0404dea36dcc9499bcb25c9adc112b7cc9a93cae41f3262"
    vectors = []
    for n in range(65):
        msg = bytes(i % 251 for i in range(n))
        vectors.append({"kind": "pattern", "msg": msg.hex(), "hash": b3(msg).hex()})
    rng = random.Random(0x5A494E43)
    for kind, n, count in (("leaf32", 32, 64), ("node64", 64, 64), ("leaf24", 24, 8)):
        for _ in range(count):
            msg = bytes(rng.getrandbits(8) for _ in range(n))
            vectors.append({"kind": kind, "msg": msg.hex(), "hash": b3(msg).hex()})
    # sign-extended leaves as the commit produces them (limbs 2,3 all-0 or all-1)
    for v in (0, 1, -1, 2**63 - 1, -(2**63), 2**90 - 12345, -(2**90) + 999):
        limbs = [(v >> (64 * i)) & (2**64 - 1) for i in range(4)]
        msg = b"".join(l.to_bytes(8, "big") for l in limbs)  # int.rs:201-210
        vectors.append({"kind": "int4_leaf", "value": str(v), "msg": msg.hex(), "hash": b3(msg).hex()})
    out = {"source": f"BLAKE3 official C implementation {version} via LLVM ({LIB})", "vectors": vectors}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "blake3_vectors.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    print(f"wrote {len(vectors)} vectors to {path}")


if __name__ == "__main__":
    main()
