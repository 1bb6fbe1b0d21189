#!/usr/bin/env python3
"""BASELINE configs[4], the PCS share of it: ZincProver::commit_z_mle_and_prove_evaluation
(src/zinc/prover.rs:305-327) through the host mirror with HOST buffers in and out, as the Rust
prover would call it (witness upload + kernels + proof download + host Keccak), and the matching
MultilinearZip::verify.  GPU box."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from zinc_amd import pcs  # noqa: E402

nv = int(sys.argv[1]) if len(sys.argv) > 1 else 20
field = pcs.FieldConfig(bench.BENCH_MODULUS, 4)
z = bench.splitmix64(0x5A494E43, 1 << nv)
r_y = field.map_to_field(np.ones(nv, dtype=np.int64))
import ctypes as C  # noqa: E402

L = pcs.lib()
for rep in range(4):
    t = pcs.KeccakTranscript()
    t.absorb(b"spartan")
    h = C.c_void_p()
    t0 = time.perf_counter()
    rc = L.zinc_commit_z_mle_and_prove_evaluation(z.ctypes.data, z.size, r_y.ctypes.data, r_y.shape[0], t._h,
                                                  field._m.ctypes.data, field.limbs, 0, C.byref(h))
    t1 = time.perf_counter()
    assert rc == 0, L.zinc_last_error()
    roots = np.zeros((L.zinc_zip_proof_num_roots(h), 32), np.uint8)
    v = np.zeros(field.limbs, np.uint64)
    proof = np.zeros(L.zinc_zip_proof_len(h), np.uint8)
    L.zinc_zip_proof_read(h, roots.ctypes.data, v.ctypes.data, proof.ctypes.data)
    L.zinc_zip_proof_free(h)
    vt = pcs.KeccakTranscript()
    vt.absorb(b"spartan")
    vp = pcs.MultilinearZip.setup(1 << nv, pcs.RaaCode(1 << nv, vt))
    tr = pcs.PcsTranscript.from_proof(proof)  # (a copy: the Rust verifier already owns the bytes)
    t2 = time.perf_counter()
    pcs.MultilinearZip.verify(vp, roots, r_y, v, field, tr)
    t3 = time.perf_counter()
    print(f"2^{nv}: commit_z_mle_and_prove_evaluation {1e3 * (t1 - t0):8.2f} ms (witness {z.nbytes >> 20} MiB up, proof "
          f"{proof.size / 2**20:.0f} MiB down, fresh host buffers), verify {1e3 * (t3 - t2):8.2f} ms (proof up)")
