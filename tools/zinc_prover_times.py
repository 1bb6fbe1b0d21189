#!/usr/bin/env python3
"""BASELINE configs[4]: the full ZincProver on the spartan_benches.rs instance (get_dummy_ccs_Z_from_z_length,
src/ccs/test_utils.rs:161-171; A = B = I, C = diag(z); 2^nv constraints) through the host mirror with HOST buffers
in and out.  Prints SpartanProver::prove (what the reference's bench times) and Prover::prove (with the Zip PCS
step), and, with --oracle, the CPU restatement's time for the Spartan part on this box (one thread).  GPU box."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from zinc_amd import pcs  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("nv", type=int, nargs="?", default=20)
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--oracle", action="store_true")
ap.add_argument("--prime", choices=["bench", "stark", "256"], default="stark")
args = ap.parse_args()
if args.oracle:
    import _oracle  # noqa: E402

    _oracle.build()  # before the first GPU call: a GPU-initialised process must not fork + exec make

PRIMES = {"bench": bench.BENCH_MODULUS,
          "stark": 3618502788666131213697322783095070105623107215331596699973092056135872020481,  # spartan_benches.rs:161
          "256": 115792089237316195423570985008687907853269984665640564039457584007913129639747}  # :152
import _ccs  # noqa: E402

inst = _ccs.dummy_ccs_from_len(1 << args.nv)
field = pcs.FieldConfig(PRIMES[args.prime], 4)
import ctypes as C  # noqa: E402
from zinc_amd import cabi  # noqa: E402

L = pcs.lib()
x, w = inst.z[:1], inst.z[2:]
arr = (cabi.SparseMatrix * inst.t)()
for k, M in enumerate(inst.matrices):
    arr[k] = cabi.SparseMatrix(M.n_rows, M.n_cols, M.row_ptr.ctypes.data, M.col_idx.ctypes.data, M.values.ctypes.data)
masks, cv = inst.masks, np.array(inst.c, dtype=np.int64)
fl, s, d = field.limbs, inst.s, inst.d
sp = dict(msgs1=np.zeros((s, d + 2, fl), np.uint64), msgs2=np.zeros((s, 3, fl), np.uint64),
          V_s=np.zeros((inst.t, fl), np.uint64), r_y=np.zeros((s, fl), np.uint64))


prep = C.c_void_p()
assert L.zinc_prover_prepare(arr, inst.t, s, field._m.ctypes.data, fl, 0, C.byref(prep)) == 0, L.zinc_last_error()


def run(with_pcs, prepared=None, keep=None):
    """the C facade only: what a Rust caller pays (CSR + z in host memory, proof kept in the returned handle)"""
    t = pcs.KeccakTranscript()
    h = C.c_void_p()
    t0 = time.perf_counter()
    rc = L.zinc_prover_prove(arr, inst.t, s, d, inst.q, masks.ctypes.data, cv.ctypes.data, x.ctypes.data, x.size,
                             w.ctypes.data, w.size, t._h, field._m.ctypes.data, fl, 0, prepared, with_pcs, sp["msgs1"].ctypes.data,
                             sp["msgs2"].ctypes.data, sp["V_s"].ctypes.data, sp["r_y"].ctypes.data, C.byref(h))
    dt = time.perf_counter() - t0
    assert rc == 0, L.zinc_last_error()
    n = 0
    if with_pcs:
        n = L.zinc_zip_proof_len(h)
        if keep is not None:  # hand the ZipProof out for the verifier timing
            keep["roots"] = np.zeros((L.zinc_zip_proof_num_roots(h), 32), np.uint8)
            keep["v"] = np.zeros(fl, np.uint64)
            keep["proof"] = np.zeros(n, np.uint8)
            L.zinc_zip_proof_read(h, keep["roots"].ctypes.data, keep["v"].ctypes.data, keep["proof"].ctypes.data)
        L.zinc_zip_proof_free(h)
    return dt, n


def verify(zp, prepared):
    """Verifier::verify through the facade (proof bytes in host memory)"""
    t = pcs.KeccakTranscript()
    t0 = time.perf_counter()
    rc = L.zinc_verifier_verify(arr, inst.t, s, d, inst.q, masks.ctypes.data, cv.ctypes.data, t._h, field._m.ctypes.data, fl, 0,
                                prepared, sp["msgs1"].ctypes.data, sp["msgs2"].ctypes.data, sp["V_s"].ctypes.data, 1,
                                zp["roots"].ctypes.data, zp["roots"].shape[0], zp["v"].ctypes.data, zp["proof"].ctypes.data,
                                zp["proof"].size, None, None, None)
    dt = time.perf_counter() - t0
    assert rc == 0, L.zinc_last_error()
    return dt


for rep in range(args.reps):
    ts, _ = run(0)
    tf, n = run(1)
    tsp, _ = run(0, prep)
    zp = {}
    tfp, _ = run(1, prep, zp)
    tv = verify(zp, prep) if args.prime != "256" else float("nan")  # (the reference rejects its own PCS proofs for that modulus)
    print(f"2^{args.nv} ({args.prime}): SpartanProver::prove {1e3 * ts:8.2f} ms   Prover::prove {1e3 * tf:8.2f} ms "
          f"(PCS proof {n / 2**20:.0f} MiB);  with the circuit prepared: {1e3 * tsp:8.2f} / {1e3 * tfp:8.2f} ms; Verifier::verify {1e3 * tv:7.2f} ms", flush=True)

L.zinc_prepared_ccs_free(prep)
if args.oracle:
    import _oracle as orc  # noqa: E402
    f = orc.make_field(PRIMES[args.prime], 4)
    o = orc.Ccs(inst)
    t0 = time.perf_counter()
    want = o.spartan_prove(f, orc.new_transcript())
    t1 = time.perf_counter()
    same = all(np.array_equal(sp[k], want[k]) for k in ("msgs1", "msgs2", "V_s", "r_y"))
    print(f"oracle (CPU restatement, 1 thread) SpartanProver::prove {t1 - t0:8.2f} s; device proof identical: {same}")
