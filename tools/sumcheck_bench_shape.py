#!/usr/bin/env python3
"""The workload of the reference's benches/sumcheck_benches.rs: rand_poly(nvars = 20, 2..4 multiplicands, 7 products)
over a 3-limb prime, MLSumcheck::prove_as_subprotocol with rand_poly_comb_fn -- through the host mirror with HOST tables
(the upload of the ~20 tables is part of the call), and the CPU restatement's time for the same proof.  GPU box."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as orc  # noqa: E402

orc.build()  # before the first GPU call: a GPU-initialised process must not fork + exec make
from zinc_amd import pcs  # noqa: E402

nv = int(sys.argv[1]) if len(sys.argv) > 1 else 20
Q, FL = 312829638388039969874974628075306023441, 3  # sumcheck_benches.rs:43
products = (2, 3, 4, 2, 3, 4, 3)                      # gen_range(2..5) seven times
rng = np.random.default_rng(7)
f = orc.make_field(Q, FL)
K = sum(products)
tables = np.zeros((K, 1 << nv, FL), dtype=np.uint64)
tables[:, :, :2] = rng.integers(0, 2**63, size=(K, 1 << nv, 2), dtype=np.int64).view(np.uint64)  # < q: canonical residues
masks, k = [], 0
for m in products:
    masks.append(sum(1 << (k + i) for i in range(m)))
    k += m
masks = np.array(masks, dtype=np.uint32)
coeffs = np.zeros((len(products), FL), dtype=np.uint64)
coeffs[:, 0] = rng.integers(1, 2**62, size=len(products))
field = pcs.FieldConfig(Q, FL)
for rep in range(3):
    t0 = time.perf_counter()
    msgs, rand = pcs.sumcheck_prove_products(pcs.KeccakTranscript(), tables, max(products), masks, coeffs, field)
    t1 = time.perf_counter()
    print(f"2^{nv}, {len(products)} products over {K} MLEs ({tables.nbytes >> 20} MiB of host tables): {1e3 * (t1 - t0):8.2f} ms", flush=True)
if "--oracle" in sys.argv:
    t0 = time.perf_counter()
    want_msgs, want_rand = orc.sumcheck_prove_products(f, tables, max(products), masks, [orc.limbs_to_int(c) for c in coeffs],
                                                       orc.new_transcript())
    t1 = time.perf_counter()
    print(f"oracle (CPU restatement, 1 thread): {t1 - t0:6.2f} s; identical: {np.array_equal(msgs, want_msgs) and np.array_equal(rand, want_rand)}")
