#!/usr/bin/env python3
"""Where the host-buffer PCS step spends its time (GPU box)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from zinc_amd import pcs

nv = int(sys.argv[1]) if len(sys.argv) > 1 else 20
field = pcs.FieldConfig(bench.BENCH_MODULUS, 4)
z = bench.splitmix64(0x5A494E43, 1 << nv)
r_y = field.map_to_field(np.ones(nv, dtype=np.int64))
for rep in range(3):
    T = [time.perf_counter()]
    def lap(): T.append(time.perf_counter())
    t = pcs.KeccakTranscript(); t.absorb(b"spartan")
    code = pcs.RaaCode(1 << nv, t); lap()
    pp = pcs.MultilinearZip.setup(1 << nv, code); lap()
    data, roots = pcs.MultilinearZip.commit(pp, z); lap()
    v = pcs.MultilinearZip.evaluate(pp, z, r_y, field); lap()
    tr = pcs.PcsTranscript()
    pcs.MultilinearZip.open(pp, z, data, r_y, field, tr); lap()
    proof = tr.into_proof(); lap()
    names = ["RaaCode::new", "setup", "commit", "evaluate", "open", "into_proof(copy)"]
    print("  ".join(f"{n} {1e3 * (b - a):.1f}" for n, a, b in zip(names, T, T[1:])), "ms")
