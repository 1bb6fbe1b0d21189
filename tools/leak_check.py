#!/usr/bin/env python3
"""Repeats the full prover + verifier (and a standalone commit/open/verify) a few hundred times and prints the
device memory in use before / after: the per-proof handles recycle their blocks, nothing may grow.  GPU box."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _ccs  # noqa: E402
import torch  # noqa: E402
from zinc_amd import cabi, pcs  # noqa: E402

n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 200
field = pcs.FieldConfig(3618502788666131213697322783095070105623107215331596699973092056135872020481, 4)
inst = _ccs.dummy_ccs_from_len(1 << 12)
prover, verifier = pcs.ZincProver(), pcs.ZincVerifier()
args = (inst.matrices, inst.s, inst.d, inst.S, inst.c)


def used():
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 2**20


def one(prepared=None):
    proof = prover.prove(*args, inst.z[:1], inst.z[2:], pcs.KeccakTranscript(), field, prepared=prepared)
    verifier.verify(*args, proof, pcs.KeccakTranscript(), field, prepared=prepared)


one()
prep = prover.prepare(inst.matrices, inst.s, field)
for i in range(20):  # warm-up: the block pools and the context cache reach their steady state
    one(prep if i % 2 else None)
torch.cuda.synchronize()
before = used()
for i in range(n_iter):
    one(prep if i % 2 else None)
    if i % 50 == 49:
        print(f"after {i + 1:4d} proofs: {used():9.1f} MiB in use", flush=True)
after = used()
print(f"device memory in use: {before:.1f} MiB before, {after:.1f} MiB after {n_iter} prove+verify rounds")
cabi.lib().zip_release_cached_memory()
pcs.lib().zinc_zip_release_cached_contexts()
print(f"after releasing the caches: {used():.1f} MiB")
assert after - before < 64, "device memory grew"
