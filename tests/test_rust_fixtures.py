"""Loader for fixtures produced by the REAL reference (integration/rust/fixture_dump.rs, run once by a maintainer
on a machine with cargo): permutation tables of `shuffle_seeded`, commit / open digests under MockTranscript and
`map_to_field` for a modulus with its top bit set.  With them the oracle -- and through it every GPU parity test --
is pinned to the Rust build itself: the shuffle's seed expansion and the signed-modulus quirk are the two pieces no
published vector covers (oracle/zip_oracle.h, include/zip_hip.h).

Skipped until tests/golden/rust_fixtures.json exists (this image has no cargo).  CPU only."""
import hashlib
import json
import os

import numpy as np
import pytest

import _oracle as orc

PATH = os.path.join(os.path.dirname(__file__), "golden", "rust_fixtures.json")
pytestmark = pytest.mark.skipif(not os.path.exists(PATH), reason="tests/golden/rust_fixtures.json not produced yet "
                                "(run integration/rust/fixture_dump.rs with cargo, see its header)")


@pytest.fixture(scope="module")
def fx():
    with open(PATH) as fh:
        return json.load(fh)


def test_fixture_format(fx):
    assert {"perm_tables", "commit_open", "map_to_field"} <= set(fx)


def test_shuffle_seeded_tables(fx):
    from zinc_amd.perm import shuffle_seeded_perm

    for e in fx["perm_tables"]:
        for impl in (orc.shuffle_perm, shuffle_seeded_perm):
            p = np.asarray(impl(e["seed"], e["len"]), dtype="<u4")
            assert [int(x) for x in p[:16]] == e["first16"][: min(16, e["len"])], (impl.__name__, e["seed"], e["len"])
            assert hashlib.sha3_256(p.tobytes()).hexdigest() == e["sha3_256_le_u32"]


def test_commit_and_open_digests(fx):
    for e in fx["commit_open"]:
        nv = e["num_vars"]
        z = orc.Zip(nv)  # MockTranscript seeds (1, 2), src/zip/pcs/tests.rs:24-37
        f = orc.make_field(int(e["modulus"]), 4)
        evals = orc.splitmix64(0x5A494E43, 1 << nv)
        rows, layers, roots = z.commit(evals)
        assert roots[0].tobytes().hex() == e["root0"]
        assert hashlib.sha3_256(roots.tobytes()).hexdigest() == e["roots_sha3_256"]
        assert hashlib.sha3_256(np.ascontiguousarray(rows, dtype="<u8").tobytes()).hexdigest() == e["rows_sha3_256"]
        point = orc.point_to_field(f, [1] * nv)
        proof, _, _ = z.open(f, evals, rows, layers, point, orc.new_transcript())
        assert proof.size == e["proof_len"]
        assert hashlib.sha3_256(proof.tobytes()).hexdigest() == e["proof_sha3_256"]


def test_map_to_field_including_the_signed_modulus_quirk(fx):
    for e in fx["map_to_field"]:
        f = orc.make_field(int(e["modulus"]), 4)
        for w, want in zip(e["inputs"], e["montgomery_be"]):
            assert orc.field_from_i64(f, int(w)).to_bytes(32, "big").hex() == want, (e["modulus"], w)
