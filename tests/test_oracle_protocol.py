"""Protocol-level properties of the CPU oracle, restating the reference's own
round-trip / tamper / size tests.  CPU only."""
import numpy as np
import pytest

import _oracle as orc

BENCH_MODULUS = 106319353542452952636349991594949358997917625194731877894581586278529202198383
TEST_MODULUS_2 = 57316695564490278656402085503


def _prove(num_vars, modulus, fl, seed=0, small=False):
    rng = np.random.default_rng(seed)
    z = orc.Zip(num_vars)
    f = orc.make_field(modulus, fl)
    n = 1 << num_vars
    if small:
        evals = rng.integers(-128, 128, size=n, dtype=np.int64)
        point_i = rng.integers(-128, 128, size=num_vars, dtype=np.int64)
    else:
        evals = orc.splitmix64(seed + 0x5A494E43, n)
        point_i = rng.integers(-(2**63), 2**63, size=num_vars, dtype=np.int64)
    rows, layers, roots = z.commit(evals)
    point = orc.point_to_field(f, point_i)
    fs = orc.new_transcript()
    proof, cols, coeffs = z.open(f, evals, rows, layers, point, fs)
    ev = z.mle_eval(f, evals, point)
    return z, f, evals, rows, layers, roots, point, proof, cols, coeffs, ev


@pytest.mark.parametrize("num_vars,modulus,fl", [(8, TEST_MODULUS_2, 2), (8, BENCH_MODULUS, 4), (7, BENCH_MODULUS, 4),
                                                 (3, TEST_MODULUS_2, 2), (10, BENCH_MODULUS, 4)])
def test_prove_verify_roundtrip(num_vars, modulus, fl):
    """src/zip/tests.rs:116-146 (test_zip_evaluation)."""
    z, f, evals, rows, layers, roots, point, proof, cols, coeffs, ev = _prove(num_vars, modulus, fl, small=(fl == 2))
    assert z.verify(f, roots, point, ev, proof) == 0


def test_proof_size_is_correct_for_parameters():
    """src/zip/pcs/commit.rs:712-775."""
    for nv in (4, 8, 9):
        z, f, *_, proof, cols, coeffs, ev = _prove(nv, BENCH_MODULUS, 4)
        expect = z.row_len * 64 + 1000 * z.num_rows * (32 + 8 + 32 * z.depth) + z.row_len * 32
        assert proof.size == expect == z.proof_len(4)


def test_wrong_evaluation_is_rejected():
    """src/zip/tests.rs:84-113 (test_failing_zip_evaluation)."""
    z, f, evals, rows, layers, roots, point, proof, cols, coeffs, ev = _prove(8, BENCH_MODULUS, 4)
    assert z.verify(f, roots, point, (ev + 1) % BENCH_MODULUS, proof) == orc.ORC_ERR_PROOF


def test_tampered_proof_is_rejected():
    """src/zip/pcs/verify_z.rs:305-400 (tampered combined row / column values)."""
    z, f, evals, rows, layers, roots, point, proof, cols, coeffs, ev = _prove(8, BENCH_MODULUS, 4)
    bad = proof.copy()
    bad[3] ^= 1  # u' limb
    assert z.verify(f, roots, point, ev, bad) == orc.ORC_ERR_PROOF
    bad = proof.copy()
    bad[z.row_len * 64 + 5] ^= 1  # first opened column value
    assert z.verify(f, roots, point, ev, bad) == orc.ORC_ERR_PROOF
    bad = proof.copy()
    bad[-1] ^= 1  # last field element of the evaluation row
    assert z.verify(f, roots, point, ev, bad) == orc.ORC_ERR_PROOF
    bad = proof.copy()
    bad[z.row_len * 64 + z.num_rows * 32 + 8 + 3] ^= 1  # a Merkle sibling hash
    assert z.verify(f, roots, point, ev, bad, check_merkle=True) == orc.ORC_ERR_PROOF
    # the reference discards the Merkle result (verify_z.rs:99): faithful mode desyncs the stream
    assert z.verify(f, roots, point, ev, bad, check_merkle=False) != 0
    assert z.verify(f, roots, point, ev, proof[:-7]) == orc.ORC_ERR_TRANSCRIPT


def test_wrong_roots_rejected_only_with_merkle_check():
    z, f, evals, rows, layers, roots, point, proof, cols, coeffs, ev = _prove(6, BENCH_MODULUS, 4)
    bad = roots.copy()
    bad[0, 0] ^= 1
    assert z.verify(f, bad, point, ev, proof, check_merkle=True) == orc.ORC_ERR_PROOF


def test_commit_is_deterministic_and_rows_match_code_definition():
    """src/zip/pcs/commit.rs:253-283, 357-398, 521-557."""
    z = orc.Zip(9)  # odd num_vars: row_len = 2 * num_rows
    assert z.row_len == 32 and z.num_rows == 16 and z.codeword_len == 64
    evals = orc.splitmix64(1, 1 << 9)
    rows, layers, roots = z.commit(evals)
    rows2, layers2, roots2 = z.commit(evals)
    assert np.array_equal(rows, rows2) and np.array_equal(layers, layers2) and np.array_equal(roots, roots2)
    for r in range(z.num_rows):
        rc, enc = z.encode_row(evals[r * z.row_len:(r + 1) * z.row_len])
        assert rc == 0
        assert np.array_equal(enc, rows[r * z.codeword_len:(r + 1) * z.codeword_len])
        tree = orc.merkle_tree(z.depth, enc)
        assert np.array_equal(tree, layers[r])
        assert np.array_equal(tree[-1], roots[r])


def test_open_stream_layout():
    """Appendix A.4 of SURVEY.md: u' | 1000 x (column values, paths) | evaluation row."""
    z, f, evals, rows, layers, roots, point, proof, cols, coeffs, ev = _prove(8, BENCH_MODULUS, 4)
    rc, uprime = z.combine_rows_int(coeffs, evals)
    assert rc == 0
    pos = z.row_len * 64
    assert proof[:pos].tobytes() == uprime.astype("<u8").tobytes()
    per_col = z.num_rows * (32 + 8 + 32 * z.depth)
    for i in (0, 1, 999):
        c = int(cols[i])
        base = pos + i * per_col
        vals = rows.reshape(z.num_rows, z.codeword_len, 4)[:, c, :]
        assert proof[base:base + z.num_rows * 32].tobytes() == vals.astype("<u8").tobytes()
        pb = base + z.num_rows * 32
        for r in (0, z.num_rows - 1):
            rec = proof[pb + r * (8 + 32 * z.depth): pb + (r + 1) * (8 + 32 * z.depth)]
            assert int.from_bytes(rec[:8].tobytes(), "big") == z.depth
            assert rec[8:].tobytes() == orc.merkle_path(z.depth, layers[r], c).tobytes()
    tail = proof[pos + 1000 * per_col:]
    lr = z.num_rows.bit_length() - 1
    q0 = orc.build_eq_x_r(f, point[z.num_vars - lr:])
    row = z.combine_rows_field(f, q0, evals)
    assert tail.tobytes() == b"".join(orc.limbs_to_int(e).to_bytes(32, "big") for e in row)


def test_evaluation_row_identity_used_by_gpu_kernel():
    """sum_r q0_mont[r] * w[r][c] mod q == the reference's
    sum_r q0[r] (x) phi(w[r][c]) (Montgomery form) -- the identity K8 relies on."""
    q = BENCH_MODULUS
    z = orc.Zip(8)
    f = orc.make_field(q, 4)
    rng = np.random.default_rng(4)
    evals = orc.splitmix64(99, 1 << 8)
    q0_vals = [int(x) for x in rng.integers(0, 2**63, size=z.num_rows)]
    q0_vals = [(v * 0x1234567890ABCDEF1234567890ABCDEF1234567) % q for v in q0_vals]
    q0 = orc.field_elems(q0_vals, 4)
    row = z.combine_rows_field(f, q0, evals)
    m = evals.reshape(z.num_rows, z.row_len)
    for c in range(z.row_len):
        assert orc.limbs_to_int(row[c]) == sum(q0_vals[r] * int(m[r, c]) for r in range(z.num_rows)) % q


def test_single_row_polynomial():
    """num_rows == 1: no proximity test, evaluation row is phi(evals) (open_z.rs:80-88,100)."""
    z = orc.Zip(0)
    assert (z.row_len, z.num_rows, z.codeword_len, z.depth) == (1, 1, 2, 1)
    f = orc.make_field(BENCH_MODULUS, 4)
    evals = np.array([-5], dtype=np.int64)
    rows, layers, roots = z.commit(evals)
    fs = orc.new_transcript()
    proof, cols, coeffs = z.open(f, evals, rows, layers, np.zeros((0, 4), dtype=np.uint64), fs)
    assert proof.size == 1000 * (32 + 8 + 32) + 32
    assert int.from_bytes(proof[-32:].tobytes(), "big") == orc.field_from_i64(f, -5)


@pytest.mark.parametrize("num_vars", [1, 6, 9, 12])
def test_streamed_commit_open_columns_equals_commit_then_open(num_vars):
    """orc_commit_open_columns (the memory-light checker of the 2^26 tests: roots + whole opening blocks, row by row)
    against orc_commit + orc_open, which restate commit.rs:50-87 and open_z.rs:124-143 and are the pinned ones."""
    z, f, evals, rows, layers, roots, point, proof, cols, coeffs, ev = _prove(num_vars, BENCH_MODULUS, 4, seed=3)
    pick = np.array([0, 1, 7, 500, 999, 998], dtype=np.int64)
    roots2, blocks = z.commit_open_columns(evals, cols[pick])
    assert np.array_equal(roots2, roots)
    per_col = z.num_rows * (32 + 8 + 32 * z.depth)
    u_bytes = z.row_len * 64 if z.num_rows > 1 else 0
    for k, i in enumerate(pick):
        assert np.array_equal(blocks[k], proof[u_bytes + i * per_col: u_bytes + (i + 1) * per_col]), (k, i)
