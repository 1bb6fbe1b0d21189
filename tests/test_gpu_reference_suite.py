"""The reference's own unit tests for this path, restated one by one against the HIP path through the
host mirror (zinc_amd/pcs.py -> libzinc_zip.so -> libzip_hip.so).  Each test cites the test it
restates (paths relative to src/zip in the reference).  Where the reference only asserts `is_ok()`,
the result is additionally compared with the CPU oracle."""
import numpy as np
import pytest

import _oracle as orc

pytestmark = pytest.mark.gpu

MODULUS = 57316695564490278656402085503  # field_config!(57316695564490278656402085503, FIELD_LIMBS = 4)
FL = 4


@pytest.fixture(scope="module")
def pcs():
    from zinc_amd import cabi, pcs as m

    if cabi.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu tests must run on the MI355X box")
    return m


def i256(v):
    """Int<4> limbs of a Python int"""
    return np.array(orc.int_to_limbs(v % (1 << 256), 4), dtype=np.uint64)


def to_int(limbs):
    return orc.limbs_to_int(limbs, signed=True)


def setup_test_params(pcs, num_vars):
    """commit.rs:222-238 / open_z.rs:177-194: MockTranscript code, evaluations 1..=2^n"""
    pp = pcs.MultilinearZip.setup(1 << num_vars, pcs.RaaCode(1 << num_vars))  # RaaCode(.., None) = MockTranscript
    assert pp.num_rows == 1 << ((num_vars + 1) // 2)  # 1 << div_ceil(num_vars, 2)
    return pp, np.arange(1, (1 << num_vars) + 1, dtype=np.int64)


def int_mle_eval(evals, point):
    """DenseMultilinearExtension::<Int>::evaluate: fold variable 0 = LSB first, exact integers"""
    cur = [int(x) for x in evals]
    for p in point:
        cur = [cur[2 * b] + int(p) * (cur[2 * b + 1] - cur[2 * b]) for b in range(len(cur) // 2)]
    return cur[0]


def field_of(pcs, values):
    return pcs.FieldConfig(MODULUS, FL).map_to_field(np.asarray(values, dtype=np.int64))


def eval_in_field(pcs, evals, point_int):
    """poly.evaluate(&point_int).map_to_field(config) for values that stay small"""
    q = MODULUS
    v = int_mle_eval(evals, point_int) % q
    return np.array(orc.int_to_limbs(v * (1 << 256) % q, FL), dtype=np.uint64)  # Montgomery form


def oracle_for(pp, seeds=(1, 2)):
    return orc.Zip(pp.num_vars, seeds=seeds)


def prove(pcs, pp, poly, data, point_int):
    field = pcs.FieldConfig(MODULUS, FL)
    t = pcs.PcsTranscript()
    pcs.MultilinearZip.open(pp, poly, data, field.map_to_field(np.asarray(point_int, dtype=np.int64)), field, t)
    return t.into_proof()


def verify(pcs, pp, roots, point_int, eval_mont, proof):
    field = pcs.FieldConfig(MODULUS, FL)
    pcs.MultilinearZip.verify(pp, roots, field.map_to_field(np.asarray(point_int, dtype=np.int64)), eval_mont, field,
                              pcs.PcsTranscript.from_proof(proof))


# ------------------------------------------------------------------------------ pcs/commit.rs
def test_commit_rejects_too_many_variables(pcs):  # commit.rs:241
    pp, _ = setup_test_params(pcs, 3)
    with pytest.raises(pcs.InvalidPcsParam):
        pcs.MultilinearZip.commit(pp, np.arange(1, 17, dtype=np.int64), num_vars=4)


def test_commit_is_deterministic(pcs):  # commit.rs:253
    pp, poly = setup_test_params(pcs, 3)
    assert np.array_equal(pcs.MultilinearZip.commit(pp, poly)[1], pcs.MultilinearZip.commit(pp, poly)[1])


def test_different_polynomials_produce_different_commitments(pcs):  # commit.rs:263
    pp, _ = setup_test_params(pcs, 3)
    r1 = pcs.MultilinearZip.commit(pp, np.full(8, 1, np.int64))[1]
    r2 = pcs.MultilinearZip.commit(pp, np.full(8, 2, np.int64))[1]
    assert not np.array_equal(r1, r2)


@pytest.mark.parametrize("nv,evals", [(4, [42] * 16), (2, [1, 2, 3, 4])])
def test_commit_succeeds_for_small_polynomials(pcs, nv, evals):  # commit.rs:276, 289
    pp, _ = setup_test_params(pcs, nv)
    evals = np.array(evals, dtype=np.int64)
    _, roots = pcs.MultilinearZip.commit(pp, evals)
    assert np.array_equal(roots, oracle_for(pp).commit(evals)[2])


def test_merkle_tree_has_correct_depth_and_count(pcs):  # commit.rs:302, 431
    pp, poly = setup_test_params(pcs, 3)
    data, roots = pcs.MultilinearZip.commit(pp, poly)
    trees = data.rows_merkle_trees
    assert len(trees) == pp.num_rows == roots.shape[0]
    assert all(t.depth == (pp.codeword_len - 1).bit_length() for t in trees)
    assert data.rows.shape[0] == pp.num_rows * pp.codeword_len


def test_commit_no_merkle_produces_empty_trees(pcs):  # commit.rs:313
    pp, poly = setup_test_params(pcs, 3)
    data, commitment_roots = pcs.MultilinearZip.commit_no_merkle(pp, poly)
    assert data.rows.shape[0] == pp.num_rows * pp.codeword_len
    assert data.rows_merkle_trees == [] and commitment_roots.shape[0] == 0


def test_batch_commit(pcs):  # commit.rs:325, 401, 496
    pp, poly = setup_test_params(pcs, 3)
    outs = pcs.MultilinearZip.batch_commit(pp, [np.arange(1, 9, dtype=np.int64), np.arange(9, 17, dtype=np.int64)])
    assert len(outs) == 2 and not np.array_equal(outs[0][1], outs[1][1])
    (bdata, broots), (sdata, sroots) = pcs.MultilinearZip.batch_commit(pp, [poly])[0], pcs.MultilinearZip.commit(pp, poly)
    assert np.array_equal(broots, sroots) and np.array_equal(bdata.rows, sdata.rows)
    assert pcs.MultilinearZip.batch_commit(pp, []) == []


def test_encode_rows_sizes_and_definition(pcs):  # commit.rs:342, 357, 416, 505
    pp, poly = setup_test_params(pcs, 3)
    enc = pcs.MultilinearZip.encode_rows(pp, poly)
    assert enc.shape[0] == pp.num_rows * pp.codeword_len
    z = oracle_for(pp)
    for i in range(pp.num_rows):  # row by row against linear_code.encode_wide
        rc, want = z.encode_row(poly[i * pp.row_len:(i + 1) * pp.row_len])
        assert rc == 0 and np.array_equal(enc[i * pp.codeword_len:(i + 1) * pp.codeword_len], want), i
    assert np.count_nonzero(enc.any(axis=1)) > 0
    # a single row (num_vars = 0 is the one-row matrix of this code)
    pp1 = pcs.MultilinearZip.setup(1, pcs.RaaCode(1))
    assert pp1.num_rows == 1
    assert pcs.MultilinearZip.encode_rows(pp1, np.array([5], dtype=np.int64)).shape[0] == pp1.codeword_len


def test_corrupted_encoding_changes_merkle_root(pcs):  # commit.rs:384
    pp, poly = setup_test_params(pcs, 3)
    data, roots = pcs.MultilinearZip.commit(pp, poly)
    rows = data.rows.copy()
    rows[0] = i256(999999)
    new_tree = pcs.MerkleTree.new(data.rows_merkle_trees[0].depth, rows[: pp.codeword_len])
    assert not np.array_equal(new_tree.root, roots[0])


@pytest.mark.parametrize("evals", [[0] * 8, [1, -1] * 4, [2**63 - 1] * 8])
def test_commit_special_polynomials(pcs, evals):  # commit.rs:473, 485, 618
    pp, _ = setup_test_params(pcs, 3)
    evals = np.array(evals, dtype=np.int64)
    data, roots = pcs.MultilinearZip.commit(pp, evals)
    rows_o, _, roots_o = oracle_for(pp).commit(evals)
    assert roots.shape[0] == pp.num_rows and len(data.rows_merkle_trees) == pp.num_rows
    assert np.array_equal(roots, roots_o) and np.array_equal(data.rows, rows_o.reshape(-1, 4))


def test_merkle_root_integrity_is_maintained(pcs):  # commit.rs:521
    pp, _ = setup_test_params(pcs, 3)
    data, roots = pcs.MultilinearZip.commit(pp, np.full(8, 42, np.int64))
    rows = data.rows
    for i, tree in enumerate(data.rows_merkle_trees):
        independent = pcs.MerkleTree.new(tree.depth, rows[i * pp.codeword_len:(i + 1) * pp.codeword_len])
        assert np.array_equal(tree.root, independent.root) and np.array_equal(roots[i], independent.root)
        assert np.array_equal(tree.layers, independent.layers)


@pytest.mark.parametrize("num_vars,expected_rows", [(2, 2), (4, 4), (6, 8), (16, 256)])
def test_matrix_dimensions_and_many_variables(pcs, num_vars, expected_rows):  # commit.rs:538, 595, 606
    pp, poly = setup_test_params(pcs, num_vars)
    assert pp.num_rows == expected_rows == 1 << (num_vars // 2) and pp.num_vars == num_vars
    _, roots = pcs.MultilinearZip.commit(pp, poly)
    assert roots.shape[0] == pp.num_rows
    if num_vars == 2:
        assert pp.row_len == 2


def test_linear_code_preserves_linearity(pcs):  # commit.rs:559, code_raa.rs:279
    pp, poly = setup_test_params(pcs, 4)
    enc = pcs.MultilinearZip.encode_rows(pp, poly)
    rl, cw = pp.row_len, pp.codeword_len
    a, b = 3, 5
    combined = a * poly[:rl] + b * poly[rl:2 * rl]
    # the combination is itself a row: encode it as row 0 of another polynomial
    enc_c = pcs.MultilinearZip.encode_rows(pp, np.concatenate([combined, np.zeros(poly.size - rl, np.int64)]))[:cw]
    want = [(a * to_int(enc[i]) + b * to_int(enc[cw + i])) for i in range(cw)]
    assert [to_int(x) for x in enc_c] == want
    # encoding the zero vector gives the zero codeword (code_raa.rs:301)
    assert not pcs.MultilinearZip.encode_rows(pp, np.zeros(poly.size, np.int64)).any()


def test_commit_panics_if_evaluations_not_multiple_of_row_len(pcs):  # commit.rs:585 (#[should_panic])
    pp, poly = setup_test_params(pcs, 4)
    with pytest.raises(pcs.ReferencePanic):
        pcs.MultilinearZip.commit(pp, poly[:15], num_vars=4)


def test_merkle_tree_new_panics_on_non_power_of_two_leaves(pcs):  # commit.rs:634
    with pytest.raises(pcs.ReferencePanic, match=r"leaves.len\(\).is_power_of_two\(\)"):
        pcs.MerkleTree.new(3, np.arange(7, dtype=np.uint64).reshape(7, 1))


def test_verifier_rejects_commitment_with_bad_proximity(pcs):  # commit.rs:643
    n = 3
    t = pcs.KeccakTranscript()
    pp = pcs.MultilinearZip.setup(1 << n, pcs.RaaCode(1 << n, t))
    rng = np.random.default_rng(0)
    evals = rng.integers(-128, 128, size=1 << n, dtype=np.int64)
    point_int = rng.integers(-(2**63), 2**63 - 1, size=n, dtype=np.int64)
    data, roots = pcs.MultilinearZip.commit(pp, evals)
    rows = data.rows.copy()
    rows[0] = i256(to_int(rows[0]) + 1)
    bad = pcs.MultilinearZipData.new(pp, rows, data.rows_merkle_trees)  # data.rows[0] += 1, trees untouched
    proof = prove(pcs, pp, evals, bad, point_int)
    field = pcs.FieldConfig(MODULUS, FL)
    ev = pcs.MultilinearZip.evaluate(pp, evals, field.map_to_field(point_int), field)
    with pytest.raises(pcs.InvalidPcsOpen):
        verify(pcs, pp, roots, point_int, ev, proof)


def test_proof_size_is_correct_for_parameters(pcs):  # commit.rs:712
    num_vars = 4
    t = pcs.KeccakTranscript()
    pp = pcs.MultilinearZip.setup(1 << num_vars, pcs.RaaCode(1 << num_vars, t))
    rng = np.random.default_rng(1)
    evals = rng.integers(-128, 128, size=1 << num_vars, dtype=np.int64)
    data, _ = pcs.MultilinearZip.commit(pp, evals)
    proof = prove(pcs, pp, evals, data, rng.integers(-(2**63), 2**63 - 1, size=num_vars, dtype=np.int64))
    depth = (pp.codeword_len - 1).bit_length()
    expected = 1 * pp.row_len * 64 + 1000 * (pp.num_rows * 32 + pp.num_rows * (8 + depth * 32)) + pp.row_len * 8 * FL
    assert proof.size == expected


# ------------------------------------------------------------------------------ pcs/open_z.rs
POINT4 = [2, 3, 4, 5]  # (0..num_vars).map(|i| i + 2)


def _roundtrip(pcs, nv, evals, point_int, eval_mont=None, data_from=None, expect_ok=True):
    pp, _ = setup_test_params(pcs, nv)
    evals = np.asarray(evals, dtype=np.int64)
    data, roots = pcs.MultilinearZip.commit(pp, evals if data_from is None else np.asarray(data_from, dtype=np.int64))
    proof = prove(pcs, pp, evals, data, point_int)
    ev = eval_in_field(pcs, evals, point_int) if eval_mont is None else eval_mont
    if expect_ok:
        verify(pcs, pp, roots, point_int, ev, proof)
    else:
        with pytest.raises(pcs.InvalidPcsOpen):
            verify(pcs, pp, roots, point_int, ev, proof)
    return pp, roots, proof


def test_successful_opening_with_correct_polynomial_and_hint(pcs):  # open_z.rs:202, verify_z.rs:273
    pp, poly = setup_test_params(pcs, 4)
    point = np.random.default_rng(2).integers(-(2**63), 2**63 - 1, size=4, dtype=np.int64)
    data, roots = pcs.MultilinearZip.commit(pp, poly)
    proof = prove(pcs, pp, poly, data, point)
    z = oracle_for(pp)
    f = orc.make_field(MODULUS, FL)
    rows_o, layers_o, _ = z.commit(poly)
    proof_o, _, _ = z.open(f, poly, rows_o, layers_o, orc.point_to_field(f, point), orc.new_transcript())
    assert np.array_equal(proof, proof_o)


def test_successful_opening_with_a_close_codeword(pcs):  # open_z.rs:222
    pp, poly = setup_test_params(pcs, 4)
    original, _ = pcs.MultilinearZip.commit(pp, poly)
    rows = original.rows.copy()
    rows[0] = i256(to_int(rows[0]) + 1)
    depth = (pp.codeword_len - 1).bit_length()
    trees = [pcs.MerkleTree.new(depth, rows[r * pp.codeword_len:(r + 1) * pp.codeword_len]) for r in range(pp.num_rows)]
    corrupted = pcs.MultilinearZipData.new(pp, rows, trees)
    proof = prove(pcs, pp, poly, corrupted, np.random.default_rng(3).integers(-(2**63), 2**63 - 1, size=4, dtype=np.int64))
    assert proof.size > 0  # `open` does not inspect the hint


def test_failed_opening_due_to_incorrect_polynomial(pcs):  # open_z.rs:261
    pp, poly1 = setup_test_params(pcs, 4)
    data, roots = pcs.MultilinearZip.commit(pp, poly1)
    poly2 = np.arange(20, 36, dtype=np.int64)
    proof = prove(pcs, pp, poly2, data, POINT4)
    with pytest.raises(pcs.InvalidPcsOpen):
        verify(pcs, pp, roots, POINT4, eval_in_field(pcs, poly1, POINT4), proof)


def test_failed_opening_due_to_a_hint_that_is_not_close(pcs):  # open_z.rs:294
    pp, poly = setup_test_params(pcs, 4)
    original, roots = pcs.MultilinearZip.commit(pp, poly)
    rows = original.rows.copy()
    for i in range(pp.codeword_len // 2 + 1):
        rows[i] = i256(to_int(rows[i]) + 1)
    depth = (pp.codeword_len - 1).bit_length()
    trees = [pcs.MerkleTree.new(depth, rows[r * pp.codeword_len:(r + 1) * pp.codeword_len]) for r in range(pp.num_rows)]
    proof = prove(pcs, pp, poly, pcs.MultilinearZipData.new(pp, rows, trees), POINT4)
    with pytest.raises(pcs.InvalidPcsOpen):
        verify(pcs, pp, roots, POINT4, eval_in_field(pcs, poly, POINT4), proof)


def test_failed_opening_due_to_oversized_polynomial(pcs):  # open_z.rs:348
    pp, poly = setup_test_params(pcs, 4)
    data, _ = pcs.MultilinearZip.commit(pp, poly)
    field = pcs.FieldConfig(MODULUS, FL)
    with pytest.raises(pcs.InvalidPcsParam):
        pcs.MultilinearZip.open(pp, np.arange(32, dtype=np.int64), data, field.map_to_field(np.arange(5, dtype=np.int64)),
                                field, pcs.PcsTranscript(), num_vars=5)


def test_failed_testing_phase_with_inconsistent_codeword(pcs):  # open_z.rs:397
    pp, poly1 = setup_test_params(pcs, 4)
    _, roots = pcs.MultilinearZip.commit(pp, poly1)
    inconsistent, _ = pcs.MultilinearZip.commit(pp, np.arange(20, 36, dtype=np.int64))
    proof = prove(pcs, pp, poly1, inconsistent, POINT4)
    with pytest.raises(pcs.InvalidPcsOpen):
        verify(pcs, pp, roots, POINT4, eval_in_field(pcs, poly1, POINT4), proof)


def test_failed_evaluation_with_incorrect_evaluation(pcs):  # open_z.rs:456, verify_z.rs:284
    pp, poly = setup_test_params(pcs, 4)
    wrong = eval_in_field(pcs, poly, POINT4)
    wrong = np.array(orc.int_to_limbs((orc.limbs_to_int(wrong) + (1 << 256)) % MODULUS, FL), dtype=np.uint64)  # + one
    _roundtrip(pcs, 4, poly, POINT4, eval_mont=wrong, expect_ok=False)


def test_opening_and_evaluation_of_the_zero_polynomial(pcs):  # open_z.rs:494, verify_z.rs:451
    _roundtrip(pcs, 4, np.zeros(16, np.int64), POINT4, eval_mont=np.zeros(FL, np.uint64))


def test_evaluation_at_the_zero_point(pcs):  # open_z.rs:529, verify_z.rs:482
    _roundtrip(pcs, 4, np.arange(1, 17, dtype=np.int64), [0, 0, 0, 0])


def test_polynomial_coefficients_at_maximum_bit_size_boundary(pcs):  # open_z.rs:559
    evals = np.arange(16, dtype=np.int64)
    evals[1] = 2**63 - 1
    _roundtrip(pcs, 4, evals, [1, 0, 0, 0])  # evaluates to evals[1]


def test_evaluation_succeeds_with_minimal_polynomial_size_mu_is_2(pcs):  # open_z.rs:596
    _roundtrip(pcs, 2, np.arange(1, 5, dtype=np.int64), [1, 2])


# ------------------------------------------------------------------------------ pcs/verify_z.rs
def _full_protocol(pcs, num_vars):
    """verify_z.rs:226-270 setup_full_protocol: evaluations 0..2^n, point i + 2"""
    poly = np.arange(1 << num_vars, dtype=np.int64)
    pp = pcs.MultilinearZip.setup(1 << num_vars, pcs.RaaCode(1 << num_vars))
    data, roots = pcs.MultilinearZip.commit(pp, poly)
    point = [i + 2 for i in range(num_vars)]
    return pp, roots, point, eval_in_field(pcs, poly, point), prove(pcs, pp, poly, data, point), poly


def test_verification_fails_with_tampered_proof(pcs):  # verify_z.rs:305
    pp, roots, point, ev, proof, _ = _full_protocol(pcs, 4)
    verify(pcs, pp, roots, point, ev, proof)
    bad = proof.copy()
    bad[bad.size // 2] ^= 0x01
    with pytest.raises(pcs.InvalidPcsOpen):
        verify(pcs, pp, roots, point, ev, bad)


def test_verification_fails_with_wrong_commitment(pcs):  # verify_z.rs:319
    pp, roots, point, ev, proof, _ = _full_protocol(pcs, 4)
    _, other = pcs.MultilinearZip.commit(pp, np.arange(100, 116, dtype=np.int64))
    with pytest.raises(pcs.InvalidPcsOpen):
        verify(pcs, pp, other, point, ev, proof)


def test_verification_fails_with_invalid_point_size(pcs):  # verify_z.rs:334
    pp, roots, point, ev, proof, _ = _full_protocol(pcs, 4)
    with pytest.raises(pcs.InvalidPcsParam):
        verify(pcs, pp, roots, [100 + i for i in range(5)], ev, proof)


def _tamper_u(pcs, flip):
    """verify_z.rs:349-398, 512-558: poly_size 8, point 0, corrupt the combined row u'"""
    poly = np.arange(8, dtype=np.int64)
    pp = pcs.MultilinearZip.setup(8, pcs.RaaCode(8))
    data, roots = pcs.MultilinearZip.commit(pp, poly)
    point = [0, 0, 0]
    proof = prove(pcs, pp, poly, data, point)
    flip(proof, pp)
    return pp, roots, point, eval_in_field(pcs, poly, point), proof


def test_verification_fails_if_proximity_check_is_invalid(pcs):  # verify_z.rs:349
    def flip(proof, pp):
        proof[64 * (pp.row_len // 2)] ^= 0x01

    pp, roots, point, ev, proof = _tamper_u(pcs, flip)
    with pytest.raises(pcs.InvalidPcsOpen, match="Proximity failure"):
        verify(pcs, pp, roots, point, ev, proof)


def test_verification_fails_if_evaluation_consistency_check_is_invalid(pcs):  # verify_z.rs:400
    poly = np.arange(8, dtype=np.int64)
    pp = pcs.MultilinearZip.setup(8, pcs.RaaCode(8))
    data, roots = pcs.MultilinearZip.commit(pp, poly)
    point = [0, 0, 0]
    proof = prove(pcs, pp, poly, data, point)
    bytes_per_field = 8 * FL
    proof[proof.size - pp.row_len * bytes_per_field + bytes_per_field // 2] ^= 0x01  # inside the first element
    with pytest.raises(pcs.InvalidPcsOpen, match="Evaluation consistency failure"):
        verify(pcs, pp, roots, point, eval_in_field(pcs, poly, point), proof)


def test_verification_fails_if_proximity_values_are_too_large(pcs):  # verify_z.rs:512
    def flip(proof, pp):
        proof[0:64] = 0xFF

    pp, roots, point, ev, proof = _tamper_u(pcs, flip)
    with pytest.raises(pcs.InvalidPcsOpen):
        verify(pcs, pp, roots, point, ev, proof)


# ------------------------------------------------------------------------------ code_raa.rs, pcs/utils.rs, tests.rs
def test_accumulate_and_repeat_through_the_encoder(pcs):  # code_raa.rs:199, 224 (via the identity permutations)
    """With both permutations the identity, encode = accumulate(accumulate(repeat(row)))."""
    from zinc_amd import cabi

    cw = 8
    ident = np.arange(cw, dtype=np.uint32)
    ctx = cabi.ZipContext(4, ident, ident, geometry_override=(4, 4, cw))
    row = np.array([-1, 5, -10, 2], dtype=np.int64)
    com, _ = ctx.commit(np.concatenate([row, np.zeros(12, np.int64)]), with_merkle=False)
    got = [to_int(x) for x in com.download()[0].reshape(-1, 4)[:cw]]
    rep = list(row) * 2                                   # repeat: [a, b, c, d, a, b, c, d]
    acc1 = list(np.cumsum(rep))
    assert acc1[:4] == [-1, 4, -6, -4]                    # the reference's expected accumulate
    assert got == [int(x) for x in np.cumsum(acc1)]


def test_shuffle_is_deterministic_for_a_given_seed(pcs):  # code_raa.rs:247
    p1, p2, p3 = (pcs.shuffle_seeded_perm(s, 10) for s in (12345, 12345, 54321))
    assert np.array_equal(p1, p2) and not np.array_equal(p1, p3)
    assert not np.array_equal(p1, np.arange(10)) and not np.array_equal(p3, np.arange(10))
    assert sorted(p1) == list(range(10))


def test_merkle_proof_of_every_leaf(pcs):  # pcs/utils.rs:340 (Int<3> leaves)
    leaves = np.random.default_rng(4).integers(0, 2**63, size=(8, 3), dtype=np.uint64)
    tree = pcs.MerkleTree.new(3, leaves)
    full = np.concatenate([tree.layers, tree.root[None]])
    for leaf in range(8):
        path = orc.merkle_path(3, full, leaf)
        assert orc.merkle_verify(3, path, tree.root, leaves[leaf], leaf) == 0


def test_zip_batch_evaluation(pcs):  # tests.rs:148
    n, m = 8, 10
    t = pcs.KeccakTranscript()
    pp = pcs.MultilinearZip.setup(1 << n, pcs.RaaCode(1 << n, t))
    rng = np.random.default_rng(5)
    mles = [rng.integers(-128, 128, size=1 << n, dtype=np.int64) for _ in range(m)]
    outs = pcs.MultilinearZip.batch_commit(pp, mles)
    datas, comms = [o[0] for o in outs], [o[1] for o in outs]
    point_int = rng.integers(-128, 128, size=n, dtype=np.int64)
    field = pcs.FieldConfig(MODULUS, FL)
    point = field.map_to_field(point_int)
    evals = [eval_in_field(pcs, mle, point_int) for mle in mles]
    transcript = pcs.PcsTranscript()
    pcs.MultilinearZip.batch_open(pp, mles, datas, [point] * m, field, transcript)
    proof = transcript.into_proof()
    vt = pcs.PcsTranscript.from_proof(proof)
    pcs.MultilinearZip.batch_verify_z(pp, comms, [point] * m, evals, vt, field)
    assert vt.position() == proof.size
