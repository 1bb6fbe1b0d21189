"""The reference's own sumcheck tests (src/sumcheck/tests.rs), one by one, with the device prover
(zip_sumcheck_* through sumcheck::prove_as_subprotocol_products / _product) and the host mirror's verifier
(sumcheck::verify_as_subprotocol).  Same field (the 2-limb prime of tests.rs:23), same shapes
(rand_poly(num_vars, (2, 5), 7)), our own PRNG."""
import numpy as np
import pytest

import _oracle as orc

pytestmark = pytest.mark.gpu

Q, FL = 57316695564490278656402085503, 2  # tests.rs:22-25
R = 1 << (64 * FL)


@pytest.fixture(scope="module")
def pcs():
    from zinc_amd import cabi, pcs as m

    if cabi.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu tests must run on the MI355X box")
    return m


def mont(v):
    return orc.int_to_limbs(v % Q * R % Q, FL)


def from_mont(limbs):
    return orc.limbs_to_int(limbs) * pow(R, -1, Q) % Q


def rand_poly(nv, rng, num_products=7, lo=2, hi=5):
    """rand_poly (sumcheck/utils.rs:27-65): products of lo..hi-1 fresh random MLEs with a random coefficient each.
    Returns (tables [K, 2^nv, FL], degree, masks, coeffs [P, FL], values as Python ints, claimed sum as limbs)."""
    vals, masks, coeffs, k, degree, total = [], [], [], 0, 0, 0
    for _ in range(num_products):
        m = int(rng.integers(lo, hi))
        degree = max(degree, m)
        mles = [[int(rng.integers(0, 2**62)) * 0x9E3779B97F4A7C15 % Q for _ in range(1 << nv)] for _ in range(m)]
        c = int(rng.integers(1, 2**62)) * 0xD1B54A32D192ED03 % Q
        ps = 0
        for b in range(1 << nv):
            t = 1
            for v in mles:
                t = t * v[b] % Q
            ps += t
        total = (total + c * ps) % Q
        vals += mles
        masks.append(sum(1 << (k + i) for i in range(m)))
        coeffs.append(c)
        k += m
    tables = np.array([[mont(x) for x in v] for v in vals], dtype=np.uint64)
    return (tables, degree, np.array(masks, dtype=np.uint32), np.array([mont(c) for c in coeffs], dtype=np.uint64), vals,
            list(zip(coeffs, masks)), np.array(mont(total), dtype=np.uint64))


def prove(pcs, poly, transcript=None):
    tables, degree, masks, coeffs = poly[:4]
    return pcs.sumcheck_prove_products(transcript or pcs.KeccakTranscript(), tables, degree, masks, coeffs, pcs.FieldConfig(Q, FL))


def verify(pcs, nv, degree, claimed, msgs, transcript=None):
    return pcs.sumcheck_verify(transcript or pcs.KeccakTranscript(), nv, degree, claimed, msgs, pcs.FieldConfig(Q, FL))


def test_full_sumcheck_protocol_works_correctly(pcs):
    """tests.rs:49-71: twenty random polynomials over three variables, prove then verify."""
    rng = np.random.default_rng(1)
    f = orc.make_field(Q, FL)
    for _ in range(20):
        poly = rand_poly(3, rng)
        msgs, rand = prove(pcs, poly)
        point, _ = verify(pcs, 3, poly[1], poly[6], msgs)
        assert np.array_equal(point, rand)
        want, _ = orc.sumcheck_prove_products(f, poly[0], poly[1], poly[2], [orc.limbs_to_int(c) for c in poly[3]], orc.new_transcript())
        assert np.array_equal(msgs, want)


def test_verifier_rejects_proof_with_incorrect_claimed_sum(pcs):
    """tests.rs:73-112"""
    poly = rand_poly(3, np.random.default_rng(2))
    msgs, _ = prove(pcs, poly)
    wrong = np.array(mont(from_mont(poly[6]) + 1), dtype=np.uint64)
    with pytest.raises(pcs.SpartanError, match="p\\(0\\) \\+ p\\(1\\)"):
        verify(pcs, 3, poly[1], wrong, msgs)


def test_verifier_rejects_proof_with_tampered_prover_message(pcs):
    """tests.rs:115-155: evaluations[0] of the first round plus one"""
    poly = rand_poly(3, np.random.default_rng(3))
    msgs, _ = prove(pcs, poly)
    msgs[0, 0] = mont(from_mont(msgs[0, 0]) + 1)
    with pytest.raises(pcs.SpartanError):
        verify(pcs, 3, poly[1], poly[6], msgs)


def test_verifier_rejects_proof_with_wrong_degree(pcs):
    """tests.rs:158-193"""
    poly = rand_poly(3, np.random.default_rng(4))
    msgs, _ = prove(pcs, poly)
    with pytest.raises(pcs.SpartanError):
        verify(pcs, 3, poly[1] - 1, poly[6], msgs)


def test_protocol_is_deterministic_with_same_transcript(pcs):
    """tests.rs:196-229"""
    poly = rand_poly(3, np.random.default_rng(5))
    a, ra = prove(pcs, poly)
    b, rb = prove(pcs, poly)
    assert np.array_equal(a, b) and np.array_equal(ra, rb)


def test_different_polynomials_produce_different_proofs(pcs):
    """tests.rs:232-275"""
    rng = np.random.default_rng(6)
    a, _ = prove(pcs, rand_poly(3, rng))
    b, _ = prove(pcs, rand_poly(3, rng))
    assert a.shape != b.shape or not np.array_equal(a, b)


def _product_of(pcs, value, nv=3, num_mles=2, degree=2):
    tables = np.array([[mont(value)] * (1 << nv)] * num_mles, dtype=np.uint64)
    return pcs.sumcheck_prove_product(pcs.KeccakTranscript(), tables, degree, pcs.FieldConfig(Q, FL))


def test_sumcheck_with_zero_polynomial(pcs):
    """tests.rs:278-323: two zero MLEs, comb = product; extract_sum(proof) is zero and the verifier accepts"""
    msgs, _ = _product_of(pcs, 0)
    assert (from_mont(msgs[0, 0]) + from_mont(msgs[0, 1])) % Q == 0  # MLSumcheck::extract_sum
    verify(pcs, 3, 2, np.array(mont(0), dtype=np.uint64), msgs)


def test_sumcheck_with_constant_polynomial(pcs):
    """tests.rs:326-371: two all-ones MLEs; the sum is the number of hypercube points"""
    msgs, _ = _product_of(pcs, 1)
    verify(pcs, 3, 2, np.array(mont(8), dtype=np.uint64), msgs)


def test_sumcheck_with_single_variable(pcs):
    """tests.rs:374-407"""
    poly = rand_poly(1, np.random.default_rng(7))
    msgs, _ = prove(pcs, poly)
    verify(pcs, 1, poly[1], poly[6], msgs)


def test_verifier_rejects_proof_if_transcript_is_tampered(pcs):
    """tests.rs:410-455"""
    poly = rand_poly(3, np.random.default_rng(8))
    msgs, _ = prove(pcs, poly)
    verify(pcs, 3, poly[1], poly[6], msgs)
    t = pcs.KeccakTranscript()
    t.absorb(b"tampering the transcript")
    with pytest.raises(pcs.SpartanError):
        verify(pcs, 3, poly[1], poly[6], msgs, transcript=t)


def test_prover_panics_if_round_exceeds_num_vars(pcs):
    """tests.rs:458-478: "Prover is not active" """
    from zinc_amd import cabi

    tables = np.array([[mont(3)] * 8] * 2, dtype=np.uint64)
    s = cabi.Sumcheck(tables, 3, 2, cabi.make_field(Q, FL))
    r = np.array(mont(5), dtype=np.uint64)
    s.round()
    s.round(r)
    s.round(r)
    with pytest.raises(cabi.ZipError, match="Prover is not active"):
        s.round(r)
    s.free()


def test_verifier_errors_on_incomplete_proof(pcs):
    """tests.rs:481-521: InvalidProofLength { expected: 3, got: 2 }"""
    poly = rand_poly(3, np.random.default_rng(9))
    msgs, _ = prove(pcs, poly)
    with pytest.raises(pcs.SpartanError, match="2 rounds, expected 3"):
        verify(pcs, 3, poly[1], poly[6], msgs[:-1])


def test_prover_handles_empty_mle_list(pcs):
    """tests.rs:524-557: no MLEs, degree 0, comb_fn == 0"""
    field = pcs.FieldConfig(Q, FL)
    msgs, _ = pcs.sumcheck_prove_products(pcs.KeccakTranscript(), np.zeros((0, 8, FL), np.uint64), 0,
                                          np.zeros(0, np.uint32), np.zeros((0, FL), np.uint64), field, nvars=3)
    assert msgs.shape == (3, 1, FL) and not msgs.any()
    verify(pcs, 3, 0, np.zeros(FL, np.uint64), msgs)


def test_prover_panics_with_zero_variables(pcs):
    """tests.rs:559-565: "Attempt to prove a constant." """
    from zinc_amd import cabi

    with pytest.raises(cabi.ZipError):
        cabi.Sumcheck(np.zeros((1, 1, FL), np.uint64), 0, 2, cabi.make_field(Q, FL))


def test_verifier_errors_on_mismatched_nvars(pcs):
    """tests.rs:567-592: InvalidProofLength { expected: 4, got: 3 }"""
    poly = rand_poly(3, np.random.default_rng(10))
    msgs, _ = prove(pcs, poly)
    with pytest.raises(pcs.SpartanError, match="3 rounds, expected 4"):
        verify(pcs, 4, poly[1], poly[6], msgs)


def test_verifier_produces_correct_subclaim(pcs):
    """tests.rs:594-639: the subclaim equals rand_poly_comb_fn over the MLEs evaluated at the point"""
    poly = rand_poly(3, np.random.default_rng(11))
    msgs, _ = prove(pcs, poly)
    point, expected = verify(pcs, 3, poly[1], poly[6], msgs)
    pt = [from_mont(p) for p in point]
    at = []
    for v in poly[4]:  # mle.evaluate(point): variable 0 = least significant bit of the index
        acc = 0
        for i, x in enumerate(v):
            w = x
            for j, r in enumerate(pt):
                w = w * (r if (i >> j) & 1 else 1 - r) % Q
            acc += w
        at.append(acc % Q)
    manual = 0
    for c, mask in poly[5]:
        t = c
        for k, a in enumerate(at):
            if (int(mask) >> k) & 1:
                t = t * a % Q
        manual += t
    assert from_mont(expected) == manual % Q


def test_zero_variable_case_returns_correct_subclaim(pcs):
    """tests.rs:641-674"""
    claimed = np.array(mont(42), dtype=np.uint64)
    point, expected = verify(pcs, 0, 2, claimed, np.zeros((0, 3, FL), np.uint64))
    assert point.shape[0] == 0 and np.array_equal(expected, claimed)
