"""CPU tests of the oracle's Spartan prover / verifier (oracle/zip_oracle.c), the restatement of
src/zinc/prover.rs + src/zinc/verifier.rs + src/sumcheck/verifier.rs.

The reference's own tests for this path are property tests (src/zinc/tests.rs: prove succeeds, the verifier accepts,
a broken witness is rejected); they are restated here, plus identities in Python big integers that pin the pieces
(interpolation, inverse, M z, the second sumcheck's table, the final equation)."""
import numpy as np
import pytest

import _ccs
import _oracle as orc

Q192 = 312829638388039969874974628075306023441          # zinc/tests.rs:28 (N = 3)
Q256 = 115792089237316195423570985008687907853269984665640564039457584007913129639747  # spartan_benches.rs:152
QSTARK = 3618502788666131213697322783095070105623107215331596699973092056135872020481  # spartan_benches.rs:161
FIELDS = [(Q192, 3), (Q256, 4), (QSTARK, 4)]


def to_int(f, q, limbs):
    """Montgomery limbs -> canonical integer."""
    R = 1 << (64 * f.fl)
    return orc.limbs_to_int(limbs) * pow(R, -1, q) % q


def mont(f, q, v):
    return v % q * (1 << (64 * f.fl)) % q


@pytest.mark.parametrize("q,fl", FIELDS)
def test_field_inv_and_interpolation(q, fl):
    f = orc.make_field(q, fl)
    rng = np.random.default_rng(5)
    for _ in range(8):
        a = int(rng.integers(1, 2**62)) * 0x9E3779B97F4A7C15 % q
        assert to_int(f, q, orc.int_to_limbs(orc.field_inv(f, mont(f, q, a)), fl)) == pow(a, -1, q)
    for deg in (1, 2, 3, 4):
        coeffs = [int(rng.integers(0, 2**62)) for _ in range(deg + 1)]
        poly = lambda x: sum(c * pow(x, i, q) for i, c in enumerate(coeffs)) % q
        p = [mont(f, q, poly(i)) for i in range(deg + 1)]
        for x in [0, 1, deg, deg + 1, 12345678901234567890123 % q, q - 1]:
            got = orc.interpolate_uni_poly(f, p, mont(f, q, x))
            assert to_int(f, q, orc.int_to_limbs(got, fl)) == poly(x), (deg, x)


def eq_table(q, r):
    """eq(x, r) over the hypercube, variable 0 = least significant bit of the index."""
    out = []
    for i in range(1 << len(r)):
        v = 1
        for j, rj in enumerate(r):
            v = v * (rj if (i >> j) & 1 else 1 - rj) % q
        out.append(v)
    return out


@pytest.mark.parametrize("q,fl", FIELDS[:2])
def test_mz_and_second_table_against_python(q, fl):
    f = orc.make_field(q, fl)
    inst = _ccs.vitalik_ccs(3)
    ccs = orc.Ccs(inst)
    z = [int(v) for v in inst.z] + [0] * (inst.m - len(inst.z))
    mz = ccs.mz(f)
    for k, M in enumerate(inst.matrices):
        for row in range(inst.m):
            want = sum(int(M.values[e]) * z[int(M.col_idx[e])] for e in range(M.row_ptr[row], M.row_ptr[row + 1])) % q
            assert to_int(f, q, mz[k, row]) == want
    rx = [7, 11, 13]
    gamma = 987654321
    eq = eq_table(q, rx)
    eq_m = orc.field_elems([mont(f, q, v) for v in eq], fl)
    assert (orc.build_eq_x_r(f, orc.field_elems([mont(f, q, v) for v in rx], fl)) == eq_m).all()
    tab = ccs.second_table(f, eq_m, orc.field_elems([mont(f, q, gamma)], fl)[0])
    for col in range(inst.m):
        want = 0
        for k, M in enumerate(inst.matrices):
            tk = sum(int(M.values[e]) * eq[row] for row in range(M.n_rows)
                     for e in range(M.row_ptr[row], M.row_ptr[row + 1]) if int(M.col_idx[e]) == col)
            want += pow(gamma, k, q) * tk
        assert to_int(f, q, tab[col]) == want % q


def _prove_verify(inst, q, fl):
    f = orc.make_field(q, fl)
    ccs = orc.Ccs(inst)
    proof = ccs.spartan_prove(f, orc.new_transcript())
    rc, pts = ccs.spartan_verify(f, proof, orc.new_transcript())
    return f, ccs, proof, rc, pts


@pytest.mark.parametrize("q,fl", FIELDS)
@pytest.mark.parametrize("log_n", [1, 2, 5, 8])
def test_dummy_spartan_prover_and_verifier(q, fl, log_n):
    """zinc/tests.rs:22-57 and :110-157 (there at n = 2^13)."""
    inst = _ccs.dummy_ccs_from_len(1 << log_n)
    f, ccs, proof, rc, pts = _prove_verify(inst, q, fl)
    assert rc == 0
    assert (pts["r_x"] == proof["r_x"]).all() and (pts["r_y"] == proof["r_y"]).all()
    # the statement the PCS then proves: e_y == lin_comb(gamma, M_k(r_x, r_y)) * z_mle(r_y)  (verifier.rs:248-269)
    # z through the reference's FieldMap (for a modulus with the top bit set that is NOT z mod q: see
    # field_from_signed_words in the oracle), the rest in Python
    z = [to_int(f, q, orc.int_to_limbs(orc.field_from_i64(f, int(v)), fl)) for v in inst.z]
    ry = [to_int(f, q, r) for r in pts["r_y"]]
    eqy = eq_table(q, ry)
    v = sum(a * b for a, b in zip(z, eqy)) % q
    assert ccs.final_check(f, pts, orc.field_elems([mont(f, q, v)], fl)[0]) == 0
    assert ccs.final_check(f, pts, orc.field_elems([mont(f, q, v + 1)], fl)[0]) == orc.ORC_ERR_PROOF


@pytest.mark.parametrize("q,fl", FIELDS)
def test_spartan_verifier_on_the_test_ccs(q, fl):
    """zinc/tests.rs:59-108: x^3 + x + 5 = y with x = 3, padded to 8 x 8."""
    f, ccs, proof, rc, pts = _prove_verify(_ccs.vitalik_ccs(3), q, fl)
    assert rc == 0
    # first claimed sum is zero (the relation holds): p(0) + p(1) == 0 in round 1
    p0, p1 = to_int(f, q, proof["msgs1"][0, 0]), to_int(f, q, proof["msgs1"][0, 1])
    assert (p0 + p1) % q == 0


@pytest.mark.parametrize("q,fl", FIELDS)
def test_failing_spartan_verifier(q, fl):
    """zinc/tests.rs:159-209: a witness that breaks the relation is rejected."""
    _, _, _, rc, _ = _prove_verify(_ccs.vitalik_ccs(3, break_witness=True), q, fl)
    assert rc == orc.ORC_ERR_PROOF


def test_tampered_proof_is_rejected():
    q, fl = FIELDS[0]
    inst = _ccs.dummy_ccs_from_len(16)
    f, ccs, proof, rc, _ = _prove_verify(inst, q, fl)
    assert rc == 0
    for key, idx in (("msgs1", (1, 2, 0)), ("msgs2", (0, 1, 0)), ("V_s", (2, 0))):
        bad = {k: v.copy() for k, v in proof.items()}
        bad[key][idx] ^= np.uint64(1)
        rc, _ = ccs.spartan_verify(f, bad, orc.new_transcript())
        assert rc == orc.ORC_ERR_PROOF, key


def test_sumcheck_verify_agrees_with_the_prover_transcript():
    """prover and verifier leave the transcript in the same state and derive the same point."""
    q, fl = FIELDS[1]
    f = orc.make_field(q, fl)
    rng = np.random.default_rng(3)
    nv = 4
    tables = np.stack([orc.field_elems([mont(f, q, int(rng.integers(0, 2**62))) for _ in range(1 << nv)], fl)
                       for _ in range(2)])
    tp, tv = orc.new_transcript(), orc.new_transcript()
    msgs, rand = orc.sumcheck_prove_product(f, tables, 2, tp)
    R = 1 << (64 * fl)
    claimed = sum(to_int(f, q, tables[0, i]) * to_int(f, q, tables[1, i]) for i in range(1 << nv)) % q
    rc, point, expected = orc.sumcheck_verify(f, nv, 2, mont(f, q, claimed), msgs, tv)
    assert rc == 0 and (point == rand).all()
    assert bytes(tp.st) == bytes(tv.st) and bytes(tp.buf)[: tp.buflen] == bytes(tv.buf)[: tv.buflen]
    # the subclaim is the product of the two MLEs at the point
    pt = [to_int(f, q, r) for r in point]
    eq = eq_table(q, pt)
    a = sum(to_int(f, q, tables[0, i]) * eq[i] for i in range(1 << nv)) % q
    b = sum(to_int(f, q, tables[1, i]) * eq[i] for i in range(1 << nv)) % q
    assert expected * pow(R, -1, q) % q == a * b % q
    rc, _, _ = orc.sumcheck_verify(f, nv, 2, mont(f, q, claimed + 1), msgs, orc.new_transcript())
    assert rc == orc.ORC_ERR_PROOF


def rand_poly(f, q, fl, nv, products, seed):
    """rand_poly (sumcheck/utils.rs:27-65) with our own PRNG: `products` = multiplicands per product, fresh MLEs each.
    Returns tables [K, 2^nv, fl] (Montgomery), masks, coefficients (Montgomery ints), the claimed sum (canonical int)."""
    rng = np.random.default_rng(seed)
    tables, masks, coeffs, total, k = [], [], [], 0, 0
    for m in products:
        vals = [[int(rng.integers(0, 2**62)) * 0x9E3779B97F4A7C15 % q for _ in range(1 << nv)] for _ in range(m)]
        c = int(rng.integers(1, 2**62)) * 0xD1B54A32D192ED03 % q
        prod_sum = 0
        for b in range(1 << nv):
            t = 1
            for v in vals:
                t = t * v[b] % q
            prod_sum += t
        total = (total + c * prod_sum) % q
        tables += [orc.field_elems([mont(f, q, x) for x in v], fl) for v in vals]
        masks.append(sum(1 << (k + i) for i in range(m)))
        coeffs.append(mont(f, q, c))
        k += m
    return np.stack(tables), np.array(masks, dtype=np.uint32), coeffs, total


@pytest.mark.parametrize("q,fl", FIELDS[:2])
@pytest.mark.parametrize("products", [(2,), (2, 3), (4, 2, 3), (2, 3, 4, 2, 3, 4, 2)])
def test_sum_of_products_sumcheck(q, fl, products):
    """The combination function of sumcheck_benches.rs / sumcheck/tests.rs (rand_poly_comb_fn): the verifier accepts
    the claimed sum computed in Python integers, and its subclaim is sum_p c_p prod_j MLE_j(point)."""
    f = orc.make_field(q, fl)
    nv = 4
    tables, masks, coeffs, total = rand_poly(f, q, fl, nv, products, seed=len(products))
    degree = max(products)
    tp, tv = orc.new_transcript(), orc.new_transcript()
    msgs, rand = orc.sumcheck_prove_products(f, tables, degree, masks, coeffs, tp)
    rc, point, expected = orc.sumcheck_verify(f, nv, degree, mont(f, q, total), msgs, tv)
    assert rc == 0 and (point == rand).all()
    eq = eq_table(q, [to_int(f, q, r) for r in point])
    at = [sum(to_int(f, q, tables[k, i]) * eq[i] for i in range(1 << nv)) % q for k in range(tables.shape[0])]
    want = 0
    for mask, c in zip(masks, coeffs):
        t = to_int(f, q, orc.int_to_limbs(c, fl))
        for k in range(tables.shape[0]):
            if (int(mask) >> k) & 1:
                t = t * at[k] % q
        want += t
    assert to_int(f, q, orc.int_to_limbs(expected, fl)) == want % q
    rc, _, _ = orc.sumcheck_verify(f, nv, degree, mont(f, q, total + 1), msgs, orc.new_transcript())
    assert rc == orc.ORC_ERR_PROOF
