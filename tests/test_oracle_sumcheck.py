"""CPU: the oracle's sumcheck prover (product combination) pinned by the protocol's own identities,
checked with independent Python big-integer arithmetic (src/sumcheck/tests.rs runs prover against
verifier; the verifier's round checks are restated here: verifier.rs:60-148)."""
import numpy as np
import pytest

import _oracle as orc

BENCH_MODULUS = 106319353542452952636349991594949358997917625194731877894581586278529202198383
TEST_MODULUS_2 = 57316695564490278656402085503


def _std(x_mont, q, fl):
    return x_mont * pow(1 << (64 * fl), -1, q) % q


def _interp(ys, x, q):
    """value at x of the polynomial through (0, ys[0]), (1, ys[1]), ... (interpolate_uni_poly, verifier.rs:161-)"""
    acc = 0
    for i, yi in enumerate(ys):
        num, den = 1, 1
        for j in range(len(ys)):
            if j != i:
                num = num * (x - j) % q
                den = den * (i - j) % q
        acc = (acc + yi * num * pow(den, -1, q)) % q
    return acc


@pytest.mark.parametrize("modulus,fl", [(BENCH_MODULUS, 4), (TEST_MODULUS_2, 2)])
@pytest.mark.parametrize("K,degree,nv", [(2, 2, 5), (3, 3, 4), (1, 1, 6), (2, 3, 3), (2, 2, 1)])
def test_sumcheck_prover_identities(modulus, fl, K, degree, nv):
    q = modulus
    f = orc.make_field(q, fl)
    rng = np.random.default_rng(nv * 10 + K)
    n = 1 << nv
    std = [[int(rng.integers(0, 2**62)) * int(rng.integers(1, 2**62)) % q for _ in range(n)] for _ in range(K)]
    R = 1 << (64 * fl)
    mles = np.stack([orc.field_elems([v * R % q for v in t], fl) for t in std])
    tr = orc.new_transcript()
    orc.absorb(tr, b"sumcheck")
    msgs, rand = orc.sumcheck_prove_product(f, mles, degree, tr)
    g = [[_std(orc.limbs_to_int(msgs[i, e]), q, fl) for e in range(degree + 1)] for i in range(nv)]
    r = [_std(orc.limbs_to_int(rand[i]), q, fl) for i in range(nv)]
    # round 1 opens the claimed sum (verifier.rs:99-107)
    claimed = 0
    for b in range(n):
        p = 1
        for t in std:
            p = p * t[b] % q
        claimed = (claimed + p) % q
    assert (g[0][0] + g[0][1]) % q == claimed
    # every later round continues the previous one at the verifier's challenge
    for i in range(1, nv):
        assert (g[i][0] + g[i][1]) % q == _interp(g[i - 1], r[i - 1], q)
    # the final claim is the product of the MLEs at the challenge point (variable 0 = LSB, dense.rs:155-164)
    final = 1
    for t in std:
        cur = t[:]
        for ri in r:
            cur = [(cur[2 * b] + ri * (cur[2 * b + 1] - cur[2 * b])) % q for b in range(len(cur) // 2)]
        final = final * cur[0] % q
    assert final == _interp(g[-1], r[-1], q)
    # the challenges are the transcript's: absorb nvars, degree, then per round the evaluations, squeeze, absorb
    replay = orc.new_transcript()
    orc.absorb(replay, b"sumcheck")
    orc.absorb_field(replay, f, orc.field_from_u128(f, nv))
    orc.absorb_field(replay, f, orc.field_from_u128(f, degree))
    for i in range(nv):
        for e in range(degree + 1):
            orc.absorb_field(replay, f, orc.limbs_to_int(msgs[i, e]))
        ri = orc.get_challenge(replay, f)
        assert ri == orc.limbs_to_int(rand[i])
        orc.absorb_field(replay, f, ri)


@pytest.mark.parametrize("modulus,fl", [(BENCH_MODULUS, 4), (TEST_MODULUS_2, 2)])
@pytest.mark.parametrize("nv", [1, 4, 6])
def test_sumcheck_prover_ccs_combination(modulus, fl, nv):
    """comb = (c0 * M0 * M1 + c1 * M2) * eq  -- sumcheck_polynomial_comb_fn_1 for an R1CS-shaped CCS
    (c = [1, -1], S = [[0, 1], [2]], zinc/utils.rs:49-94), degree d + 1 = 3."""
    q = modulus
    f = orc.make_field(q, fl)
    rng = np.random.default_rng(nv)
    n, K, degree = 1 << nv, 4, 3
    std = [[int(rng.integers(0, 2**62)) * int(rng.integers(1, 2**62)) % q for _ in range(n)] for _ in range(K)]
    R = 1 << (64 * fl)
    mles = np.stack([orc.field_elems([v * R % q for v in t], fl) for t in std])
    c = [1, q - 1]
    masks = [0b011, 0b100]
    tr = orc.new_transcript()
    msgs, rand = orc.sumcheck_prove(f, mles, degree, masks, [x * R % q for x in c], tr)
    g = [[_std(orc.limbs_to_int(msgs[i, e]), q, fl) for e in range(degree + 1)] for i in range(nv)]
    r = [_std(orc.limbs_to_int(rand[i]), q, fl) for i in range(nv)]
    comb = lambda v: (c[0] * v[0] * v[1] + c[1] * v[2]) * v[3] % q
    claimed = sum(comb([t[b] for t in std]) for b in range(n)) % q
    assert (g[0][0] + g[0][1]) % q == claimed
    for i in range(1, nv):
        assert (g[i][0] + g[i][1]) % q == _interp(g[i - 1], r[i - 1], q)
    finals = []
    for t in std:
        cur = t[:]
        for ri in r:
            cur = [(cur[2 * b] + ri * (cur[2 * b + 1] - cur[2 * b])) % q for b in range(len(cur) // 2)]
        finals.append(cur[0])
    assert comb(finals) == _interp(g[-1], r[-1], q)
