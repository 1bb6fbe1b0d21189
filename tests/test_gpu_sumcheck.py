"""GPU: the product sumcheck prover (zip_sumcheck_*) against the oracle's
MLSumcheck::prove_as_subprotocol, round messages and challenges bit for bit."""
import numpy as np
import pytest

import _oracle as orc

pytestmark = pytest.mark.gpu

BENCH_MODULUS = 106319353542452952636349991594949358997917625194731877894581586278529202198383
TEST_MODULUS_2 = 57316695564490278656402085503
MOD_NO_SPARE = (1 << 256) - 189
MOD_3LIMB = (1 << 190) - 11 * (1 << 64) - 59


@pytest.fixture(scope="module")
def mods():
    from zinc_amd import cabi, pcs

    if cabi.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu tests must run on the MI355X box")
    return cabi, pcs


def _tables(f, fl, modulus, K, nv, seed):
    """K tables of 2^nv canonical field elements (Montgomery limbs): the witness-like one from i64, the rest random"""
    rng = np.random.default_rng(seed)
    n = 1 << nv
    out = np.zeros((K, n, fl), dtype=np.uint64)
    for k in range(K):
        if k == K - 1:  # z_mle: map_to_field of integers (zinc/prover.rs:299)
            w = orc.splitmix64(seed + k, n)
            for i in range(n):
                out[k, i] = orc.int_to_limbs(orc.field_from_i64(f, int(w[i])), fl)
        else:
            vals = [int.from_bytes(rng.bytes(40), "little") % modulus for _ in range(n)]
            out[k] = orc.field_elems(vals, fl)
    return out


@pytest.mark.parametrize("modulus,fl", [(BENCH_MODULUS, 4), (TEST_MODULUS_2, 2), (MOD_NO_SPARE, 4), (MOD_3LIMB, 3)])
@pytest.mark.parametrize("K,degree,nv", [(2, 2, 10), (1, 1, 3), (3, 3, 7), (2, 3, 1), (4, 4, 5), (2, 2, 2)])
def test_sumcheck_rounds_equal_the_oracle(mods, modulus, fl, K, degree, nv):
    cabi, pcs = mods
    f = orc.make_field(modulus, fl)
    mles = _tables(f, fl, modulus, K, nv, seed=nv * 7 + K)
    to = orc.new_transcript()
    orc.absorb(to, b"sumcheck-2")
    msgs_o, rand_o = orc.sumcheck_prove_product(f, mles, degree, to)
    # through the host mirror (transcript on the host, rounds on the device)
    t = pcs.KeccakTranscript()
    t.absorb(b"sumcheck-2")
    msgs, rand = pcs.sumcheck_prove_product(t, mles, degree, pcs.FieldConfig(modulus, fl))
    assert np.array_equal(msgs, msgs_o)
    assert np.array_equal(rand, rand_o)
    assert t.get_u64() == orc.lib().orc_tr_get_u64(orc.C.byref(to))  # same Fiat-Shamir state afterwards


def test_sumcheck_device_resident_tables_2pow20(mods):
    """ZincProver's second sumcheck shape at 2^20 (CCS s = 20): two tables in HBM, read in place."""
    torch = pytest.importorskip("torch")
    cabi, pcs = mods
    nv, fl, K, degree = 20, 4, 2, 2
    f = orc.make_field(BENCH_MODULUS, fl)
    rng = np.random.default_rng(5)
    n = 1 << nv
    # canonical residues below 2^250 < q, as raw Montgomery limbs
    mles = rng.integers(0, 1 << 62, size=(K, n, fl), dtype=np.uint64)
    mles[..., fl - 1] >>= np.uint64(6)
    to = orc.new_transcript()
    msgs_o, rand_o = orc.sumcheck_prove_product(f, mles, degree, to)
    dev = [torch.from_numpy(mles[k].view(np.int64)).cuda() for k in range(K)]
    before = [d.clone() for d in dev]
    sc = cabi.Sumcheck(dev, nv, degree, cabi.make_field(BENCH_MODULUS, fl))
    r = None
    for i in range(nv):
        ev = sc.round(r)
        assert np.array_equal(ev, msgs_o[i]), i
        r = rand_o[i]
    with pytest.raises(cabi.ZipError):  # "Prover is not active" (prover.rs:91-93)
        sc.round(r)
    sc.free()
    assert all(torch.equal(a, b) for a, b in zip(dev, before))  # the caller's tables were only read


def test_sumcheck_usage_errors(mods):
    cabi, pcs = mods
    zf = cabi.make_field(BENCH_MODULUS, 4)
    m = np.zeros((2, 8, 4), dtype=np.uint64)
    with pytest.raises(cabi.ZipError):
        cabi.Sumcheck(m, 0, 2, zf)        # "Attempt to prove a constant." (prover.rs:47-49)
    with pytest.raises(cabi.ZipError):
        cabi.Sumcheck(m, 3, 5, zf)        # degree beyond the supported 1..4
    sc = cabi.Sumcheck(m, 3, 2, zf)
    with pytest.raises(cabi.ZipError):
        sc.round(np.ones(4, dtype=np.uint64))  # "first round should be prover first." (prover.rs:69-71)
    sc.round()
    with pytest.raises(cabi.ZipError):
        sc.round()                         # "verifier message is empty" (prover.rs:87-89)


@pytest.mark.parametrize("modulus,fl", [(BENCH_MODULUS, 4), (TEST_MODULUS_2, 2), (MOD_3LIMB, 3)])
@pytest.mark.parametrize("nv", [1, 6, 11])
def test_sumcheck_ccs_combination_equals_the_oracle(mods, modulus, fl, nv):
    """ZincProver's first sumcheck (zinc/prover.rs:241-259): comb = (M0 * M1 - M2) * eq for an R1CS-shaped CCS
    (c = [1, -1], S = [[0, 1], [2]]; sumcheck_polynomial_comb_fn_1, zinc/utils.rs:77-94), degree 3."""
    cabi, pcs = mods
    f = orc.make_field(modulus, fl)
    mles = _tables(f, fl, modulus, 4, nv, seed=nv + 40)
    R = 1 << (64 * fl)
    c = [1 * R % modulus, (modulus - 1) * R % modulus]
    S = [[0, 1], [2]]
    to = orc.new_transcript()
    orc.absorb(to, b"sumcheck-1")
    msgs_o, rand_o = orc.sumcheck_prove(f, mles, 3, [0b011, 0b100], c, to)
    t = pcs.KeccakTranscript()
    t.absorb(b"sumcheck-1")
    msgs, rand = pcs.sumcheck_prove_ccs(t, mles, 3, orc.field_elems(c, fl), S, pcs.FieldConfig(modulus, fl))
    assert np.array_equal(msgs, msgs_o) and np.array_equal(rand, rand_o)
    assert t.get_u64() == orc.lib().orc_tr_get_u64(orc.C.byref(to))


def test_sumcheck_ccs_general_terms_and_zero_coefficients(mods):
    """three terms over three MLEs + eq, one coefficient zero (skipped, zinc/utils.rs:80-82), degree 4"""
    cabi, pcs = mods
    modulus, fl, nv = BENCH_MODULUS, 4, 7
    f = orc.make_field(modulus, fl)
    mles = _tables(f, fl, modulus, 4, nv, seed=9)
    R = 1 << (64 * fl)
    c_std = [5, 0, modulus - 7]
    c = [x * R % modulus for x in c_std]
    S = [[0, 1, 2], [1], [0, 2]]
    to = orc.new_transcript()
    msgs_o, rand_o = orc.sumcheck_prove(f, mles, 4, [0b111, 0b101], [c[0], c[2]], to)
    t = pcs.KeccakTranscript()
    msgs, rand = pcs.sumcheck_prove_ccs(t, mles, 4, orc.field_elems(c, fl), S, pcs.FieldConfig(modulus, fl))
    assert np.array_equal(msgs, msgs_o) and np.array_equal(rand, rand_o)
    # device-resident tables through the ABI
    torch = pytest.importorskip("torch")
    dev = [torch.from_numpy(mles[k].view(np.int64)).cuda() for k in range(4)]
    comb = cabi.make_comb([0b111, 0b101], orc.field_elems([c[0], c[2]], fl))
    sc = cabi.Sumcheck(dev, nv, 4, cabi.make_field(modulus, fl), comb=comb)
    r = None
    for i in range(nv):
        assert np.array_equal(sc.round(r), msgs_o[i]), i
        r = rand_o[i]
    with pytest.raises(cabi.ZipError):
        cabi.Sumcheck(dev, nv, 4, cabi.make_field(modulus, fl), comb=cabi.make_comb([0b10000], orc.field_elems([c[0]], fl)))


# ------------------------------------------------------------------ sum of products (benches/sumcheck_benches.rs)
def _rand_poly_tables(modulus, fl, nv, products, seed):
    """rand_poly (sumcheck/utils.rs:27-65) with our own PRNG: fresh MLEs per product, a random coefficient each.
    Canonical residues as Montgomery limbs: the limbs below the modulus' top limb random, that one below 2^62."""
    rng = np.random.default_rng(seed)
    top = (modulus.bit_length() - 1) // 64  # index of the modulus' highest non-zero limb
    assert (modulus >> (64 * top)) > (1 << 62)

    def residues(shape):
        a = np.zeros(shape + (fl,), dtype=np.uint64)
        a[..., :top] = rng.integers(0, 2**64, size=shape + (top,), dtype=np.uint64)
        a[..., top] = rng.integers(0, 2**62, size=shape, dtype=np.uint64)
        return a

    K = sum(products)
    tables = residues((K, 1 << nv))
    masks, k = [], 0
    for m in products:
        masks.append(sum(1 << (k + i) for i in range(m)))
        k += m
    return tables, np.array(masks, dtype=np.uint32), residues((len(products),))


@pytest.mark.parametrize("modulus,fl", [(312829638388039969874974628075306023441, 3), (BENCH_MODULUS, 4)])
@pytest.mark.parametrize("nv,products", [(1, (2,)), (3, (1, 2)), (6, (2, 3, 4)), (10, (2, 3, 4, 2, 3, 4, 2)), (9, (4, 4, 4, 4, 4, 4, 4))])
def test_sum_of_products_equals_the_oracle(mods, modulus, fl, nv, products):
    """MLSumcheck::prove_as_subprotocol with rand_poly_comb_fn: every round message and challenge."""
    _, pcs = mods
    f = orc.make_field(modulus, fl)
    tables, masks, coeffs = _rand_poly_tables(modulus, fl, nv, products, seed=nv)
    degree = max(products)
    want_msgs, want_rand = orc.sumcheck_prove_products(f, tables, degree, masks, [orc.limbs_to_int(c) for c in coeffs],
                                                       orc.new_transcript())
    msgs, rand = pcs.sumcheck_prove_products(pcs.KeccakTranscript(), tables, degree, masks, coeffs, pcs.FieldConfig(modulus, fl))
    assert np.array_equal(msgs, want_msgs) and np.array_equal(rand, want_rand)


def test_sum_of_products_bad_shapes(mods):
    _, pcs = mods
    field = pcs.FieldConfig(BENCH_MODULUS, 4)
    tables, masks, coeffs = _rand_poly_tables(BENCH_MODULUS, 4, 3, (2, 2), seed=1)
    with pytest.raises(pcs.InvalidPcsParam):  # five multiplicands in one product
        big, _, c1 = _rand_poly_tables(BENCH_MODULUS, 4, 3, (5,), seed=2)
        pcs.sumcheck_prove_products(pcs.KeccakTranscript(), big, 5, np.array([31], dtype=np.uint32), c1, field)
    with pytest.raises(pcs.ReferencePanic):   # a product refers to an MLE that does not exist
        pcs.sumcheck_prove_products(pcs.KeccakTranscript(), tables, 2, np.array([3, 1 << 7], dtype=np.uint32), coeffs, field)


def test_round_begin_end_protocol(mods):
    """zip_sumcheck_round_begin / _end: a round is enqueued once and collected once; begin + end == round."""
    cabi, _ = mods
    L = cabi.lib()
    field = cabi.make_field(BENCH_MODULUS, 4)
    tables, _, _ = _rand_poly_tables(BENCH_MODULUS, 4, 5, (2,), seed=3)
    a = cabi.Sumcheck(tables, 5, 2, field)
    b = cabi.Sumcheck(tables, 5, 2, field)
    out = np.zeros((3, 4), dtype=np.uint64)
    assert L.zip_sumcheck_round_end(a._h, out.ctypes.data) != 0          # nothing in flight
    assert L.zip_sumcheck_round_begin(a._h, None) == 0
    assert L.zip_sumcheck_round_begin(a._h, None) != 0                    # the first one has not been collected
    assert L.zip_sumcheck_round_end(a._h, out.ctypes.data) == 0
    assert np.array_equal(out, b.round())
    r = np.array([5, 0, 0, 0], dtype=np.uint64)
    assert L.zip_sumcheck_round_begin(a._h, r.ctypes.data) == 0
    assert L.zip_sumcheck_round_end(a._h, out.ctypes.data) == 0
    assert np.array_equal(out, b.round(r))
    a.free()
    b.free()


@pytest.mark.parametrize("quad", ["1", "2", "0"])
@pytest.mark.parametrize("modulus,fl", [(BENCH_MODULUS, 4), (TEST_MODULUS_2, 2), (MOD_NO_SPARE, 4), (MOD_3LIMB, 3)])
@pytest.mark.parametrize("K,nv", [(1, 1), (2, 2), (3, 3), (4, 9), (2, 13), (4, 12)])
def test_degree_3_rounds_on_both_kernels(mods, monkeypatch, quad, modulus, fl, K, nv):
    """The folding rounds of a degree-3 sumcheck run on sumcheck_round_quad_kernel (four lanes per hypercube point, one
    evaluation point each; the quad shares the folded pairs) by default, every round with ZIP_HIP_SUMCHECK_QUAD=2, none
    with =0 (the one-thread-per-point kernel): all give the oracle's messages, in one-launch rounds (<= 64 workgroups) and in
    rounds with the separate reduction."""
    cabi, pcs = mods
    monkeypatch.setenv("ZIP_HIP_SUMCHECK_QUAD", quad)
    f = orc.make_field(modulus, fl)
    mles = _tables(f, fl, modulus, K, nv, seed=nv * 11 + K)
    to = orc.new_transcript()
    msgs_o, rand_o = orc.sumcheck_prove_product(f, mles, 3, to)
    t = pcs.KeccakTranscript()
    msgs, rand = pcs.sumcheck_prove_product(t, mles, 3, pcs.FieldConfig(modulus, fl))
    assert np.array_equal(msgs, msgs_o) and np.array_equal(rand, rand_o)


@pytest.mark.parametrize("quad", ["1", "2", "0"])
@pytest.mark.parametrize("modulus,fl", [(BENCH_MODULUS, 4), (MOD_NO_SPARE, 4), (MOD_3LIMB, 3)])
@pytest.mark.parametrize("K,nv", [(1, 2), (2, 9), (3, 13), (2, 18)])
def test_degree_2_rounds_on_both_kernels(mods, monkeypatch, quad, modulus, fl, K, nv):
    """Degree 2 (ZincProver's second sumcheck): the quad kernel with its fourth lane idle takes the folding rounds of at
    most 2^16 points by default, every round with ZIP_HIP_SUMCHECK_QUAD=2, none with =0."""
    torch = pytest.importorskip("torch")
    cabi, pcs = mods
    monkeypatch.setenv("ZIP_HIP_SUMCHECK_QUAD", quad)
    f = orc.make_field(modulus, fl)
    if nv <= 13:
        mles = _tables(f, fl, modulus, K, nv, seed=nv * 13 + K)
    else:
        rng = np.random.default_rng(nv)
        mles = rng.integers(0, 1 << 62, size=(K, 1 << nv, fl), dtype=np.uint64)
        mles[..., fl - 1] >>= np.uint64(6 if fl == 4 else 4)
    to = orc.new_transcript()
    msgs_o, rand_o = orc.sumcheck_prove_product(f, mles, 2, to)
    dev = [torch.from_numpy(mles[k].view(np.int64)).cuda() for k in range(K)]
    sc = cabi.Sumcheck(dev, nv, 2, cabi.make_field(modulus, fl))
    r = None
    for i in range(nv):
        assert np.array_equal(sc.round(r), msgs_o[i]), i
        r = rand_o[i]
    sc.free()


@pytest.mark.parametrize("quad", ["1", "2", "0"])
def test_ccs_sumcheck_2pow18_device_tables_on_both_kernels(mods, monkeypatch, quad):
    """ZincProver's first sumcheck shape -- (M0 z * M1 z - M2 z) * eq, degree 3, four tables in HBM -- at 2^18: every round
    message equals the oracle's, the caller's tables are only read."""
    torch = pytest.importorskip("torch")
    cabi, pcs = mods
    monkeypatch.setenv("ZIP_HIP_SUMCHECK_QUAD", quad)
    nv, fl, K = 18, 4, 4
    f = orc.make_field(BENCH_MODULUS, fl)
    rng = np.random.default_rng(18)
    mles = rng.integers(0, 1 << 62, size=(K, 1 << nv, fl), dtype=np.uint64)
    mles[..., fl - 1] >>= np.uint64(6)
    R = 1 << (64 * fl)
    c = [1 * R % BENCH_MODULUS, (BENCH_MODULUS - 1) * R % BENCH_MODULUS]
    to = orc.new_transcript()
    msgs_o, rand_o = orc.sumcheck_prove(f, mles, 3, [0b011, 0b100], c, to)
    dev = [torch.from_numpy(mles[k].view(np.int64)).cuda() for k in range(K)]
    before = [d.clone() for d in dev]
    sc = cabi.Sumcheck(dev, nv, 3, cabi.make_field(BENCH_MODULUS, fl), comb=cabi.make_comb([0b011, 0b100], orc.field_elems(c, fl)))
    r = None
    for i in range(nv):
        assert np.array_equal(sc.round(r), msgs_o[i]), i
        r = rand_o[i]
    sc.free()
    assert all(torch.equal(a, b) for a, b in zip(dev, before))
