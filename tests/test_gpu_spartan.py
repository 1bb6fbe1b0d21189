"""ZincProver on the device (BASELINE configs[4]): the field loops of SpartanProver::prove (zip_ccs_*), the two
sumchecks (zip_sumcheck_*) and the Zip PCS step, driven by the host mirror with the transcript on the host, against
the oracle's restatement of src/zinc/prover.rs and checked by the oracle's restatement of src/zinc/verifier.rs.
Restates src/zinc/tests.rs."""
import numpy as np
import pytest

import _ccs
import _oracle as orc

pytestmark = pytest.mark.gpu

Q192 = 312829638388039969874974628075306023441          # zinc/tests.rs:28
Q256 = 115792089237316195423570985008687907853269984665640564039457584007913129639747  # spartan_benches.rs:152 (top bit set)
QSTARK = 3618502788666131213697322783095070105623107215331596699973092056135872020481  # spartan_benches.rs:161
Q128 = 57316695564490278656402085503
FIELDS = [(Q192, 3), (Q256, 4), (QSTARK, 4), (Q128, 2)]


@pytest.fixture(scope="module")
def mods():
    from zinc_amd import cabi, pcs

    if cabi.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu tests must run on the MI355X box")
    return cabi, pcs


def _instances():
    return [("dummy2", _ccs.dummy_ccs_from_len(2)), ("dummy8", _ccs.dummy_ccs_from_len(8)),
            ("dummy1k", _ccs.dummy_ccs_from_len(1 << 10, seed=77)), ("vitalik", _ccs.vitalik_ccs(3)),
            ("dummy8k", _ccs.dummy_ccs_from_len(1 << 13, seed=13))]  # >= 12 variables: the split eq() tables


@pytest.mark.parametrize("q,fl", FIELDS)
@pytest.mark.parametrize("name,inst", _instances())
def test_ccs_tables_equal_the_oracle(mods, q, fl, name, inst):
    """z_ccs in F_q, M_k z, eq(beta), eq(r_x), the second sumcheck's table and V_s, one by one."""
    cabi, _ = mods
    f = orc.make_field(q, fl)
    o = orc.Ccs(inst)
    d = cabi.Ccs(inst.matrices, inst.s, cabi.make_field(q, fl))
    d.set_z(inst.z)
    zf = d.download(cabi.CCS_Z_FIELD)
    for i in range(inst.m):
        want = orc.field_from_i64(f, int(inst.z[i])) if i < len(inst.z) else 0
        assert orc.limbs_to_int(zf[i]) == want, i
    mz_o = o.mz(f)
    for k in range(inst.t):
        assert np.array_equal(d.download(cabi.CCS_MZ, k), mz_o[k]), k
    rng = np.random.default_rng(inst.s)
    r = orc.field_elems([orc.field_from_i64(f, int(v)) for v in rng.integers(-2**62, 2**62, size=inst.s)], fl)
    gamma = orc.field_elems([orc.field_from_i64(f, 0x1234567890ABCDEF)], fl)[0]
    eq_o = orc.build_eq_x_r(f, r)
    d.eq_table(r, 0)
    assert np.array_equal(d.download(cabi.CCS_EQ, 0), eq_o)
    vs = d.second_table(r, gamma)
    assert np.array_equal(d.download(cabi.CCS_EQ, 1), eq_o)
    assert np.array_equal(d.download(cabi.CCS_SECOND), o.second_table(f, eq_o, gamma))
    R_inv = pow(1 << (64 * fl), -1, q)
    for k in range(inst.t):  # V_s[k] = <Mz_k, eq(r_x)>, in Python integers (Montgomery: a*b*R^-1)
        acc = sum(orc.limbs_to_int(a) * orc.limbs_to_int(b) for a, b in zip(mz_o[k], eq_o)) * R_inv % q
        assert orc.limbs_to_int(vs[k]) == acc
    d.free()


def test_ccs_shape_errors(mods):
    cabi, _ = mods
    inst = _ccs.dummy_ccs_from_len(8)
    field = cabi.make_field(Q192, 3)
    with pytest.raises(cabi.ZipError):  # n_cols != 2^s: LengthsNotEqual in mat_vec_mul
        cabi.Ccs(inst.matrices, 2, field)
    d = cabi.Ccs(inst.matrices, 3, field)
    with pytest.raises(cabi.ZipError):
        d.set_z(np.zeros(9, dtype=np.int64))
    with pytest.raises(cabi.ZipError):  # nothing built yet
        d.table(cabi.CCS_SECOND)
    d.free()


def _device_spartan(pcs, inst, q, fl, label=b"", with_pcs=False):
    field = pcs.FieldConfig(q, fl)
    t = pcs.KeccakTranscript()
    if label:
        t.absorb(label)
    prover = pcs.ZincProver()
    x, w = inst.z[:1], inst.z[2:]  # z = (x, 1, w), pub_io_len = 1
    fn = prover.prove if with_pcs else prover.spartan_prove
    return fn(inst.matrices, inst.s, inst.d, inst.S, inst.c, x, w, t, field), t


@pytest.mark.parametrize("q,fl", FIELDS)
@pytest.mark.parametrize("name,inst", _instances())
def test_spartan_prove_equals_the_oracle(mods, q, fl, name, inst):
    """SpartanProver::prove: every round message of both sumchecks, V_s and r_y; the verifier accepts."""
    _, pcs = mods
    f = orc.make_field(q, fl)
    o = orc.Ccs(inst)
    want = o.spartan_prove(f, orc.new_transcript())
    got, _ = _device_spartan(pcs, inst, q, fl)
    for key in ("msgs1", "msgs2", "V_s", "r_y"):
        assert np.array_equal(got[key], want[key]), key
    rc, pts = o.spartan_verify(f, got, orc.new_transcript())
    assert rc == 0 and np.array_equal(pts["r_y"], got["r_y"])


def test_dummy_spartan_prover_and_verifier_reference_size(mods):
    """zinc/tests.rs:22-57 and :110-157: n = 2^13, the 192-bit prime, prove then verify."""
    _, pcs = mods
    inst = _ccs.dummy_ccs_from_len(1 << 13)
    got, _ = _device_spartan(pcs, inst, Q192, 3)
    rc, _ = orc.Ccs(inst).spartan_verify(orc.make_field(Q192, 3), got, orc.new_transcript())
    assert rc == 0


def test_spartan_verifier_and_failing_verifier(mods):
    """zinc/tests.rs:59-108 and :159-209 (x^3 + x + 5 = y at x = 3; then w_ccs[3] = 0)."""
    _, pcs = mods
    f = orc.make_field(Q192, 3)
    good = _ccs.vitalik_ccs(3)
    got, _ = _device_spartan(pcs, good, Q192, 3)
    assert orc.Ccs(good).spartan_verify(f, got, orc.new_transcript())[0] == 0
    bad = _ccs.vitalik_ccs(3, break_witness=True)
    got, _ = _device_spartan(pcs, bad, Q192, 3)  # the prover still succeeds (tests.rs:184-193)
    assert np.array_equal(got["msgs1"], orc.Ccs(bad).spartan_prove(f, orc.new_transcript())["msgs1"])
    assert orc.Ccs(bad).spartan_verify(f, got, orc.new_transcript())[0] == orc.ORC_ERR_PROOF


@pytest.mark.parametrize("q,fl,log_n", [(Q192, 3, 4), (QSTARK, 4, 10), (Q256, 4, 8), (Q192, 3, 13)])
def test_zinc_prove_end_to_end(mods, q, fl, log_n):
    """Prover::prove (prover.rs:50-88): the Spartan proof, then RaaCode::new from the same transcript, commit, the
    evaluation v and the PCS proof -- all equal to the oracle's; the oracle's ZincVerifier steps
    (SpartanVerifier::verify, MultilinearZip::verify, the final equation of verify_pcs_proof) accept."""
    _, pcs = mods
    inst = _ccs.dummy_ccs_from_len(1 << log_n, seed=log_n)
    f = orc.make_field(q, fl)
    o = orc.Ccs(inst)
    # oracle prover
    ko = orc.new_transcript()
    orc.absorb(ko, b"zinc")
    want = o.spartan_prove(f, ko)
    s1 = orc.lib().orc_tr_get_u64(orc.C.byref(ko))
    s2 = orc.lib().orc_tr_get_u64(orc.C.byref(ko))
    z = orc.Zip(log_n, seeds=(s1, s2))
    rows, layers, roots_o = z.commit(inst.z)
    proof_o, _, _ = z.open(f, inst.z, rows, layers, want["r_y"], orc.new_transcript())
    v_o = z.mle_eval(f, inst.z, want["r_y"])
    # device prover
    got, _ = _device_spartan(pcs, inst, q, fl, label=b"zinc", with_pcs=True)
    for key in ("msgs1", "msgs2", "V_s", "r_y"):
        assert np.array_equal(got[key], want[key]), key
    zp = got["zip_proof"]
    assert np.array_equal(zp["z_comm"], roots_o)
    assert orc.limbs_to_int(zp["v"]) == v_o
    assert np.array_equal(zp["pcs_proof"], proof_o)
    # oracle verifier over the device's proof
    kv = orc.new_transcript()
    orc.absorb(kv, b"zinc")
    rc, pts = o.spartan_verify(f, got, kv)
    assert rc == 0
    t1 = orc.lib().orc_tr_get_u64(orc.C.byref(kv))
    t2 = orc.lib().orc_tr_get_u64(orc.C.byref(kv))
    assert (t1, t2) == (s1, s2)
    pcs_rc = z.verify(f, zp["z_comm"], pts["r_y"], orc.limbs_to_int(zp["v"]), zp["pcs_proof"])
    if q == Q256:  # the reference rejects its own PCS proofs for a modulus with the top bit set
        assert pcs_rc != 0
    else:
        assert pcs_rc == 0
    assert o.final_check(f, pts, zp["v"]) == 0
    wrong = zp["v"].copy()
    wrong[0] ^= np.uint64(1)
    assert o.final_check(f, pts, wrong) == orc.ORC_ERR_PROOF


def test_prepared_circuit_is_reused_across_proofs(mods):
    """PreparedCcs: the matrices stay on the device; proofs for different witnesses of the circuit equal the
    unprepared ones (and the oracle's)."""
    _, pcs = mods
    field = pcs.FieldConfig(QSTARK, 4)
    f = orc.make_field(QSTARK, 4)
    prover = pcs.ZincProver()
    base = _ccs.vitalik_ccs(3)
    prep = prover.prepare(base.matrices, base.s, field)
    for x in (3, 5, -7):
        inst = _ccs.vitalik_ccs(x)
        want = orc.Ccs(inst).spartan_prove(f, orc.new_transcript())
        got = prover.spartan_prove(inst.matrices, inst.s, inst.d, inst.S, inst.c, inst.z[:1], inst.z[2:],
                                   pcs.KeccakTranscript(), field, prepared=prep)
        for key in ("msgs1", "msgs2", "V_s", "r_y"):
            assert np.array_equal(got[key], want[key]), (x, key)
    other = pcs.FieldConfig(Q192, 3)
    with pytest.raises(pcs.ReferencePanic):
        prover.spartan_prove(base.matrices, base.s, base.d, base.S, base.c, base.z[:1], base.z[2:],
                             pcs.KeccakTranscript(), other, prepared=prep)


def test_ccs_tables_at_2_pow_20(mods):
    """MiB-sized index arrays whose byte length is not a multiple of the copy threads (row_ptr has 2^20 + 1 entries):
    the last row of M z depends on the last word of row_ptr."""
    cabi, _ = mods
    inst = _ccs.dummy_ccs_from_len(1 << 20, seed=20)
    f = orc.make_field(QSTARK, 4)
    want = orc.Ccs(inst).mz(f)
    for rep in range(2):  # the second handle runs on recycled device blocks and pinned buffers
        d = cabi.Ccs(inst.matrices, inst.s, cabi.make_field(QSTARK, 4))
        d.set_z(inst.z)
        for k in range(inst.t):
            assert np.array_equal(d.download(cabi.CCS_MZ, k), want[k]), (rep, k)
        d.free()
    cabi.lib().zip_release_cached_memory()


# ------------------------------------------------------------------------------------------------ verifier
@pytest.mark.parametrize("q,fl", FIELDS)
@pytest.mark.parametrize("name,inst", _instances())
def test_matrix_mles_at_a_point_equal_the_oracle(mods, q, fl, name, inst):
    """V_xy of verify_pcs_proof (verifier.rs:248-261): mle[M_k](r_x, r_y) without the dense 2^(2s) table."""
    cabi, _ = mods
    f = orc.make_field(q, fl)
    rng = np.random.default_rng(inst.s + 40)
    rx = orc.field_elems([orc.field_from_i64(f, int(v)) for v in rng.integers(-2**62, 2**62, size=inst.s)], fl)
    ry = orc.field_elems([orc.field_from_i64(f, int(v)) for v in rng.integers(-2**62, 2**62, size=inst.s)], fl)
    d = cabi.Ccs(inst.matrices, inst.s, cabi.make_field(q, fl))
    assert np.array_equal(d.eval_matrices(rx, ry), orc.Ccs(inst).eval_matrices(f, rx, ry))
    d.free()


def _args(inst):
    return inst.matrices, inst.s, inst.d, inst.S, inst.c


@pytest.mark.parametrize("q,fl", FIELDS)
@pytest.mark.parametrize("name,inst", _instances())
def test_spartan_verifier_mirror_equals_the_oracle(mods, q, fl, name, inst):
    """SpartanVerifier::verify in the host mirror: same verification points as the oracle on honest proofs, same
    rejections on tampered ones."""
    _, pcs = mods
    f = orc.make_field(q, fl)
    field = pcs.FieldConfig(q, fl)
    o = orc.Ccs(inst)
    proof, _ = _device_spartan(pcs, inst, q, fl)
    rc, want = o.spartan_verify(f, proof, orc.new_transcript())
    assert rc == 0
    got = pcs.ZincVerifier().spartan_verify(*_args(inst), proof, pcs.KeccakTranscript(), field)
    assert np.array_equal(got["rx_ry"], np.concatenate([want["r_x"], want["r_y"]]))
    assert np.array_equal(got["e_y"], want["e_y"]) and np.array_equal(got["gamma"], want["gamma"])
    for key, idx in (("msgs1", (0, 1, 0)), ("msgs1", (inst.s - 1, 3, 0)), ("msgs2", (0, 0, 0)), ("V_s", (1, 0))):
        bad = {k: v.copy() for k, v in proof.items()}
        bad[key][idx] ^= np.uint64(2)
        assert o.spartan_verify(f, bad, orc.new_transcript())[0] == orc.ORC_ERR_PROOF
        with pytest.raises(pcs.SpartanError):
            pcs.ZincVerifier().spartan_verify(*_args(inst), bad, pcs.KeccakTranscript(), field)
    # the last round of the second sumcheck only moves e_y, which SpartanVerifier::verify hands on unchecked (the
    # final equation of verify_pcs_proof catches it): accepted here by the reference, the oracle and the mirror
    bad = {k: v.copy() for k, v in proof.items()}
    bad["msgs2"][inst.s - 1, 2, 0] ^= np.uint64(2)
    rc, moved = o.spartan_verify(f, bad, orc.new_transcript())
    got = pcs.ZincVerifier().spartan_verify(*_args(inst), bad, pcs.KeccakTranscript(), field)
    assert rc == 0 and np.array_equal(got["e_y"], moved["e_y"]) and not np.array_equal(moved["e_y"], want["e_y"])


def test_failing_spartan_verifier_mirror(mods):
    """zinc/tests.rs:159-209 with the mirror's verifier."""
    _, pcs = mods
    bad = _ccs.vitalik_ccs(3, break_witness=True)
    proof, _ = _device_spartan(pcs, bad, Q192, 3)
    with pytest.raises(pcs.SpartanError):
        pcs.ZincVerifier().spartan_verify(*_args(bad), proof, pcs.KeccakTranscript(), pcs.FieldConfig(Q192, 3))


@pytest.mark.parametrize("q,fl,log_n", [(Q192, 3, 4), (QSTARK, 4, 10), (Q128, 2, 6), (Q256, 4, 8)])
def test_zinc_prove_then_verify_on_the_device(mods, q, fl, log_n):
    """Prover::prove then Verifier::verify (without its field draw), both through the mirror; the verdicts and the
    transcript state afterwards are the oracle's."""
    _, pcs = mods
    inst = _ccs.dummy_ccs_from_len(1 << log_n, seed=log_n + 1)
    field = pcs.FieldConfig(q, fl)
    proof, _ = _device_spartan(pcs, inst, q, fl, label=b"zv", with_pcs=True)
    # the oracle's verifier over the same proof, for the expected verdict and the final transcript state
    f = orc.make_field(q, fl)
    o = orc.Ccs(inst)
    kv = orc.new_transcript()
    orc.absorb(kv, b"zv")
    rc, pts = o.spartan_verify(f, proof, kv)
    assert rc == 0
    s1 = orc.lib().orc_tr_get_u64(orc.C.byref(kv))
    s2 = orc.lib().orc_tr_get_u64(orc.C.byref(kv))
    zp = proof["zip_proof"]
    pcs_rc = orc.Zip(log_n, seeds=(s1, s2)).verify(f, zp["z_comm"], pts["r_y"], orc.limbs_to_int(zp["v"]), zp["pcs_proof"])

    vt = pcs.KeccakTranscript()
    vt.absorb(b"zv")
    verifier = pcs.ZincVerifier()
    if pcs_rc != 0:  # the modulus with the top bit set: the reference rejects its own PCS proofs
        assert q == Q256
        with pytest.raises(pcs.InvalidPcsOpen):
            verifier.verify(*_args(inst), proof, vt, field)
        return
    got = verifier.verify(*_args(inst), proof, vt, field)
    assert np.array_equal(got["rx_ry"][inst.s:], pts["r_y"])
    assert vt.get_u64() == orc.lib().orc_tr_get_u64(orc.C.byref(kv))  # same Fiat-Shamir state after verification

    def fresh():
        t = pcs.KeccakTranscript()
        t.absorb(b"zv")
        return t

    # a different circuit (C = diag(z) with one entry changed): both sumchecks and the PCS still pass, the final
    # equation lin_comb(gamma, V_xy) * v == e_y (verifier.rs:264-269) does not
    other = _ccs.dummy_ccs(inst.z.copy())
    other.matrices[2].values = other.matrices[2].values.copy()
    other.matrices[2].values[3] += 1
    with pytest.raises(pcs.SpartanError, match="e_y"):
        verifier.verify(*_args(other), proof, fresh(), field)
    assert o.final_check(f, pts, zp["v"]) == 0 and orc.Ccs(other).final_check(f, pts, zp["v"]) == orc.ORC_ERR_PROOF
    # a flipped byte in the PCS proof, a wrong evaluation, a tampered round message
    bad = dict(proof, zip_proof=dict(zp, pcs_proof=zp["pcs_proof"].copy()))
    bad["zip_proof"]["pcs_proof"][zp["pcs_proof"].size // 3] ^= 1
    with pytest.raises(pcs.InvalidPcsOpen):
        verifier.verify(*_args(inst), bad, fresh(), field)
    wrong_v = zp["v"].copy()
    wrong_v[0] ^= np.uint64(1)
    with pytest.raises(pcs.InvalidPcsOpen, match="Evaluation consistency failure"):
        verifier.verify(*_args(inst), dict(proof, zip_proof=dict(zp, v=wrong_v)), fresh(), field)
    m2 = proof["msgs2"].copy()
    m2[1, 0, 0] ^= np.uint64(1)
    with pytest.raises(pcs.SpartanError):
        verifier.verify(*_args(inst), dict(proof, msgs2=m2), fresh(), field)
    # with the circuit prepared once
    prep = pcs.ZincProver().prepare(inst.matrices, inst.s, field)
    verifier.verify(*_args(inst), proof, fresh(), field, prepared=prep)


def test_zinc_prover_at_2_pow_20_constraints(mods):
    """BASELINE configs[4] at its full size: the spartan_benches.rs instance with 2^20 constraints, Stark prime.
    SpartanProver::prove equals the oracle's message for message; Prover::prove -> Verifier::verify round trip on the
    device with the circuit prepared once; the PCS proof has the reference's length (commit.rs:712-775)."""
    _, pcs = mods
    log_n = 20
    inst = _ccs.dummy_ccs_from_len(1 << log_n)
    f = orc.make_field(QSTARK, 4)
    field = pcs.FieldConfig(QSTARK, 4)
    want = orc.Ccs(inst).spartan_prove(f, orc.new_transcript())
    prover = pcs.ZincProver()
    prep = prover.prepare(inst.matrices, inst.s, field)
    x, w = inst.z[:1], inst.z[2:]
    proof = prover.prove(*_args(inst), x, w, pcs.KeccakTranscript(), field, prepared=prep)
    for key in ("msgs1", "msgs2", "V_s", "r_y"):
        assert np.array_equal(proof[key], want[key]), key
    row_len, num_rows, depth = 1024, 1024, 11
    assert proof["zip_proof"]["pcs_proof"].size == row_len * 64 + 1000 * num_rows * (32 + 8 + 32 * depth) + row_len * 32
    pts = pcs.ZincVerifier().verify(*_args(inst), proof, pcs.KeccakTranscript(), field, prepared=prep)
    assert np.array_equal(pts["rx_ry"][log_n:], want["r_y"])
    bad = dict(proof, V_s=proof["V_s"].copy())
    bad["V_s"][0, 0] ^= np.uint64(1)
    with pytest.raises(pcs.SpartanError):
        pcs.ZincVerifier().verify(*_args(inst), bad, pcs.KeccakTranscript(), field, prepared=prep)


def test_one_prepared_circuit_shared_by_concurrent_provers(mods):
    """Proofs from several threads over the same PreparedCcs are serialised inside and stay correct."""
    import threading

    _, pcs = mods
    field = pcs.FieldConfig(Q192, 3)
    f = orc.make_field(Q192, 3)
    base = _ccs.vitalik_ccs(3)
    prover = pcs.ZincProver()
    prep = prover.prepare(base.matrices, base.s, field)
    xs = [2, 3, 4, 5, 6, 7]
    out, errs = {}, []

    def work(x):
        try:
            inst = _ccs.vitalik_ccs(x)
            for _ in range(3):
                out[x] = prover.spartan_prove(*_args(inst), inst.z[:1], inst.z[2:], pcs.KeccakTranscript(), field, prepared=prep)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    threads = [threading.Thread(target=work, args=(x,)) for x in xs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    for x in xs:
        want = orc.Ccs(_ccs.vitalik_ccs(x)).spartan_prove(f, orc.new_transcript())
        for key in ("msgs1", "msgs2", "V_s", "r_y"):
            assert np.array_equal(out[x][key], want[key]), (x, key)


def test_prover_rejects_what_the_reference_prover_cannot_handle(mods):
    """Shapes on which the reference's own prover panics or silently mis-indexes are refused, not computed."""
    _, pcs = mods
    field = pcs.FieldConfig(Q192, 3)
    inst = _ccs.dummy_ccs_from_len(8)
    prover = pcs.ZincProver()
    x, w = inst.z[:1], inst.z[2:]

    def run(**kw):
        a = dict(matrices=inst.matrices, s=inst.s, d=inst.d, S=inst.S, c=inst.c, public_input=x, w_ccs=w)
        a.update(kw)
        return prover.spartan_prove(a["matrices"], a["s"], a["d"], a["S"], a["c"], a["public_input"], a["w_ccs"],
                                    pcs.KeccakTranscript(), field)

    run()  # the honest call works
    with pytest.raises(pcs.InvalidPcsParam):  # a zero coefficient shifts the MLE list comb_fn_1 indexes (zinc/utils.rs:66-88)
        run(c=[1, 0])
    run(S=[[1, 0], [2]])  # the same multiset: fine
    with pytest.raises(pcs.InvalidPcsParam):  # S must enumerate the matrices in order: comb_fn_1 reads vals[j] by matrix number
        run(S=[[2], [0, 1]])
    with pytest.raises(pcs.ReferencePanic):   # z longer than the matrices are wide: LengthsNotEqual (ccs/utils.rs:52-59)
        run(w_ccs=np.zeros(20, dtype=np.int64))
    with pytest.raises(pcs.ReferencePanic):   # n_cols != 2^s
        run(s=2)
    wide = _ccs.CsrMatrix(8, 8, [[(1, 9)]])   # a column index outside the matrix: the reference indexes out of bounds
    with pytest.raises(pcs.ReferencePanic):
        run(matrices=[wide, inst.matrices[1], inst.matrices[2]])
    with pytest.raises(pcs.InvalidPcsParam):  # degree above what the round kernel is instantiated for
        run(d=6)
