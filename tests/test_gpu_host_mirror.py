"""End to end through the host mirror (the shim's role): RaaCode::new from a Keccak transcript,
MultilinearZip::{setup, commit, open} with a fresh PcsTranscript, exactly the call sequence of
ZincProver::commit_z_mle_and_prove_evaluation (src/zinc/prover.rs:305-328) -- proof bytes, roots
and the final Fiat-Shamir state must equal the oracle's.  Reads like src/zip/tests.rs."""
import numpy as np
import pytest

import _oracle as orc

pytestmark = pytest.mark.gpu

BENCH_MODULUS = 106319353542452952636349991594949358997917625194731877894581586278529202198383
TEST_MODULUS_2 = 57316695564490278656402085503


@pytest.fixture(scope="module")
def pcs():
    from zinc_amd import cabi, pcs as m

    if cabi.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu tests must run on the MI355X box")
    return m


def _oracle_flow(nv, modulus, fl, evals, point_i, label):
    ko = orc.new_transcript()
    orc.absorb(ko, label)
    s1 = orc.lib().orc_tr_get_u64(orc.C.byref(ko))
    s2 = orc.lib().orc_tr_get_u64(orc.C.byref(ko))
    z = orc.Zip(nv, seeds=(s1, s2))
    f = orc.make_field(modulus, fl)
    rows, layers, roots = z.commit(evals)
    point = orc.point_to_field(f, point_i) if nv else np.zeros((0, fl), np.uint64)
    fs = orc.new_transcript()  # PcsTranscript::new()
    proof, _, _ = z.open(f, evals, rows, layers, point, fs)
    return z, f, roots, point, proof, orc.lib().orc_tr_get_u64(orc.C.byref(fs))


@pytest.mark.parametrize("nv,modulus,fl", [(8, TEST_MODULUS_2, 2), (8, BENCH_MODULUS, 4), (3, TEST_MODULUS_2, 2),
                                            (11, BENCH_MODULUS, 4), (0, BENCH_MODULUS, 4), (16, BENCH_MODULUS, 4)])
def test_zip_evaluation_end_to_end(pcs, nv, modulus, fl):
    """src/zip/tests.rs:116-146 (test_zip_evaluation) with the GPU prover and the oracle verifier."""
    rng = np.random.default_rng(nv)
    evals = rng.integers(-128, 128, size=1 << nv, dtype=np.int64) if fl == 2 else orc.splitmix64(nv, 1 << nv)
    point_i = rng.integers(-128, 128, size=nv, dtype=np.int64)
    z, fo, roots_o, point, proof_o, probe_o = _oracle_flow(nv, modulus, fl, evals, point_i, b"zinc-amd")

    transcript = pcs.KeccakTranscript()
    transcript.absorb(b"zinc-amd")
    linear_code = pcs.RaaCode(1 << nv, transcript)
    param = pcs.MultilinearZip.setup(1 << nv, linear_code)
    assert (param.num_rows, param.row_len, param.codeword_len) == (z.num_rows, z.row_len, z.codeword_len)
    data, roots = pcs.MultilinearZip.commit(param, evals)
    assert np.array_equal(roots, roots_o)
    field = pcs.FieldConfig(modulus, fl)
    pcs_transcript = pcs.PcsTranscript()
    pcs.MultilinearZip.open(param, evals, data, field.map_to_field(point_i) if nv else np.zeros((0, fl), np.uint64),
                            field, pcs_transcript)
    proof = pcs_transcript.into_proof()
    assert proof.size == z.proof_len(fl)
    assert np.array_equal(proof, proof_o)
    assert pcs_transcript.probe() == probe_o  # the evaluation row was absorbed identically
    if nv > 0:
        assert z.verify(fo, roots, point, z.mle_eval(fo, evals, point), proof) == 0


def test_failing_zip_commitment(pcs):
    """src/zip/tests.rs:43-59: a 4-variate polynomial against 3-variate parameters is InvalidPcsParam."""
    param = pcs.MultilinearZip.setup(8, pcs.RaaCode(8))
    with pytest.raises(pcs.InvalidPcsParam):
        pcs.MultilinearZip.commit(param, np.arange(16, dtype=np.int64), num_vars=4)
    data, roots = pcs.MultilinearZip.commit(param, np.arange(8, dtype=np.int64))  # tests.rs:26-41
    assert roots.shape == (4, 32)


def test_commit_wrong_size_panics_like_the_reference(pcs):
    """commit.rs:56-63 assert_eq! on the evaluation count."""
    param = pcs.MultilinearZip.setup(1 << 8, pcs.RaaCode(1 << 8))
    with pytest.raises(pcs.ReferencePanic):
        pcs.MultilinearZip.commit(param, np.arange(100, dtype=np.int64), num_vars=8)


def test_open_with_wrong_point_length_is_invalid_param(pcs):
    """pcs/utils.rs:48-56."""
    param = pcs.MultilinearZip.setup(1 << 8, pcs.RaaCode(1 << 8))
    evals = np.arange(256, dtype=np.int64)
    data, _ = pcs.MultilinearZip.commit(param, evals)
    field = pcs.FieldConfig(BENCH_MODULUS, 4)
    with pytest.raises(pcs.InvalidPcsParam):
        pcs.MultilinearZip.open(param, evals, data, field.map_to_field([1, 2, 3]), field, pcs.PcsTranscript())


def test_commit_no_merkle_then_open_fails(pcs):
    param = pcs.MultilinearZip.setup(1 << 8, pcs.RaaCode(1 << 8))
    evals = np.arange(256, dtype=np.int64)
    data, roots = pcs.MultilinearZip.commit(param, evals, with_merkle=False)
    assert roots is None
    field = pcs.FieldConfig(BENCH_MODULUS, 4)
    with pytest.raises(pcs.InvalidPcsParam):
        pcs.MultilinearZip.open(param, evals, data, field.map_to_field([1] * 8), field, pcs.PcsTranscript())


@pytest.mark.parametrize("nv,modulus,fl", [(8, TEST_MODULUS_2, 2), (10, BENCH_MODULUS, 4), (13, BENCH_MODULUS, 4)])
def test_prover_pcs_step_and_verifier_round_trip(pcs, nv, modulus, fl):
    """ZincProver::commit_z_mle_and_prove_evaluation (zinc/prover.rs:305-327) produces the oracle's
    ZipProof {z_comm, v, pcs_proof}; MultilinearZip::verify (verify_z.rs:19-38) accepts it, leaves the
    Fiat-Shamir state where the oracle's verifier leaves it, and rejects it once tampered with."""
    rng = np.random.default_rng(nv + 100)
    evals = rng.integers(-128, 128, size=1 << nv, dtype=np.int64) if fl == 2 else orc.splitmix64(nv + 9, 1 << nv)
    point_i = rng.integers(-128, 128, size=nv, dtype=np.int64)
    z, fo, roots_o, point, proof_o, probe_o = _oracle_flow(nv, modulus, fl, evals, point_i, b"spartan")
    v_o = z.mle_eval(fo, evals, point)

    field = pcs.FieldConfig(modulus, fl)
    transcript = pcs.KeccakTranscript()
    transcript.absorb(b"spartan")
    r_y = field.map_to_field(point_i)
    roots, v, proof = pcs.commit_z_mle_and_prove_evaluation(evals, r_y, transcript, field)
    assert np.array_equal(roots, roots_o)
    assert orc.limbs_to_int(v) == v_o
    assert np.array_equal(proof, proof_o)

    # verifier side: the code comes from the same main-transcript state
    vt = pcs.KeccakTranscript()
    vt.absorb(b"spartan")
    vp = pcs.MultilinearZip.setup(1 << nv, pcs.RaaCode(1 << nv, vt))
    t = pcs.PcsTranscript.from_proof(proof)
    pcs.MultilinearZip.verify(vp, roots, r_y, v, field, t)
    assert t.position() == proof.size
    fs = orc.new_transcript()
    assert z.verify(fo, roots_o, point, v_o, proof_o, fs=fs) == 0
    assert t.probe() == orc.lib().orc_tr_get_u64(orc.C.byref(fs))

    bad = proof.copy()
    bad[bad.size // 2] ^= 4
    with pytest.raises(pcs.InvalidPcsOpen):
        pcs.MultilinearZip.verify(vp, roots, r_y, v, field, pcs.PcsTranscript.from_proof(bad))
    with pytest.raises(pcs.InvalidPcsOpen, match="Evaluation consistency failure"):
        wrong = v.copy()
        wrong[0] ^= np.uint64(1)
        pcs.MultilinearZip.verify(vp, roots, r_y, wrong, field, pcs.PcsTranscript.from_proof(proof))
    with pytest.raises(pcs.InvalidPcsOpen):
        pcs.MultilinearZip.verify(vp, roots, r_y, v, field, pcs.PcsTranscript.from_proof(proof[:-8]))
    with pytest.raises(pcs.InvalidPcsParam):
        pcs.MultilinearZip.verify(vp, roots, r_y[:-1], v, field, pcs.PcsTranscript.from_proof(proof))


def test_evaluate_matches_map_to_field_then_evaluate(pcs):
    nv = 9
    field = pcs.FieldConfig(BENCH_MODULUS, 4)
    evals = orc.splitmix64(3, 1 << nv)
    point_i = np.arange(-4, nv - 4, dtype=np.int64)
    param = pcs.MultilinearZip.setup(1 << nv, pcs.RaaCode(1 << nv))
    v = pcs.MultilinearZip.evaluate(param, evals, field.map_to_field(point_i), field)
    fo = orc.make_field(BENCH_MODULUS, 4)
    assert orc.limbs_to_int(v) == orc.Zip(nv).mle_eval(fo, evals, orc.point_to_field(fo, point_i))
    with pytest.raises(pcs.InvalidPcsParam):
        pcs.MultilinearZip.evaluate(param, evals, field.map_to_field(point_i[:-1]), field)


def test_concurrent_commits_share_a_cached_context(pcs):
    """commit is callable from many threads at once in the reference (commit.rs:439-470); setup hands
    the same cached device context to all of them, and the library serialises the calls on it."""
    import threading

    nv = 10
    z = orc.Zip(nv, seeds=(1, 2))
    want = {}
    polys = [orc.splitmix64(100 + i, 1 << nv) for i in range(6)]
    for i, p in enumerate(polys):
        want[i] = z.commit(p)[2]
    got, errs = {}, []

    def work(i):
        try:
            pp = pcs.MultilinearZip.setup(1 << nv, pcs.RaaCode(1 << nv))
            for _ in range(3):
                _, roots = pcs.MultilinearZip.commit(pp, polys[i])
            got[i] = roots
        except Exception as e:  # pragma: no cover
            errs.append(e)

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(polys))]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs
    for i in range(len(polys)):
        assert np.array_equal(got[i], want[i])
