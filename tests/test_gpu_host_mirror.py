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
