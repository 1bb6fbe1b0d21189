"""GPU: the verifier-side kernels (zip_verify) and the witness MLE evaluation
(item 2) against the CPU oracle, through the C ABI."""
import numpy as np
import pytest

import _oracle as orc

pytestmark = pytest.mark.gpu

BENCH_MODULUS = 106319353542452952636349991594949358997917625194731877894581586278529202198383
TEST_MODULUS_2 = 57316695564490278656402085503
MOD_NO_SPARE = (1 << 256) - 189  # benches/spartan_benches.rs:134-137; read as a negative Int by the reference
MOD_3LIMB = (1 << 190) - 11 * (1 << 64) - 59
FIELDS = [(BENCH_MODULUS, 4), (TEST_MODULUS_2, 2), (MOD_NO_SPARE, 4), (MOD_3LIMB, 3)]


@pytest.fixture(scope="module")
def cabi():
    from zinc_amd import cabi as m

    if m.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu tests must run on the MI355X box")
    return m


def _instance(num_vars, modulus, fl, seed=0, small=False):
    z = orc.Zip(num_vars)
    f = orc.make_field(modulus, fl)
    n = 1 << num_vars
    if small:
        evals = np.random.default_rng(seed).integers(-128, 128, size=n, dtype=np.int64)
    else:
        evals = orc.splitmix64(0x5A494E43 + seed, n).copy()
        evals[: min(n, 3)] = np.array([-(2**63), 2**63 - 1, -1], dtype=np.int64)[: min(n, 3)]
    point = orc.point_to_field(f, np.random.default_rng(seed + 1).integers(-50, 50, size=num_vars, dtype=np.int64))
    rows, layers, roots = z.commit(evals)
    proof, cols, coeffs = z.open(f, evals, rows, layers, point, orc.new_transcript())
    lr = z.num_rows.bit_length() - 1
    q0 = orc.build_eq_x_r(f, point[num_vars - lr:]) if lr else None
    q1 = orc.build_eq_x_r(f, point[: num_vars - lr]) if num_vars - lr else None
    ev = z.mle_eval(f, evals, point)
    return z, f, evals, point, roots, proof, cols, coeffs, q0, q1, ev


def _verify(cabi, ctx, z, fl, modulus, roots, proof, cols, coeffs, q0, q1, ev):
    return ctx.verify(roots, proof, coeffs if z.num_rows > 1 else None, cols, q0, q1,
                      np.array(orc.int_to_limbs(ev, fl), dtype=np.uint64), cabi.make_field(modulus, fl))


def _ctx(cabi, z):
    return cabi.ZipContext(z.num_vars, z.perm1, z.perm2, geometry_override=(z.row_len, z.num_rows, z.codeword_len))


@pytest.mark.parametrize("modulus,fl", FIELDS)
@pytest.mark.parametrize("num_vars", [2, 3, 4, 8, 9, 12])
def test_verify_agrees_with_the_oracle_on_honest_proofs(cabi, num_vars, modulus, fl):
    z, f, evals, point, roots, proof, cols, coeffs, q0, q1, ev = _instance(num_vars, modulus, fl, small=(fl == 2))
    orc_rc = z.verify(f, roots, point, ev, proof)  # the oracle's verifier (check_merkle on)
    ctx = _ctx(cabi, z)
    rep = _verify(cabi, ctx, z, fl, modulus, roots, proof, cols, coeffs, q0, q1, ev)
    if modulus == MOD_NO_SPARE:
        # 2^256 - 189 is a negative Int<4> inside the reference's `%=` (field.rs:550-557): FieldMap reduces
        # by 189 instead, is no longer additive, and the reference rejects its own proofs.  Same here.
        assert orc_rc != 0 and rep["verdict"] == cabi.VERIFY_PROXIMITY_Q0 and rep["bad_merkle_paths"] == 0
    else:
        assert orc_rc == 0
        assert rep == {"verdict": cabi.VERIFY_ACCEPT, "column": 0, "bad_merkle_paths": 0, "malformed_paths": 0}
    # and the witness MLE evaluation ZincProver computes before open (prover.rs:317-319)
    got = ctx.mle_eval(evals, q0, q1, cabi.make_field(modulus, fl))
    assert orc.limbs_to_int(got) == ev


@pytest.mark.parametrize("num_vars", [0, 1])
def test_verify_one_column_matrix_follows_the_reference(cabi, num_vars):
    """row_len == 1 leaves q_1 empty (pcs/utils.rs:252-276): <row, q_1> = 0, so the reference's own
    verifier only accepts a zero evaluation there (verify_z.rs:145-149); the device agrees."""
    z, f, evals, point, roots, proof, cols, coeffs, q0, q1, ev = _instance(num_vars, BENCH_MODULUS, 4)
    assert z.row_len == 1 and q1 is None and ev != 0
    ctx = _ctx(cabi, z)
    assert z.verify(f, roots, point, ev, proof) != 0
    assert _verify(cabi, ctx, z, 4, BENCH_MODULUS, roots, proof, cols, coeffs, q0, q1, ev)["verdict"] == cabi.VERIFY_EVAL_CONSISTENCY
    assert z.verify(f, roots, point, 0, proof) == 0
    assert _verify(cabi, ctx, z, 4, BENCH_MODULUS, roots, proof, cols, coeffs, q0, q1, 0)["verdict"] == cabi.VERIFY_ACCEPT
    # the prover-side evaluation is still the witness MLE at the point
    assert orc.limbs_to_int(ctx.mle_eval(evals, q0, q1, cabi.make_field(BENCH_MODULUS, 4))) == ev


def test_verify_device_resident_proof_2pow16(cabi):
    torch = pytest.importorskip("torch")
    z, f, evals, point, roots, proof, cols, coeffs, q0, q1, ev = _instance(16, BENCH_MODULUS, 4)
    ctx = _ctx(cabi, z)
    zf = cabi.make_field(BENCH_MODULUS, 4)
    # the proof the HIP prover wrote, still in HBM
    com, roots_g = ctx.commit(evals)
    out = torch.empty(ctx.proof_len(len(cols), 4), dtype=torch.uint8, device="cuda")
    com.open(evals, coeffs, cols, q0, zf, out=out)
    assert np.array_equal(roots_g, roots)
    rep = ctx.verify(roots_g, out, coeffs, cols, q0, q1, np.array(orc.int_to_limbs(ev, 4), dtype=np.uint64), zf)
    assert rep["verdict"] == cabi.VERIFY_ACCEPT and rep["bad_merkle_paths"] == 0


def _tampered(proof, at, xor=1):
    p = proof.copy()
    p[at] ^= xor
    return p


@pytest.mark.parametrize("modulus,fl", [(BENCH_MODULUS, 4), (TEST_MODULUS_2, 2), (MOD_3LIMB, 3)])
def test_verify_rejects_what_the_oracle_rejects(cabi, modulus, fl):
    """One flipped bit in each section of the stream (verify_z.rs:305-400, zip/tests.rs:116-146)."""
    nv = 10
    z, f, evals, point, roots, proof, cols, coeffs, q0, q1, ev = _instance(nv, modulus, fl, seed=5, small=(fl == 2))
    ctx = _ctx(cabi, z)
    R, C, d = z.num_rows, z.row_len, z.depth
    u_bytes = C * 64
    col_bytes = R * (32 + 8 + 32 * d)
    run = lambda p, e=ev, r=roots: _verify(cabi, ctx, z, fl, modulus, r, p, cols, coeffs, q0, q1, e)
    orc_rejects = lambda p, e=ev, r=roots: z.verify(f, r, point, e, p) != 0
    k, r = 3, 7
    cases = {
        "combined row u'": (_tampered(proof, 5 * 64 + 1), cabi.VERIFY_PROXIMITY_TESTING),
        "column value": (_tampered(proof, u_bytes + k * col_bytes + r * 32), cabi.VERIFY_PROXIMITY_TESTING),
        "path node": (_tampered(proof, u_bytes + k * col_bytes + R * 32 + r * (8 + 32 * d) + 8 + 32 * 4 + 3), cabi.VERIFY_MERKLE),
        "path length prefix": (_tampered(proof, u_bytes + k * col_bytes + R * 32 + r * (8 + 32 * d) + 7), cabi.VERIFY_MALFORMED),
        "evaluation row": (_tampered(proof, proof.size - 9), cabi.VERIFY_EVAL_CONSISTENCY),
    }
    for name, (p, want) in cases.items():
        rep = run(p)
        assert rep["verdict"] == want, (name, rep)
        assert orc_rejects(p), name
    rep = run(proof, e=(ev + 1) % modulus)  # a wrong claimed evaluation
    assert rep["verdict"] == cabi.VERIFY_EVAL_CONSISTENCY and orc_rejects(proof, e=(ev + 1) % modulus)
    bad_roots = roots.copy()
    bad_roots[2, 0] ^= 1  # a wrong commitment
    rep = run(proof, r=bad_roots)
    assert rep["verdict"] == cabi.VERIFY_MERKLE and rep["bad_merkle_paths"] == len(cols) and orc_rejects(proof, r=bad_roots)
    assert run(proof[:-1])["verdict"] == cabi.VERIFY_MALFORMED and orc_rejects(proof[:-1])
    assert run(proof)["verdict"] == cabi.VERIFY_ACCEPT


def test_verify_evaluation_row_consistent_but_wrong_rows(cabi):
    """A proof for another polynomial under this commitment: the q0 proximity check must catch what
    the evaluation-consistency check cannot (verify_z.rs:165-188)."""
    nv, modulus, fl = 8, BENCH_MODULUS, 4
    z, f, evals, point, roots, proof, cols, coeffs, q0, q1, ev = _instance(nv, modulus, fl, seed=2)
    ctx = _ctx(cabi, z)
    # replace the evaluation row by one that still satisfies <row, q1> = ev' for ev' = its own product
    row = proof[-z.row_len * 32:].copy().reshape(z.row_len, 32)
    row[0, 31] ^= 1  # big-endian: lowest byte of element 0
    p = proof.copy()
    p[-z.row_len * 32:] = row.reshape(-1)
    limbs = np.array([[int.from_bytes(bytes(e[8 * (3 - i): 8 * (4 - i)]), "big") for i in range(4)] for e in row], dtype=np.uint64)
    acc = 0
    for c in range(z.row_len):
        acc = orc.field_add(f, acc, orc.field_mul(f, orc.limbs_to_int(limbs[c]), orc.limbs_to_int(q1[c])))
    rep = _verify(cabi, ctx, z, fl, modulus, roots, p, cols, coeffs, q0, q1, acc)
    assert rep["verdict"] == cabi.VERIFY_PROXIMITY_Q0
    assert z.verify(f, roots, point, acc, p) != 0


@pytest.mark.parametrize("modulus,fl", FIELDS + [((1 << 256) - 2**200 - 1, 4), ((1 << 128) - 159, 2)])
def test_field_map_of_full_width_column_entries(cabi, modulus, fl):
    """FieldMap for Int<4> over the whole 256-bit range (attacker-chosen column entries), including the
    reference's reduction by 2^256 - q for moduli with the top bit set (conversion.rs:86-100)."""
    rng = np.random.default_rng(11)
    vals = rng.integers(0, 1 << 64, size=(512, 4), dtype=np.uint64)
    vals[0] = 0
    vals[1] = [0, 0, 0, 1 << 63]                       # -2^255
    vals[2] = [2**64 - 1] * 4                           # -1
    vals[3] = [2**64 - 1, 2**64 - 1, 2**64 - 1, 2**63 - 1]  # 2^255 - 1
    vals[4] = orc.int_to_limbs(modulus % (1 << 256), 4)
    vals[5] = orc.int_to_limbs((modulus - 1) % (1 << 256), 4)
    vals[6:70, 1:] = 0                                  # 64-bit magnitudes
    vals[70:130, 2:] = 0                                # 128-bit
    vals[130:190, 3] = 0                                # 192-bit
    f = orc.make_field(modulus, fl)
    ctx = cabi.ZipContext(4, orc.shuffle_perm(1, 8), orc.shuffle_perm(2, 8))
    got = ctx.field_map_int256(vals, cabi.make_field(modulus, fl))
    want = np.stack([orc.field_from_int(f, v) for v in vals])
    assert np.array_equal(got, want)
