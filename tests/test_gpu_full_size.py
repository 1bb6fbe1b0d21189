"""BASELINE.json's full sizes on the GPU.  2^24 (configs[2]): every row, tree node, root and proof byte against the
oracle's whole commit + open (seconds on the box's host cores).  2^26 (configs[3], 12 GiB of rows + trees): all 8192
roots and 64 whole opening blocks against the oracle's row-by-row pass (orc_commit_open_columns), sampled trees,
exact proof length, the verifiers on the whole stream, linearity."""
import os

import numpy as np
import pytest

import _oracle as orc

pytestmark = pytest.mark.gpu

BENCH_MODULUS = 106319353542452952636349991594949358997917625194731877894581586278529202198383


@pytest.fixture(scope="module")
def env():
    torch = pytest.importorskip("torch")
    from zinc_amd import cabi

    if cabi.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu tests must run on the MI355X box")
    return cabi, torch


def _dev_view(torch, ptr, shape, typestr):
    h = type("_H", (), {})()
    h.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (ptr, False), "version": 2}
    return torch.as_tensor(h, device="cuda")


def _check_sampled_rows(z, evals, rows_t, layers_t, roots, sample):
    for r in sample:
        rc, enc = z.encode_row(evals[r * z.row_len:(r + 1) * z.row_len])
        assert rc == 0
        assert np.array_equal(rows_t[r].cpu().numpy().view(np.uint64).reshape(z.codeword_len, 4), enc), r
        tree = orc.merkle_tree(z.depth, enc)
        got = layers_t[r].cpu().numpy().reshape(2 * z.codeword_len, 32)
        assert np.array_equal(got[: 2 * z.codeword_len - 1], tree), r  # root included at slot 2cw-2
        assert np.array_equal(roots[r], tree[-1]), r


def _equal_in_slabs(torch, dev_t, host_a, what, slab=512):
    """dev_t (device, [rows, ...]) == host_a (numpy, same shape), uploaded in slabs of rows."""
    for r in range(0, host_a.shape[0], slab):
        assert torch.equal(dev_t[r:r + slab], torch.from_numpy(host_a[r:r + slab]).cuda()), (what, r)


def test_commit_open_2pow24(env):
    """configs[2]: commit + open at 2^24 (row_len = num_rows = 4096, codeword 8192, depth 13) against the oracle's
    WHOLE commit and WHOLE proof (a few seconds on the box's host cores): every encoded row, every tree node, all 4096
    roots and every byte of the 1.74 GiB stream, for the plain two calls, the hinted / packed commit, the one call and
    the self-hinted plain commit of the unchanged prover flow (commit.rs:78-86, open_z.rs:22-40)."""
    cabi, torch = env
    nv = 24
    z = orc.Zip(nv)
    f = orc.make_field(BENCH_MODULUS, 4)
    zf = cabi.make_field(BENCH_MODULUS, 4)
    evals = orc.splitmix64(0x5A494E43, 1 << nv)
    point = orc.point_to_field(f, [1] * nv)  # as in the bench (zip_benches.rs:143)
    rows_o, layers_o, roots_o = z.commit(evals)
    proof_o, cols, coeffs = z.open(f, evals, rows_o, layers_o, point, orc.new_transcript())  # fresh PcsTranscript
    lr = z.num_rows.bit_length() - 1
    q0 = orc.build_eq_x_r(f, point[nv - lr:])
    assert proof_o.size == z.proof_len(4) == 4096 * 64 + 1000 * 4096 * (32 + 8 + 32 * 13) + 4096 * 32  # commit.rs:712-737

    ctx = cabi.ZipContext(nv, z.perm1, z.perm2)
    ctx.set_speculation(False)  # the first commit below is the plain one: everything materialised
    d_evals = torch.from_numpy(evals).cuda()
    com, roots = ctx.commit(d_evals)
    assert np.array_equal(roots, roots_o)  # all 4096
    rows_p, layers_p, _ = com.device_ptrs()
    ctx.synchronize()
    rows_t = _dev_view(torch, rows_p, (z.num_rows, z.codeword_len * 4), "<i8")
    layers_t = _dev_view(torch, layers_p, (z.num_rows, 2 * z.codeword_len, 32), "|u1")
    _equal_in_slabs(torch, rows_t, rows_o.view(np.int64).reshape(z.num_rows, z.codeword_len * 4), "rows")
    nodes = 2 * z.codeword_len - 1  # MerkleTree.layers + the root at slot 2cw - 2 (the device keeps one pad slot more)
    for r in range(0, z.num_rows, 512):
        assert torch.equal(layers_t[r:r + 512, :nodes], torch.from_numpy(layers_o[r:r + 512]).cuda()), ("layers", r)
    del rows_o, layers_o
    # determinism (commit.rs:253-283)
    com2, roots2 = ctx.commit(d_evals)
    assert np.array_equal(roots, roots2)
    com2.free()

    # the plain open of the plain commit: every byte
    proof = com.open(d_evals, coeffs, cols, q0, zf)
    assert proof.size == proof_o.size and np.array_equal(proof, proof_o)
    com.free()
    ev = z.mle_eval(f, evals, point)
    assert z.verify(f, roots, point, ev, proof, check_merkle=True) == 0
    bad = proof.copy()
    bad[proof.size // 2] ^= 1
    assert z.verify(f, roots, point, ev, bad, check_merkle=True) != 0
    del bad

    # zip_commit_open (packed + row-interleaved openings, and the natural places), zip_commit_hinted + zip_open, and
    # the plain zip_commit that hints itself with the columns of the ctx's last opening: the same 1.74 GiB, byte for
    # byte against the ORACLE's stream (poisoned buffer first)
    d_ref = torch.from_numpy(proof_o).cuda()
    for how in ("one_call", "one_call_unpacked", "hinted", "self_hinted"):
        os.environ["ZIP_HIP_PACKED"] = "0" if how == "one_call_unpacked" else "1"
        try:
            d_one = torch.full((proof.size,), 0x33, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()  # the fill runs on torch's stream, the library on its own
            if how.startswith("one_call"):
                _, roots_one, _ = ctx.commit_open(d_evals, coeffs, cols, q0, zf, out=d_one)
            else:
                ctx.set_speculation(how == "self_hinted")
                c3, roots_one = ctx.commit(d_evals, hint_cols=cols if how == "hinted" else None)
                c3.open(d_evals, coeffs, cols, q0, zf, out=d_one)
                c3.free()
            ctx.synchronize()
        finally:
            os.environ.pop("ZIP_HIP_PACKED", None)
        assert np.array_equal(roots_one, roots_o), how
        assert torch.equal(d_one, d_ref), how
        del d_one
    del d_ref

    # the device verifier (zip_verify) agrees with the oracle's on the full-size stream, and the witness
    # MLE evaluation (prover.rs:317-319) equals the oracle's
    q1 = orc.build_eq_x_r(f, point[: nv - lr])
    ev_limbs = np.array(orc.int_to_limbs(ev, 4), dtype=np.uint64)
    d_proof = torch.from_numpy(proof).cuda()
    rep = ctx.verify(roots, d_proof, coeffs, cols, q0, q1, ev_limbs, zf)
    assert rep == {"verdict": cabi.VERIFY_ACCEPT, "column": 0, "bad_merkle_paths": 0, "malformed_paths": 0}
    d_proof[proof.size // 2] ^= 1
    rep = ctx.verify(roots, d_proof, coeffs, cols, q0, q1, ev_limbs, zf)
    assert rep["verdict"] != cabi.VERIFY_ACCEPT
    assert orc.limbs_to_int(ctx.mle_eval(d_evals, q0, q1, zf)) == ev
    del d_proof

    # linearity of the proximity row: unit coefficients select a witness row (combine_rows definition)
    unit = np.zeros(z.num_rows, dtype=np.int64)
    unit[777] = 1
    u = ctx.open_testing(d_evals, unit)
    w = evals[777 * z.row_len:778 * z.row_len]
    assert np.array_equal(u[:, 0].view(np.int64), w)
    assert np.array_equal(u[:, 1:], np.repeat(((w < 0) * np.uint64(0xFFFFFFFFFFFFFFFF)).astype(np.uint64)[:, None], 7, axis=1))


def test_commit_2pow26_geometry_sampled(env):
    """configs[3] geometry (row_len 8192, codeword 16384, depth 14: raa_commit16_kernel), on a 1024-row slice =
    what one of 8 GPUs owns of a 2^26 commit."""
    cabi, torch = env
    nv, rows = 26, 1024
    zfull = orc.Zip(nv, perm1=np.zeros(1, np.uint32), perm2=np.zeros(1, np.uint32))
    assert (zfull.row_len, zfull.num_rows, zfull.codeword_len) == (8192, 8192, 16384)
    z = orc.Zip(nv, geometry=(8192, rows, 16384))
    evals = orc.splitmix64(26, rows * 8192)
    ctx = cabi.ZipContext(nv, z.perm1, z.perm2, row_begin=2048, row_count=rows)
    com, roots = ctx.commit(torch.from_numpy(evals).cuda())
    rows_p, layers_p, _ = com.device_ptrs()
    ctx.synchronize()
    rows_t = _dev_view(torch, rows_p, (rows, z.codeword_len * 4), "<i8")
    layers_t = _dev_view(torch, layers_p, (rows, 2 * z.codeword_len * 32), "|u1")
    _check_sampled_rows(z, evals, rows_t, layers_t, roots, [0, 511, 1023])


def _squeeze_open_inputs(z, f, nv):
    """What a fresh PcsTranscript yields at the start of open (open_z.rs:104,118): the proximity coefficients and
    the column indices; plus q0 for point = [1; nv] as in the bench (zip_benches.rs:143)."""
    fs = orc.new_transcript()
    coeffs = np.zeros(z.num_rows, dtype=np.int64)
    for r in range(z.num_rows):
        orc.lib().orc_tr_get_integer_challenge(orc.C.byref(fs), 1, coeffs[r:].ctypes.data_as(orc.C.POINTER(orc.C.c_uint64)))
    cols = np.array([orc.get_challenge(fs, f) % (1 << 32) % z.codeword_len for _ in range(1000)], dtype=np.uint32)
    point = orc.point_to_field(f, [1] * nv)
    lr = z.num_rows.bit_length() - 1
    return coeffs, cols, point, orc.build_eq_x_r(f, point[nv - lr:]), orc.build_eq_x_r(f, point[: nv - lr])


def _host(x):
    return x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)


def _diff_proof_columns(z, evals, proof_t, cols, pick_cols, pick_rows, u_bytes):
    """Byte diff of opened-column blocks of a device-resident proof against oracle-built rows and paths
    (open_z.rs:124-143): for the picked openings, the value and the whole path record of every picked row."""
    per_col = z.num_rows * (32 + 8 + 32 * z.depth)
    rec_bytes = 8 + 32 * z.depth
    enc, trees = {}, {}
    for r in pick_rows:
        rc, e = z.encode_row(evals[r * z.row_len:(r + 1) * z.row_len])
        assert rc == 0
        enc[r], trees[r] = e, orc.merkle_tree(z.depth, e)
    for i in pick_cols:
        c = int(cols[i])
        base = u_bytes + i * per_col
        for r in pick_rows:
            val = _host(proof_t[base + r * 32: base + (r + 1) * 32])
            assert val.tobytes() == enc[r][c].astype("<u8").tobytes(), (i, c, r)
            o = base + z.num_rows * 32 + r * rec_bytes
            rec = _host(proof_t[o: o + rec_bytes])
            assert int.from_bytes(rec[:8].tobytes(), "big") == z.depth
            assert rec[8:].tobytes() == orc.merkle_path(z.depth, trees[r][: (2 << z.depth) - 1], c).tobytes(), (i, c, r)


def test_commit_open_2pow26_full_on_one_gpu(env):
    """configs[3] at its stated size on ONE device: 2^26 coefficients = 8192 rows x 8192, codeword 16384, depth 14
    (0.5 GiB witness, 2 GiB of 16-byte row entries, 8 GiB of trees, a 3.7 GiB proof).  ALL 8192 roots and 64 WHOLE
    opening blocks (8192 values + 8192 path records each) against the oracle's row-by-row pass, sampled trees,
    determinism, exact proof length (commit.rs:712-737), and the device verifier on the whole stream (every byte of it
    is read there) -- for the plain and for the hinted (packed, row-interleaved) commit."""
    cabi, torch = env
    nv = 26
    z = orc.Zip(nv)
    assert (z.row_len, z.num_rows, z.codeword_len, z.depth) == (8192, 8192, 16384, 14)
    f = orc.make_field(BENCH_MODULUS, 4)
    zf = cabi.make_field(BENCH_MODULUS, 4)
    evals = orc.splitmix64(0x5A494E43 + 26, 1 << nv)
    coeffs, cols, point, q0, q1 = _squeeze_open_inputs(z, f, nv)
    ctx = cabi.ZipContext(nv, z.perm1, z.perm2)
    d_evals = torch.from_numpy(evals).cuda()
    sample = [0, 1, 255, 256, 4095, 4096, 8191, 5555]
    pick = np.unique(np.concatenate([[0, 1, 2, 499, 500, 997, 998, 999], np.arange(7, 1000, 17)]))[:64]
    assert pick.size == 64
    roots_o, blocks_o = z.commit_open_columns(evals, cols[pick])  # the oracle: every root, 64 whole opening blocks
    d_blocks = torch.from_numpy(blocks_o).cuda()
    per_col = z.num_rows * (32 + 8 + 32 * z.depth)

    def check_blocks(proof_t, what):
        for k, i in enumerate(pick):
            o = z.row_len * 64 + int(i) * per_col
            assert torch.equal(proof_t[o:o + per_col], d_blocks[k]), (what, int(i), int(cols[i]))

    # plain commit: everything materialised
    ctx.set_speculation(False)
    com, roots = ctx.commit(d_evals)
    assert np.array_equal(roots, roots_o)  # all 8192
    _, layers_p, _ = com.device_ptrs(rows=False)
    ctx.synchronize()
    layers_t = _dev_view(torch, layers_p, (z.num_rows, 2 * z.codeword_len * 32), "|u1")
    for r in sample:
        rc, enc = z.encode_row(evals[r * z.row_len:(r + 1) * z.row_len])
        assert rc == 0
        tree = orc.merkle_tree(z.depth, enc)
        got = layers_t[r].cpu().numpy().reshape(2 * z.codeword_len, 32)
        assert np.array_equal(got[: 2 * z.codeword_len - 1], tree), r
        assert np.array_equal(roots[r], tree[-1]), r
    proof = torch.empty(ctx.proof_len(1000, 4), dtype=torch.uint8, device="cuda")
    assert proof.numel() == z.proof_len(4) == 8192 * 64 + 1000 * 8192 * (32 + 8 + 32 * 14) + 8192 * 32
    com.open(d_evals, coeffs, cols, q0, zf, out=proof)
    ctx.synchronize()
    check_blocks(proof, "plain")
    ev = z.mle_eval(f, evals, point)
    ev_limbs = np.array(orc.int_to_limbs(ev, 4), dtype=np.uint64)
    rep = ctx.verify(roots, proof, coeffs, cols, q0, q1, ev_limbs, zf)
    assert rep == {"verdict": cabi.VERIFY_ACCEPT, "column": 0, "bad_merkle_paths": 0, "malformed_paths": 0}
    com.free()
    # the hinted commit (columns known up front): same roots (determinism, commit.rs:253-283), same proof bytes
    com2, roots2 = ctx.commit(d_evals, hint_cols=cols)
    assert np.array_equal(roots_o, roots2)
    proof2 = torch.full_like(proof, 0x33)
    torch.cuda.synchronize()
    com2.open(d_evals, coeffs, cols, q0, zf, out=proof2)
    ctx.synchronize()
    check_blocks(proof2, "hinted")
    assert torch.equal(proof, proof2)
    proof2[proof2.numel() // 2] ^= 1
    rep = ctx.verify(roots, proof2, coeffs, cols, q0, q1, ev_limbs, zf)
    assert rep["verdict"] != cabi.VERIFY_ACCEPT
    com2.free()


@pytest.mark.parametrize("hinted", [False, True, "unpacked", "one_call", "one_call_unpacked"])
def test_commit_open_2pow22_pipelined_gather_byte_diff(env, hinted, monkeypatch):
    """2^22 with the default chunking (two chunks: the gather of the first runs beside the hashing of the second):
    sampled proof blocks byte for byte against oracle-built rows and paths -- plain commit, hinted commit (packed
    openings, and the natural places with ZIP_HIP_PACKED=0) and zip_commit_open (the same two)."""
    cabi, torch = env
    nv = 22
    z = orc.Zip(nv)
    f = orc.make_field(BENCH_MODULUS, 4)
    zf = cabi.make_field(BENCH_MODULUS, 4)
    evals = orc.splitmix64(0x5A494E43 + 22, 1 << nv)
    coeffs, cols, point, q0, q1 = _squeeze_open_inputs(z, f, nv)
    ctx = cabi.ZipContext(nv, z.perm1, z.perm2)
    d_evals = torch.from_numpy(evals).cuda()
    proof = torch.full((ctx.proof_len(1000, 4),), 0x55, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()  # the fill runs on torch's stream, the library on its own
    monkeypatch.setenv("ZIP_HIP_PACKED", "0" if hinted in ("unpacked", "one_call_unpacked") else "1")
    if hinted in ("one_call", "one_call_unpacked"):
        _, _, com = ctx.commit_open(d_evals, coeffs, cols, q0, zf, out=proof, want_roots=False, keep=True)
    else:
        com, _ = ctx.commit(d_evals, want_roots=False, hint_cols=cols if hinted else None)  # asynchronous
        com.open(d_evals, coeffs, cols, q0, zf, out=proof)
    ctx.synchronize()
    _diff_proof_columns(z, evals, proof, cols, [0, 1, 333, 500, 666, 998, 999], [0, 1, 255, 256, 1023, 1024, 2047],
                        z.row_len * 64)
    _, _, roots = com.download(rows=False, layers=False)
    ev = z.mle_eval(f, evals, point)
    rep = ctx.verify(roots, proof, coeffs, cols, q0, q1, np.array(orc.int_to_limbs(ev, 4), dtype=np.uint64), zf)
    assert rep == {"verdict": cabi.VERIFY_ACCEPT, "column": 0, "bad_merkle_paths": 0, "malformed_paths": 0}
    com.free()


def test_handle_opened_after_its_chunk_counters_were_recycled(env):
    """The chunk counters of a pipelined commit come from a ring of 64 pre-zeroed slots per ctx.  A handle that is opened
    only after 70 further commits finds its counters gone (new ring epoch): its openings then wait for the whole commit
    instead of polling them -- same bytes as an immediate open."""
    cabi, torch = env
    nv = 22
    z = orc.Zip(nv)
    f = orc.make_field(BENCH_MODULUS, 4)
    zf = cabi.make_field(BENCH_MODULUS, 4)
    evals = orc.splitmix64(0x5A494E43 + 7, 1 << nv)
    coeffs, cols, point, q0, q1 = _squeeze_open_inputs(z, f, nv)
    ctx = cabi.ZipContext(nv, z.perm1, z.perm2)
    d_evals = torch.from_numpy(evals).cuda()
    first = torch.empty(ctx.proof_len(1000, 4), dtype=torch.uint8, device="cuda")
    late = torch.full_like(first, 0x77)
    torch.cuda.synchronize()
    kept, _ = ctx.commit(d_evals, want_roots=False, hint_cols=cols)
    for _ in range(70):
        com, _ = ctx.commit(d_evals, want_roots=False, hint_cols=cols)
        com.free()
    now, _ = ctx.commit(d_evals, want_roots=False, hint_cols=cols)
    now.open(d_evals, coeffs, cols, q0, zf, out=first)
    kept.open(d_evals, coeffs, cols, q0, zf, out=late)
    ctx.synchronize()
    assert torch.equal(first, late)
    _diff_proof_columns(z, evals, late, cols, [0, 500, 999], [0, 1023, 1024, 2047], z.row_len * 64)
    kept.free()
    now.free()
