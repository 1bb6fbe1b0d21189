"""Pins the CPU oracle (oracle/zip_oracle.c) against every golden vector and
known-answer test available for the path, and against Python
big-integer models of the arithmetic.  CPU only."""
import hashlib
import json
import os
import random

import numpy as np
import pytest

import _oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BENCH_MODULUS = 106319353542452952636349991594949358997917625194731877894581586278529202198383  # benches/zip_benches.rs:253
TEST_MODULUS_2 = 57316695564490278656402085503  # src/zip/tests.rs:63 (FIELD_LIMBS = 2)


# ---------------------------------------------------------------- BLAKE3 / Keccak
def test_blake3_empty_kat():
    assert orc.blake3(b"").hex() == "af1349b9f5f9a1a6a0404dea36dcc9499bcb25c9adc112b7cc9a93cae41f3262"


def test_blake3_golden_vectors():
    with open(os.path.join(GOLDEN, "blake3_vectors.json")) as f:
        gold = json.load(f)
    assert "1.8.2" in gold["source"]  # the version the reference pins (Cargo.toml:30)
    assert len(gold["vectors"]) >= 200
    for v in gold["vectors"]:
        assert orc.blake3(bytes.fromhex(v["msg"])).hex() == v["hash"], v["kind"]


def test_blake3_rejects_multi_block():
    out = (orc.C.c_uint8 * 32)()
    assert orc.lib().orc_blake3_hash_block(b"\0" * 65, orc.C.c_size_t(65), out) == orc.ORC_ERR_PARAM


def test_keccak_permutation_matches_hashlib_sha3():
    rng = random.Random(1)
    for n in (0, 1, 31, 135, 136, 137, 271, 272, 273, 1000):
        data = bytes(rng.getrandbits(8) for _ in range(n))
        assert orc.keccak256(data, domain=6) == hashlib.sha3_256(data).digest()


def test_keccak256_kats():
    assert orc.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert orc.keccak256(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"


def test_keccak_transcript_reference_kat():
    """src/transcript.rs:214-234 (test_keccak_transcript)."""
    p = 3618502788666131213697322783095070105623107215331596699973092056135872020481
    f = orc.make_field(p, 4)
    k = orc.new_transcript()
    orc.absorb(k, b"This is a test string!")
    ch_mont = orc.get_challenge(k, f)
    expected = 693058076479703886486101269644733982722902192016595549603371045888466087870
    R = 1 << 256
    assert ch_mont == expected * R % p
    # independent model of transcript.rs:72-133
    h = orc.keccak256(b"This is a test string!")
    lo, hi = int.from_bytes(h[:16], "big"), int.from_bytes(h[16:], "big")
    keep = (p.bit_length() - 1) - 128
    assert (lo + (1 << 128) * (hi & ((1 << keep) - 1))) % p == expected


def test_transcript_integer_challenge_model():
    """transcript.rs:40-55,142-155 against a Python model built on the pinned hash."""
    k = orc.new_transcript()
    orc.absorb(k, b"zinc")
    state = b"zinc"
    for _ in range(5):
        got = orc.lib().orc_tr_get_u64(orc.C.byref(k))
        ch = orc.keccak256(state + (0).to_bytes(4, "big"))[:8]
        assert got == int.from_bytes(ch, "little")
        state += b"\x12" + ch + b"\x34"


# ------------------------------------------------------------------------ field
def test_montgomery_reference_kat():
    """src/field/config.rs:338-345."""
    f = orc.make_field(695962179703626800597079116051991347, 4)
    assert orc.field_mul(f, 423024736033, 246308734) == 504579159360957705315139767875358506


@pytest.mark.parametrize("modulus,fl", [(BENCH_MODULUS, 4), (TEST_MODULUS_2, 2), ((1 << 256) - 189, 4),
                                        (695962179703626800597079116051991347, 4), ((1 << 61) - 1, 1)])
def test_field_config_and_arithmetic_vs_python(modulus, fl):
    f = orc.make_field(modulus, fl)
    R = 1 << (64 * fl)
    assert orc.limbs_to_int(f.r[:fl]) == R % modulus
    assert orc.limbs_to_int(f.r2[:fl]) == R * R % modulus
    assert (f.inv * modulus + 1) % (1 << 64) == 0
    assert f.has_spare_bit == int(modulus < (1 << (64 * fl - 1)))
    rng = random.Random(7)
    rinv = pow(R, -1, modulus)
    for _ in range(200):
        a, b = rng.randrange(modulus), rng.randrange(modulus)
        assert orc.field_mul(f, a, b) == a * b * rinv % modulus
        assert orc.field_add(f, a, b) == (a + b) % modulus
    for a in (0, 1, modulus - 1):
        for b in (0, 1, modulus - 1):
            assert orc.field_mul(f, a, b) == a * b * rinv % modulus
            assert orc.field_add(f, a, b) == (a + b) % modulus


@pytest.mark.parametrize("modulus,fl", [(BENCH_MODULUS, 4), (TEST_MODULUS_2, 2)])
def test_map_to_field_i64(modulus, fl):
    """conversion.rs:86-100 + field.rs:536-568: phi(w) = w * R mod q (canonical)."""
    f = orc.make_field(modulus, fl)
    R = 1 << (64 * fl)
    rng = random.Random(3)
    vals = [0, 1, -1, 7, -7, 2**63 - 1, -(2**63)] + [rng.randrange(-(2**63), 2**63) for _ in range(100)]
    for w in vals:
        assert orc.field_from_i64(f, w) == (w * R) % modulus


def test_map_to_field_signed_modulus_quirk():
    """A modulus with its top bit set is read as a NEGATIVE Int<N> inside `%=`
    (field.rs:550-557 with F::I = Int<N>, field.rs:280), so |w| is first reduced
    modulo 2^(64N) - q.  Harmless for the bench modulus (2^256 - q > 2^64) but
    observable for q = 2^256 - 189 (benches/spartan_benches.rs:134-137)."""
    q = (1 << 256) - 189
    f = orc.make_field(q, 4)
    R = 1 << 256
    for w in (5, 188, 189, 190, 1000, -1000, 2**63 - 1):
        mag = abs(w) % 189
        expect = (mag * R) % q
        if w < 0 and expect:
            expect = q - expect
        assert orc.field_from_i64(f, w) == expect


def test_build_eq_x_r_vs_model():
    """sumcheck/utils.rs:117-177: variable 0 <-> least significant index bit."""
    q = BENCH_MODULUS
    f = orc.make_field(q, 4)
    R = 1 << 256
    rng = random.Random(11)
    for nv in (1, 2, 5):
        r = [rng.randrange(q) for _ in range(nv)]
        got = orc.build_eq_x_r(f, orc.field_elems([x * R % q for x in r], 4))
        for i in range(1 << nv):
            e = 1
            for t in range(nv):
                e = e * (r[t] if (i >> t) & 1 else (1 - r[t])) % q
            assert orc.limbs_to_int(got[i]) == e * R % q


# -------------------------------------------------------------------- RAA / ints
def _identity_zip(num_vars):
    probe = orc.Zip(num_vars)
    ident = np.arange(probe.codeword_len, dtype=np.uint32)
    return orc.Zip(num_vars, perm1=ident, perm2=ident)


def test_accumulate_reference_kats():
    """src/zip/code_raa.rs:223-244: accumulate == inclusive prefix sum.  With
    identity permutations and rep=1 a single-row encode is accumulate(accumulate(x))."""
    for inp in ([1, 2, 3, 4], [5, 0, 2, 0], [-1, 5, -10, 2]):
        z = orc.Zip(4, perm1=np.arange(4, dtype=np.uint32), perm2=np.arange(4, dtype=np.uint32), rep=1)
        assert z.row_len == 4 and z.codeword_len == 4
        rc, out = z.encode_row(np.array(inp, dtype=np.int64))
        assert rc == 0
        got = [orc.limbs_to_int(o, signed=True) for o in out]
        assert got == list(np.cumsum(np.cumsum(inp)))


def test_repeat_reference_kat():
    """src/zip/code_raa.rs:199-221: [10,20] x3 -> [10,20,10,20,10,20].  The second
    scan is undone by differencing so `repeat` is observed in isolation."""
    ident = np.arange(6, dtype=np.uint32)
    z = orc.Zip(2, perm1=ident, perm2=ident, rep=3)  # row_len = 2
    assert z.row_len == 2 and z.codeword_len == 6
    rc, out = z.encode_row(np.array([10, 20], dtype=np.int64))
    got = np.array([orc.limbs_to_int(o, signed=True) for o in out])
    assert list(np.diff(np.diff(got, prepend=0), prepend=0)) == [10, 20, 10, 20, 10, 20]


def test_encode_matches_python_model_and_bounds():
    rng = np.random.default_rng(5)
    z = orc.Zip(10)
    row = rng.integers(-(2**63), 2**63, size=z.row_len, dtype=np.int64)
    rc, out = z.encode_row(row)
    assert rc == 0
    t = [int(row[j % z.row_len]) for j in range(z.codeword_len)]
    t = [t[p] for p in z.perm1]
    t = list(np.cumsum(np.array(t, dtype=object)))
    t = [t[p] for p in z.perm2]
    t = list(np.cumsum(np.array(t, dtype=object)))
    assert [orc.limbs_to_int(o, signed=True) for o in out] == t
    bound = 64 + z.num_vars + 2  # code_raa.rs:53-72
    assert all(abs(v) < (1 << bound) for v in t)


def test_encoding_preserves_linearity():
    """src/zip/code_raa.rs:279-299."""
    rng = np.random.default_rng(9)
    z = orc.Zip(8)
    a = rng.integers(-(2**40), 2**40, size=z.row_len, dtype=np.int64)
    b = rng.integers(-(2**40), 2**40, size=z.row_len, dtype=np.int64)
    ea = [orc.limbs_to_int(o, True) for o in z.encode_row(a)[1]]
    eb = [orc.limbs_to_int(o, True) for o in z.encode_row(b)[1]]
    eab = [orc.limbs_to_int(o, True) for o in z.encode_row(a + b)[1]]
    assert eab == [x + y for x, y in zip(ea, eb)]


def test_expand_sign_extension_kats():
    """src/zip/utils.rs:164-234: expand == sign extension (-1 stays -1)."""
    ident = np.arange(2, dtype=np.uint32)
    z = orc.Zip(0, perm1=ident, perm2=ident)  # row_len 1, cw 2
    rc, out = z.encode_row(np.array([-1], dtype=np.int64), out_limbs=8)
    assert rc == 0
    assert list(out[0]) == [0xFFFFFFFFFFFFFFFF] * 8
    assert orc.limbs_to_int(out[1], True) == -3  # scan(scan([-1, -1]))
    rc, out = z.encode_row(np.array([123], dtype=np.int64), out_limbs=3)
    assert list(out[0]) == [123, 0, 0]


def test_wide_overflow_is_reported():
    """crypto-bigint checked_add panics (int.rs:122-134); the oracle returns an error."""
    ident = np.arange(4, dtype=np.uint32)
    z = orc.Zip(4, perm1=ident, perm2=ident, rep=1)
    rc, _ = z.encode_row(np.array([2**63 - 1] * 4, dtype=np.int64), out_limbs=1)
    assert rc == orc.ORC_ERR_OVERFLOW


def test_combine_rows_reference_kats():
    """src/zip/pcs/utils.rs:301-337."""
    for coeffs, evals, expect in (
        ([1, 2], [3, 4, 5, 6], [3 + 2 * 5, 4 + 2 * 6]),
        ([3, 4], [2, 4, 6, 8], [3 * 2 + 4 * 6, 3 * 4 + 4 * 8]),
        ([1000, -500], [2000, -3000, 4000, -5000], [1000 * 2000 - 500 * 4000, 1000 * -3000 + 500 * 5000]),
    ):
        z = orc.Zip(2)  # row_len 2, num_rows 2
        assert (z.row_len, z.num_rows) == (2, 2)
        rc, out = z.combine_rows_int(coeffs, evals)
        assert rc == 0
        assert [orc.limbs_to_int(o, True) for o in out] == expect


def test_combine_rows_full_range():
    rng = np.random.default_rng(21)
    z = orc.Zip(8)
    coeffs = rng.integers(-(2**63), 2**63, size=z.num_rows, dtype=np.int64)
    evals = rng.integers(-(2**63), 2**63, size=1 << 8, dtype=np.int64)
    rc, out = z.combine_rows_int(coeffs, evals)
    assert rc == 0
    m = evals.reshape(z.num_rows, z.row_len)
    for c in range(z.row_len):
        assert orc.limbs_to_int(out[c], True) == sum(int(coeffs[r]) * int(m[r, c]) for r in range(z.num_rows))


# ----------------------------------------------------------------------- Merkle
def test_merkle_layout_and_all_proofs():
    """src/zip/pcs/utils.rs:340-363 (Int<3> leaves, every leaf proves) + layout A.3."""
    rng = np.random.default_rng(2)
    depth = 6
    leaves = rng.integers(0, 2**64, size=(1 << depth, 3), dtype=np.uint64)
    layers = orc.merkle_tree(depth, leaves)
    assert layers.shape[0] == (2 << depth) - 1
    for i in range(1 << depth):  # leaf hashing: limbs LE order, each limb big-endian
        msg = b"".join(int(l).to_bytes(8, "big") for l in leaves[i])
        assert bytes(layers[i]) == orc.blake3(msg)
    off = 0
    for d in range(depth, 0, -1):
        w = 1 << d
        for i in range(w // 2):
            assert bytes(layers[off + w + i]) == orc.blake3(bytes(layers[off + 2 * i]) + bytes(layers[off + 2 * i + 1]))
        off += w
    root = layers[-1]
    for i in range(1 << depth):
        path = orc.merkle_path(depth, layers, i)
        assert orc.merkle_verify(depth, path, root, leaves[i], i) == 0
        assert orc.merkle_verify(depth, path, root, leaves[i], i ^ 1) != 0
    bad = leaves[5].copy()
    bad[0] ^= 1
    assert orc.merkle_verify(depth, orc.merkle_path(depth, layers, 5), root, bad, 5) != 0


# ------------------------------------------------------------- shuffle: pinned piece by piece
def _rand_vectors():
    import json
    import os

    with open(os.path.join(os.path.dirname(__file__), "golden", "rand_vectors.json")) as fh:
        return json.load(fh)


def test_shuffle_is_permutation_deterministic_and_seed_dependent():
    """src/zip/code_raa.rs:246-276 (the reference's own tests of shuffle_seeded pin no permutation)."""
    a = orc.shuffle_perm(12345, 512)
    assert sorted(a) == list(range(512))
    assert np.array_equal(a, orc.shuffle_perm(12345, 512))
    assert not np.array_equal(a, orc.shuffle_perm(54321, 512))
    assert not np.array_equal(a, np.arange(512))
    assert list(orc.shuffle_perm(1, 1)) == [0]


def test_chacha12_block_matches_the_published_vector():
    v = _rand_vectors()["chacha12_zero_key_block0"]
    assert orc.kat_chacha12_block(v["key_words"], v["counter"]).hex() == v["keystream_hex"]


def test_stdrng_is_chacha12_with_rands_own_construction_vector():
    v = _rand_vectors()["stdrng_construction"]
    assert orc.kat_stdrng_u64(v["seed_bytes"], 1) == [int(v["next_u64"])]


def test_pcg32_matches_oneills_demo_outputs():
    v = _rand_vectors()["pcg32_demo"]
    assert [f"{x:08x}" for x in orc.kat_pcg32(v["state"], v["stream"], 6)] == v["outputs_hex"]


def test_shuffle_algorithm_matches_rands_value_stability_vector():
    """IncreasingUniform + Canon's-method random_range + the swap order of SliceRandom::shuffle, driven by the
    generator rand's own test uses (Pcg32): the expected permutation is rand 0.9's value_stability_slice."""
    v = _rand_vectors()["shuffle_value_stability"]
    got = orc.kat_shuffle_pcg32(v["pcg32_state"], int(v["pcg32_stream"]), v["len"])
    # shuffle of [0..13): shuffled[j] = x[perm[j]] with x the identity
    assert list(got) == v["shuffled"]


def _pcg32_from_seed_next_u64(seed16):
    """rand_pcg::Lcg64Xsh32::from_seed(seed).next_u64(), in Python integers: state = LE u64 of bytes 0..8, increment =
    LE u64 of bytes 8..16 | 1, `state += inc; step()`; next_u64 = two XSH-RR outputs, low word first."""
    M, MUL = (1 << 64) - 1, 6364136223846793005
    inc = int.from_bytes(bytes(seed16[8:16]), "little") | 1
    state = ((int.from_bytes(bytes(seed16[:8]), "little") + inc) * MUL + inc) & M
    words = []
    for _ in range(2):
        xs, rot = (((state >> 18) ^ state) >> 27) & 0xFFFFFFFF, state >> 59
        words.append(((xs >> rot) | (xs << ((32 - rot) & 31))) & 0xFFFFFFFF)
        state = (state * MUL + inc) & M
    return words[0] | (words[1] << 32)


def test_seed_from_u64_matches_rand_pcgs_construction_vector():
    """rand_core::SeedableRng::seed_from_u64 -- the link zip/utils.rs:139-142 goes through and the last one that had no
    external anchor -- held to rand_pcg's own test_lcg64xsh32_construction: `Lcg64Xsh32::seed_from_u64(0).next_u64()`.
    The provided method is generic: Pcg32 takes the first 16 bytes of the expansion StdRng takes 32 of."""
    v = _rand_vectors()["pcg32_seed_from_u64"]
    assert _pcg32_from_seed_next_u64(v["from_seed_bytes"]) == int(v["from_seed_next_u64"])  # pins from_seed / next_u64 above
    words = orc.kat_seed_from_u64(v["seed_from_u64_seed"])  # the oracle's expansion (8 words; Pcg32 uses 4)
    seed16 = b"".join(w.to_bytes(4, "little") for w in words[:4])
    assert _pcg32_from_seed_next_u64(seed16) == int(v["seed_from_u64_next_u64"])


def test_seed_from_u64_link_is_the_documented_pcg32_expansion():
    """The oracle's seed expansion IS the procedure its header describes (PCG32 steps with rand_core's increment, output
    of the NEW state), written a second time in Python, for more seeds than the published vector covers;
    `orc_shuffle_seeded_perm` = that seed -> ChaCha12 -> the pinned shuffle."""
    M = (1 << 64) - 1
    for seed in (0, 1, 2, 0xDEADBEEF, M):
        state, want = seed, []
        for _ in range(8):
            state = (state * 6364136223846793005 + 11634580027462260723) & M
            xs = (((state >> 18) ^ state) >> 27) & 0xFFFFFFFF
            rot = state >> 59
            want.append(((xs >> rot) | (xs << ((32 - rot) & 31))) & 0xFFFFFFFF)
        assert orc.kat_seed_from_u64(seed) == want
        # and the ChaCha12 stream under that key starts the shuffle: first block reproduced through the block KAT
        blk = orc.kat_chacha12_block(want, 0)
        assert orc.kat_stdrng_u64(b"".join(w.to_bytes(4, "little") for w in want), 1)[0] == int.from_bytes(blk[:8], "little")
