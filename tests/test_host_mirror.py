"""The C++ host mirror (zinc_amd/host, libzinc_zip.so) against the oracle: Fiat-Shamir transcript,
field constants, map_to_field, eq tensor, permutation expansion, RaaCode geometry.  CPU only; the
device-backed parts (commit/open) are in test_gpu_host_mirror.py."""
import numpy as np
import pytest

import _oracle as orc
from zinc_amd import cabi, pcs

BENCH_MODULUS = 106319353542452952636349991594949358997917625194731877894581586278529202198383
TEST_MODULUS_2 = 57316695564490278656402085503
STARK = 3618502788666131213697322783095070105623107215331596699973092056135872020481


def test_libraries_load_and_export_every_declared_symbol():
    import re, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for header, lib, symbols in (("zip_hip.h", cabi.lib(), cabi.EXPORTED_SYMBOLS), ("zinc_zip_host.h", pcs.lib(), pcs.EXPORTED_SYMBOLS)):
        text = open(os.path.join(root, "include", header)).read()
        declared = set(re.findall(r"\b(zi(?:p|nc)_[a-z0-9_]+)\s*\(", text))
        declared -= {"zip_ctx", "zip_commitment"}
        assert declared, header
        assert declared == set(symbols), (header, declared ^ set(symbols))
        for s in declared:
            assert hasattr(lib, s), s
    assert cabi.lib().zip_abi_version() == 3


def test_no_gpu_means_loud_failure_not_fallback():
    if cabi.device_count() > 0:
        pytest.skip("a GPU is present")
    z = orc.Zip(8)
    with pytest.raises(cabi.ZipError) as e:
        cabi.ZipContext(8, z.perm1, z.perm2)
    assert e.value.code == cabi.ZIP_ERR_NO_DEVICE
    with pytest.raises(pcs.DeviceError):
        pcs.MultilinearZip.setup(1 << 8, pcs.RaaCode(1 << 8))


def test_geometry_and_param_checks_need_no_gpu():
    z = orc.Zip(8)
    bad = z.perm1.copy()
    bad[3] = bad[4]
    with pytest.raises(cabi.ZipError) as e:
        cabi.ZipContext(8, bad, z.perm2)
    assert e.value.code == cabi.ZIP_ERR_INVALID_PARAM
    with pytest.raises(cabi.ZipError) as e:
        cabi.ZipContext(8, z.perm1, z.perm2, k_limbs=2)
    assert e.value.code == cabi.ZIP_ERR_UNSUPPORTED
    for nv in range(0, 27):
        assert cabi.geometry(nv) == (orc.Zip(nv, perm1=np.zeros(1, np.uint32), perm2=np.zeros(1, np.uint32)).row_len,) + \
            tuple(getattr(orc.Zip(nv, perm1=np.zeros(1, np.uint32), perm2=np.zeros(1, np.uint32)), k) for k in ("num_rows", "codeword_len"))


def test_keccak_transcript_matches_oracle_and_reference_kat():
    f = pcs.FieldConfig(STARK, 4)
    t = pcs.KeccakTranscript()
    t.absorb(b"This is a test string!")
    ch = orc.limbs_to_int(t.get_challenge(f))
    assert ch == 693058076479703886486101269644733982722902192016595549603371045888466087870 * (1 << 256) % STARK  # transcript.rs:214-234
    for modulus, fl in ((BENCH_MODULUS, 4), (TEST_MODULUS_2, 2), ((1 << 61) - 1, 1), ((1 << 190) - 11 * (1 << 64) - 59, 3)):
        fo, fp = orc.make_field(modulus, fl), pcs.FieldConfig(modulus, fl)
        ko, kp = orc.new_transcript(), pcs.KeccakTranscript()
        orc.absorb(ko, b"zinc"), kp.absorb(b"zinc")
        for _ in range(3):
            assert orc.lib().orc_tr_get_u64(orc.C.byref(ko)) == kp.get_u64()
            assert orc.get_challenge(ko, fo) == orc.limbs_to_int(kp.get_challenge(fp))
        got = kp.get_integer_challenges(5)
        exp = np.zeros(5, np.int64)
        for i in range(5):
            orc.lib().orc_tr_get_integer_challenge(orc.C.byref(ko), 1, exp[i:].ctypes.data_as(orc.C.POINTER(orc.C.c_uint64)))
        assert np.array_equal(got, exp)


@pytest.mark.parametrize("modulus,fl", [(BENCH_MODULUS, 4), (TEST_MODULUS_2, 2), ((1 << 256) - 189, 4)])
def test_field_constants_map_to_field_and_eq(modulus, fl):
    f = pcs.FieldConfig(modulus, fl)
    R = 1 << (64 * fl)
    r, r2, inv = f.constants()
    assert orc.limbs_to_int(r) == R % modulus and orc.limbs_to_int(r2) == R * R % modulus
    assert (inv * modulus + 1) % (1 << 64) == 0
    a, b = 0x1234567890ABCDEF1234567 % modulus, (modulus - 5)
    assert orc.limbs_to_int(f.mul(orc.field_elems([a], fl)[0], orc.field_elems([b], fl)[0])) == a * b * pow(R, -1, modulus) % modulus
    vals = np.array([0, 1, -1, 5, -7, 2**63 - 1, -(2**63), 188, 189, 190], dtype=np.int64)
    fo = orc.make_field(modulus, fl)
    got = f.map_to_field(vals)
    for i, v in enumerate(vals):
        assert orc.limbs_to_int(got[i]) == orc.field_from_i64(fo, int(v))
    pt = f.map_to_field(np.array([3, -4, 17, 1, 0], dtype=np.int64))
    assert np.array_equal(f.build_eq_x_r(pt), orc.build_eq_x_r(fo, pt))


def test_shuffle_seeded_three_implementations_agree():
    """C++ host mirror == Python mirror == oracle restatement.  The oracle's pieces are pinned by published vectors
    (tests/test_oracle_kats.py, tests/golden/rand_vectors.json), rand_core's seed_from_u64 included; agreement on whole
    permutations for many (seed, length) extends those pins to the other two restatements."""
    from zinc_amd.perm import shuffle_seeded_perm as py_perm
    for seed, n in ((1, 512), (2, 8192), (12345, 10), (2**63 + 11, 64), (7, 1), (9, 2), (3, 16384), (0, 13)):
        a = pcs.shuffle_seeded_perm(seed, n)
        assert sorted(a) == list(range(n))
        assert np.array_equal(a, orc.shuffle_perm(seed, n))
        assert np.array_equal(a, py_perm(seed, n))


def test_python_permutation_module_against_the_published_vectors():
    """zinc_amd/perm.py (what bench.py and the tests feed the C ABI) against the same vectors as the oracle."""
    import json
    import os

    from zinc_amd import perm

    with open(os.path.join(os.path.dirname(__file__), "golden", "rand_vectors.json")) as fh:
        v = json.load(fh)
    c = v["chacha12_zero_key_block0"]
    rng = perm.ChaCha12Rng(key_words=c["key_words"])
    assert b"".join(rng.next_u32().to_bytes(4, "little") for _ in range(16)).hex() == c["keystream_hex"]
    sc = v["stdrng_construction"]
    words = [int.from_bytes(bytes(sc["seed_bytes"][4 * i:4 * i + 4]), "little") for i in range(8)]
    rng = perm.ChaCha12Rng(key_words=words)
    lo, hi = rng.next_u32(), rng.next_u32()
    assert lo | (hi << 32) == int(sc["next_u64"])
    pd = v["pcg32_demo"]
    g = perm.Pcg32(pd["state"], pd["stream"])
    assert [f"{g.next_u32():08x}" for _ in range(6)] == pd["outputs_hex"]
    sv = v["shuffle_value_stability"]
    got = perm.shuffle_perm_with(perm.Pcg32(sv["pcg32_state"], int(sv["pcg32_stream"])), sv["len"])
    assert [int(x) for x in got] == sv["shuffled"]
    # seed_from_u64: rand_pcg's construction vector, through this module's own expansion
    ps = v["pcg32_seed_from_u64"]
    assert perm.Pcg32.from_seed(ps["from_seed_bytes"]).next_u64() == int(ps["from_seed_next_u64"])
    seed16 = b"".join(w.to_bytes(4, "little") for w in perm.seed_from_u64_words(ps["seed_from_u64_seed"], 4))
    assert perm.Pcg32.from_seed(seed16).next_u64() == int(ps["seed_from_u64_next_u64"])
    assert perm.ChaCha12Rng(seed_u64=5).key == perm.seed_from_u64_words(5, 8)


def test_cpp_mirror_seed_expansion_against_rand_pcgs_vector():
    """The C++ host mirror's seed_from_u64 (zinc_kat_seed_from_u64) against the same published vector."""
    import json
    import os

    from zinc_amd import perm

    with open(os.path.join(os.path.dirname(__file__), "golden", "rand_vectors.json")) as fh:
        ps = json.load(fh)["pcg32_seed_from_u64"]
    words = pcs.kat_seed_from_u64(ps["seed_from_u64_seed"], 4)
    seed16 = b"".join(w.to_bytes(4, "little") for w in words)
    assert perm.Pcg32.from_seed(seed16).next_u64() == int(ps["seed_from_u64_next_u64"])
    for seed in (1, 2, 0xDEADBEEF, 2**64 - 1):
        assert pcs.kat_seed_from_u64(seed, 8) == perm.seed_from_u64_words(seed, 8) == orc.kat_seed_from_u64(seed)


def test_raa_code_new_geometry_and_seeds():
    """code_raa.rs:35-86: row_len, MockTranscript seeds (1, 2), Keccak-derived seeds, width assert."""
    for nv in (0, 3, 8, 9, 16, 20, 24, 26):
        c = pcs.RaaCode(1 << nv)
        assert (c.row_len, c.perm_1_seed, c.perm_2_seed) == (orc.Zip(nv, perm1=np.zeros(1, np.uint32), perm2=np.zeros(1, np.uint32)).row_len, 1, 2)
        assert c.s.num_column_opening == 1000 and c.s.num_proximity_testing == 1 and c.s.repetition_factor == 2
    t, ko = pcs.KeccakTranscript(), orc.new_transcript()
    t.absorb(b"seed"), orc.absorb(ko, b"seed")
    c = pcs.RaaCode(1 << 10, t)
    assert c.perm_1_seed == orc.lib().orc_tr_get_u64(orc.C.byref(ko))
    assert c.perm_2_seed == orc.lib().orc_tr_get_u64(orc.C.byref(ko))
