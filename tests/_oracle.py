"""ctypes wrapper around oracle/_build/libzip_oracle.so (the CPU restatement).

Test infrastructure: imported only by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB_PATH = os.path.join(ORACLE_DIR, "_build", "libzip_oracle.so")

ORC_MAX_FL = 8
ORC_OK, ORC_ERR_OVERFLOW, ORC_ERR_PARAM, ORC_ERR_PROOF, ORC_ERR_TRANSCRIPT, ORC_ERR_ALLOC = 0, -1, -2, -3, -4, -5


class Keccak(C.Structure):
    _fields_ = [("st", C.c_uint64 * 25), ("buf", C.c_uint8 * 136), ("buflen", C.c_uint32)]


class Field(C.Structure):
    _fields_ = [
        ("fl", C.c_uint32),
        ("modulus", C.c_uint64 * ORC_MAX_FL),
        ("r", C.c_uint64 * ORC_MAX_FL),
        ("r2", C.c_uint64 * ORC_MAX_FL),
        ("inv", C.c_uint64),
        ("has_spare_bit", C.c_int),
    ]


class Params(C.Structure):
    _fields_ = [
        ("num_vars", C.c_uint32), ("row_len", C.c_uint32), ("num_rows", C.c_uint32),
        ("codeword_len", C.c_uint32), ("rep", C.c_uint32), ("depth", C.c_uint32),
        ("n_limbs", C.c_uint32), ("k_limbs", C.c_uint32), ("m_limbs", C.c_uint32),
        ("perm1", C.POINTER(C.c_uint32)), ("perm2", C.POINTER(C.c_uint32)),
        ("num_column_opening", C.c_uint32), ("num_proximity_testing", C.c_uint32),
    ]


def _stale():
    return not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(
        os.path.getmtime(os.path.join(ORACLE_DIR, f)) for f in ("zip_oracle.c", "zip_oracle.h"))


def build(force=False):
    """Runs `make -C oracle` when the library is missing or older than its sources.  This forks and execs, which a
    process that has initialised the GPU must never do on the GPU boxes: call it FIRST (pytest_configure, the top of
    bench.main(), smoke() and the tools do), never lazily."""
    if force or _stale():
        # Under `rocprofv3 ... -- python3 tool.py` the profiler's preloaded library has initialised the GPU before this
        # line runs: then even this early call must not fork + exec.
        preload = os.environ.get("LD_PRELOAD", "")
        if "rocprof" in preload or any(k.startswith(("ROCPROFILER_", "ROCPROF_", "ROCP_")) for k in os.environ):
            raise RuntimeError(f"{_LIB_PATH} is missing or stale and this process runs under a profiler: "
                               "run `make -C oracle` first (a GPU-initialised process must not fork + exec)")
        subprocess.run(["make", "-B", "-C", ORACLE_DIR], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib():
    """The loaded oracle.  Never builds: a missing library is an error that names the fix."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(f"{_LIB_PATH} is missing: run `make -C oracle` (or _oracle.build()) before any GPU call")
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_tr_get_u64.restype = C.c_uint64
        _lib.orc_proof_len.restype = C.c_size_t
    return _lib


def int_to_limbs(v, n):
    """Two's-complement little-endian limbs of a Python int."""
    v &= (1 << (64 * n)) - 1
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]


def limbs_to_int(limbs, signed=False):
    v = 0
    for i, l in enumerate(limbs):
        v |= int(l) << (64 * i)
    if signed and limbs is not None and (int(limbs[-1]) >> 63):
        v -= 1 << (64 * len(limbs))
    return v


def _u64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def _u32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def _u8p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def blake3(msg: bytes) -> bytes:
    out = (C.c_uint8 * 32)()
    rc = lib().orc_blake3_hash_block(msg, C.c_size_t(len(msg)), out)
    assert rc == 0
    return bytes(out)


def keccak256(data: bytes, domain=1) -> bytes:
    k = Keccak()
    lib().orc_keccak_init(C.byref(k))
    lib().orc_keccak_update(C.byref(k), data, C.c_size_t(len(data)))
    out = (C.c_uint8 * 32)()
    lib().orc_keccak_finalize_copy(C.byref(k), C.c_uint8(domain), out)
    return bytes(out)


def new_transcript() -> Keccak:
    k = Keccak()
    lib().orc_keccak_init(C.byref(k))
    return k


def absorb(k: Keccak, data: bytes):
    lib().orc_keccak_update(C.byref(k), data, C.c_size_t(len(data)))


def make_field(modulus: int, fl: int) -> Field:
    f = Field()
    m = (C.c_uint64 * fl)(*int_to_limbs(modulus, fl))
    rc = lib().orc_field_new(C.byref(f), fl, m)
    assert rc == 0, rc
    return f


def field_elems(values, fl):
    """Python ints (already Montgomery / raw residues) -> uint64 array [n, fl]."""
    a = np.zeros((len(values), fl), dtype=np.uint64)
    for i, v in enumerate(values):
        a[i] = int_to_limbs(v, fl)
    return a


def field_from_i64(f: Field, v: int) -> int:
    out = (C.c_uint64 * ORC_MAX_FL)()
    lib().orc_field_from_i64(C.byref(f), C.c_int64(v), out)
    return limbs_to_int(out[: f.fl])


def field_from_int(f: Field, limbs) -> np.ndarray:
    """FieldMap for an n-limb two's-complement Int (conversion.rs:86-100): Montgomery limbs."""
    v = np.ascontiguousarray(limbs, dtype=np.uint64)
    out = (C.c_uint64 * ORC_MAX_FL)()
    lib().orc_field_from_int(C.byref(f), _u64p(v), v.size, out)
    return np.array(out[: f.fl], dtype=np.uint64)


def field_mul(f: Field, a: int, b: int) -> int:
    aa = (C.c_uint64 * ORC_MAX_FL)(*int_to_limbs(a, f.fl))
    bb = (C.c_uint64 * ORC_MAX_FL)(*int_to_limbs(b, f.fl))
    lib().orc_field_mul(C.byref(f), aa, bb)
    return limbs_to_int(aa[: f.fl])


def field_add(f: Field, a: int, b: int) -> int:
    aa = (C.c_uint64 * ORC_MAX_FL)(*int_to_limbs(a, f.fl))
    bb = (C.c_uint64 * ORC_MAX_FL)(*int_to_limbs(b, f.fl))
    lib().orc_field_add(C.byref(f), aa, bb)
    return limbs_to_int(aa[: f.fl])


def get_challenge(k: Keccak, f: Field) -> int:
    out = (C.c_uint64 * ORC_MAX_FL)()
    lib().orc_tr_get_challenge(C.byref(k), C.byref(f), out)
    return limbs_to_int(out[: f.fl])


def absorb_field(k: Keccak, f: Field, value_mont: int):
    v = (C.c_uint64 * ORC_MAX_FL)(*int_to_limbs(value_mont, f.fl))
    lib().orc_tr_absorb_field(C.byref(k), C.byref(f), v)


def field_from_u128(f: Field, value: int) -> int:
    out = (C.c_uint64 * ORC_MAX_FL)()
    lib().orc_field_from_u128(C.byref(f), C.c_uint64(value & (2**64 - 1)), C.c_uint64(value >> 64), out)
    return limbs_to_int(out[: f.fl])


def sumcheck_prove_product(f: Field, mles: np.ndarray, degree: int, transcript: Keccak):
    """MLSumcheck::prove_as_subprotocol with comb_fn = product.  mles: [K, 2^nv, fl] Montgomery limbs (copied).
    Returns (msgs [nv, degree+1, fl], randomness [nv, fl])."""
    m = np.ascontiguousarray(mles, dtype=np.uint64).copy()
    K, n, fl = m.shape
    nv = n.bit_length() - 1
    msgs = np.zeros((nv, degree + 1, fl), dtype=np.uint64)
    rand = np.zeros((nv, fl), dtype=np.uint64)
    rc = lib().orc_sumcheck_prove_product(C.byref(f), _u64p(m), K, nv, degree, C.byref(transcript), _u64p(msgs), _u64p(rand))
    assert rc == 0, rc
    return msgs, rand


def sumcheck_prove(f: Field, mles: np.ndarray, degree: int, term_masks, coeffs_mont, transcript: Keccak):
    """prove_as_subprotocol with comb = (sum_t coeffs[t] * prod_{j in mask t} vals[j]) * vals[-1]
    (sumcheck_polynomial_comb_fn_1, zinc/utils.rs:77-94).  coeffs_mont: Python ints (Montgomery)."""
    m = np.ascontiguousarray(mles, dtype=np.uint64).copy()
    K, n, fl = m.shape
    nv = n.bit_length() - 1
    msgs = np.zeros((nv, degree + 1, fl), dtype=np.uint64)
    rand = np.zeros((nv, fl), dtype=np.uint64)
    masks = np.ascontiguousarray(term_masks, dtype=np.uint32)
    cf = field_elems(list(coeffs_mont), fl)
    rc = lib().orc_sumcheck_prove(C.byref(f), _u64p(m), K, nv, degree, masks.size, _u32p(masks), _u64p(cf),
                                  C.byref(transcript), _u64p(msgs), _u64p(rand))
    assert rc == 0, rc
    return msgs, rand


class Sparse(C.Structure):
    _fields_ = [("n_rows", C.c_uint32), ("n_cols", C.c_uint32), ("row_ptr", C.POINTER(C.c_uint32)),
                ("col_idx", C.POINTER(C.c_uint32)), ("values", C.POINTER(C.c_int64))]


class CcsStruct(C.Structure):
    _fields_ = [("m", C.c_uint32), ("n", C.c_uint32), ("s", C.c_uint32), ("s_prime", C.c_uint32),
                ("t", C.c_uint32), ("q", C.c_uint32), ("d", C.c_uint32), ("M", C.POINTER(Sparse)),
                ("S_masks", C.POINTER(C.c_uint32)), ("c", C.POINTER(C.c_int64))]


class Ccs:
    """orc_ccs over a tests/_ccs.CcsInstance (keeps the numpy arrays alive)."""

    def __init__(self, inst):
        self.inst = inst
        self._mats = (Sparse * inst.t)()
        for k, m in enumerate(inst.matrices):
            self._mats[k] = Sparse(m.n_rows, m.n_cols, _u32p(m.row_ptr), _u32p(m.col_idx),
                                   m.values.ctypes.data_as(C.POINTER(C.c_int64)))
        self._masks = inst.masks
        self._c = np.array(inst.c, dtype=np.int64)
        self.struct = CcsStruct(inst.m, inst.n, inst.s, inst.s_prime, inst.t, inst.q, inst.d, self._mats,
                                _u32p(self._masks), self._c.ctypes.data_as(C.POINTER(C.c_int64)))

    def _z(self):
        z = self.inst.z
        return z.ctypes.data_as(C.POINTER(C.c_int64)), len(z)

    def mz(self, f: Field) -> np.ndarray:
        out = np.zeros((self.inst.t, self.inst.m, f.fl), dtype=np.uint64)
        zp, zl = self._z()
        rc = lib().orc_ccs_mz(C.byref(f), C.byref(self.struct), zp, zl, _u64p(out))
        assert rc == 0, rc
        return out

    def second_table(self, f: Field, eq_rx: np.ndarray, gamma: np.ndarray) -> np.ndarray:
        out = np.zeros((self.inst.m, f.fl), dtype=np.uint64)
        rc = lib().orc_ccs_second_table(C.byref(f), C.byref(self.struct), _u64p(np.ascontiguousarray(eq_rx)),
                                        _u64p(np.ascontiguousarray(gamma)), _u64p(out))
        assert rc == 0, rc
        return out

    def spartan_prove(self, f: Field, transcript: Keccak):
        """SpartanProver::prove.  Returns dict(msgs1, r_x, msgs2, r_y, V_s) of uint64 limb arrays."""
        i, fl = self.inst, f.fl
        out = dict(msgs1=np.zeros((i.s, i.d + 2, fl), dtype=np.uint64), r_x=np.zeros((i.s, fl), dtype=np.uint64),
                   msgs2=np.zeros((i.s, 3, fl), dtype=np.uint64), r_y=np.zeros((i.s, fl), dtype=np.uint64),
                   V_s=np.zeros((i.t, fl), dtype=np.uint64))
        zp, zl = self._z()
        rc = lib().orc_spartan_prove(C.byref(f), C.byref(self.struct), zp, zl, C.byref(transcript), _u64p(out["msgs1"]),
                                     _u64p(out["r_x"]), _u64p(out["msgs2"]), _u64p(out["r_y"]), _u64p(out["V_s"]))
        assert rc == 0, rc
        return out

    def spartan_verify(self, f: Field, proof, transcript: Keccak):
        """SpartanVerifier::verify.  Returns (rc, dict(r_x, r_y, e_y, gamma))."""
        i, fl = self.inst, f.fl
        pts = dict(r_x=np.zeros((i.s, fl), dtype=np.uint64), r_y=np.zeros((i.s_prime, fl), dtype=np.uint64),
                   e_y=np.zeros(fl, dtype=np.uint64), gamma=np.zeros(fl, dtype=np.uint64))
        rc = lib().orc_spartan_verify(C.byref(f), C.byref(self.struct), _u64p(np.ascontiguousarray(proof["msgs1"])),
                                      _u64p(np.ascontiguousarray(proof["msgs2"])), _u64p(np.ascontiguousarray(proof["V_s"])),
                                      C.byref(transcript), _u64p(pts["r_x"]), _u64p(pts["r_y"]), _u64p(pts["e_y"]),
                                      _u64p(pts["gamma"]))
        return rc, pts

    def eval_matrices(self, f: Field, r_x, r_y) -> np.ndarray:
        out = np.zeros((self.inst.t, f.fl), dtype=np.uint64)
        rc = lib().orc_ccs_eval_matrices(C.byref(f), C.byref(self.struct), _u64p(np.ascontiguousarray(r_x)),
                                         _u64p(np.ascontiguousarray(r_y)), _u64p(out))
        assert rc == 0, rc
        return out

    def final_check(self, f: Field, pts, v: np.ndarray) -> int:
        return lib().orc_spartan_final_check(C.byref(f), C.byref(self.struct), _u64p(pts["r_x"]), _u64p(pts["r_y"]),
                                             _u64p(pts["gamma"]), _u64p(np.ascontiguousarray(v)), _u64p(pts["e_y"]))


def field_inv(f: Field, a_mont: int) -> int:
    a = (C.c_uint64 * ORC_MAX_FL)(*int_to_limbs(a_mont, f.fl))
    out = (C.c_uint64 * ORC_MAX_FL)()
    lib().orc_field_inv(C.byref(f), a, out)
    return limbs_to_int(out[: f.fl])


def interpolate_uni_poly(f: Field, p_mont, x_mont: int) -> int:
    p = field_elems(list(p_mont), f.fl)
    x = (C.c_uint64 * ORC_MAX_FL)(*int_to_limbs(x_mont, f.fl))
    out = (C.c_uint64 * ORC_MAX_FL)()
    lib().orc_interpolate_uni_poly(C.byref(f), _u64p(p), len(p_mont), x, out)
    return limbs_to_int(out[: f.fl])


def sumcheck_verify(f: Field, nvars, degree, claimed_mont: int, msgs: np.ndarray, transcript: Keccak):
    """verify_as_subprotocol.  Returns (rc, point [nvars, fl], expected_evaluation int (Montgomery))."""
    point = np.zeros((max(nvars, 1), f.fl), dtype=np.uint64)
    exp = (C.c_uint64 * ORC_MAX_FL)()
    cl = (C.c_uint64 * ORC_MAX_FL)(*int_to_limbs(claimed_mont, f.fl))
    rc = lib().orc_sumcheck_verify(C.byref(f), nvars, degree, cl, _u64p(np.ascontiguousarray(msgs)), C.byref(transcript),
                                   _u64p(point), exp)
    return rc, point[:nvars], limbs_to_int(exp[: f.fl])


def sumcheck_prove_products(f: Field, mles: np.ndarray, degree: int, masks, coeffs_mont, transcript: Keccak):
    """prove_as_subprotocol with rand_poly_comb_fn (sumcheck/utils.rs:67-78): sum_p coeffs[p] * prod_{j in masks[p]} vals[j]."""
    m = np.ascontiguousarray(mles, dtype=np.uint64).copy()
    K, n, fl = m.shape
    nv = n.bit_length() - 1
    msgs = np.zeros((nv, degree + 1, fl), dtype=np.uint64)
    rand = np.zeros((nv, fl), dtype=np.uint64)
    mk = np.ascontiguousarray(masks, dtype=np.uint32)
    cf = field_elems(list(coeffs_mont), fl)
    rc = lib().orc_sumcheck_prove_products(C.byref(f), _u64p(m), K, nv, degree, mk.size, _u32p(mk), _u64p(cf),
                                           C.byref(transcript), _u64p(msgs), _u64p(rand))
    assert rc == 0, rc
    return msgs, rand


def build_eq_x_r(f: Field, r: np.ndarray) -> np.ndarray:
    nvars = r.shape[0]
    out = np.zeros((1 << nvars, f.fl), dtype=np.uint64)
    rc = lib().orc_build_eq_x_r(C.byref(f), _u64p(np.ascontiguousarray(r)), nvars, _u64p(out))
    assert rc == 0
    return out


def kat_chacha12_block(key_words, counter) -> bytes:
    key = (C.c_uint32 * 8)(*key_words)
    out = (C.c_uint32 * 16)()
    lib().orc_kat_chacha12_block(key, C.c_uint64(counter), out)
    return b"".join(int(w).to_bytes(4, "little") for w in out)


def kat_stdrng_u64(seed_bytes, n):
    seed = (C.c_uint8 * 32)(*seed_bytes)
    out = (C.c_uint64 * n)()
    lib().orc_kat_stdrng_from_seed_u64(seed, n, out)
    return [int(x) for x in out]


def kat_pcg32(state, stream, n):
    out = (C.c_uint32 * n)()
    lib().orc_kat_pcg32(C.c_uint64(state), C.c_uint64(stream), n, out)
    return [int(x) for x in out]


def kat_shuffle_pcg32(state, stream, length):
    perm = np.zeros(length, dtype=np.uint32)
    lib().orc_kat_shuffle_pcg32(C.c_uint64(state), C.c_uint64(stream), length, _u32p(perm))
    return perm


def kat_seed_from_u64(state):
    out = (C.c_uint32 * 8)()
    lib().orc_kat_seed_from_u64(C.c_uint64(state), out)
    return [int(x) for x in out]


def shuffle_perm(seed: int, length: int) -> np.ndarray:
    perm = np.zeros(length, dtype=np.uint32)
    lib().orc_shuffle_seeded_perm(C.c_uint64(seed), length, _u32p(perm))
    return perm


class Zip:
    """Geometry + permutations for one polynomial size (RaaCode::new + setup)."""

    def __init__(self, num_vars, perm1=None, perm2=None, seeds=(1, 2), n_limbs=1, rep=2, geometry=None):
        self.p = Params()
        probe = Params()
        lib().orc_params_init(C.byref(probe), num_vars, n_limbs, rep, None, None)
        cw = geometry[2] if geometry else probe.codeword_len
        self.perm1 = np.ascontiguousarray(perm1 if perm1 is not None else shuffle_perm(seeds[0], cw), dtype=np.uint32)
        self.perm2 = np.ascontiguousarray(perm2 if perm2 is not None else shuffle_perm(seeds[1], cw), dtype=np.uint32)
        rc = lib().orc_params_init(C.byref(self.p), num_vars, n_limbs, rep, _u32p(self.perm1), _u32p(self.perm2))
        if rc != 0:
            raise ValueError(f"orc_params_init failed: {rc}")
        if geometry:  # (row_len, num_rows, codeword_len): any consistent shape, e.g. a row shard
            self.p.row_len, self.p.num_rows, self.p.codeword_len = geometry
            self.p.depth = (geometry[2] - 1).bit_length() if geometry[2] > 1 else 0
        for name in ("num_vars", "row_len", "num_rows", "codeword_len", "rep", "depth", "n_limbs", "k_limbs", "m_limbs"):
            setattr(self, name, getattr(self.p, name))
        self.tree_hashes = (2 << self.depth) - 1

    def encode_row(self, row, in_limbs=None, out_limbs=None):
        in_limbs = in_limbs or self.n_limbs
        out_limbs = out_limbs or self.k_limbs
        row = np.ascontiguousarray(row).view(np.uint64).reshape(-1)
        out = np.zeros((self.codeword_len, out_limbs), dtype=np.uint64)
        rc = lib().orc_raa_encode_row(_u64p(row), in_limbs, self.row_len, self.rep, _u32p(self.perm1),
                                      _u32p(self.perm2), _u64p(out), out_limbs)
        return rc, out

    def commit(self, evals, merkle=True):
        evals = np.ascontiguousarray(evals, dtype=np.int64)
        assert evals.size == self.num_rows * self.row_len
        rows = np.zeros((self.num_rows * self.codeword_len, self.k_limbs), dtype=np.uint64)
        layers = roots = None
        if merkle:
            layers = np.zeros((self.num_rows, self.tree_hashes, 32), dtype=np.uint8)
            roots = np.zeros((self.num_rows, 32), dtype=np.uint8)
        rc = lib().orc_commit(C.byref(self.p), _u64p(evals.view(np.uint64)), _u64p(rows),
                              _u8p(layers) if merkle else None, _u8p(roots) if merkle else None)
        assert rc == 0, rc
        return rows, layers, roots

    def commit_open_columns(self, evals, cols):
        """Every root and the whole opening block of each picked column, row by row (no rows / layers kept): the
        checker of the sizes whose full commit does not fit a test (2^26: 12 GiB)."""
        evals = np.ascontiguousarray(evals, dtype=np.int64)
        cols = np.ascontiguousarray(cols, dtype=np.uint32)
        assert evals.size == self.num_rows * self.row_len
        roots = np.zeros((self.num_rows, 32), dtype=np.uint8)
        per_col = self.num_rows * (8 * self.k_limbs + 8 + 32 * self.depth)
        blocks = np.zeros((cols.size, per_col), dtype=np.uint8)
        rc = lib().orc_commit_open_columns(C.byref(self.p), _u64p(evals.view(np.uint64)), _u32p(cols), int(cols.size),
                                           _u8p(roots), _u8p(blocks))
        assert rc == 0, rc
        return roots, blocks

    def proof_len(self, fl):
        return lib().orc_proof_len(C.byref(self.p), fl)

    def combine_rows_int(self, coeffs, evals):
        coeffs = np.ascontiguousarray(coeffs, dtype=np.int64)
        evals = np.ascontiguousarray(evals, dtype=np.int64)
        out = np.zeros((self.row_len, self.m_limbs), dtype=np.uint64)
        rc = lib().orc_combine_rows_int(_u64p(coeffs.view(np.uint64)), 1, _u64p(evals.view(np.uint64)), 1,
                                        self.num_rows, self.row_len, _u64p(out), self.m_limbs)
        return rc, out

    def combine_rows_field(self, f: Field, q0, evals):
        evals = np.ascontiguousarray(evals, dtype=np.int64)
        q0 = np.ascontiguousarray(q0, dtype=np.uint64)
        out = np.zeros((self.row_len, f.fl), dtype=np.uint64)
        lib().orc_combine_rows_field(C.byref(f), _u64p(q0), _u64p(evals.view(np.uint64)), 1,
                                     self.num_rows, self.row_len, _u64p(out))
        return out

    def open(self, f: Field, evals, rows, layers, point, fs: Keccak):
        evals = np.ascontiguousarray(evals, dtype=np.int64)
        point = np.ascontiguousarray(point, dtype=np.uint64)
        cap = self.proof_len(f.fl)
        proof = np.zeros(cap, dtype=np.uint8)
        plen = C.c_size_t(0)
        cols = np.zeros(self.p.num_column_opening, dtype=np.uint32)
        coeffs = np.zeros(self.num_rows, dtype=np.int64)
        rc = lib().orc_open(C.byref(self.p), C.byref(f), _u64p(evals.view(np.uint64)), _u64p(rows),
                            _u8p(layers), _u64p(point), C.byref(fs), _u8p(proof), C.c_size_t(cap),
                            C.byref(plen), _u32p(cols), _u64p(coeffs.view(np.uint64)))
        assert rc == 0, rc
        assert plen.value == cap, (plen.value, cap)
        return proof, cols, coeffs

    def verify(self, f: Field, roots, point, eval_mont, proof, fs: Keccak = None, check_merkle=True):
        fs = fs or new_transcript()
        point = np.ascontiguousarray(point, dtype=np.uint64)
        ev = np.array(int_to_limbs(eval_mont, f.fl), dtype=np.uint64)
        proof = np.ascontiguousarray(proof, dtype=np.uint8)
        roots = np.ascontiguousarray(roots, dtype=np.uint8)
        return lib().orc_verify(C.byref(self.p), C.byref(f), _u8p(roots), _u64p(point), _u64p(ev),
                                C.byref(fs), _u8p(proof), C.c_size_t(proof.size), int(check_merkle))

    def mle_eval(self, f: Field, evals, point) -> int:
        evals = np.ascontiguousarray(evals, dtype=np.int64)
        point = np.ascontiguousarray(point, dtype=np.uint64)
        out = (C.c_uint64 * ORC_MAX_FL)()
        lib().orc_mle_eval_field(C.byref(f), _u64p(evals.view(np.uint64)), 1, self.num_vars, _u64p(point), out)
        return limbs_to_int(out[: f.fl])


def merkle_tree(depth, leaves: np.ndarray):
    leaves = np.ascontiguousarray(leaves, dtype=np.uint64)
    leaf_limbs = leaves.shape[1]
    layers = np.zeros(((2 << depth) - 1, 32), dtype=np.uint8)
    rc = lib().orc_merkle_tree(depth, _u64p(leaves), leaf_limbs, _u8p(layers))
    assert rc == 0
    return layers


def merkle_path(depth, layers, leaf):
    path = np.zeros((depth, 32), dtype=np.uint8)
    lib().orc_merkle_path(depth, _u8p(np.ascontiguousarray(layers)), leaf, _u8p(path))
    return path


def merkle_verify(depth, path, root, leaf_limbs_arr, leaf_index):
    leaf = np.ascontiguousarray(leaf_limbs_arr, dtype=np.uint64)
    return lib().orc_merkle_verify(depth, _u8p(np.ascontiguousarray(path)), _u8p(np.ascontiguousarray(root)),
                                   _u64p(leaf), leaf.size, leaf_index)


def point_to_field(f: Field, ints):
    """`vec![..i64..].map_to_field(config)` -> uint64 [n, fl] Montgomery limbs."""
    return field_elems([field_from_i64(f, int(v)) for v in ints], f.fl)


def splitmix64(seed, n):
    """Synthetic witness generator shared by tests and bench."""
    out = np.empty(n, dtype=np.uint64)
    x = np.uint64(seed)
    idx = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    out[:] = z
    return out.view(np.int64)
