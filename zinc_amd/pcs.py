"""ctypes binding of libzinc_zip.so (include/zinc_zip_host.h): the C++ mirror of the host side of
zinc::zip -- KeccakTranscript, RaaCode::new, MultilinearZip::{setup, commit, open}, PcsTranscript --
driving the HIP library.  Names and argument meaning follow the reference (src/zip/pcs/*.rs)."""
import ctypes as C
import os

import numpy as np

from . import cabi

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libzinc_zip.so")
OK, ERR_INVALID_PARAM, ERR_PANIC, ERR_DEVICE, ERR_NULL, ERR_INVALID_OPEN, ERR_SPARTAN = 0, -1, -2, -3, -4, -5, -6

EXPORTED_SYMBOLS = (
    "zinc_last_error", "zinc_transcript_new", "zinc_transcript_free", "zinc_transcript_absorb",
    "zinc_transcript_get_u64", "zinc_transcript_get_integer_challenges", "zinc_transcript_get_challenge",
    "zinc_field_constants", "zinc_field_mul", "zinc_map_to_field_i64", "zinc_build_eq_x_r",
    "zinc_shuffle_seeded_perm", "zinc_kat_seed_from_u64", "zinc_raa_code_new", "zinc_zip_setup", "zinc_zip_params_free",
    "zinc_zip_params_geometry", "zinc_zip_commit", "zinc_zip_data_free", "zinc_pcs_transcript_new",
    "zinc_pcs_transcript_free", "zinc_pcs_transcript_len", "zinc_pcs_transcript_copy",
    "zinc_pcs_transcript_probe", "zinc_zip_open", "zinc_pcs_transcript_from_proof", "zinc_pcs_transcript_position",
    "zinc_zip_verify", "zinc_zip_evaluate", "zinc_commit_z_mle_and_prove_evaluation", "zinc_zip_proof_len",
    "zinc_zip_proof_num_roots", "zinc_zip_proof_read", "zinc_zip_proof_free", "zinc_zip_release_cached_contexts", "zinc_sumcheck_prove_product", "zinc_sumcheck_prove_ccs", "zinc_zip_data_download", "zinc_zip_data_upload", "zinc_merkle_tree_new",
    "zinc_prover_prove", "zinc_prover_prepare", "zinc_prepared_ccs_free", "zinc_verifier_verify",
    "zinc_sumcheck_prove_products", "zinc_sumcheck_verify",
)


class InvalidPcsParam(ValueError):
    """zip::Error::InvalidPcsParam"""


class InvalidPcsOpen(ValueError):
    """zip::Error::InvalidPcsOpen / a transcript read error: the proof is rejected."""


class SpartanError(ValueError):
    """SpartanError / SumCheckError of the verifier (src/zinc/errors.rs)"""


class ReferencePanic(AssertionError):
    """A place where the reference panics (assert! / expect)."""


class DeviceError(RuntimeError):
    pass


class RaaCodeStruct(C.Structure):
    _fields_ = [("row_len", C.c_uint32), ("repetition_factor", C.c_uint32), ("num_column_opening", C.c_uint32),
                ("num_proximity_testing", C.c_uint32), ("perm_1_seed", C.c_uint64), ("perm_2_seed", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        cabi.lib()  # loads torch's HIP runtime first, then libzip_hip.so
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -m zinc_amd.build`")
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.zinc_last_error.restype = C.c_char_p
        L.zinc_transcript_new.restype = vp
        L.zinc_transcript_free.argtypes = [vp]
        L.zinc_transcript_absorb.argtypes = [vp, C.c_char_p, C.c_size_t]
        L.zinc_transcript_get_u64.argtypes = [vp]
        L.zinc_transcript_get_u64.restype = C.c_uint64
        L.zinc_transcript_get_integer_challenges.argtypes = [vp, C.c_size_t, vp]
        L.zinc_transcript_get_challenge.argtypes = [vp, vp, C.c_uint32, vp]
        L.zinc_field_constants.argtypes = [vp, C.c_uint32, vp, vp, vp]
        L.zinc_field_mul.argtypes = [vp, C.c_uint32, vp, vp, vp]
        L.zinc_map_to_field_i64.argtypes = [vp, C.c_uint32, vp, C.c_size_t, vp]
        L.zinc_build_eq_x_r.argtypes = [vp, C.c_uint32, vp, C.c_uint32, vp]
        L.zinc_shuffle_seeded_perm.argtypes = [C.c_uint64, C.c_uint32, vp]
        L.zinc_shuffle_seeded_perm.restype = None
        L.zinc_kat_seed_from_u64.argtypes = [C.c_uint64, vp, C.c_uint32]
        L.zinc_kat_seed_from_u64.restype = None
        L.zinc_raa_code_new.argtypes = [C.c_uint64, vp, C.POINTER(RaaCodeStruct)]
        L.zinc_zip_setup.argtypes = [C.c_uint64, C.POINTER(RaaCodeStruct), C.c_int32, C.POINTER(vp)]
        L.zinc_zip_params_free.argtypes = [vp]
        L.zinc_zip_params_geometry.argtypes = [vp] + [C.POINTER(C.c_uint32)] * 4
        L.zinc_zip_commit.argtypes = [vp, vp, C.c_size_t, C.c_uint32, C.c_int32, vp, C.POINTER(vp)]
        L.zinc_zip_data_free.argtypes = [vp]
        L.zinc_pcs_transcript_new.restype = vp
        L.zinc_pcs_transcript_free.argtypes = [vp]
        L.zinc_pcs_transcript_len.argtypes = [vp]
        L.zinc_pcs_transcript_len.restype = C.c_size_t
        L.zinc_pcs_transcript_copy.argtypes = [vp, vp]
        L.zinc_pcs_transcript_probe.argtypes = [vp]
        L.zinc_pcs_transcript_probe.restype = C.c_uint64
        L.zinc_zip_open.argtypes = [vp, vp, C.c_size_t, C.c_uint32, vp, vp, C.c_size_t, vp, C.c_uint32, vp]
        L.zinc_pcs_transcript_from_proof.argtypes = [vp, C.c_size_t]
        L.zinc_pcs_transcript_from_proof.restype = vp
        L.zinc_pcs_transcript_position.argtypes = [vp]
        L.zinc_pcs_transcript_position.restype = C.c_size_t
        L.zinc_zip_verify.argtypes = [vp, vp, vp, C.c_size_t, vp, vp, C.c_uint32, vp]
        L.zinc_zip_evaluate.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, vp, C.c_uint32, vp]
        L.zinc_commit_z_mle_and_prove_evaluation.argtypes = [vp, C.c_size_t, vp, C.c_size_t, vp, vp, C.c_uint32,
                                                             C.c_int32, C.POINTER(vp)]
        L.zinc_zip_proof_len.argtypes = [vp]
        L.zinc_zip_proof_len.restype = C.c_size_t
        L.zinc_zip_proof_num_roots.argtypes = [vp]
        L.zinc_zip_proof_num_roots.restype = C.c_size_t
        L.zinc_zip_proof_read.argtypes = [vp, vp, vp, vp]
        L.zinc_zip_proof_read.restype = None
        L.zinc_zip_proof_free.argtypes = [vp]
        L.zinc_sumcheck_prove_ccs.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp,
                                              C.c_uint32, C.c_int32, vp, vp]
        L.zinc_zip_data_download.argtypes = [vp, vp, vp, vp]
        L.zinc_zip_data_upload.argtypes = [vp, vp, vp, vp, C.POINTER(vp)]
        L.zinc_merkle_tree_new.argtypes = [C.c_uint32, vp, C.c_size_t, C.c_uint32, C.c_int32, vp]
        L.zinc_sumcheck_prove_product.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, C.c_uint32, C.c_int32, vp, vp]
        L.zinc_prover_prove.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp, C.c_size_t, vp,
                                        C.c_size_t, vp, vp, C.c_uint32, C.c_int32, vp, C.c_int32, vp, vp, vp, vp,
                                        C.POINTER(vp)]
        L.zinc_prover_prepare.argtypes = [vp, C.c_uint32, C.c_uint32, vp, C.c_uint32, C.c_int32, C.POINTER(vp)]
        L.zinc_prepared_ccs_free.argtypes = [vp]
        L.zinc_prepared_ccs_free.restype = None
        L.zinc_sumcheck_verify.argtypes = [vp, C.c_uint32, C.c_uint32, vp, vp, C.c_uint32, C.c_uint32, vp, C.c_uint32, vp, vp]
        L.zinc_sumcheck_prove_products.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp,
                                                   C.c_uint32, C.c_int32, vp, vp]
        L.zinc_verifier_verify.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp, vp, C.c_uint32,
                                           C.c_int32, vp, vp, vp, vp, C.c_int32, vp, C.c_size_t, vp, vp, C.c_size_t,
                                           vp, vp, vp]
        _lib = L
    return _lib


def _check(rc):
    if rc == OK:
        return
    msg = lib().zinc_last_error().decode()
    if rc == ERR_INVALID_PARAM:
        raise InvalidPcsParam(msg)
    if rc == ERR_INVALID_OPEN:
        raise InvalidPcsOpen(msg)
    if rc == ERR_PANIC:
        raise ReferencePanic(msg)
    if rc == ERR_SPARTAN:
        raise SpartanError(msg)
    raise DeviceError(msg)


def _limbs(value: int, n: int) -> np.ndarray:
    return np.array([(value >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)], dtype=np.uint64)


class FieldConfig:
    """FieldConfig::new(modulus) with FIELD_LIMBS = limbs."""

    def __init__(self, modulus: int, limbs: int):
        self.modulus, self.limbs = modulus, limbs
        self._m = _limbs(modulus, limbs)

    def constants(self):
        r, r2 = np.zeros(self.limbs, np.uint64), np.zeros(self.limbs, np.uint64)
        inv = C.c_uint64()
        _check(lib().zinc_field_constants(self._m.ctypes.data, self.limbs, r.ctypes.data, r2.ctypes.data, C.byref(inv)))
        return r, r2, inv.value

    def mul(self, a: np.ndarray, b: np.ndarray) -> np.ndarray:
        out = np.zeros(self.limbs, np.uint64)
        a, b = np.ascontiguousarray(a, np.uint64), np.ascontiguousarray(b, np.uint64)
        _check(lib().zinc_field_mul(self._m.ctypes.data, self.limbs, a.ctypes.data, b.ctypes.data, out.ctypes.data))
        return out

    def map_to_field(self, values) -> np.ndarray:
        v = np.ascontiguousarray(values, dtype=np.int64)
        out = np.zeros((v.size, self.limbs), np.uint64)
        _check(lib().zinc_map_to_field_i64(self._m.ctypes.data, self.limbs, v.ctypes.data, v.size, out.ctypes.data))
        return out

    def build_eq_x_r(self, r: np.ndarray) -> np.ndarray:
        r = np.ascontiguousarray(r, np.uint64)
        out = np.zeros((1 << r.shape[0], self.limbs), np.uint64)
        _check(lib().zinc_build_eq_x_r(self._m.ctypes.data, self.limbs, r.ctypes.data, r.shape[0], out.ctypes.data))
        return out


class KeccakTranscript:
    def __init__(self):
        self._h = lib().zinc_transcript_new()

    def __del__(self):
        try:  # (at interpreter shutdown the module globals may already be gone)
            if getattr(self, "_h", None):
                lib().zinc_transcript_free(self._h)
                self._h = None
        except Exception:
            pass

    def absorb(self, data: bytes):
        lib().zinc_transcript_absorb(self._h, data, len(data))

    def get_u64(self) -> int:
        return lib().zinc_transcript_get_u64(self._h)

    def get_integer_challenges(self, n: int) -> np.ndarray:
        out = np.zeros(n, np.int64)
        lib().zinc_transcript_get_integer_challenges(self._h, n, out.ctypes.data)
        return out

    def get_challenge(self, field: FieldConfig) -> np.ndarray:
        out = np.zeros(field.limbs, np.uint64)
        _check(lib().zinc_transcript_get_challenge(self._h, field._m.ctypes.data, field.limbs, out.ctypes.data))
        return out


def kat_seed_from_u64(seed: int, n_words: int):
    out = np.zeros(n_words, np.uint32)
    lib().zinc_kat_seed_from_u64(seed, out.ctypes.data, n_words)
    return [int(x) for x in out]


def shuffle_seeded_perm(seed: int, length: int) -> np.ndarray:
    out = np.zeros(length, np.uint32)
    lib().zinc_shuffle_seeded_perm(seed, length, out.ctypes.data)
    return out


class RaaCode:
    """RaaCode::new(&DefaultLinearCodeSpec, poly_size, transcript); transcript=None -> MockTranscript."""

    def __init__(self, poly_size: int, transcript: KeccakTranscript = None):
        self.s = RaaCodeStruct()
        _check(lib().zinc_raa_code_new(poly_size, transcript._h if transcript else None, C.byref(self.s)))
        self.poly_size = poly_size

    row_len = property(lambda self: self.s.row_len)
    perm_1_seed = property(lambda self: self.s.perm_1_seed)
    perm_2_seed = property(lambda self: self.s.perm_2_seed)

    def codeword_len(self):
        return self.s.row_len * self.s.repetition_factor


class PcsTranscript:
    def __init__(self, _handle=None):
        self._h = _handle if _handle is not None else lib().zinc_pcs_transcript_new()

    @classmethod
    def from_proof(cls, proof) -> "PcsTranscript":
        p = np.ascontiguousarray(proof, dtype=np.uint8)
        return cls(lib().zinc_pcs_transcript_from_proof(p.ctypes.data, p.size))

    def position(self) -> int:
        return lib().zinc_pcs_transcript_position(self._h)

    def __del__(self):
        try:  # (at interpreter shutdown the module globals may already be gone)
            if getattr(self, "_h", None):
                lib().zinc_pcs_transcript_free(self._h)
                self._h = None
        except Exception:
            pass

    def into_proof(self) -> np.ndarray:
        out = np.zeros(lib().zinc_pcs_transcript_len(self._h), np.uint8)
        if out.size:
            lib().zinc_pcs_transcript_copy(self._h, out.ctypes.data)
        return out

    def probe(self) -> int:
        return lib().zinc_pcs_transcript_probe(self._h)


class MerkleTree:
    """MerkleTree {root, depth, layers} (src/zip/pcs/utils.rs:67-85)."""

    def __init__(self, depth: int, layers: np.ndarray, root: np.ndarray):
        self.depth, self.layers, self.root = depth, layers, root

    @classmethod
    def new(cls, depth: int, leaves, device: int = 0) -> "MerkleTree":
        lv = np.ascontiguousarray(leaves, dtype=np.uint64)
        lv = lv.reshape(lv.shape[0], -1)
        out = np.zeros(((2 << depth) - 1, 32), np.uint8)
        _check(lib().zinc_merkle_tree_new(depth, lv.ctypes.data, lv.shape[0], lv.shape[1], device, out.ctypes.data))
        return cls(depth, out[:-1], out[-1])


class MultilinearZipData:
    """MultilinearZipData {rows, rows_merkle_trees} (structs.rs:33-38), resident on the device; `rows` and
    `rows_merkle_trees` copy it to the host, `new` builds a handle from (possibly modified) host data."""

    def __init__(self, handle, pp=None, has_trees=True):
        self._h, self._pp, self._has_trees = handle, pp, has_trees

    @property
    def rows(self) -> np.ndarray:
        pp = self._pp
        out = np.zeros((pp.num_rows * pp.codeword_len, 4), np.uint64)
        _check(lib().zinc_zip_data_download(self._h, out.ctypes.data, None, None))
        return out

    @property
    def rows_merkle_trees(self):
        pp = self._pp
        if not self._has_trees:
            return []
        depth = pp.codeword_len.bit_length() - 1
        layers = np.zeros((pp.num_rows, 2 * pp.codeword_len - 2, 32), np.uint8)
        roots = np.zeros((pp.num_rows, 32), np.uint8)
        _check(lib().zinc_zip_data_download(self._h, None, layers.ctypes.data, roots.ctypes.data))
        return [MerkleTree(depth, layers[r], roots[r]) for r in range(pp.num_rows)]

    @classmethod
    def new(cls, pp, rows, rows_merkle_trees) -> "MultilinearZipData":
        r = np.ascontiguousarray(rows, dtype=np.uint64)
        layers = np.ascontiguousarray(np.stack([t.layers for t in rows_merkle_trees]), dtype=np.uint8)
        roots = np.ascontiguousarray(np.stack([t.root for t in rows_merkle_trees]), dtype=np.uint8)
        h = C.c_void_p()
        _check(lib().zinc_zip_data_upload(pp._h, r.ctypes.data, layers.ctypes.data, roots.ctypes.data, C.byref(h)))
        return cls(h, pp, True)

    def __del__(self):
        try:  # (at interpreter shutdown the module globals may already be gone)
            if getattr(self, "_h", None):
                lib().zinc_zip_data_free(self._h)
                self._h = None
        except Exception:
            pass


class MultilinearZipParams:
    def __init__(self, handle):
        self._h = handle
        g = [C.c_uint32() for _ in range(4)]
        lib().zinc_zip_params_geometry(handle, *[C.byref(x) for x in g])
        self.num_vars, self.num_rows, self.row_len, self.codeword_len = [x.value for x in g]

    def __del__(self):
        try:  # (at interpreter shutdown the module globals may already be gone)
            if getattr(self, "_h", None):
                lib().zinc_zip_params_free(self._h)
                self._h = None
        except Exception:
            pass


class MultilinearZip:
    @staticmethod
    def setup(poly_size: int, linear_code: RaaCode, device: int = 0) -> MultilinearZipParams:
        h = C.c_void_p()
        _check(lib().zinc_zip_setup(poly_size, C.byref(linear_code.s), device, C.byref(h)))
        return MultilinearZipParams(h)

    @staticmethod
    def commit(pp: MultilinearZipParams, evaluations, num_vars: int = None, with_merkle: bool = True):
        ev = np.ascontiguousarray(evaluations, dtype=np.int64)
        nv = num_vars if num_vars is not None else max(ev.size, 1).bit_length() - 1
        roots = np.zeros((pp.num_rows, 32), np.uint8)
        h = C.c_void_p()
        _check(lib().zinc_zip_commit(pp._h, ev.ctypes.data, ev.size, nv, int(with_merkle), roots.ctypes.data, C.byref(h)))
        return MultilinearZipData(h, pp, with_merkle), (roots if with_merkle else None)

    @staticmethod
    def commit_no_merkle(pp: MultilinearZipParams, evaluations, num_vars: int = None):
        """commit.rs:104-119: (data with empty trees, commitment with no roots)"""
        data, _ = MultilinearZip.commit(pp, evaluations, num_vars, with_merkle=False)
        return data, np.zeros((0, 32), np.uint8)

    @staticmethod
    def encode_rows(pp: MultilinearZipParams, evaluations) -> np.ndarray:
        """commit.rs:158-183: the encoded rows as a flat vector of Int<4>"""
        return MultilinearZip.commit_no_merkle(pp, evaluations)[0].rows

    @staticmethod
    def batch_commit(pp: MultilinearZipParams, polys):
        """commit.rs:134-142: a plain loop over the polynomials"""
        return [MultilinearZip.commit(pp, p) for p in polys]

    @staticmethod
    def batch_open(pp: MultilinearZipParams, polys, datas, points, field: "FieldConfig", transcript: "PcsTranscript"):
        """open_z.rs:43-58"""
        for p, d, pt in zip(polys, datas, points):
            MultilinearZip.open(pp, p, d, pt, field, transcript)

    @staticmethod
    def batch_verify_z(vp: MultilinearZipParams, comms, points, evals, transcript: "PcsTranscript", field: "FieldConfig"):
        """verify_z.rs:40-58"""
        for c, pt, ev in zip(comms, points, evals):
            MultilinearZip.verify(vp, c, pt, ev, field, transcript)

    @staticmethod
    def open(pp: MultilinearZipParams, evaluations, commit_data: MultilinearZipData, point: np.ndarray,
             field: FieldConfig, transcript: PcsTranscript, num_vars: int = None):
        ev = np.ascontiguousarray(evaluations, dtype=np.int64)
        nv = num_vars if num_vars is not None else max(ev.size, 1).bit_length() - 1
        pt = np.ascontiguousarray(point, dtype=np.uint64).reshape(-1, field.limbs) if np.size(point) else np.zeros((0, field.limbs), np.uint64)
        _check(lib().zinc_zip_open(pp._h, ev.ctypes.data, ev.size, nv, commit_data._h, pt.ctypes.data, pt.shape[0],
                                   field._m.ctypes.data, field.limbs, transcript._h))

    @staticmethod
    def verify(vp: MultilinearZipParams, roots, point: np.ndarray, eval_mont, field: FieldConfig,
               transcript: PcsTranscript):
        """MultilinearZip::verify: returns None when the proof is accepted, raises InvalidPcsOpen otherwise."""
        r = np.ascontiguousarray(roots, dtype=np.uint8)
        pt = np.ascontiguousarray(point, dtype=np.uint64).reshape(-1, field.limbs) if np.size(point) else np.zeros((0, field.limbs), np.uint64)
        ev = np.ascontiguousarray(eval_mont, dtype=np.uint64)
        _check(lib().zinc_zip_verify(vp._h, r.ctypes.data, pt.ctypes.data, pt.shape[0], ev.ctypes.data,
                                     field._m.ctypes.data, field.limbs, transcript._h))

    @staticmethod
    def evaluate(pp: MultilinearZipParams, evaluations, point: np.ndarray, field: FieldConfig) -> np.ndarray:
        """z_mle.map_to_field(config).evaluate(point) (zinc/prover.rs:317-319)."""
        ev = np.ascontiguousarray(evaluations, dtype=np.int64)
        pt = np.ascontiguousarray(point, dtype=np.uint64).reshape(-1, field.limbs) if np.size(point) else np.zeros((0, field.limbs), np.uint64)
        out = np.zeros(field.limbs, np.uint64)
        _check(lib().zinc_zip_evaluate(pp._h, ev.ctypes.data, ev.size, pt.ctypes.data, pt.shape[0], field._m.ctypes.data,
                                       field.limbs, out.ctypes.data))
        return out


def commit_z_mle_and_prove_evaluation(z_evals, r_y: np.ndarray, transcript: KeccakTranscript, field: FieldConfig,
                                      device: int = 0):
    """ZincProver::commit_z_mle_and_prove_evaluation (src/zinc/prover.rs:305-327) -> (roots, v, pcs_proof)."""
    ev = np.ascontiguousarray(z_evals, dtype=np.int64)
    pt = np.ascontiguousarray(r_y, dtype=np.uint64).reshape(-1, field.limbs) if np.size(r_y) else np.zeros((0, field.limbs), np.uint64)
    h = C.c_void_p()
    _check(lib().zinc_commit_z_mle_and_prove_evaluation(ev.ctypes.data, ev.size, pt.ctypes.data, pt.shape[0],
                                                        transcript._h, field._m.ctypes.data, field.limbs, device,
                                                        C.byref(h)))
    try:
        roots = np.zeros((lib().zinc_zip_proof_num_roots(h), 32), np.uint8)
        v = np.zeros(field.limbs, np.uint64)
        proof = np.zeros(lib().zinc_zip_proof_len(h), np.uint8)
        lib().zinc_zip_proof_read(h, roots.ctypes.data, v.ctypes.data, proof.ctypes.data)
    finally:
        lib().zinc_zip_proof_free(h)
    return roots, v, proof


def sumcheck_prove_product(transcript: KeccakTranscript, mles, degree: int, field: FieldConfig, device: int = 0):
    """MLSumcheck::prove_as_subprotocol for comb_fn = product of the MLEs (src/sumcheck.rs:56-112).
    mles: [K, 2^nv, limbs] Montgomery limbs.  Returns (msgs [nv, degree+1, limbs], randomness [nv, limbs])."""
    m = np.ascontiguousarray(mles, dtype=np.uint64)
    K, n, fl = m.shape
    nv = n.bit_length() - 1
    ptrs = (C.c_void_p * K)(*[m[k].ctypes.data for k in range(K)])
    msgs = np.zeros((nv, degree + 1, fl), np.uint64)
    rand = np.zeros((nv, fl), np.uint64)
    _check(lib().zinc_sumcheck_prove_product(transcript._h, ptrs, K, nv, degree, field._m.ctypes.data, fl, device,
                                             msgs.ctypes.data, rand.ctypes.data))
    return msgs, rand


def sumcheck_prove_ccs(transcript: KeccakTranscript, mles, degree: int, c, S, field: FieldConfig, device: int = 0):
    """prove_as_subprotocol with sumcheck_polynomial_comb_fn_1 (zinc/utils.rs:77-94): c = ccs.c as [n_terms, limbs]
    Montgomery limbs, S = ccs.S (lists of positions in `mles`, the eq() MLE last)."""
    m = np.ascontiguousarray(mles, dtype=np.uint64)
    K, n, fl = m.shape
    nv = n.bit_length() - 1
    ptrs = (C.c_void_p * K)(*[m[k].ctypes.data for k in range(K)])
    cv = np.ascontiguousarray(c, dtype=np.uint64).reshape(len(S), fl)
    masks = np.array([sum(1 << j for j in s) for s in S], dtype=np.uint32)
    msgs = np.zeros((nv, degree + 1, fl), np.uint64)
    rand = np.zeros((nv, fl), np.uint64)
    _check(lib().zinc_sumcheck_prove_ccs(transcript._h, ptrs, K, nv, degree, len(S), cv.ctypes.data, masks.ctypes.data,
                                         field._m.ctypes.data, fl, device, msgs.ctypes.data, rand.ctypes.data))
    return msgs, rand


def sumcheck_prove_products(transcript: KeccakTranscript, mles, degree: int, masks, coeffs, field: FieldConfig,
                            device: int = 0, nvars: int = None):
    """prove_as_subprotocol with rand_poly_comb_fn (sumcheck/utils.rs:67-78): sum_p coeffs[p] * prod_{j in masks[p]} vals[j].
    mles: [K, 2^nv, limbs] Montgomery limbs; coeffs: [P, limbs]; masks: P bit masks over the K MLEs."""
    m = np.ascontiguousarray(mles, dtype=np.uint64)
    K, n, fl = m.shape
    nv = nvars if nvars is not None else n.bit_length() - 1
    ptrs = (C.c_void_p * max(K, 1))(*[m[k].ctypes.data for k in range(K)])
    mk = np.ascontiguousarray(masks, dtype=np.uint32)
    cv = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(mk.size, fl) if mk.size else np.zeros((1, fl), np.uint64)
    msgs = np.zeros((nv, degree + 1, fl), np.uint64)
    rand = np.zeros((nv, fl), np.uint64)
    if not mk.size:
        mk = np.zeros(1, np.uint32)  # (a valid pointer; n_products == 0 says nothing is read)
        n_products = 0
    else:
        n_products = mk.size
    _check(lib().zinc_sumcheck_prove_products(transcript._h, ptrs, K, nv, degree, n_products, cv.ctypes.data, mk.ctypes.data,
                                              field._m.ctypes.data, fl, device, msgs.ctypes.data, rand.ctypes.data))
    return msgs, rand


def sumcheck_verify(transcript: KeccakTranscript, nvars: int, degree: int, claimed_sum, msgs, field: FieldConfig):
    """MLSumcheck::verify_as_subprotocol -> (point [nvars, limbs], expected_evaluation [limbs]); raises SpartanError.
    msgs: [rounds, evaluations per round, limbs] -- whatever the proof holds, also when that is the wrong shape."""
    fl = field.limbs
    m = np.ascontiguousarray(msgs, dtype=np.uint64).reshape(-1, np.shape(msgs)[1] if np.ndim(msgs) == 3 else 0, fl) \
        if np.size(msgs) else np.zeros((0, 0, fl), np.uint64)
    cs = np.ascontiguousarray(claimed_sum, dtype=np.uint64).reshape(fl)
    point = np.zeros((max(nvars, 1), fl), np.uint64)
    expected = np.zeros(fl, np.uint64)
    _check(lib().zinc_sumcheck_verify(transcript._h, nvars, degree, cs.ctypes.data, m.ctypes.data if m.size else None,
                                      m.shape[0], m.shape[1], field._m.ctypes.data, fl, point.ctypes.data, expected.ctypes.data))
    return point[:nvars], expected


class ZincProver:
    """ZincProver (src/zinc/structs.rs:33-47, src/zinc/prover.rs).  A CCS is given as
      matrices  Statement_Z.constraints: objects with n_rows, n_cols, row_ptr, col_idx (uint32), values (int64)  [CSR]
      s, d, S, c  CCS_Z with m = n = 2^s = 2^s_prime
      public_input, w_ccs  Statement_Z.public_input, Witness_Z.w_ccs (z = x || 1 || w)"""

    def __init__(self, device: int = 0):
        self.device = device

    @staticmethod
    def _abi_matrices(matrices):
        keep = [(np.ascontiguousarray(M.row_ptr, dtype=np.uint32), np.ascontiguousarray(M.col_idx, dtype=np.uint32),
                 np.ascontiguousarray(M.values, dtype=np.int64)) for M in matrices]
        arr = (cabi.SparseMatrix * len(matrices))()
        for k, (M, (rp, ci, va)) in enumerate(zip(matrices, keep)):
            arr[k] = cabi.SparseMatrix(M.n_rows, M.n_cols, rp.ctypes.data, ci.ctypes.data, va.ctypes.data)
        return arr, keep

    def prepare(self, matrices, s, field: "FieldConfig") -> "PreparedCcs":
        """The circuit's matrices in F_q on the device, for every later proof of that circuit (PreparedCcs)."""
        arr, _keep = self._abi_matrices(matrices)
        h = C.c_void_p()
        _check(lib().zinc_prover_prepare(arr, len(matrices), s, field._m.ctypes.data, field.limbs, self.device, C.byref(h)))
        return PreparedCcs(h)

    def _run(self, matrices, s, d, S, c, public_input, w_ccs, transcript, field, with_pcs, prepared=None):
        t, fl = len(matrices), field.limbs
        arr, _keep = self._abi_matrices(matrices)
        masks = np.array([sum(1 << j for j in Si) for Si in S], dtype=np.uint32)
        cv = np.array(c, dtype=np.int64)
        x = np.ascontiguousarray(public_input, dtype=np.int64)
        w = np.ascontiguousarray(w_ccs, dtype=np.int64)
        out = dict(msgs1=np.zeros((s, d + 2, fl), np.uint64), msgs2=np.zeros((s, 3, fl), np.uint64),
                   V_s=np.zeros((t, fl), np.uint64), r_y=np.zeros((s, fl), np.uint64))
        h = C.c_void_p()
        _check(lib().zinc_prover_prove(arr, t, s, d, len(S), masks.ctypes.data, cv.ctypes.data, x.ctypes.data, x.size,
                                       w.ctypes.data, w.size, transcript._h, field._m.ctypes.data, fl, self.device,
                                       prepared._h if prepared is not None else None, 1 if with_pcs else 0, out["msgs1"].ctypes.data, out["msgs2"].ctypes.data,
                                       out["V_s"].ctypes.data, out["r_y"].ctypes.data, C.byref(h)))
        if with_pcs:
            try:
                roots = np.zeros((lib().zinc_zip_proof_num_roots(h), 32), np.uint8)
                v = np.zeros(fl, np.uint64)
                proof = np.zeros(lib().zinc_zip_proof_len(h), np.uint8)
                lib().zinc_zip_proof_read(h, roots.ctypes.data, v.ctypes.data, proof.ctypes.data)
            finally:
                lib().zinc_zip_proof_free(h)
            out["zip_proof"] = dict(z_comm=roots, v=v, pcs_proof=proof)
        return out

    def spartan_prove(self, matrices, s, d, S, c, public_input, w_ccs, transcript: KeccakTranscript, field: FieldConfig,
                      prepared=None):
        """prepare_for_random_field_piop + SpartanProver::prove (prover.rs:130-161) -> SpartanProof fields + r_y."""
        return self._run(matrices, s, d, S, c, public_input, w_ccs, transcript, field, False, prepared)

    def prove(self, matrices, s, d, S, c, public_input, w_ccs, transcript: KeccakTranscript, field: FieldConfig,
              prepared=None):
        """Prover::prove (prover.rs:50-88) -> ZincProof {spartan_proof, zip_proof}."""
        return self._run(matrices, s, d, S, c, public_input, w_ccs, transcript, field, True, prepared)


class ZincVerifier:
    """ZincVerifier (src/zinc/verifier.rs) without the draw_random_field check of Verifier::verify."""

    def __init__(self, device: int = 0):
        self.device = device

    def _run(self, matrices, s, d, S, c, proof, transcript, field, with_pcs, prepared):
        t, fl = len(matrices), field.limbs
        arr, _keep = ZincProver._abi_matrices(matrices)
        masks = np.array([sum(1 << j for j in Si) for Si in S], dtype=np.uint32)
        cv = np.array(c, dtype=np.int64)
        m1 = np.ascontiguousarray(proof["msgs1"], dtype=np.uint64)
        m2 = np.ascontiguousarray(proof["msgs2"], dtype=np.uint64)
        vs = np.ascontiguousarray(proof["V_s"], dtype=np.uint64)
        assert m1.shape == (s, d + 2, fl) and m2.shape == (s, 3, fl) and vs.shape == (t, fl)
        pts = dict(rx_ry=np.zeros((2 * s, fl), np.uint64), e_y=np.zeros(fl, np.uint64), gamma=np.zeros(fl, np.uint64))
        roots = v = pp = None
        if with_pcs:
            zp = proof["zip_proof"]
            roots = np.ascontiguousarray(zp["z_comm"], dtype=np.uint8)
            v = np.ascontiguousarray(zp["v"], dtype=np.uint64)
            pp = np.ascontiguousarray(zp["pcs_proof"], dtype=np.uint8)
        _check(lib().zinc_verifier_verify(arr, t, s, d, len(S), masks.ctypes.data, cv.ctypes.data, transcript._h,
                                          field._m.ctypes.data, fl, self.device,
                                          prepared._h if prepared is not None else None, m1.ctypes.data, m2.ctypes.data,
                                          vs.ctypes.data, 1 if with_pcs else 0,
                                          roots.ctypes.data if with_pcs else None, roots.shape[0] if with_pcs else 0,
                                          v.ctypes.data if with_pcs else None, pp.ctypes.data if with_pcs else None,
                                          pp.size if with_pcs else 0, pts["rx_ry"].ctypes.data, pts["e_y"].ctypes.data,
                                          pts["gamma"].ctypes.data))
        return pts

    def spartan_verify(self, matrices, s, d, S, c, proof, transcript: KeccakTranscript, field: FieldConfig):
        """SpartanVerifier::verify (verifier.rs:105-139) -> VerificationPoints; raises SpartanError."""
        return self._run(matrices, s, d, S, c, proof, transcript, field, False, None)

    def verify(self, matrices, s, d, S, c, proof, transcript: KeccakTranscript, field: FieldConfig, prepared=None):
        """Verifier::verify (verifier.rs:45-76); raises SpartanError / InvalidPcsOpen."""
        return self._run(matrices, s, d, S, c, proof, transcript, field, True, prepared)


class PreparedCcs:
    def __init__(self, handle):
        self._h = handle

    def __del__(self):
        try:  # (at interpreter shutdown the module globals may already be gone)
            if getattr(self, "_h", None):
                lib().zinc_prepared_ccs_free(self._h)
                self._h = None
        except Exception:
            pass
