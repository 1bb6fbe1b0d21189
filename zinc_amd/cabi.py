"""ctypes binding of libzip_hip.so (include/zip_hip.h).

Thin by design: the product is the HIP library behind the C ABI; this module only
marshals numpy arrays (host memory) and raw device pointers (e.g. torch tensors'
``data_ptr()``) across it.  There is NO CPU fallback: if the shared library is
missing or no gfx950 device is present the calls fail loudly.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ZIP_HIP_LIB_PATH") or os.path.join(_PKG, "lib", "libzip_hip.so")  # override: A/B runs of two builds

ZIP_OK = 0
ZIP_ERR_INVALID_PARAM = -1
ZIP_ERR_SHAPE = -2
ZIP_ERR_HIP = -3
ZIP_ERR_NO_DEVICE = -4
ZIP_ERR_UNSUPPORTED = -5
ZIP_ERR_ALLOC = -6
ZIP_ERR_NULL = -7
MEM_HOST, MEM_DEVICE = 0, 1

EXPORTED_SYMBOLS = (
    "zip_abi_version", "zip_strerror", "zip_device_count", "zip_release_cached_memory", "zip_host_register", "zip_host_unregister", "zip_ctx_create", "zip_ctx_destroy",
    "zip_ctx_last_error", "zip_ctx_synchronize", "zip_ctx_stream", "zip_ctx_set_speculation", "zip_commit", "zip_commit_hinted", "zip_commit_open", "zip_commit_open_begin", "zip_job_wait", "zip_commitment_free",
    "zip_commitment_device_ptrs", "zip_commit_download", "zip_commitment_upload", "zip_open_testing",
    "zip_open_columns", "zip_open_eval", "zip_proof_len", "zip_open", "zip_open_shard", "zip_sum_partials", "zip_merkle_trees",
    "zip_ctx_set_profiling", "zip_ctx_profile_read", "zip_ctx_commit_clock",
    "zip_mctx_create", "zip_mctx_destroy", "zip_mctx_last_error", "zip_mctx_shards", "zip_mctx_shard_ctx", "zip_mctx_set_witness",
    "zip_mctx_commit_open", "zip_mctx_shard_openings", "zip_mctx_ends", "zip_mctx_roots", "zip_mctx_roots_path", "zip_verify", "zip_mle_eval", "zip_commitment_mle_eval", "zip_field_map_int256",
    "zip_open_stream", "zip_sumcheck_init", "zip_sumcheck_round", "zip_sumcheck_round_begin", "zip_sumcheck_round_end", "zip_sumcheck_last_error", "zip_sumcheck_free",
    "zip_ccs_create", "zip_ccs_free", "zip_ccs_last_error", "zip_ccs_set_z", "zip_ccs_eq_table",
    "zip_ccs_second_table", "zip_ccs_table", "zip_ccs_download", "zip_ccs_eval_matrices",
)


class Job:
    """A zip_commit_open in flight (ZipContext.commit_open_begin)."""

    def __init__(self, ctx, handle, keep):
        self._ctx, self._h, self._keep = ctx, handle, keep

    def wait(self, want_roots=False):
        """-> roots (numpy) or None; the proof is in the buffer given to commit_open_begin."""
        assert self._h is not None, "job already waited for"
        roots = np.zeros((self._ctx.rows_local, 32), dtype=np.uint8) if want_roots else None
        h, self._h = self._h, None
        rc = lib().zip_job_wait(h, roots.ctypes.data if roots is not None else None)
        self._keep = None
        self._ctx._check(rc, "zip_job_wait")
        return roots

    def close(self):
        """Collects a job nobody waited for (an exception between begin and wait): frees its slot of the ctx and keeps
        the buffers alive until the GPU is done with them."""
        if self._h is not None:
            h, self._h = self._h, None
            try:
                lib().zip_job_wait(h, None)
            finally:
                self._keep = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass


class ZipError(RuntimeError):
    def __init__(self, code, what, detail=""):
        self.code = code
        super().__init__(f"{what}: {strerror(code)} ({code})" + (f": {detail}" if detail else ""))


class ZipParams(C.Structure):
    _fields_ = [
        ("num_vars", C.c_uint32), ("row_len", C.c_uint32), ("num_rows", C.c_uint32),
        ("codeword_len", C.c_uint32), ("rep", C.c_uint32),
        ("n_limbs", C.c_uint32), ("k_limbs", C.c_uint32), ("m_limbs", C.c_uint32),
        ("perm1", C.POINTER(C.c_uint32)), ("perm2", C.POINTER(C.c_uint32)),
        ("device", C.c_int32), ("row_begin", C.c_uint32), ("row_count", C.c_uint32),
    ]


class ZipField(C.Structure):
    _fields_ = [("limbs", C.c_uint32), ("modulus", C.c_uint64 * 8)]


class VerifyReport(C.Structure):
    _fields_ = [("verdict", C.c_int32), ("column", C.c_uint32), ("bad_merkle_paths", C.c_uint32),
                ("malformed_paths", C.c_uint32)]


VERIFY_ACCEPT, VERIFY_PROXIMITY_TESTING, VERIFY_EVAL_CONSISTENCY, VERIFY_PROXIMITY_Q0 = 0, 1, 2, 3
VERIFY_MERKLE, VERIFY_MALFORMED, VERIFY_OVERFLOW = 4, 5, 6


class SumcheckComb(C.Structure):
    """zip_sumcheck_comb: (sum_t coeff[t] * prod_{j in term_mask[t]} vals[j]) * vals[-1]"""
    _fields_ = [("n_terms", C.c_uint32), ("term_mask", C.c_uint32 * 8), ("coeff", (C.c_uint64 * 8) * 8)]


def make_comb(term_masks, coeffs_limbs):
    """term_masks: ints; coeffs_limbs: [n_terms, limbs] Montgomery limbs"""
    c = SumcheckComb()
    c.n_terms = len(term_masks)
    cl = np.asarray(coeffs_limbs, dtype=np.uint64).reshape(len(term_masks), -1)
    for t, m in enumerate(term_masks):
        c.term_mask[t] = int(m)
        for i in range(cl.shape[1]):
            c.coeff[t][i] = int(cl[t, i])
    return c


class SparseMatrix(C.Structure):
    """zip_sparse_matrix: SparseMatrix<Int<1>> as CSR"""
    _fields_ = [("n_rows", C.c_uint32), ("n_cols", C.c_uint32), ("row_ptr", C.c_void_p), ("col_idx", C.c_void_p),
                ("values", C.c_void_p)]


CCS_Z_FIELD, CCS_MZ, CCS_EQ, CCS_SECOND = 0, 1, 2, 3

PROOF_SINK = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t)


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char_p), ("launches", C.c_uint32), ("total_ms", C.c_float)]


_lib = None


def lib():
    """Loads the HIP library; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        # PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64.so.1.  Loading ours
        # first would put TWO HIP runtimes in the process (torch then sees "No HIP GPUs").
        # With torch's already mapped, the loader resolves our NEEDED sonames to the same copy.
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m zinc_amd.build` "
            "(__graft_entry__.build()).  The Zip HIP path has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, u8p, u32p, u64p, i64p = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p
    L.zip_abi_version.restype = C.c_int32
    L.zip_strerror.restype = C.c_char_p
    L.zip_strerror.argtypes = [C.c_int32]
    L.zip_device_count.restype = C.c_int32
    L.zip_release_cached_memory.restype = None
    L.zip_host_register.argtypes = [C.c_void_p, C.c_size_t]
    L.zip_host_unregister.argtypes = [C.c_void_p]
    L.zip_host_unregister.restype = None
    L.zip_ctx_create.argtypes = [C.POINTER(ZipParams), C.POINTER(vp)]
    L.zip_ctx_destroy.argtypes = [vp]
    L.zip_ctx_destroy.restype = None
    L.zip_ctx_last_error.argtypes = [vp]
    L.zip_ctx_last_error.restype = C.c_char_p
    L.zip_ctx_synchronize.argtypes = [vp]
    L.zip_ctx_stream.argtypes = [vp]
    L.zip_ctx_stream.restype = vp
    L.zip_commit.argtypes = [vp, i64p, C.c_size_t, C.c_int, C.c_int32, u8p, C.POINTER(vp)]
    L.zip_commit_hinted.argtypes = [vp, i64p, C.c_size_t, C.c_int, u32p, C.c_uint32, u8p, C.POINTER(vp)]
    L.zip_commit_open.argtypes = [vp, i64p, C.c_size_t, C.c_int, i64p, u32p, C.c_uint32, u64p, C.POINTER(ZipField), u8p, vp,
                                  C.c_int, C.POINTER(vp)]
    L.zip_commit_open.restype = C.c_int32
    L.zip_commit_open_begin.argtypes = [vp, i64p, C.c_size_t, i64p, u32p, C.c_uint32, u64p, C.POINTER(ZipField), vp, C.POINTER(vp)]
    L.zip_commit_open_begin.restype = C.c_int32
    L.zip_job_wait.argtypes = [vp, u8p]
    L.zip_job_wait.restype = C.c_int32
    L.zip_verify.argtypes = [vp, u8p, vp, C.c_int, C.c_size_t, i64p, u32p, C.c_uint32, u64p, u64p, u64p,
                             C.POINTER(ZipField), C.POINTER(VerifyReport)]
    L.zip_mle_eval.argtypes = [vp, i64p, C.c_int, u64p, u64p, C.POINTER(ZipField), u64p]
    L.zip_commitment_mle_eval.argtypes = [vp, u64p, u64p, C.POINTER(ZipField), u64p]
    L.zip_field_map_int256.argtypes = [vp, u64p, C.c_uint32, C.POINTER(ZipField), u64p]
    L.zip_open_stream.argtypes = [vp, i64p, C.c_int, i64p, u32p, C.c_uint32, u64p, C.POINTER(ZipField), PROOF_SINK,
                                  vp, C.c_size_t]
    L.zip_sumcheck_init.argtypes = [C.c_int32, vp, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(SumcheckComb),
                                    C.POINTER(ZipField), C.POINTER(vp)]
    L.zip_sumcheck_round.argtypes = [vp, u64p, u64p]
    L.zip_sumcheck_round_begin.argtypes = [vp, u64p]
    L.zip_sumcheck_round_end.argtypes = [vp, u64p]
    L.zip_sumcheck_last_error.argtypes = [vp]
    L.zip_sumcheck_last_error.restype = C.c_char_p
    L.zip_sumcheck_free.argtypes = [vp]
    L.zip_sumcheck_free.restype = None
    L.zip_ccs_create.argtypes = [C.c_int32, C.POINTER(SparseMatrix), C.c_uint32, C.c_uint32, C.POINTER(ZipField),
                                 C.POINTER(vp)]
    L.zip_ccs_free.argtypes = [vp]
    L.zip_ccs_free.restype = None
    L.zip_ccs_last_error.argtypes = [vp]
    L.zip_ccs_last_error.restype = C.c_char_p
    L.zip_ccs_set_z.argtypes = [vp, i64p, C.c_size_t, C.c_int]
    L.zip_ccs_eq_table.argtypes = [vp, u64p, C.c_uint32]
    L.zip_ccs_second_table.argtypes = [vp, u64p, u64p, u64p]
    L.zip_ccs_table.argtypes = [vp, C.c_int, C.c_uint32, C.POINTER(vp)]
    L.zip_ccs_eval_matrices.argtypes = [vp, u64p, u64p, u64p]
    L.zip_ccs_download.argtypes = [vp, C.c_int, C.c_uint32, u64p]
    L.zip_commitment_free.argtypes = [vp]
    L.zip_commitment_free.restype = None
    L.zip_commitment_device_ptrs.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.zip_commit_download.argtypes = [vp, u64p, u8p, u8p]
    L.zip_commitment_upload.argtypes = [vp, u64p, u8p, u8p, C.POINTER(vp)]
    L.zip_open_testing.argtypes = [vp, i64p, C.c_int, i64p, u64p, C.c_int]
    L.zip_open_columns.argtypes = [vp, u32p, C.c_uint32, u8p, C.c_int]
    L.zip_open_eval.argtypes = [vp, i64p, C.c_int, u64p, C.POINTER(ZipField), u64p, C.c_int]
    L.zip_proof_len.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.zip_proof_len.restype = C.c_size_t
    L.zip_open.argtypes = [vp, i64p, C.c_int, i64p, u32p, C.c_uint32, u64p, C.POINTER(ZipField), u8p, C.c_int]
    L.zip_open_shard.argtypes = [vp, i64p, i64p, u32p, C.c_uint32, u64p, C.POINTER(ZipField), u64p, u64p, u8p]
    L.zip_open_shard.restype = C.c_int32
    L.zip_sum_partials.argtypes = [vp, u64p, u64p, C.c_uint32, C.POINTER(ZipField), u64p, u64p]
    L.zip_merkle_trees.argtypes = [C.c_int32, u64p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, u8p]
    L.zip_ctx_set_profiling.argtypes = [vp, C.c_int32]
    L.zip_ctx_profile_read.argtypes = [vp, C.POINTER(KernelTime), C.c_uint32]
    L.zip_ctx_commit_clock.argtypes = [vp, C.POINTER(C.c_double)]
    L.zip_ctx_commit_clock.restype = C.c_int32
    L.zip_mctx_create.argtypes = [C.POINTER(ZipParams), C.c_int32, C.POINTER(C.c_int32), C.POINTER(vp)]
    L.zip_mctx_destroy.argtypes = [vp]
    L.zip_mctx_destroy.restype = None
    L.zip_mctx_last_error.argtypes = [vp]
    L.zip_mctx_last_error.restype = C.c_char_p
    L.zip_mctx_shards.argtypes = [vp]
    L.zip_mctx_shards.restype = C.c_uint32
    L.zip_mctx_shard_ctx.argtypes = [vp, C.c_uint32]
    L.zip_mctx_shard_ctx.restype = vp
    L.zip_mctx_set_witness.argtypes = [vp, i64p, C.c_size_t]
    L.zip_mctx_commit_open.argtypes = [vp, i64p, i64p, u32p, C.c_uint32, u64p, C.POINTER(ZipField), u8p, u8p]
    L.zip_mctx_shard_openings.argtypes = [vp, C.c_uint32, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(C.c_uint32),
                                          C.POINTER(C.c_uint32)]
    L.zip_mctx_ends.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.zip_mctx_roots.argtypes = [vp, C.c_uint32, C.POINTER(vp)]
    L.zip_mctx_roots.restype = C.c_int32
    L.zip_mctx_roots_path.argtypes = [vp]
    L.zip_mctx_roots_path.restype = C.c_char_p
    for fn in ("zip_mctx_create", "zip_mctx_set_witness", "zip_mctx_commit_open", "zip_mctx_shard_openings", "zip_mctx_ends"):
        getattr(L, fn).restype = C.c_int32
    for fn in ("zip_ctx_create", "zip_ctx_synchronize", "zip_commit", "zip_commit_hinted", "zip_commitment_device_ptrs",
               "zip_commit_download", "zip_commitment_upload", "zip_open_testing", "zip_open_columns",
               "zip_open_eval", "zip_open", "zip_sum_partials", "zip_merkle_trees", "zip_ctx_set_profiling",
               "zip_ctx_profile_read", "zip_verify", "zip_mle_eval", "zip_field_map_int256", "zip_open_stream", "zip_sumcheck_init", "zip_sumcheck_round"):
        getattr(L, fn).restype = C.c_int32
    _lib = L
    return L


def strerror(code):
    return lib().zip_strerror(code).decode()


def device_count():
    return lib().zip_device_count()


def make_field(modulus: int, limbs: int) -> ZipField:
    f = ZipField()
    f.limbs = limbs
    for i in range(8):
        f.modulus[i] = (modulus >> (64 * i)) & 0xFFFFFFFFFFFFFFFF if i < limbs else 0
    return f


def _ptr(x):
    """(address, mem_kind) of a numpy array (host), a torch tensor, or None."""
    if x is None:
        return None, MEM_HOST
    if isinstance(x, np.ndarray):
        assert x.flags["C_CONTIGUOUS"]
        return x.ctypes.data, MEM_HOST
    if hasattr(x, "data_ptr"):  # torch tensor
        assert x.is_contiguous()
        return x.data_ptr(), (MEM_DEVICE if x.is_cuda else MEM_HOST)
    raise TypeError(type(x))


def geometry(num_vars, rep=2):
    """RaaCode::new + MultilinearZip::setup geometry (code_raa.rs:43,113; structs.rs:82)."""
    n = 1 << num_vars
    isqrt = int(np.floor(np.sqrt(n)))
    while isqrt * isqrt > n:
        isqrt -= 1
    while (isqrt + 1) * (isqrt + 1) <= n:
        isqrt += 1
    row_len = 1
    while row_len < isqrt:
        row_len <<= 1
    num_rows = 1
    while num_rows < n // row_len:
        num_rows <<= 1
    return row_len, num_rows, row_len * rep


class ZipContext:
    """One geometry on one GPU (zip_ctx).  perm1/perm2: uint32 permutation tables."""

    def __init__(self, num_vars, perm1, perm2, device=0, rep=2, row_begin=0, row_count=0,
                 n_limbs=1, k_limbs=4, m_limbs=8, geometry_override=None):
        L = lib()
        row_len, num_rows, cw = geometry_override or geometry(num_vars, rep)
        self.num_vars, self.row_len, self.num_rows, self.codeword_len, self.rep = num_vars, row_len, num_rows, cw, rep
        self.depth = max(cw - 1, 0).bit_length() if cw > 1 else 0
        self.rows_local = row_count or num_rows
        self.row_begin = row_begin
        self.k_limbs, self.m_limbs = k_limbs, m_limbs
        self._perm1 = np.ascontiguousarray(perm1, dtype=np.uint32)
        self._perm2 = np.ascontiguousarray(perm2, dtype=np.uint32)
        p = ZipParams(num_vars, row_len, num_rows, cw, rep, n_limbs, k_limbs, m_limbs,
                      self._perm1.ctypes.data_as(C.POINTER(C.c_uint32)),
                      self._perm2.ctypes.data_as(C.POINTER(C.c_uint32)), device, row_begin, row_count)
        h = C.c_void_p()
        rc = L.zip_ctx_create(C.byref(p), C.byref(h))
        if rc != ZIP_OK:
            raise ZipError(rc, "zip_ctx_create")
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            lib().zip_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != ZIP_OK:
            raise ZipError(rc, what, lib().zip_ctx_last_error(self._h).decode())

    def synchronize(self):
        self._check(lib().zip_ctx_synchronize(self._h), "zip_ctx_synchronize")

    @property
    def stream(self):
        return lib().zip_ctx_stream(self._h)

    def commit(self, evals, with_merkle=True, want_roots=True, hint_cols=None):
        """MultilinearZip::commit / commit_no_merkle.  evals: int64 numpy array or CUDA tensor.
        hint_cols: the column indices the caller is going to open (zip_commit_hinted): same roots, same handle, but
        the kernel skips the stores no opening of those columns reads."""
        ptr, kind = _ptr(evals)
        n = evals.size if isinstance(evals, np.ndarray) else evals.numel()
        roots = np.zeros((self.rows_local, 32), dtype=np.uint8) if (with_merkle and want_roots) else None
        h = C.c_void_p()
        rp = roots.ctypes.data if roots is not None else None
        if hint_cols is not None:
            assert with_merkle, "a hint only makes sense for a commitment that will be opened"
            hc = np.ascontiguousarray(hint_cols, dtype=np.uint32)
            rc = lib().zip_commit_hinted(self._h, ptr, n, kind, hc.ctypes.data, hc.size, rp, C.byref(h))
            self._check(rc, "zip_commit_hinted")
        else:
            rc = lib().zip_commit(self._h, ptr, n, kind, int(with_merkle), rp, C.byref(h))
            self._check(rc, "zip_commit")
        return Commitment(self, h, bool(with_merkle)), roots

    def commit_open(self, evals, coeffs, cols, q0_mont, field, out=None, want_roots=True, keep=False):
        """zip_commit_open: MultilinearZip::commit + open in one call (prover.rs:305-328).
        -> (proof, roots or None, Commitment or None)."""
        ptr, kind = _ptr(evals)
        n = evals.size if isinstance(evals, np.ndarray) else evals.numel()
        cols = np.ascontiguousarray(cols, dtype=np.uint32)
        coeffs_c = np.ascontiguousarray(coeffs, dtype=np.int64) if coeffs is not None else None
        q0 = np.ascontiguousarray(q0_mont, dtype=np.uint64) if q0_mont is not None else None
        roots = np.zeros((self.rows_local, 32), dtype=np.uint8) if want_roots else None
        res = out if out is not None else np.zeros(self.proof_len(cols.size, field.limbs), dtype=np.uint8)
        optr, okind = _ptr(res)
        h = C.c_void_p()
        rc = lib().zip_commit_open(self._h, ptr, n, kind, coeffs_c.ctypes.data if coeffs_c is not None else None,
                                   cols.ctypes.data, cols.size, q0.ctypes.data if q0 is not None else None, C.byref(field),
                                   roots.ctypes.data if roots is not None else None, optr, okind,
                                   C.byref(h) if keep else None)
        self._check(rc, "zip_commit_open")
        return res, roots, (Commitment(self, h, True) if keep else None)

    def commit_open_prepared(self, evals, coeffs, cols, q0_mont, field, out):
        """The same zip_commit_open call as commit_open(..., out=out, want_roots=False, keep=False), with its arguments
        marshalled ONCE: returns a function that makes the call (a prover that proves many witnesses of one shape into the
        same buffers pays the numpy / ctypes conversions, ~10 us, once instead of per proof).  The arrays are kept alive by
        the returned function; `evals` and `out` are read / written in place at every call."""
        ptr, kind = _ptr(evals)
        n = evals.size if isinstance(evals, np.ndarray) else evals.numel()
        cols = np.ascontiguousarray(cols, dtype=np.uint32)
        coeffs_c = np.ascontiguousarray(coeffs, dtype=np.int64) if coeffs is not None else None
        q0 = np.ascontiguousarray(q0_mont, dtype=np.uint64) if q0_mont is not None else None
        optr, okind = _ptr(out)
        fn = lib().zip_commit_open
        args = (self._h, ptr, n, kind, coeffs_c.ctypes.data if coeffs_c is not None else None, cols.ctypes.data, cols.size,
                q0.ctypes.data if q0 is not None else None, C.byref(field), None, optr, okind, None)
        keep = (evals, cols, coeffs_c, q0, field, out)
        check = self._check

        def call(_keep=keep):
            rc = fn(*args)
            if rc != ZIP_OK:
                check(rc, "zip_commit_open")

        return call

    def commit_open_begin(self, evals_d, coeffs, cols, q0_mont, field, out_d):
        """zip_commit_open_begin: the whole commit + open of a DEVICE witness into a DEVICE proof buffer is enqueued; the
        returned Job's wait() collects it.  Two jobs per ctx may be in flight."""
        ptr, kind = _ptr(evals_d)
        optr, okind = _ptr(out_d)
        assert kind == MEM_DEVICE and okind == MEM_DEVICE, "jobs take device memory"
        cols = np.ascontiguousarray(cols, dtype=np.uint32)
        coeffs_c = np.ascontiguousarray(coeffs, dtype=np.int64) if coeffs is not None else None
        q0 = np.ascontiguousarray(q0_mont, dtype=np.uint64) if q0_mont is not None else None
        h = C.c_void_p()
        rc = lib().zip_commit_open_begin(self._h, ptr, evals_d.numel(), coeffs_c.ctypes.data if coeffs_c is not None else None,
                                         cols.ctypes.data, cols.size, q0.ctypes.data if q0 is not None else None, C.byref(field),
                                         optr, C.byref(h))
        self._check(rc, "zip_commit_open_begin")
        return Job(self, h, (evals_d, out_d))

    def upload_commitment(self, rows, layers=None, roots=None):
        # keep every (possibly copied) array alive across the call
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        layers_c = None if layers is None else np.ascontiguousarray(layers, dtype=np.uint8)
        roots_c = None if roots is None else np.ascontiguousarray(roots, dtype=np.uint8)
        h = C.c_void_p()
        rc = lib().zip_commitment_upload(
            self._h, rows.ctypes.data, None if layers_c is None else layers_c.ctypes.data,
            None if roots_c is None else roots_c.ctypes.data, C.byref(h))
        self._check(rc, "zip_commitment_upload")
        return Commitment(self, h, layers is not None)

    def open_testing(self, evals, coeffs, out=None):
        ptr, kind = _ptr(evals)
        coeffs = np.ascontiguousarray(coeffs, dtype=np.int64)
        assert coeffs.size == self.rows_local
        res = out if out is not None else np.zeros((self.row_len, self.m_limbs), dtype=np.uint64)
        optr, okind = _ptr(res)
        self._check(lib().zip_open_testing(self._h, ptr, kind, coeffs.ctypes.data, optr, okind), "zip_open_testing")
        return res

    def open_eval(self, evals, q0_mont, field: ZipField, out=None):
        ptr, kind = _ptr(evals)
        q0p = None
        if q0_mont is not None:
            q0_mont = np.ascontiguousarray(q0_mont, dtype=np.uint64)
            assert q0_mont.size == self.rows_local * field.limbs
            q0p = q0_mont.ctypes.data
        res = out if out is not None else np.zeros((self.row_len, field.limbs), dtype=np.uint64)
        optr, okind = _ptr(res)
        self._check(lib().zip_open_eval(self._h, ptr, kind, q0p, C.byref(field), optr, okind), "zip_open_eval")
        return res

    def verify(self, roots, proof, coeffs, cols, q0_mont, q1_mont, eval_mont, field: ZipField):
        """MultilinearZip::verify on the device; returns the zip_verify_report as a dict."""
        roots_c = np.ascontiguousarray(roots, dtype=np.uint8)
        cols_c = np.ascontiguousarray(cols, dtype=np.uint32)
        coeffs_c = None if coeffs is None else np.ascontiguousarray(coeffs, dtype=np.int64)
        q0 = None if q0_mont is None else np.ascontiguousarray(q0_mont, dtype=np.uint64)
        q1 = None if q1_mont is None else np.ascontiguousarray(q1_mont, dtype=np.uint64)
        ev = np.ascontiguousarray(eval_mont, dtype=np.uint64)
        if isinstance(proof, np.ndarray):
            proof = np.ascontiguousarray(proof, dtype=np.uint8)
        pptr, pkind = _ptr(proof)
        plen = proof.size if isinstance(proof, np.ndarray) else proof.numel()
        rep = VerifyReport()
        rc = lib().zip_verify(self._h, roots_c.ctypes.data, pptr, pkind, plen,
                              None if coeffs_c is None else coeffs_c.ctypes.data, cols_c.ctypes.data, cols_c.size,
                              None if q0 is None else q0.ctypes.data, None if q1 is None else q1.ctypes.data,
                              ev.ctypes.data, C.byref(field), C.byref(rep))
        self._check(rc, "zip_verify")
        return {"verdict": rep.verdict, "column": rep.column, "bad_merkle_paths": rep.bad_merkle_paths,
                "malformed_paths": rep.malformed_paths}

    def mle_eval(self, evals, q0_mont, q1_mont, field: ZipField):
        ptr, kind = _ptr(evals)
        q0 = None if q0_mont is None else np.ascontiguousarray(q0_mont, dtype=np.uint64)
        q1 = None if q1_mont is None else np.ascontiguousarray(q1_mont, dtype=np.uint64)
        out = np.zeros(field.limbs, dtype=np.uint64)
        self._check(lib().zip_mle_eval(self._h, ptr, kind, None if q0 is None else q0.ctypes.data,
                                       None if q1 is None else q1.ctypes.data, C.byref(field), out.ctypes.data),
                    "zip_mle_eval")
        return out

    def field_map_int256(self, values, field: ZipField):
        v = np.ascontiguousarray(values, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros((v.shape[0], field.limbs), dtype=np.uint64)
        self._check(lib().zip_field_map_int256(self._h, v.ctypes.data, v.shape[0], C.byref(field), out.ctypes.data),
                    "zip_field_map_int256")
        return out

    def proof_len(self, n_cols, field_limbs):
        return lib().zip_proof_len(self._h, n_cols, field_limbs)

    def sum_partials(self, uparts, fparts, n_parts, field, uprime_out, row_out):
        up = _ptr(uparts)[0] if uparts is not None else None
        fp = _ptr(fparts)[0] if fparts is not None else None
        uo = _ptr(uprime_out)[0] if uprime_out is not None else None
        ro = _ptr(row_out)[0] if row_out is not None else None
        self._check(lib().zip_sum_partials(self._h, up, fp, n_parts, C.byref(field) if field is not None else None,
                                           uo, ro), "zip_sum_partials")

    def set_profiling(self, on=True):
        self._check(lib().zip_ctx_set_profiling(self._h, int(on)), "zip_ctx_set_profiling")

    def set_speculation(self, on=True):
        """zip_ctx_set_speculation: plain commits hint themselves with the columns of the ctx's last opening.  Default: only
        commits of a HOST witness do; on=True extends it to DEVICE witnesses (which must then outlive their handles
        unchanged), on=False switches it off."""
        self._check(lib().zip_ctx_set_speculation(self._h, int(on)), "zip_ctx_set_speculation")

    def profile_read(self):
        buf = (KernelTime * 32)()
        n = lib().zip_ctx_profile_read(self._h, buf, 32)
        if n < 0:
            self._check(n, "zip_ctx_profile_read")
        return {buf[i].name.decode(): (buf[i].launches, buf[i].total_ms) for i in range(min(n, 32))}

    def commit_clock_mhz(self):
        """Shader clock held during the most recent profiled commit kernel (0.0 if none)."""
        mhz = C.c_double(0.0)
        self._check(lib().zip_ctx_commit_clock(self._h, C.byref(mhz)), "zip_ctx_commit_clock")
        return mhz.value


class ZipMultiContext:
    """zip_mctx: one polynomial's commit + open over several GPUs from ONE process (devices may repeat)."""

    def __init__(self, num_vars, perm1, perm2, devices, geometry_override=None):
        row_len, num_rows, cw = geometry_override or geometry(num_vars)
        self.num_vars, self.row_len, self.num_rows, self.codeword_len = num_vars, row_len, num_rows, cw
        self.depth = cw.bit_length() - 1
        p1 = np.ascontiguousarray(perm1, dtype=np.uint32)
        p2 = np.ascontiguousarray(perm2, dtype=np.uint32)
        p = ZipParams(num_vars, row_len, num_rows, cw, cw // row_len, 1, 4, 8,
                      p1.ctypes.data_as(C.POINTER(C.c_uint32)), p2.ctypes.data_as(C.POINTER(C.c_uint32)), 0, 0, 0)
        devs = (C.c_int32 * len(devices))(*devices)
        h = C.c_void_p()
        rc = lib().zip_mctx_create(C.byref(p), len(devices), devs, C.byref(h))
        if rc:
            raise ZipError(rc, "zip_mctx_create")
        self._h = h

    def _check(self, rc, what):
        if rc:
            raise ZipError(rc, what, (lib().zip_mctx_last_error(self._h) or b"").decode())

    def proof_len(self, n_cols, fl):
        u = self.row_len * 64 if self.num_rows > 1 else 0
        return u + n_cols * self.num_rows * (32 + 8 + 32 * self.depth) + self.row_len * 8 * fl

    def set_witness(self, evals):
        evals = np.ascontiguousarray(evals, dtype=np.int64)
        self._check(lib().zip_mctx_set_witness(self._h, evals.ctypes.data, evals.size), "zip_mctx_set_witness")

    def commit_open(self, evals, coeffs, cols, q0_mont, field, want_proof=True, want_roots=True):
        """evals: host int64 array or None (witness placed with set_witness).  -> (proof or None, roots or None)."""
        ev = np.ascontiguousarray(evals, dtype=np.int64) if evals is not None else None
        cols = np.ascontiguousarray(cols, dtype=np.uint32)
        co = np.ascontiguousarray(coeffs, dtype=np.int64) if coeffs is not None else None
        q0 = np.ascontiguousarray(q0_mont, dtype=np.uint64) if q0_mont is not None else None
        proof = np.zeros(self.proof_len(cols.size, field.limbs), dtype=np.uint8) if want_proof else None
        roots = np.zeros((self.num_rows, 32), dtype=np.uint8) if want_roots else None
        rc = lib().zip_mctx_commit_open(self._h, ev.ctypes.data if ev is not None else None,
                                        co.ctypes.data if co is not None else None, cols.ctypes.data, cols.size,
                                        q0.ctypes.data if q0 is not None else None, C.byref(field),
                                        roots.ctypes.data if roots is not None else None,
                                        proof.ctypes.data if proof is not None else None)
        self._check(rc, "zip_mctx_commit_open")
        return proof, roots

    def shards(self):
        return lib().zip_mctx_shards(self._h)

    def roots_ptr(self, shard):
        """Device address (on shard's device) of the gathered commitment: the roots of ALL rows, [num_rows][32]."""
        p = C.c_void_p()
        self._check(lib().zip_mctx_roots(self._h, shard, C.byref(p)), "zip_mctx_roots")
        return p.value

    def roots_path(self):
        """How the roots reached every device in the last commit_open: "rccl", "copies" or "none"."""
        return (lib().zip_mctx_roots_path(self._h) or b"").decode()

    def shard_profile(self, s, on=None):
        """Profiling hooks of shard s's context: on=True/False switches, None reads {kernel: (launches, ms)}."""
        h = lib().zip_mctx_shard_ctx(self._h, s)
        if on is not None:
            lib().zip_ctx_set_profiling(h, int(on))
            return None
        buf = (KernelTime * 32)()
        n = lib().zip_ctx_profile_read(h, buf, 32)
        return {buf[i].name.decode(): (buf[i].launches, buf[i].total_ms) for i in range(max(0, min(n, 32)))}

    def close(self):
        if self._h:
            lib().zip_mctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class Commitment:
    """Device-resident MultilinearZipData (+ roots) behind a zip_commitment handle."""

    def __init__(self, ctx: ZipContext, handle, has_merkle):
        self.ctx, self._h, self.has_merkle = ctx, handle, has_merkle

    def free(self):
        if getattr(self, "_h", None) and self.ctx._h:
            lib().zip_commitment_free(self._h)
        self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def device_ptrs(self, rows=True, layers=True):
        """(rows, layers, roots) device pointers.  Asking for `rows` expands the handle's 16-byte entries into the
        Int<4> array once (a full-size copy on the device); asking for rows or layers of a HINTED handle first re-runs
        the commit in full.  Callers that only need the roots pass rows=False, layers=False (None for those)."""
        r, l, t = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self.ctx._check(lib().zip_commitment_device_ptrs(self._h, C.byref(r) if rows else None,
                                                         C.byref(l) if layers else None, C.byref(t)),
                        "zip_commitment_device_ptrs")
        return (r.value if rows else None), (l.value if layers else None), t.value

    def roots_ptr(self):
        """Device pointer of the row_count x 32-byte Merkle roots (never materialises or completes anything)."""
        return self.device_ptrs(rows=False, layers=False)[2]

    def download(self, rows=True, layers=True, roots=True):
        c = self.ctx
        R, cw = c.rows_local, c.codeword_len
        rows_a = np.zeros((R * cw, c.k_limbs), dtype=np.uint64) if rows else None
        layers_a = np.zeros((R, 2 * cw - 2, 32), dtype=np.uint8) if (layers and self.has_merkle) else None
        roots_a = np.zeros((R, 32), dtype=np.uint8) if (roots and self.has_merkle) else None
        rc = lib().zip_commit_download(self._h, rows_a.ctypes.data if rows_a is not None else None,
                                       layers_a.ctypes.data if layers_a is not None else None,
                                       roots_a.ctypes.data if roots_a is not None else None)
        c._check(rc, "zip_commit_download")
        return rows_a, layers_a, roots_a

    def open_columns(self, cols, out=None):
        c = self.ctx
        cols = np.ascontiguousarray(cols, dtype=np.uint32)
        nbytes = cols.size * c.rows_local * (8 * c.k_limbs + 8 + 32 * c.depth)
        res = out if out is not None else np.zeros(nbytes, dtype=np.uint8)
        optr, okind = _ptr(res)
        c._check(lib().zip_open_columns(self._h, cols.ctypes.data, cols.size, optr, okind), "zip_open_columns")
        return res

    def open_stream(self, evals, coeffs, cols, q0_mont, field: ZipField, sink, chunk_bytes=0):
        """zip_open_stream: sink(memoryview) is called for each piece of the proof, in stream order;
        return a truthy value from it to abort."""
        c = self.ctx
        ptr, kind = _ptr(evals)
        cols = np.ascontiguousarray(cols, dtype=np.uint32)
        coeffs_c = np.ascontiguousarray(coeffs, dtype=np.int64) if coeffs is not None else None
        q0 = np.ascontiguousarray(q0_mont, dtype=np.uint64) if q0_mont is not None else None

        def _sink(_user, p, n):
            piece = (C.c_uint8 * n).from_address(p) if n else b""
            return 1 if sink(memoryview(piece)) else 0

        cb = PROOF_SINK(_sink)
        rc = lib().zip_open_stream(self._h, ptr, kind, coeffs_c.ctypes.data if coeffs_c is not None else None,
                                   cols.ctypes.data, cols.size, q0.ctypes.data if q0 is not None else None,
                                   C.byref(field), cb, None, chunk_bytes)
        c._check(rc, "zip_open_stream")

    def open_shard(self, evals_d, coeffs, cols, q0_mont, field: ZipField, uprime_part, row_part, wire):
        """zip_open_shard: both partial row combinations (one witness pass) + this shard's rows of every opened column;
        evals_d / outputs are device tensors, coeffs / q0_mont the shard's host slices."""
        c = self.ctx
        cols = np.ascontiguousarray(cols, dtype=np.uint32)
        coeffs_c = np.ascontiguousarray(coeffs, dtype=np.int64) if coeffs is not None else None
        q0 = np.ascontiguousarray(q0_mont, dtype=np.uint64) if q0_mont is not None else None
        rc = lib().zip_open_shard(self._h, _ptr(evals_d)[0], coeffs_c.ctypes.data if coeffs_c is not None else None,
                                  cols.ctypes.data, cols.size, q0.ctypes.data if q0 is not None else None, C.byref(field),
                                  _ptr(uprime_part)[0] if uprime_part is not None else None, _ptr(row_part)[0], _ptr(wire)[0])
        c._check(rc, "zip_open_shard")

    def open(self, evals, coeffs, cols, q0_mont, field: ZipField, out=None):
        """Whole proof stream of MultilinearZip::open (the field elements still need absorbing)."""
        c = self.ctx
        ptr, kind = _ptr(evals)
        cols = np.ascontiguousarray(cols, dtype=np.uint32)
        coeffs_c = np.ascontiguousarray(coeffs, dtype=np.int64) if coeffs is not None else None
        cp = coeffs_c.ctypes.data if coeffs_c is not None else None
        q0 = np.ascontiguousarray(q0_mont, dtype=np.uint64) if q0_mont is not None else None
        total = c.proof_len(cols.size, field.limbs)
        res = out if out is not None else np.zeros(total, dtype=np.uint8)
        optr, okind = _ptr(res)
        rc = lib().zip_open(self._h, ptr, kind, cp, cols.ctypes.data, cols.size,
                            q0.ctypes.data if q0 is not None else None, C.byref(field), optr, okind)
        c._check(rc, "zip_open")
        return res


def merkle_trees(leaves, depth, device=0):
    """MerkleTree::new over [num_trees, 2^depth, limbs] uint64 leaves; returns [num_trees, (2<<depth)-1, 32]."""
    leaves = np.ascontiguousarray(leaves, dtype=np.uint64)
    if leaves.ndim == 2:
        leaves = leaves[None]
    num_trees, n, limbs = leaves.shape
    assert n == 1 << depth
    out = np.zeros((num_trees, (2 << depth) - 1, 32), dtype=np.uint8)
    rc = lib().zip_merkle_trees(device, leaves.ctypes.data, limbs, depth, num_trees, MEM_HOST, out.ctypes.data)
    if rc != ZIP_OK:
        raise ZipError(rc, "zip_merkle_trees")
    return out


class Sumcheck:
    """Device prover state of one sumcheck (zip_sumcheck_*): product of the MLEs, or the CCS form `comb`.  mles: list of CUDA int64/uint64 tensors
    (read in place) or one numpy array [K, 2^nv, limbs] (copied)."""

    def __init__(self, mles, num_vars, degree, field: ZipField, device=0, comb: "SumcheckComb" = None):
        self.field, self.degree = field, degree
        if isinstance(mles, np.ndarray):
            self._keep = np.ascontiguousarray(mles, dtype=np.uint64)
            ptrs = [self._keep[k].ctypes.data for k in range(self._keep.shape[0])]
            kind = MEM_HOST
        else:  # CUDA tensors, or raw device addresses (zip_ccs tables)
            self._keep = list(mles)
            ptrs = [t if isinstance(t, int) else t.data_ptr() for t in self._keep]
            kind = MEM_DEVICE
        arr = (C.c_void_p * len(ptrs))(*ptrs)
        h = C.c_void_p()
        rc = lib().zip_sumcheck_init(device, arr, kind, len(ptrs), num_vars, degree,
                                     C.byref(comb) if comb is not None else None, C.byref(field), C.byref(h))
        if rc != ZIP_OK:
            raise ZipError(rc, "zip_sumcheck_init", strerror(rc))
        self._h = h

    def round(self, r_prev=None):
        out = np.zeros((self.degree + 1, self.field.limbs), dtype=np.uint64)
        rp = None if r_prev is None else np.ascontiguousarray(r_prev, dtype=np.uint64)
        rc = lib().zip_sumcheck_round(self._h, None if rp is None else rp.ctypes.data, out.ctypes.data)
        if rc != ZIP_OK:
            raise ZipError(rc, "zip_sumcheck_round", lib().zip_sumcheck_last_error(self._h).decode())
        return out

    def free(self):
        if getattr(self, "_h", None):
            lib().zip_sumcheck_free(self._h)
        self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Ccs:
    """zip_ccs: the CCS matrices and the tables of SpartanProver::prove in HBM.
    matrices: objects with n_rows, n_cols, row_ptr (uint32), col_idx (uint32), values (int64) numpy arrays."""

    def __init__(self, matrices, s, field: ZipField, device=0):
        self.field, self.s, self.t, self.m = field, s, len(matrices), 1 << s
        self._keep = [(np.ascontiguousarray(M.row_ptr, dtype=np.uint32), np.ascontiguousarray(M.col_idx, dtype=np.uint32),
                       np.ascontiguousarray(M.values, dtype=np.int64)) for M in matrices]
        arr = (SparseMatrix * self.t)()
        for k, (M, (rp, ci, va)) in enumerate(zip(matrices, self._keep)):
            arr[k] = SparseMatrix(M.n_rows, M.n_cols, rp.ctypes.data, ci.ctypes.data, va.ctypes.data)
        h = C.c_void_p()
        rc = lib().zip_ccs_create(device, arr, self.t, s, C.byref(field), C.byref(h))
        if rc != ZIP_OK:
            raise ZipError(rc, "zip_ccs_create", strerror(rc))
        self._h = h

    def _check(self, rc, what):
        if rc != ZIP_OK:
            raise ZipError(rc, what, lib().zip_ccs_last_error(self._h).decode())

    def set_z(self, z):
        """z: numpy int64 (host) or a CUDA int64 tensor"""
        if isinstance(z, np.ndarray):
            z = np.ascontiguousarray(z, dtype=np.int64)
            self._check(lib().zip_ccs_set_z(self._h, z.ctypes.data, z.size, MEM_HOST), "zip_ccs_set_z")
        else:
            self._check(lib().zip_ccs_set_z(self._h, z.data_ptr(), z.numel(), MEM_DEVICE), "zip_ccs_set_z")

    def eq_table(self, r, slot):
        r = np.ascontiguousarray(r, dtype=np.uint64)
        assert r.shape == (self.s, self.field.limbs)
        self._check(lib().zip_ccs_eq_table(self._h, r.ctypes.data, slot), "zip_ccs_eq_table")

    def second_table(self, r_x, gamma):
        """Returns V_s [t, limbs]."""
        r_x = np.ascontiguousarray(r_x, dtype=np.uint64)
        gamma = np.ascontiguousarray(gamma, dtype=np.uint64)
        assert r_x.shape == (self.s, self.field.limbs) and gamma.shape == (self.field.limbs,)
        vs = np.zeros((self.t, self.field.limbs), dtype=np.uint64)
        self._check(lib().zip_ccs_second_table(self._h, r_x.ctypes.data, gamma.ctypes.data, vs.ctypes.data),
                    "zip_ccs_second_table")
        return vs

    def eval_matrices(self, r_x, r_y):
        """mle[M_k](r_x, r_y) for every matrix -> [t, limbs]"""
        r_x = np.ascontiguousarray(r_x, dtype=np.uint64)
        r_y = np.ascontiguousarray(r_y, dtype=np.uint64)
        assert r_x.shape == r_y.shape == (self.s, self.field.limbs)
        out = np.zeros((self.t, self.field.limbs), dtype=np.uint64)
        self._check(lib().zip_ccs_eval_matrices(self._h, r_x.ctypes.data, r_y.ctypes.data, out.ctypes.data),
                    "zip_ccs_eval_matrices")
        return out

    def table(self, which, index=0) -> int:
        """Device address of a table of 2^s field elements."""
        p = C.c_void_p()
        self._check(lib().zip_ccs_table(self._h, which, index, C.byref(p)), "zip_ccs_table")
        return int(p.value)

    def download(self, which, index=0):
        out = np.zeros((self.m, self.field.limbs), dtype=np.uint64)
        self._check(lib().zip_ccs_download(self._h, which, index, out.ctypes.data), "zip_ccs_download")
        return out

    def free(self):
        if getattr(self, "_h", None):
            lib().zip_ccs_free(self._h)
        self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
