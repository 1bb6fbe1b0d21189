"""CCS instances of the reference's tests as plain CSR arrays (test data, no arithmetic).

  dummy_ccs(z)          get_dummy_ccs_Z_from_z            src/ccs/test_utils.rs:89-121
  dummy_ccs_from_len    get_dummy_ccs_Z_from_z_length     src/ccs/test_utils.rs:161-171 (our own PRNG)
  vitalik_ccs(x)        get_test_ccs_stuff_Z              src/ccs/ccs_z.rs:231-318 (x^3 + x + 5 = y, padded to 8)
"""
import numpy as np


class CsrMatrix:
    def __init__(self, n_rows, n_cols, rows):
        """rows: list (len <= n_rows) of lists of (value, col) -- SparseMatrix.coeffs."""
        self.n_rows, self.n_cols = n_rows, n_cols
        ptr, cols, vals = [0], [], []
        for r in range(n_rows):
            for v, c in (rows[r] if r < len(rows) else []):
                cols.append(c)
                vals.append(v)
            ptr.append(len(cols))
        self.row_ptr = np.array(ptr, dtype=np.uint32)
        self.col_idx = np.array(cols, dtype=np.uint32)
        self.values = np.array(vals, dtype=np.int64)

    @classmethod
    def diagonal(cls, values):
        n = len(values)
        m = cls.__new__(cls)
        m.n_rows = m.n_cols = n
        m.row_ptr = np.arange(n + 1, dtype=np.uint32)
        m.col_idx = np.arange(n, dtype=np.uint32)
        m.values = np.ascontiguousarray(values, dtype=np.int64)
        return m


class CcsInstance:
    def __init__(self, m, n, s, s_prime, d, matrices, S, c, z):
        self.m, self.n, self.s, self.s_prime, self.d = m, n, s, s_prime, d
        self.matrices, self.S, self.c = matrices, S, list(c)
        self.z = np.ascontiguousarray(z, dtype=np.int64)  # x || 1 || w
        self.t, self.q = len(matrices), len(S)

    @property
    def masks(self):
        return np.array([sum(1 << j for j in Si) for Si in self.S], dtype=np.uint32)


def dummy_ccs(z):
    z = np.ascontiguousarray(z, dtype=np.int64)
    n = len(z)
    s = n.bit_length() - 1
    assert 1 << s == n
    ident = CsrMatrix.diagonal(np.ones(n, dtype=np.int64))
    return CcsInstance(n, n, s, s, 2, [ident, ident, CsrMatrix.diagonal(z)], [[0, 1], [2]], [1, -1], z)


def dummy_ccs_from_len(n, seed=0x5A494E43):
    import _oracle  # SplitMix64 stream only
    z = _oracle.splitmix64(seed, n).view(np.int64).copy()
    z[1] = 1  # pub_io_len = 1: z = (x, 1, w)
    return dummy_ccs(z)


def _dense(rows):
    return [[(v, c) for c, v in enumerate(r) if v] for r in rows]


def vitalik_ccs(x, break_witness=False):
    A = _dense([[1, 0, 0, 0, 0, 0], [0, 0, 0, 1, 0, 0], [1, 0, 0, 0, 1, 0], [0, 5, 0, 0, 0, 1]])
    B = _dense([[1, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0], [0, 1, 0, 0, 0, 0], [0, 1, 0, 0, 0, 0]])
    Cm = _dense([[0, 0, 0, 1, 0, 0], [0, 0, 0, 0, 1, 0], [0, 0, 0, 0, 0, 1], [0, 0, 1, 0, 0, 0]])
    z = [x, 1, x ** 3 + x + 5, x * x, x ** 3, x ** 3 + x]
    if break_witness:  # zinc/tests.rs:166-169: wit.w_ccs[3] = 0
        z[2 + 3] = 0
    mats = [CsrMatrix(8, 8, m) for m in (A, B, Cm)]  # ccs.pad(.., 8): pad_rows / pad_cols only bump the sizes
    return CcsInstance(8, 8, 3, 3, 2, mats, [[0, 1], [2]], [1, -1], z)
