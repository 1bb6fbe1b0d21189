"""Row sharding of ONE polynomial over the GPUs of a node: one process per GPU,
torch.distributed (backend "nccl" == RCCL over xGMI on ROCm; "gloo" for CPU rehearsals).

SURVEY.md §8e: every witness row is encoded and hashed independently (commit.rs:71-74,172-178),
so rank g owns rows [g*R/G, (g+1)*R/G) -- a contiguous slice of the evaluation vector.  The path
has exactly two exchange steps, both tiny next to the per-GPU work:

  commit : all-gather of the per-rank Merkle roots                  (R * 32 bytes in total)
  open   : all-gather of the per-rank partial row combinations      (G * row_len * (64 + 8*FL) bytes)
           followed by an exact local sum (512-bit integer adds / modular adds are associative,
           so the result is bit-identical to the unsharded one)

Column openings need no exchange: a rank emits its row slice of every opened column and
`assemble_columns` interleaves the slices into proof-stream order wherever the proof is collected.

The compute goes through a `backend` (default: the HIP library via zinc_amd.cabi).  Tests on a
GPU-less box inject a backend of their own; this module never computes anything itself.
"""
import numpy as np


def shard_rows(num_rows: int, world: int, rank: int):
    """Rows [begin, begin + count) of rank `rank`; num_rows is a power of two, world must divide it."""
    if world < 1 or num_rows % world:
        raise ValueError(f"world size {world} does not divide num_rows {num_rows}")
    per = num_rows // world
    return rank * per, per


def assemble_columns(wire_shards, n_cols: int, rows_per_rank: int, k_bytes: int, rec_bytes: int) -> np.ndarray:
    """Interleaves per-rank column openings (each: n_cols x [rows_per_rank values | rows_per_rank
    path records]) into the reference's stream order: per column all values, then all records
    (open_z.rs:124-143)."""
    vals, recs = [], []
    for w in wire_shards:
        w = np.asarray(w, dtype=np.uint8).reshape(n_cols, rows_per_rank * (k_bytes + rec_bytes))
        vals.append(w[:, : rows_per_rank * k_bytes])
        recs.append(w[:, rows_per_rank * k_bytes:])
    return np.concatenate(vals + recs, axis=1).reshape(-1)


class HipBackend:
    """The product backend: a row-shard zip_ctx on this process's GPU."""

    def __init__(self, num_vars, perm1, perm2, row_begin, row_count, device):
        import torch

        from . import cabi

        self.torch, self.cabi = torch, cabi
        self.ctx = cabi.ZipContext(num_vars, perm1, perm2, device=device, row_begin=row_begin, row_count=row_count)
        self.device = torch.device("cuda", device)

    def geometry(self):
        c = self.ctx
        return c.row_len, c.num_rows, c.codeword_len, c.depth

    def commit(self, evals, cols=None):
        """Asynchronous: the persistent commit kernel is enqueued and the handle returned; `roots(com)` waits for it.
        cols: the columns the open will ask for (known before the commit in the prover's flow): a hinted commit."""
        com, _ = self.ctx.commit(evals, want_roots=False, hint_cols=cols)
        return com

    def roots(self, com):
        """This shard's roots as a device tensor, once its commit kernel has finished."""
        roots_ptr = com.roots_ptr()
        self.ctx.synchronize()
        holder = type("_H", (), {})()
        holder.__cuda_array_interface__ = {"shape": (self.ctx.rows_local, 32), "typestr": "|u1",
                                           "data": (roots_ptr, False), "version": 2}
        return self.torch.as_tensor(holder, device=self.device)

    def open_shard(self, com, evals, coeffs, cols, q0, field, upart, fpart):
        """One call: both partial row combinations in ONE witness pass + this shard's rows of every opened column,
        pipelined behind the commit kernel (zip_open_shard)."""
        wire = self.torch.empty(len(cols) * self.ctx.rows_local * (32 + 8 + 32 * self.ctx.depth), dtype=self.torch.uint8,
                                device=self.device)
        if isinstance(evals, np.ndarray):  # the entry point takes the shard's witness rows in device memory
            evals = self.torch.from_numpy(np.ascontiguousarray(evals, dtype=np.int64)).to(self.device)
        com.open_shard(evals, coeffs, cols, q0, field, upart, fpart, wire)
        return wire

    def empty(self, shape, dtype):
        return self.torch.empty(shape, dtype=dtype, device=self.device)

    def open_testing(self, evals, coeffs, out):
        self.ctx.open_testing(evals, coeffs, out=out)

    def open_eval(self, evals, q0, field, out):
        self.ctx.open_eval(evals, q0, field, out=out)

    def open_columns(self, com, cols):
        out = self.torch.empty(len(cols) * self.ctx.rows_local * (32 + 8 + 32 * self.ctx.depth), dtype=self.torch.uint8,
                               device=self.device)
        com.open_columns(cols, out=out)
        return out

    def sum_partials(self, uparts, fparts, n_parts, field, uprime_out, row_out):
        self.torch.cuda.synchronize()
        self.ctx.sum_partials(uparts, fparts, n_parts, field, uprime_out, row_out)
        self.ctx.synchronize()


class RowShardedZip:
    """commit / open of one polynomial whose rows are sharded over `group` (one rank per GPU)."""

    def __init__(self, num_vars, perm1, perm2, group=None, backend=None, device=None):
        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.group = torch, dist, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        from .cabi import geometry

        self.row_len, self.num_rows, self.codeword_len = geometry(num_vars)
        self.row_begin, self.row_count = shard_rows(self.num_rows, self.world, self.rank)
        if backend is None:
            if device is None:
                device = torch.cuda.current_device()
            backend = HipBackend(num_vars, perm1, perm2, self.row_begin, self.row_count, device)
        self.backend = backend

    def local_slice(self, evals_full):
        """This rank's contiguous slice of the evaluation vector."""
        return evals_full[self.row_begin * self.row_len:(self.row_begin + self.row_count) * self.row_len]

    def _all_gather(self, t):
        if self.world == 1:
            return t.reshape((1,) + tuple(t.shape))
        # concatenated along dim 0 (the layout both RCCL and gloo accept), viewed as [world, ...]
        out = self.backend.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), t.dtype)
        if t.is_cuda and self.dist.get_backend(self.group) == "gloo":
            # rehearsal on a box without RCCL peers (several ranks on one GPU): the exchange goes through the host
            host = self.torch.empty(out.shape, dtype=out.dtype)
            self.dist.all_gather_into_tensor(host, t.contiguous().cpu(), group=self.group)
            out.copy_(host)
        else:
            self.dist.all_gather_into_tensor(out, t.contiguous(), group=self.group)
        return out.reshape((self.world,) + tuple(t.shape))

    def commit(self, evals_local, cols=None, gather_roots=True):
        """-> (local commitment handle, roots of ALL rows [num_rows, 32] on every rank).
        cols: the opening's columns when they are known before the commit (prover.rs:316): a hinted commit.
        gather_roots=False returns (handle, None) at once -- the commit kernel keeps running, an open enqueued now is
        pipelined behind it -- and `gather_roots(com)` collects the commitment afterwards."""
        if hasattr(self.backend, "roots"):
            com = self.backend.commit(evals_local, cols)
            return com, (self.gather_roots(com) if gather_roots else None)
        com, roots_local = self.backend.commit(evals_local)  # (a test backend: synchronous)
        return com, self._all_gather(roots_local).reshape(self.num_rows, 32)

    def gather_roots(self, com):
        """The one exchange of the commit: every rank's roots, all-gathered (RCCL over xGMI)."""
        return self._all_gather(self.backend.roots(com)).reshape(self.num_rows, 32)

    def open(self, com, evals_local, coeffs, cols, q0_mont, field):
        """coeffs / q0_mont: the FULL challenge vectors (every rank derives the same transcript).
        -> (u' [row_len, 8], evaluation row [row_len, FL] Montgomery limbs, local column-opening wire bytes)."""
        t = self.torch
        sl = slice(self.row_begin, self.row_begin + self.row_count)
        fl = field.limbs
        upart = self.backend.empty((self.row_len, 8), t.int64)
        fpart = self.backend.empty((self.row_len, fl), t.int64)
        if hasattr(self.backend, "open_shard"):  # one call: one witness pass, openings pipelined behind the commit
            wire = self.backend.open_shard(com, evals_local, np.ascontiguousarray(coeffs[sl]), cols,
                                           np.ascontiguousarray(q0_mont[sl]), field, upart, fpart)
        else:
            self.backend.open_testing(evals_local, np.ascontiguousarray(coeffs[sl]), upart)
            self.backend.open_eval(evals_local, np.ascontiguousarray(q0_mont[sl]), field, fpart)
            wire = self.backend.open_columns(com, cols)
        uall, fall = self._all_gather(upart), self._all_gather(fpart)
        uprime = self.backend.empty((self.row_len, 8), t.int64)
        row = self.backend.empty((self.row_len, fl), t.int64)
        self.backend.sum_partials(uall, fall, self.world, field, uprime, row)
        return uprime, row, wire
