"""Worker of tests/test_dist_gloo.py: world_size ranks over gloo on the CPU.  The compute backend
is the ORACLE (this is a rehearsal of the sharding / collective logic of zinc_amd.dist, not of the
kernels): sharded commit + open must reproduce the unsharded oracle bit for bit."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import _oracle as orc  # noqa: E402
from zinc_amd.dist import RowShardedZip, assemble_columns, shard_rows  # noqa: E402

MODULUS = 106319353542452952636349991594949358997917625194731877894581586278529202198383


class OracleBackend:
    """CPU stand-in with the interface of zinc_amd.dist.HipBackend."""

    def __init__(self, zfull, row_begin, row_count):
        self.z = orc.Zip(zfull.num_vars, perm1=zfull.perm1, perm2=zfull.perm2,
                         geometry=(zfull.row_len, row_count, zfull.codeword_len))
        self.f = orc.make_field(MODULUS, 4)

    def commit(self, evals):
        rows, layers, roots = self.z.commit(evals)
        return (rows, layers), torch.from_numpy(roots.copy())

    def empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype)

    def open_testing(self, evals, coeffs, out):
        rc, u = self.z.combine_rows_int(coeffs, evals)
        assert rc == 0
        out.copy_(torch.from_numpy(u.view(np.int64)))

    def open_eval(self, evals, q0, field, out):
        out.copy_(torch.from_numpy(self.z.combine_rows_field(self.f, q0, evals).view(np.int64)))

    def open_columns(self, com, cols):
        rows, layers = com
        z = self.z
        parts = []
        for c in cols:
            vals = rows.reshape(z.num_rows, z.codeword_len, 4)[:, int(c), :].astype("<u8").tobytes()
            recs = b"".join(int(z.depth).to_bytes(8, "big") + orc.merkle_path(z.depth, layers[r], int(c)).tobytes()
                            for r in range(z.num_rows))
            parts.append(vals + recs)
        return torch.from_numpy(np.frombuffer(b"".join(parts), dtype=np.uint8).copy())

    def sum_partials(self, uparts, fparts, n_parts, field, uprime_out, row_out):
        u = uparts.numpy().view(np.uint64)
        fp = fparts.numpy().view(np.uint64)
        for c in range(u.shape[1]):
            s = sum(orc.limbs_to_int(u[g, c], signed=True) for g in range(n_parts))
            uprime_out[c] = torch.from_numpy(np.array(orc.int_to_limbs(s, 8), dtype=np.uint64).view(np.int64))
            m = sum(orc.limbs_to_int(fp[g, c]) for g in range(n_parts)) % MODULUS
            row_out[c] = torch.from_numpy(np.array(orc.int_to_limbs(m, 4), dtype=np.uint64).view(np.int64))


def main():
    nv = int(sys.argv[1])
    hip = len(sys.argv) > 2 and sys.argv[2] == "--hip"  # the PRODUCT backend, every rank on GPU 0 (tests/test_gpu_dist.py)
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    zfull = orc.Zip(nv)
    f = orc.make_field(MODULUS, 4)
    evals = orc.splitmix64(0x5A494E43, 1 << nv)
    begin, count = shard_rows(zfull.num_rows, world, rank)
    if hip:
        from zinc_amd import cabi
        from zinc_amd.dist import HipBackend

        torch.cuda.set_device(0)
        backend = HipBackend(nv, zfull.perm1, zfull.perm2, begin, count, 0)
    else:
        backend = OracleBackend(zfull, begin, count)
    sharded = RowShardedZip(nv, zfull.perm1, zfull.perm2, backend=backend)
    assert (sharded.row_begin, sharded.row_count) == (begin, count)

    point = orc.point_to_field(f, [1] * nv)
    rows_o, layers_o, roots_o = zfull.commit(evals)
    proof_o, cols, coeffs = zfull.open(f, evals, rows_o, layers_o, point, orc.new_transcript())
    lr = zfull.num_rows.bit_length() - 1
    q0 = orc.build_eq_x_r(f, point[nv - lr:])

    class _F:
        limbs = 4

    field = cabi.make_field(MODULUS, 4) if hip else _F()
    if hip:
        # the prover's order: the columns are known before the commit (hinted commit, enqueued), the shard's open is
        # pipelined behind it (zip_open_shard: one witness pass), the roots are all-gathered at the end
        com, none = sharded.commit(sharded.local_slice(evals), cols, gather_roots=False)
        assert none is None
        uprime, row, wire = sharded.open(com, sharded.local_slice(evals), coeffs, cols, q0, field)
        roots_all = sharded.gather_roots(com)
    else:
        com, roots_all = sharded.commit(sharded.local_slice(evals))
        uprime, row, wire = sharded.open(com, sharded.local_slice(evals), coeffs, cols, q0, field)
    assert np.array_equal(roots_all.cpu().numpy(), roots_o), "all-gathered roots differ"
    uprime, row, wire = uprime.cpu(), row.cpu(), wire.cpu()
    wires = [torch.empty_like(wire) for _ in range(world)]
    dist.all_gather(wires, wire)
    if rank == 0:
        z = zfull
        body = assemble_columns([w.numpy() for w in wires], len(cols), count, 32, 8 + 32 * z.depth)
        row_be = b"".join(orc.limbs_to_int(r).to_bytes(32, "big") for r in row.numpy().view(np.uint64))
        proof = np.concatenate([uprime.numpy().view(np.uint8).reshape(-1), body, np.frombuffer(row_be, dtype=np.uint8)])
        assert proof.size == proof_o.size and np.array_equal(proof, proof_o), "sharded proof differs from the unsharded one"
        print("DIST_OK", world, nv, proof.size, "hip" if hip else "oracle", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
