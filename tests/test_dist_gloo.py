"""world_size-2 (and 4) rehearsal of the row-sharded path over gloo on the CPU."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from zinc_amd.dist import assemble_columns, shard_rows

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,nv", [(2, 8), (4, 10)])
def test_row_sharded_commit_open_over_gloo(world, nv):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(HERE, "_dist_worker.py"), str(nv)]
    env = dict(os.environ, OMP_NUM_THREADS="2")
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert f"DIST_OK {world} {nv}" in res.stdout


def test_shard_rows_and_assemble_columns():
    assert shard_rows(4096, 8, 3) == (1536, 512)
    with pytest.raises(ValueError):
        shard_rows(4096, 3, 0)
    # two ranks, 3 columns, 2 rows per rank, 4-byte values, 6-byte records
    n_cols, per, kb, rb = 3, 2, 4, 6
    shards = [np.arange(n_cols * per * (kb + rb), dtype=np.uint8) + 100 * g for g in range(2)]
    out = assemble_columns(shards, n_cols, per, kb, rb).reshape(n_cols, 2 * per * (kb + rb))
    for c in range(n_cols):
        s0, s1 = shards[0].reshape(n_cols, -1)[c], shards[1].reshape(n_cols, -1)[c]
        expect = np.concatenate([s0[: per * kb], s1[: per * kb], s0[per * kb:], s1[per * kb:]])
        assert np.array_equal(out[c], expect)
