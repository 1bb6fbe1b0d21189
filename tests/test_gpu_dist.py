"""The torch.distributed variant of row sharding with the PRODUCT backend: zinc_amd.dist.RowShardedZip + HipBackend,
two ranks over gloo, both on GPU 0, real all-gathers (staged through the host under gloo), exact sum of the partial
rows on the device -- the sharded roots and proof equal the unsharded oracle's.  The two-rank program runs from
conftest.pytest_collection_finish, before this process touches the GPU; here only its result is looked at.
(The single-process path, zip_mctx, is tested in test_gpu_parity.py.)"""
import pytest

pytestmark = pytest.mark.gpu


def test_row_sharded_hip_backend_two_ranks_over_gloo(request):
    res = getattr(request.config, "_zinc_gpu_dist", None)
    assert res is not None, "conftest did not run the two-rank program"
    assert res["returncode"] == 0, (res["stdout"][-2000:] + res["stderr"][-4000:])
    assert "DIST_OK 2 12" in res["stdout"] and "hip" in res["stdout"]
