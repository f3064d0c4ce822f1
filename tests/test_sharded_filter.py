"""SURVEY 8(e), optional mode: ONE filter whose features are dealt to two ranks, one all-reduce of [A | b] per frame
(ingvio_frame_run_phase / ingvio_info_set, ingvio_amd/parallel.py::sharded_frame_update).  Two processes (gloo) on the GPU box."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("clones,feats", [(11, 150), (30, 120)])
def test_feature_sharded_filter_two_ranks(clones, feats):
    env = dict(os.environ, SHARD_CLONES=str(clones), SHARD_FEATS=str(feats), HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29540 + clones), os.path.join(ROOT, "tests", "worker_sharded_filter.py")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    assert "SHARDED world=2" in p.stdout and "replicas_equal=True" in p.stdout, p.stdout[-2000:]


def test_shard_features_partition():
    """CPU: the feature deal is a partition of the frame's features (every feature on exactly one rank)."""
    import numpy as np
    from ingvio_amd.parallel import shard_features
    frame = dict(pf=np.zeros((10, 3)), anchor=np.arange(10), obs_mask=np.arange(10), uv=np.zeros((10, 4, 4)), dof=np.arange(10), other=1)
    seen = []
    for r in range(3):
        loc, keep = shard_features(frame, 3, r)
        assert loc["other"] == 1 and len(loc["pf"]) == len(keep) and np.array_equal(loc["anchor"], keep)
        seen += list(keep)
    assert sorted(seen) == list(range(10))
