import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def frame_from_golden(z, prefix):
    keys = ["clone_idx", "clone_R", "clone_p", "pf", "anchor", "obs_mask", "uv", "dof", "stereo",
            "R_cl2cr", "t_cl2cr", "noise", "chi2_table"]
    fr = {k: z[prefix + k] for k in keys}
    fr["stereo"] = int(fr["stereo"]); fr["noise"] = float(fr["noise"])
    return fr


def rel_err(A, B):
    return float(np.linalg.norm(np.asarray(A) - np.asarray(B)) / max(np.linalg.norm(np.asarray(B)), 1e-300))


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.build()
    return oracle
