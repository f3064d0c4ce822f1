"""The alternative code paths (kept for A/B measurements) must stay CORRECT.  Since round 5 they are NOT in the product
library: libingvio_hip.so carries the shipped kernels and one stereo gate fallback (INGVIO_GATE=4); everything else is compiled
only with -DINGVIO_ALT_KERNELS into build_var/alt/libingvio_hip.so (ingvio_amd/build.py::build_alt), which these tests load through
INGVIO_HIP_LIB.  Each path is switched on through its environment variable in a child process (the switches are read once per
process) that runs the parity tests covering it.  One toggle per case:

  INGVIO_GATE=3            first-generation gate (k_feat_gate3 / k_feat_gate3_big) instead of the difference-coordinate one
  INGVIO_GATE=4            the difference-coordinate gate with ONE feature per wave (k_feat_gate4, round 3) instead of four (k_feat_gate5)
  INGVIO_INFO_GAUGE=off    full-size symmetric solve (no reduction to difference coordinates of a reference clone)
  INGVIO_INFO_SOLVE=gj     Gauss-Jordan on A Pcc + s^2 I
  INGVIO_APPLY_TW=2        k_info_apply with two tile columns per step
  INGVIO_LM_FRONT=split    landmark update with k_lm_build + k_lm_products (compacting) instead of the fused front
  INGVIO_LM_SOLVE=sweep    landmark / dense-H update on the Cholesky sweep out of L2 instead of the register-resident solve
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
ALT_LIB = os.path.join(ROOT, "build_var", "alt", "libingvio_hip.so")
PRODUCT_ONLY = {("INGVIO_GATE", "4")}                    # the one fallback the product library keeps

CASES = [
    ("INGVIO_GATE", "4", ["tests/test_gpu_parity.py", "tests/test_gpu_pinning.py", "-k",
                          "test_full_n249_batch_vs_oracle or test_msckf_small or test_window_size_classes or test_ragged or test_gate or test_sigma_and_prior_scale_sweep"]),
    ("INGVIO_GATE", "3", ["tests/test_gpu_parity.py", "-k", "test_full_n249_batch_vs_oracle or test_large_window_vs_oracle or test_msckf_small"]),
    ("INGVIO_INFO_GAUGE", "off", ["tests/test_gpu_parity.py", "-k", "test_full_n249_batch_vs_oracle or test_large_window_vs_oracle or test_window_size_classes"]),
    ("INGVIO_INFO_SOLVE", "gj", ["tests/test_gpu_parity.py", "-k", "test_full_n249_batch_vs_oracle or test_window_size_classes"]),
    ("INGVIO_APPLY_TW", "2", ["tests/test_gpu_parity.py", "-k", "test_full_n249_batch_vs_oracle or test_window_size_classes or test_consecutive_frames"]),
    ("INGVIO_LM_FRONT", "split", ["tests/test_landmark_batch.py"]),
    ("INGVIO_LM_SOLVE", "sweep", ["tests/test_landmark_batch.py"]),
]


@pytest.mark.parametrize("var,value,args", CASES, ids=["%s=%s" % (c[0], c[1]) for c in CASES])
def test_alternative_path_stays_correct(var, value, args):
    env = dict(os.environ)
    env[var] = value
    if (var, value) not in PRODUCT_ONLY:
        if not os.path.exists(ALT_LIB):
            pytest.skip("build_var/alt/libingvio_hip.so not built (python -c 'from ingvio_amd import build; build.build_alt()')")
        env["INGVIO_HIP_LIB"] = ALT_LIB
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"] + args, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    tail = "\n".join(r.stdout.strip().splitlines()[-15:])
    assert r.returncode == 0, "%s=%s:\n%s\n%s" % (var, value, tail, r.stderr[-2000:])
    assert " passed" in tail and " failed" not in tail, tail
