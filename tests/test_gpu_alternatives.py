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
  INGVIO_GRAM=3            k_feat_gram3 (round 6): one operand panel Z = D^-1/2 L^-1 B (B^T Ns^-1 B = Z^T Z), double-buffered, one barrier per batch
  INGVIO_BIG_GEMM=full     the three products of the large-window solve over all of K (the product skips the chunks in front of a block's first
                           row / column: triangular operands, GemmArgs::k_from, round 6)
  INGVIO_FEW=off           few filters (up to 64 per launch) on the kernels of a full batch: chunk partials added inside the solve, k_info_apply
                           (the product sums them with k_chunk_sum first and applies with one wave per tile, k_apply_*_flat)
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
    ("INGVIO_FEW", "off", ["tests/test_gpu_parity.py", "tests/test_gpu_pinning.py", "-k",
                           "test_msckf_small or test_window_size_classes or test_ragged or test_consecutive_frames or test_sigma_and_prior_scale_sweep"]),
    ("INGVIO_GRAM", "3", ["tests/test_gpu_parity.py", "tests/test_gpu_pinning.py", "-k",
                          "test_full_n249_batch_vs_oracle or test_msckf_small or test_window_size_classes or test_ragged or test_mono_gate or test_sigma_and_prior_scale_sweep"]),
    ("INGVIO_BIG_GEMM", "full", ["tests/test_gpu_parity.py", "tests/test_gpu_pinning.py", "-k", "large_window or config5 or big"]),
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


FEW_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import bench
from ingvio_amd import capi, synth
B = int(sys.argv[2])
ctx = capi.Context(batch=B, n_max=256, c_max=11, f_max=150, m_max=64)
filters, steps, frames, infos = bench.build_batch(ctx, B, 0, 150, 11, 6, 52)
ctx.snapshot(); pr = synth.PARAMS
ctx.frame_stage(0, steps, frames, filters[0].sigma(), 1, pr["sigma_cb"], pr["sigma_rw"])
ctx.frame_run(restore_prior=True)
dx, acc, rows = ctx.frame_fetch()
np.savez(sys.argv[1], P=np.stack([np.asarray(ctx.cov_get(b)) for b in range(B)]), dx=dx, acc=acc, rows=rows)
"""


@pytest.mark.parametrize("B", [1, 3])
def test_few_filter_path_is_bit_identical_to_the_full_batch_kernels(tmp_path, B):
    """k_chunk_sum + k_apply_T_flat / k_apply_sym_flat (what launches of up to 64 filters take) form the same sums and products in the same order as
    the chunk loop of k_info_solve and k_info_apply: posterior, dx and accept masks of a whole frame (N = 249, 150 features, fused
    marginalisation) are equal to the last bit."""
    if not os.path.exists(ALT_LIB):
        pytest.skip("build_var/alt/libingvio_hip.so not built")
    import numpy as np
    outs = []
    for tag, extra in (("few", {}), ("full", {"INGVIO_HIP_LIB": ALT_LIB, "INGVIO_FEW": "off"})):
        env = dict(os.environ); env.update(extra)
        out = str(tmp_path / ("%s.npz" % tag))
        r = subprocess.run([sys.executable, "-c", FEW_CHILD % ROOT, out, str(B)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(out))
    a, b = outs
    assert np.array_equal(a["acc"], b["acc"]) and np.array_equal(a["rows"], b["rows"]) and a["acc"].sum() > 50 * B
    assert np.array_equal(a["dx"], b["dx"]), float(np.max(np.abs(a["dx"] - b["dx"])))
    assert np.array_equal(a["P"], b["P"]), float(np.max(np.abs(a["P"] - b["P"])))


@pytest.mark.parametrize("parts", [2, 3])
def test_split_frame_step_is_bit_identical(parts):
    """ingvio_set_frame_parts (round 6): the batch dealt to slices on their own streams, throughput segments chained by events, the slices
    NOT joined at the end of ingvio_frame_run - consecutive steps, a fetch in between, a stage, ingvio_sync: every result equal to the
    unsplit step's to the last bit (same kernels, same arguments per filter)."""
    import numpy as np
    from ingvio_amd import capi, host, synth
    B, C, F = 48, 11, 30
    N = 21 + 6 * C
    ctx = capi.Context(batch=B, n_max=((N + 15) // 16) * 16, c_max=C, f_max=F, m_max=32)
    cases = [synth.build_case(lambda P, b=b: capi.DeviceCov(ctx, b, P), host.imu_transition, seed=300 + b, F=F, C=C, n_gnss=0, n_landmarks=0) for b in range(B)]
    ctx.snapshot()
    stage = lambda: ctx.frame_stage(0, [c[1] for c in cases], [c[2] for c in cases], cases[0][1]["sigma"])
    stage()
    outs = []
    for p in (1, parts):
        ctx.set_frame_parts(p)
        for _ in range(3):                                              # pipelined steps, nothing joined in between
            ctx.frame_run(restore_prior=True)
        dx, acc, rows = ctx.frame_fetch()
        stage()                                                         # an entry point other than frame_run joins the slices first
        ctx.frame_run(restore_prior=True)
        ctx.frame_run(restore_prior=False)                              # and a step that continues from the posterior
        ctx.sync()
        dx2, acc2, rows2 = ctx.frame_fetch()
        outs.append((dx, acc, rows, dx2, acc2, [ctx.cov_get(b) for b in (0, B // 2 - 1, B // 2, B - 1)]))
        ctx.restore()
    a, b = outs
    assert a[1].sum() > 10 * B
    for x, y in zip(a[:5], b[:5]):
        assert np.array_equal(x, y)
    for x, y in zip(a[5], b[5]):
        assert np.array_equal(x, y)
    ctx.close()
