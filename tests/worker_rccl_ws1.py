"""RCCL at world size 1 on the GPU box (run by tests/test_gpu_pinning.py::test_rccl_world_size_1 in a fresh interpreter: torch must
initialise its HIP runtime before libingvio_hip.so is loaded, the order bench.py uses): the timing barrier / MAX all-reduce /
all-gather of the multi-GPU harness, the feature-sharded filter's device-resident exchange through RCCL, and the protocol of the
split frame step."""
import os, sys
sys.path.insert(0, os.environ["INGVIO_ROOT"])
import numpy as np
from ingvio_amd.parallel import Group
grp = Group(backend="nccl", force_init=True)                 # torch first, as bench.py does at N > 1
assert grp.world == 1 and grp.backend == "nccl"
grp.barrier()
assert grp.max_over_ranks(1.25) == 1.25
summ = grp.gather_summaries([1.0, 2.0, 3.0])
assert summ.shape == (1, 3) and summ[0, 1] == 2.0
assert grp.shard(4096) == (0, 4096) and grp.gather_scalars(0.5) == [0.5]
# the covariance engine in the same process as an initialised RCCL communicator (one process per GPU)
from ingvio_amd import capi
ctx = capi.Context(batch=2, n_max=32, c_max=2, f_max=4, m_max=16)
P = np.eye(21) * 0.5
ctx.cov_set(0, P); ctx.cov_set(1, 2 * P)
assert np.array_equal(ctx.cov_get(1), 2 * P)
grp.barrier()
# the feature-sharded filter's ONE exchange step through RCCL itself, in place on the library's device buffer
# (ingvio_info_reduce -> all_reduce on a zero-copy view -> ingvio_info_commit), against the host-staged variant
from ingvio_amd import host, synth
from ingvio_amd.parallel import sharded_frame_update
ctx2 = capi.Context(batch=1, n_max=96, c_max=6, f_max=24, m_max=64)
flt, step, frame, info = synth.build_case(lambda Q: capi.DeviceCov(ctx2, 0, Q), host.imu_transition, seed=5, F=24, C=6, n_gnss=0, n_landmarks=0)
ctx2.snapshot()
dx1, acc1, rows1 = sharded_frame_update(ctx2, grp, 0, step, frame, step["sigma"], 0, 0.2, 0.2, device_exchange=True)
P1 = ctx2.cov_get(0)
ctx2.restore()
dx2, acc2, rows2 = sharded_frame_update(ctx2, grp, 0, step, frame, step["sigma"], 0, 0.2, 0.2, device_exchange=False)
P2 = ctx2.cov_get(0)
assert rows1 == rows2 == 36 and np.array_equal(acc1, acc2) and len(acc1) > 10
assert np.allclose(P1, P2, rtol=0, atol=1e-14 * np.abs(P2).max()) and np.allclose(dx1, dx2, rtol=0, atol=1e-12)
# protocol of the split step: phase 2 without phase 1, a second phase 1, or any covariance-changing call in between is refused
ctx2.restore()
ctx2.frame_stage(0, [step], [frame], step["sigma"], 0, 0.2, 0.2)
for bad in (lambda: ctx2.frame_run_phase(2), ):
    try:
        bad(); raise SystemExit("phase 2 without phase 1 was accepted")
    except capi.IngvioError as e:
        assert e.code == capi.E_ARG
ctx2.frame_run_phase(1)
for bad in (lambda: ctx2.frame_run_phase(1), lambda: ctx2.frame_run(), lambda: ctx2.cov_set(0, P2), lambda: ctx2.snapshot(),
            lambda: ctx2.marginalize(0, [21], 6)):
    try:
        bad(); raise SystemExit("a mutating call was accepted between the phases")
    except capi.IngvioError as e:
        assert e.code == capi.E_ARG
assert ctx2.cov_get(0).shape[0] == P2.shape[0] + 6          # reading is allowed: cloned, not yet updated
ctx2.frame_run_phase(2)
assert np.allclose(ctx2.cov_get(0), P2, rtol=0, atol=1e-14 * np.abs(P2).max())
try:
    ctx2.frame_run_phase(2); raise SystemExit("phase 2 twice was accepted")
except capi.IngvioError as e:
    assert e.code == capi.E_ARG
ctx2.close()
ctx.close(); grp.close()
print("RCCL_OK")
