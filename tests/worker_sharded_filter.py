"""Feature-sharded single filter (optional mode), run as `python -m torch.distributed.run --nproc-per-node 2 ...`:
both ranks replicate ONE filter (on the same GPU when only one is present), each gates + accumulates half of the features, ONE
all-reduce of [A | b], identical solve + apply.  Rank 0 compares with the unsharded frame on a second context."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ingvio_amd.parallel import Group, sharded_frame_update      # noqa: E402  (torch.distributed first: its HIP runtime must load before ours)

grp = Group(backend="gloo", force_init=True)
from ingvio_amd import capi, host, synth                         # noqa: E402

C = int(os.environ.get("SHARD_CLONES", "11")); F = int(os.environ.get("SHARD_FEATS", "150"))
n_lm = 52 if C <= 16 else 100
n_max = ((21 + 6 + 3 * n_lm + 6 * C + 15) // 16) * 16
ctx = capi.Context(batch=1, n_max=n_max, c_max=C, f_max=F, m_max=64)
flt, step, frame, info = synth.build_case(lambda P: capi.DeviceCov(ctx, 0, P), host.imu_transition, seed=77, F=F, C=C, n_landmarks=n_lm)
pr = synth.PARAMS
dx, acc_ids, rows = sharded_frame_update(ctx, grp, 0, step, frame, step["sigma"], step["enable_gnss"], pr["sigma_cb"], pr["sigma_rw"])
P = ctx.cov_get(0)
# every replica ends with the same posterior
chk = grp.gather_summaries([float(np.abs(P).sum()), float(np.abs(dx).sum()), float(len(acc_ids))])
ok = bool(np.allclose(chk[:, 0], chk[0, 0], rtol=1e-13) and np.allclose(chk[:, 1], chk[0, 1], rtol=1e-10))
if grp.rank == 0:
    ref = capi.Context(batch=1, n_max=n_max, c_max=C, f_max=F, m_max=64)
    flt2, step2, frame2, _ = synth.build_case(lambda Q: capi.DeviceCov(ref, 0, Q), host.imu_transition, seed=77, F=F, C=C, n_landmarks=n_lm)
    ref.frame_stage(0, [step2], [frame2], step2["sigma"], step2["enable_gnss"], pr["sigma_cb"], pr["sigma_rw"], max_accept=0, compress_rule=1)
    ref.frame_run()
    dxr, accr, rowsr = ref.frame_fetch()
    Pr = ref.cov_get(0)
    e_cov = float(np.linalg.norm(P - Pr) / np.linalg.norm(Pr))
    e_dx = float(np.linalg.norm(dx[:Pr.shape[0] + 6] - dxr[0, :Pr.shape[0] + 6]) / max(np.linalg.norm(dxr[0]), 1e-300))
    n_acc = int(chk[:, 2].sum())
    print("SHARDED world=%d C=%d F=%d  replicas_equal=%s  cov_err=%.2e  dx_err=%.2e  accepted %d (unsharded %d)  n=%d"
          % (grp.world, C, F, ok, e_cov, e_dx, n_acc, int(accr[0, :F].sum()), P.shape[0]))
    assert ok and e_cov < 1e-10 and e_dx < 1e-8 and n_acc == int(accr[0, :F].sum()) and P.shape == Pr.shape
    ref.close()
ctx.close()
grp.close()
