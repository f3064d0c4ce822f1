"""Does the gate's time depend on how the covariance allocation happens to land?  Creates the bench's 512-filter context several
times in one process with perturbing allocations in between and prints the gate's time for each (INGVIO_P_PAD=0: power-of-two stride
between the filters' covariances, the layout of rounds 1-3 until this test; default: padded stride)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import bench
from ingvio_amd import capi, synth
B, F, C = 512, 150, 11
keep = []
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    if trial:      # perturb the allocator: an extra context of another size stays alive
        keep.append(capi.Context(batch=37 + 11 * trial, n_max=96 + 16 * trial, c_max=11, f_max=64, m_max=32))
    ctx = capi.Context(batch=B, n_max=256, c_max=C, f_max=F, m_max=64)
    filters, steps, frames, infos = bench.build_batch(ctx, B, 0, F, C, 6, 52)
    ctx.snapshot(); pr = synth.PARAMS
    ctx.frame_stage(0, steps, frames, filters[0].sigma(), filters[0].enable_gnss, pr["sigma_cb"], pr["sigma_rw"], max_accept=0, compress_rule=1)
    for _ in range(3): ctx.frame_run(restore_prior=True)
    ctx.profile_select(None); ctx.profile_reset(); ctx.profile_enable(True)
    for _ in range(5): ctx.frame_run(restore_prior=True)
    ctx.sync(); ctx.profile_enable(False)
    prof = ctx.profile_get()
    ms = lambda k: prof[k][0] / max(prof[k][1], 1) if k in prof else float("nan")
    print("trial %d  gate %.4f ms  gram %.4f  solve %.4f  apply %.4f" % (trial, ms("gate"), ms("gram"), ms("solve"), ms("apply")), flush=True)
    ctx.close()
