"""Shader-clock phases of k_lm_front for the bench's landmark workload (needs an INGVIO_DBG_STAMPS build, INGVIO_DBG_TU=l)."""
import sys
sys.path.insert(0, "/root/repo")
import bench
from ingvio_amd import capi, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
F, C, n_gnss, n_lm = 150, 11, 6, 52
ctx = capi.Context(batch=B, n_max=256, c_max=C, f_max=F, m_max=64)
filters, steps, frames, infos = bench.build_batch(ctx, B, 0, F, C, n_gnss, n_lm, lm_sigma=0.05)
ctx.snapshot(); pr = synth.PARAMS
ctx.frame_stage(0, steps, frames, filters[0].sigma(), filters[0].enable_gnss, pr["sigma_cb"], pr["sigma_rw"], max_accept=0, compress_rule=1)
lms = [synth.make_landmarks(infos[b]["rng"], filters[b], frames[b], n_lm) for b in range(B)]
Rlr, tlr = synth.t_cl2cr()
ctx.landmark_stage(0, lms, True, pr["visual_noise"], 9.487729036781154, Rlr, tlr, in_frame=True)
for _ in range(3):
    ctx.frame_run(restore_prior=True)
d = ctx.debug_read(8)
print("B=%d k_lm_front [rows to LDS, A(0), B(0), chunks 1.., gate + map]" % B, [d[1] - d[0], d[2] - d[1], d[3] - d[2], d[4] - d[3], d[5] - d[4]], "total", d[5] - d[0])
import os
if os.environ.get("INGVIO_DBG_TU") == "m":
    print("B=%d k_lm_factor panel 1 [(A) diag tile, wait, (B) row panel, (C) trailing]" % B, [d[2] - d[1], d[3] - d[2], d[4] - d[3], d[5] - d[4]], "all panels", d[6] - d[0])
