#!/usr/bin/env python3
"""GPU bring-up diagnostic: runs every C-ABI entry point against the oracle / golden vectors and
prints the errors (no asserts) so one gpurun call gives the whole picture.  Not a pytest file."""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import frame_from_golden, load_golden, rel_err  # noqa: E402
from ingvio_amd import capi, synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def section(name, fn):
    t0 = time.time()
    try:
        fn()
        print("[%s] done in %.2fs" % (name, time.time() - t0), flush=True)
    except Exception:
        print("[%s] EXCEPTION" % name)
        traceback.print_exc()
        sys.stdout.flush()


def main():
    ctx = capi.Context(batch=4, n_max=256, c_max=11, f_max=160, m_max=64)
    print("ctx ok ldp", ctx.ldp, flush=True)

    def t_prop():
        z = load_golden("propagate")
        for c in ("c0", "c1", "c2"):
            ctx.cov_set(1, z[c + "_P"])
            ctx.propagate(1, z[c + "_Phi"], z[c + "_G"], float(z[c + "_dt"]), z[c + "_sigma"], 1, z[c + "_gnss_idx"],
                          float(z[c + "_scb"]), float(z[c + "_srw"]))
            P = ctx.cov_get(1)
            print("  propagate", c, "rel", rel_err(P, z[c + "_Pn"]), "asym", np.abs(P - P.T).max())
        ctx.cov_set(0, z["d_P"])
        ctx.propagate(0, z["d_Phi"], z["d_G"], float(z["d_dt"]), z["d_sigma"])
        print("  propagate d rel", rel_err(ctx.cov_get(0), z["d_Pn"]))
        # fused k steps vs oracle sequence, with gnss
        rng = np.random.default_rng(1)
        n = 60
        A = rng.standard_normal((n, n)); P0 = A @ A.T / n + 0.05 * np.eye(n)
        k = 7
        Phis = np.eye(15) + 0.02 * rng.standard_normal((k, 15, 15)); Gs = rng.uniform(-1, 1, (k, 15, 12)); dts = rng.uniform(0.004, 0.006, k)
        gi = [21, 40, -1, 22, 30]
        sig = [0.004, 0.08, 0.0002, 0.008]
        oc = orc.Cov(P0)
        for s in range(k):
            oc.propagate(Phis[s], Gs[s], dts[s], sig, 1, gi, 0.2, 0.2)
        ctx.cov_set(2, P0)
        ctx.propagate(2, Phis, Gs, dts, sig, 1, gi, 0.2, 0.2, fused=True)
        print("  fused k=7 gnss rel", rel_err(ctx.cov_get(2), oc.P))

    def t_struct():
        z = load_golden("augment")
        ctx.cov_set(0, z["P"])
        idx = ctx.augment(0, z["R_i2w"])
        P = ctx.cov_get(0)
        print("  augment idx", idx, "n", ctx.n(0), "rel", rel_err(P, z["Pn"]), "asym", np.abs(P - P.T).max())
        z = load_golden("ekf")
        ctx.cov_set(0, z["P"])
        print("  marginal eq", np.array_equal(ctx.marginal(0, z["vidx"], z["vsize"]), z["P_small"]))
        ctx.marginalize(0, [int(z["marg_idx"])], int(z["marg_size"]))
        print("  marginalize eq", np.array_equal(ctx.cov_get(0), z["P_marg"]), "n", ctx.n(0))
        i2 = ctx.append_independent(0, 2.5 * np.eye(3))
        P = ctx.cov_get(0)
        print("  append idx", i2, "n", ctx.n(0), "blk ok", np.allclose(P[-3:, -3:], 2.5 * np.eye(3)), "zeros", np.abs(P[:-3, -3:]).max())

    def t_ekf():
        z = load_golden("ekf")
        for name, R, Pn, dxg in (("scalar", 0.5, z["Pn"], z["dx"]), ("diag", z["Rd"], z["Pn_d"], z["dx_d"]), ("full", z["Rf"], z["Pn_f"], z["dx_f"])):
            ctx.cov_set(3, z["P"])
            dx, rc = ctx.ekf_update(3, z["vidx"], z["vsize"], z["H"], z["res"], R)
            P = ctx.cov_get(3)
            print("  ekf", name, "rc", rc, "P rel", rel_err(P, Pn), "dx rel", rel_err(dx, dxg), "asym", np.abs(P - P.T).max())
        ctx.cov_set(3, z["P"])
        g = ctx.chi2_gamma(3, z["vidx"], z["vsize"], z["H"], z["res"], 0.5)
        print("  gamma", g, "golden", float(z["gamma"]))
        z = load_golden("gnss")
        ctx.cov_set(3, z["P"])
        dx, rc = ctx.ekf_update(3, z["vidx"], z["vsize"], z["H"], z["res"], z["Rdiag"])
        print("  gnss update rc", rc, "P rel", rel_err(ctx.cov_get(3), z["Pn"]), "dx rel", rel_err(dx, z["dx"]))

    def t_msckf_small():
        z = load_golden("msckf_small")
        for name in ["stereo_ragged", "mono_ragged", "stereo_cap", "selected_q10", "keyframe_like"]:
            fr = frame_from_golden(z, name + "_")
            kw = dict(zip(("max_accept", "compress_rule", "selected_variant"), [int(x) for x in z[name + "_kw"]]))
            ctx.cov_set(0, z[name + "_P"])
            dx, acc, gam, rows = ctx.msckf_update(0, fr, **kw)
            F = len(fr["dof"]); n = z[name + "_P"].shape[0]
            P = ctx.cov_get(0)
            ev = ~np.isnan(z[name + "_gamma"])
            print("  %s rows %d acc_eq %s gamma rel %.2e P rel %.2e dx rel %.2e asym %.1e" % (
                name, rows[0], np.array_equal(acc[0, :F], z[name + "_acc"]),
                np.abs(gam[0, :F][ev] / z[name + "_gamma"][ev] - 1).max(), rel_err(P, z[name + "_Pn"]),
                rel_err(dx[0, :n], z[name + "_dx"]), np.abs(P - P.T).max()))

    def t_config2():
        z = load_golden("config2_n87")
        fr = frame_from_golden(z, "fr_")
        ctx.cov_set(1, z["P_pre_update"])
        t0 = time.time()
        dx, acc, gam, rows = ctx.msckf_update(1, fr)
        t1 = time.time()
        P = ctx.cov_get(1)
        print("  config2 N=87 rows", rows[0], "acc_eq", np.array_equal(acc[0, :150], z["acc"]), "gamma rel",
              np.abs(gam[0, :150] / z["gamma"] - 1).max(), "P rel", rel_err(P, z["Pn"]), "dx rel", rel_err(dx[0, :87], z["dx"]),
              "wall %.1f ms" % ((t1 - t0) * 1e3))
        ctx.cov_set(1, z["P_pre_update"])
        dx, acc, gam, rows = ctx.msckf_update(1, fr, max_accept=20, compress_rule=0)
        print("  as_written cap20 acc_eq", np.array_equal(acc[0, :150], z["acc_aw"]), "P rel", rel_err(ctx.cov_get(1), z["Pn_aw"]))
        # whole frame via stage/run
        step = dict(Phi=list(z["step_Phi"]), G=list(z["step_G"]), dt=list(z["step_dt"]), sigma=list(z["step_sigma"]),
                    R_i2w=z["step_R_i2w"], marg_idx=int(z["step_marg_idx"]))
        for b in range(4):
            ctx.cov_set(b, z["P_prior"])
        ctx.snapshot()
        ctx.frame_stage(0, [step] * 4, [fr] * 4, step["sigma"])
        ctx.frame_run(restore_prior=True)
        ctx.frame_run(restore_prior=True)
        dx, acc, rows = ctx.frame_fetch()
        for b in (0, 3):
            print("  frame b%d n %d rows %d P_final rel %.2e dx rel %.2e" % (b, ctx.n(b), rows[b], rel_err(ctx.cov_get(b), z["P_final"]),
                                                                             rel_err(dx[b, :87], z["dx"])))

    def t_qr():
        rng = np.random.default_rng(3)
        A = rng.standard_normal((500, 66)); b = rng.standard_normal(500)
        Ht, rt = ctx.qr_compress(A, b)
        print("  qr HtH rel", rel_err(Ht.T @ Ht, A.T @ A), "Htr rel", rel_err(Ht.T @ rt, A.T @ b), "lower", np.abs(np.tril(Ht, -1)).max())

    def t_full():
        ctx2 = capi.Context(batch=8, n_max=256, c_max=11, f_max=160, m_max=64)
        cases = []
        for b in range(8):
            flt, step, frame, info = synth.build_case(lambda P: orc.Cov(P, ld=256), orc.imu_transition, seed=b)
            cases.append((flt, step, frame, info))
            ctx2.cov_set(b, flt.cov.P)
        ctx2.snapshot()
        ctx2.frame_stage(0, [c[1] for c in cases], [c[2] for c in cases], cases[0][1]["sigma"], 1, 0.2, 0.2)
        ctx2.profile_enable(True)
        for it in range(3):
            ctx2.frame_run(restore_prior=True)
        ctx2.sync()
        t0 = time.time()
        for it in range(5):
            ctx2.frame_run(restore_prior=True)
        ctx2.sync()
        print("  8 filters N=249: %.3f ms / step (wall, profiled)" % ((time.time() - t0) / 5 * 1e3))
        print("  profile:", {k: "%.3f ms/%d" % v for k, v in ctx2.profile_get().items()})
        dx, acc, rows = ctx2.frame_fetch()
        for b in range(8):
            flt, step, frame, info = cases[b]
            dxo, acco, gamo, m = orc.frame_update(flt.cov, step, frame, max_accept=0, compress_rule=1)
            if b in (0, 5, 7):
                print("  b%d n %d rows %d acc_eq %s P rel %.2e dx rel %.2e" % (b, ctx2.n(b), rows[b], np.array_equal(acc[b, :150], acco),
                                                                              rel_err(ctx2.cov_get(b), flt.cov.P), rel_err(dx[b, :249], dxo)))
        ctx2.close()

    section("propagate", t_prop)
    section("struct", t_struct)
    section("ekf", t_ekf)
    section("msckf_small", t_msckf_small)
    section("config2", t_config2)
    section("qr", t_qr)
    section("full249", t_full)


if __name__ == "__main__":
    main()
