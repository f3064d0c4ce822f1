"""The C++ host shim (ingvio_amd/csrc/host): CPU tests for the pure-host pieces, and the GPU run of
tests/cpp/test_host_shim.cpp (the reference's gtests re-stated against shim + HIP backend)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import load_golden, ROOT


def test_host_gamma_and_imu_transition_match_golden():
    """AuxGammaFunc + ImuPropagator::stateAndCovTransition (analytic) of the shim vs golden vectors."""
    from ingvio_amd import host
    z = load_golden("closed_forms")
    for v, g in zip(z["vs"], z["gamma"]):
        for m in range(4):
            assert np.abs(host.gamma(v, m) - g[m]).max() < 1e-14
    for i in range(len(z["tr_dt"])):
        R, p, v, Phi, G = host.imu_transition(z["tr_R"][i], z["tr_p"][i], z["tr_v"][i], z["tr_bg"][i], z["tr_ba"][i],
                                              z["tr_gyro"][i], z["tr_acc"][i], z["tr_g"][i], float(z["tr_dt"][i]))
        assert np.abs(R - z["tr_Rn"][i]).max() < 1e-13 and np.abs(p - z["tr_pn"][i]).max() < 1e-12
        assert np.abs(v - z["tr_vn"][i]).max() < 1e-12
        assert np.abs(Phi - z["tr_Phi"][i]).max() < 1e-11 and np.abs(G - z["tr_G"][i]).max() < 1e-13


def test_host_imu_transition_rk4_branch_matches_golden():
    """stateAndCovTransition(isAnalytic=false) (ImuPropagator.cpp:163-229): the shim's closed-form Phi and matrix mid-point
    rotations vs the quaternion / dense-Taylor restatement of oracle/gen_golden.py (rk4_transition.npz)."""
    from ingvio_amd import host
    z = load_golden("rk4_transition")
    for i in range(len(z["dt"])):
        R, p, v, Phi, G = host.imu_transition(z["R"][i], z["p"][i], z["v"][i], z["bg"][i], z["ba"][i], z["gyro"][i],
                                              z["acc"][i], z["g"][i], float(z["dt"][i]), analytic=False)
        assert np.abs(R - z["Rn"][i]).max() < 1e-13 and np.abs(p - z["pn"][i]).max() < 1e-12
        assert np.abs(v - z["vn"][i]).max() < 1e-12
        assert np.abs(Phi - z["Phi"][i]).max() < 1e-12 and np.abs(G - z["G"][i]).max() < 1e-13
        # and it is a different integrator than the analytic one (second-order agreement only)
        Ra, pa, va, Phia, _ = host.imu_transition(z["R"][i], z["p"][i], z["v"][i], z["bg"][i], z["ba"][i], z["gyro"][i],
                                                  z["acc"][i], z["g"][i], float(z["dt"][i]))
        assert np.abs(R - Ra).max() < 1e-13 and np.abs(p - pa).max() < 50 * z["dt"][i] ** 3 + 1e-14
    # TestPropagator.cpp:159-182 (oneStepProp): dt = 1, 0.1, ... 1e-4 — both distances shrink with every step
    e_state, e_phi = np.inf, np.inf
    for k in range(5):
        a = [z[n][0] for n in ("R", "p", "v", "bg", "ba", "gyro", "acc", "g")]
        Ra, pa, va, Phia, _ = host.imu_transition(*a, 10.0 ** -k)
        Rr, pr, vr, Phir, _ = host.imu_transition(*a, 10.0 ** -k, analytic=False)
        es = np.sqrt(((Ra - Rr) ** 2).sum() + ((pa - pr) ** 2).sum() + ((va - vr) ** 2).sum())
        ep = np.linalg.norm(Phia - Phir)
        assert es < e_state and ep < e_phi
        e_state, e_phi = es, ep


def test_host_chi2_quantile_matches_boost_values():
    """UpdateBase::setChiSquaredTable (Update.cpp:27-34) without Boost: equals scipy/Boost quantiles."""
    from ingvio_amd import host
    z = load_golden("closed_forms")
    got = np.array([host.chi2_quantile(k, 0.95) for k in range(1, 201)])
    assert np.abs(got - z["chi2_095"]).max() < 1e-9


def test_gnss_rows_yaw_offset_column_is_the_derivative_of_the_prediction():
    """is_adjust_yof = 1 (GnssUpdate.cpp:164-167, 239-242; GnssManager::dotRw2enu, GnssManager.cpp:101-113): column 9 of a
    pseudo-range row is d(range)/d(yaw offset) with the receiver at Renu2ecef Rz(yo) p + anchor, of a Doppler row the same with
    v — checked by central differences of the geometric prediction, no closed form of the derivative in the test.  Without the
    flag the column stays zero (every shipped config)."""
    from ingvio_amd import host
    rng = np.random.default_rng(12)
    ns = 6
    yo = 0.7
    Q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    R_e2e = Q * np.sign(np.linalg.det(Q))
    Rz = lambda a: np.array([[np.cos(a), -np.sin(a), 0.0], [np.sin(a), np.cos(a), 0.0], [0.0, 0.0, 1.0]])
    anchor = np.array([-2.2e6, 5.0e6, 3.2e6])
    p_w = np.array([12.0, -7.0, 1.5]); v_w = np.array([1.2, 0.4, -0.1])
    sats = anchor + 2.0e7 * rng.standard_normal((ns, 3))
    sat_v = 3.0e3 * rng.standard_normal((ns, 3))
    rcv = lambda a: R_e2e @ Rz(a) @ p_w + anchor
    los = np.array([(s - rcv(yo)) / np.linalg.norm(s - rcv(yo)) for s in sats])
    g = dict(los=los, sys=np.array([0, 0, 2, 3, 3, 0]), res_pos=rng.standard_normal(ns), res_vel=rng.standard_normal(ns),
             sin_el=np.full(ns, 0.7), ura=np.full(ns, 2.0), psr_std=np.ones(ns), dopp_std_mps=np.full(ns, 0.1),
             R_w2ecef=R_e2e @ Rz(yo), p_w=p_w, v_w=v_w, idx_se23=0, idx_yof=21, idx_cb=[22, -1, 23, 24], idx_fs=25)
    vidx, vsize, H0, res0, Rd0 = host.gnss_rows(g)
    assert not H0[:, 9].any()
    g.update(adjust_yof=True, R_enu2ecef=R_e2e, yaw_offset=yo)
    vidx1, vsize1, H, res, Rd = host.gnss_rows(g)
    assert np.array_equal(vidx, vidx1) and np.array_equal(res, res0) and np.array_equal(Rd, Rd0)
    mask = np.ones(H.shape[1], dtype=bool); mask[9] = False
    assert np.array_equal(H[:, mask], H0[:, mask])                          # only the yaw-offset column changes
    h = 1e-4
    LD = np.longdouble                                                      # a 2e7 m range differenced over 1e-4 rad needs more than 53 bits
    rng_of = lambda a, i: np.sqrt((((sats[i] - anchor).astype(LD) - (R_e2e @ Rz(a) @ p_w).astype(LD)) ** 2).sum())
    # range rate with the line of sight held at the linearisation point (dopp_res differentiates the velocity term only)
    rr_of = lambda a, i: los[i] @ (sat_v[i] - R_e2e @ Rz(a) @ v_w)
    for i in range(ns):
        assert abs(H[i, 9] - float(rng_of(yo + h, i) - rng_of(yo - h, i)) / (2 * h)) < 1e-6 * np.linalg.norm(p_w)
        assert abs(H[ns + i, 9] - (rr_of(yo + h, i) - rr_of(yo - h, i)) / (2 * h)) < 1e-8 * np.linalg.norm(v_w)


def test_c_abi_exports_every_declared_symbol():
    """The library loads without a GPU and exports every function include/ingvio_hip.h declares."""
    import re
    from ingvio_amd import capi
    hdr = open(os.path.join(ROOT, "include", "ingvio_hip.h")).read()
    declared = set(re.findall(r"\b(ingvio_[a-z0-9_]+)\s*\(", hdr)) - {"ingvio_ctx_desc"}
    L = capi.lib()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)


def test_context_creation_fails_loudly_without_gpu():
    from ingvio_amd import capi
    import ctypes
    n = ctypes.c_int(0)
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        has_gpu = hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0
    except OSError:
        has_gpu = False
    if has_gpu:
        pytest.skip("GPU present")
    with pytest.raises(capi.IngvioError):
        capi.Context()


@pytest.mark.gpu
def test_reference_gtests_against_shim_and_hip_backend():
    exe = os.path.join(ROOT, "ingvio_amd", "lib", "test_host_shim")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert "0 failures" in r.stdout


def test_ros1_adapter_conversions_on_mock_messages():
    """ros1/include/RosAdapter.h (the field-by-field core of ros1/src/ingvio_node.cpp) instantiated with mock structs shaped like the
    ROS messages: frames / IMU / odometry, gnss_comm's satellite numbering and L1 selection, ephemeris records, best-ephemeris choice,
    tracking counters, evaluated records -> GnssMeas.  Built without ROS by ingvio_amd/build.py."""
    exe = os.path.join(ROOT, "ingvio_amd", "lib", "test_ros_adapter")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "0 failures" in r.stdout, r.stdout[-2000:] + r.stderr[-1000:]
