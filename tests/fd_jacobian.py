"""Finite-difference MSCKF measurement Jacobians under the reference's retractions — a checker that shares
NOTHING with oracle/ingvio_oracle.c or oracle/gen_golden.py (no analytic Jacobian is written down here).

What the reference defines (and what a kernel for K3/K4 has to reproduce):

* measurement of feature p_f seen from clone (R, p) with the stereo rig (RemoveLostUpdate.cpp:446-450,503):
      q = R^T (p_f - p),  q_r = R_lr q + t_lr,  h = [q_x/q_z, q_y/q_z, q_rx/q_rz, q_ry/q_rz]      (mono: first two)
* retraction of a clone pose (SE3::update, PoseState.cpp:79-88):
      R <- Gamma0(dth) R,   p <- Gamma0(dth) p + Gamma1(dth) dp
* retraction of an anchored landmark (AnchoredLandmark::update, AnchoredLandmark.cpp:227-243):
      p_f <- Gamma0(dth_anchor) p_f + Gamma1(dth_anchor) dp_f
  with Gamma_m(th) = sum_k [th]x^k / (k + m)!  (AuxGammaFunc.cpp:46-110).

The Jacobian of the stacked measurement with respect to the error state (window clones: 6 columns each, [theta, p]) and to
dp_f is obtained by central differences of h(x [+] dx).  The residual is z - h(x), so H = +dh/d(dx).

The "Selected" forms (SwMargUpdate.cpp:127,302; KeyframeUpdate.cpp:523,673) ASSIGN the anchor pose's six
columns from a block whose translation part is never filled: when the anchor is itself an observing pose, its own
translation columns are zeroed.  `selected_variant=True` applies exactly that overwrite to the finite-difference matrix.
"""
import numpy as np


def skew(v):
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


def gamma(th, m):
    """Gamma_m(th) by its defining series (40 terms: exact to rounding for |th| < 1)."""
    X = skew(np.asarray(th, dtype=np.float64))
    out = np.zeros((3, 3))
    term = np.eye(3)
    fact = 1.0
    for i in range(1, m + 1):
        fact *= i
    for k in range(40):
        out += term / fact
        term = term @ X
        fact *= (k + m + 1)
    return out


def project(R, p, pf, Rlr, tlr, stereo):
    q = R.T @ (pf - p)
    out = [q[0] / q[2], q[1] / q[2]]
    if stereo:
        qr = Rlr @ q + tlr
        out += [qr[0] / qr[2], qr[1] / qr[2]]
    return np.array(out)


def stacked_measurement(clone_R, clone_p, pf, slots, Rlr, tlr, stereo):
    return np.concatenate([project(clone_R[s], clone_p[s], pf, Rlr, tlr, stereo) for s in slots])


def retract(clone_R, clone_p, pf, anchor, dx, dpf):
    """x [+] dx for the window (dx [C, 6] = [dtheta, dp] per clone) and the landmark anchored at slot `anchor`."""
    Rn, pn = [], []
    for c in range(len(clone_R)):
        if not dx[c].any():                       # Gamma_0(0) = Gamma_1(0) = I: untouched clone (keeps the differences cheap)
            Rn.append(clone_R[c]); pn.append(clone_p[c])
            continue
        G0 = gamma(dx[c, :3], 0)
        Rn.append(G0 @ clone_R[c])
        pn.append(G0 @ clone_p[c] + gamma(dx[c, :3], 1) @ dx[c, 3:])
    if not dx[anchor, :3].any():
        return np.stack(Rn), np.stack(pn), pf + dpf
    G0a = gamma(dx[anchor, :3], 0)
    pfn = G0a @ pf + gamma(dx[anchor, :3], 1) @ dpf
    return np.stack(Rn), np.stack(pn), pfn


def _d5(f, h):
    """five-point central difference: truncation O(h^4), so h ~ 1e-3 leaves ~1e-12 instead of the 1e-8 of the 3-point rule"""
    return (-f(2 * h) + 8 * f(h) - 8 * f(-h) + f(-2 * h)) / (12 * h)


def feature_fd(frame, j, selected_variant=False, h=1e-3):
    """Returns (Hx [rows, 6C], Hf [rows, 3], r [rows], slots) for feature j of a frame dict (ingvio_amd.synth layout)."""
    cR = np.asarray(frame["clone_R"], dtype=np.float64).reshape(-1, 3, 3)
    cp = np.asarray(frame["clone_p"], dtype=np.float64).reshape(-1, 3)
    C = cR.shape[0]
    pf = np.asarray(frame["pf"], dtype=np.float64)[j]
    a = int(frame["anchor"][j])
    mask = int(frame["obs_mask"][j])
    stereo = bool(frame.get("stereo", 1))
    Rlr = np.asarray(frame["R_cl2cr"], dtype=np.float64).reshape(3, 3)
    tlr = np.asarray(frame["t_cl2cr"], dtype=np.float64).reshape(3)
    slots = [s for s in range(C) if (mask >> s) & 1]
    rpo = 4 if stereo else 2
    z = np.concatenate([np.asarray(frame["uv"], dtype=np.float64)[j, s, :rpo] for s in slots]) if slots else np.zeros(0)

    def hfun(dx, dpf):
        Rn, pn, pfn = retract(cR, cp, pf, a, dx, dpf)
        return stacked_measurement(Rn, pn, pfn, slots, Rlr, tlr, stereo)

    rows = rpo * len(slots)
    Hx = np.zeros((rows, 6 * C))
    Hf = np.zeros((rows, 3))
    zero6, zero3 = np.zeros((C, 6)), np.zeros(3)
    for c in range(C):
        for k in range(6):
            d = zero6.copy(); d[c, k] = 1.0
            Hx[:, 6 * c + k] = _d5(lambda t: hfun(t * d, zero3), h)
    for k in range(3):
        d = zero3.copy(); d[k] = 1.0
        Hf[:, k] = _d5(lambda t: hfun(zero6, t * d), h)
    r = z - hfun(zero6, zero3)
    if selected_variant and a in slots:
        Hx[:, 6 * a + 3:6 * a + 6] = 0.0          # Q10: the anchor's six columns are overwritten by [theta part | 0]
    return Hx, Hf, r, slots


def nullspace_projector(Hf):
    """I - Hf (Hf^T Hf)^-1 Hf^T = V V^T for ANY orthonormal basis V of the left nullspace (JacobiSVD's included)."""
    Q, _ = np.linalg.qr(Hf)
    return np.eye(Hf.shape[0]) - Q @ Q.T


def feature_info_fd(frame, j, selected_variant=False):
    """(A_j, b_j, H_j, r_j): A_j = H_j^T H_j [6C, 6C], b_j = H_j^T r_j, and an explicit projected pair (SVD basis)."""
    Hx, Hf, r, slots = feature_fd(frame, j, selected_variant)
    U, _, _ = np.linalg.svd(Hf, full_matrices=True)
    V = U[:, 3:]
    Hj, rj = V.T @ Hx, V.T @ r
    return Hj.T @ Hj, Hj.T @ rj, Hj, rj


def gate_gamma(P, frame, Hj, rj):
    """gamma = r_j^T (H_j Pcc H_j^T + sigma^2 I)^-1 r_j on the prior P (Update.cpp:36-56)."""
    cols = np.concatenate([np.arange(i, i + 6) for i in frame["clone_idx"]])
    S = Hj @ P[np.ix_(cols, cols)] @ Hj.T + frame["noise"] ** 2 * np.eye(Hj.shape[0])
    return float(rj @ np.linalg.solve(S, rj))


def ekf_update_dense(P, frame, H, r):
    """Plain Kalman update with the stacked (uncompressed) H on the window columns: K = P Hc^T S^-1,
    P <- P - K Hc P, symmetrised (StateManager.cpp:359-411).  Returns (P_post, dx)."""
    cols = np.concatenate([np.arange(i, i + 6) for i in frame["clone_idx"]])
    PHt = P[:, cols] @ H.T
    S = H @ P[np.ix_(cols, cols)] @ H.T + frame["noise"] ** 2 * np.eye(H.shape[0])
    K = np.linalg.solve(S, PHt.T).T
    Pn = P - K @ PHt.T
    return 0.5 * (Pn + Pn.T), K @ r
