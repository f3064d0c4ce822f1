"""ctypes binding of libingvio_host.so — the C++14 host shim (ingvio_amd/csrc/host/) that mirrors
the reference's State / StateManager / ImuPropagator / Update classes above the HIP C ABI."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libingvio_host.so")
c_dp = C.POINTER(C.c_double)
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libingvio_host.so not built; run __graft_entry__.build()")
        _lib = C.CDLL(LIB_PATH)
        _lib.ingvio_host_chi2_quantile.restype = C.c_double
    return _lib


def _d(a):
    return a.ctypes.data_as(c_dp)


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def gamma(v, m=0):
    out = np.zeros(9)
    lib().ingvio_host_gamma(_d(_f(v)), C.c_int(m), _d(out))
    return out.reshape(3, 3)


def imu_transition(R, p, v, bg, ba, gyro, acc, gravity, dt):
    """ImuPropagator::stateAndCovTransition (analytic).  Returns (R', p', v', Phi[15,15], G[15,12])."""
    R = _f(R).copy(); p = _f(p).copy(); v = _f(v).copy()
    Phi = np.zeros(225); G = np.zeros(180)
    lib().ingvio_host_imu_transition(_d(R), _d(p), _d(v), _d(_f(bg)), _d(_f(ba)), _d(_f(gyro)), _d(_f(acc)),
                                     _d(_f(gravity)), C.c_double(dt), _d(Phi), _d(G))
    return R, p, v, Phi.reshape(15, 15, order="F"), G.reshape(15, 12, order="F")


def chi2_quantile(dof, p=0.95):
    """UpdateBase's table entry: boost::math::quantile(chi_squared(dof), p) without Boost."""
    return lib().ingvio_host_chi2_quantile(C.c_int(dof), C.c_double(p))
