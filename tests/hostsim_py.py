"""Loader for tests/hostsim/libhostsim.so: the HIP kernels' per-lane arithmetic (cpi_math.hpp)
compiled for the host and driven lane-by-lane.  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "hostsim", "hostsim.cpp")
_LIB = os.path.join(_HERE, "hostsim", "libhostsim.so")
_HDR = os.path.join(os.path.dirname(_HERE), "cpi_amd", "csrc", "cpi_math.hpp")
_cache = {}


def lib():
    if "l" not in _cache:
        if (not os.path.exists(_LIB)) or os.path.getmtime(_LIB) < max(os.path.getmtime(_SRC), os.path.getmtime(_HDR)):
            subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-Wno-unknown-pragmas",
                                   "-ffp-contract=off", "-o", _LIB, _SRC])
        _cache["l"] = C.CDLL(os.environ.get("CPI_HOSTSIM_LIB") or _LIB)   # env: an instrumented build (tests/tools/sanitize.sh)
    return _cache["l"]


def dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


GRAV = np.array([0.0, 0.0, 9.8])
SIG = np.array([0.005, 4e-6, 0.01, 2e-4])


def mean(model, jac, avg, L, kn, lin, q):
    W, n1, _ = kn.shape
    raw = np.zeros((W, 308))
    lib().hs_mean(model, jac, avg, L, C.c_long(W), n1 - 1, dp(kn), dp(lin), dp(q), dp(GRAV), dp(raw))
    return raw


def cov(model, avg, kn, lin, q):
    W, n1, _ = kn.shape
    raw = np.zeros((W, 308))
    lib().hs_cov(model, avg, C.c_long(W), n1 - 1, dp(kn), dp(lin), dp(q), dp(SIG), dp(GRAV), dp(raw))
    return raw


def factor(model, rec, xi, xj):
    F = rec.shape[0]
    err = np.zeros((F, 15)); H1 = np.zeros((F, 225)); H2 = np.zeros((F, 225))
    lib().hs_factor(model, C.c_long(F), dp(rec), dp(xi), dp(xj), dp(err), dp(H1), dp(H2))
    return err, H1, H2


def hessian(model, rec, xi, xj, R):
    """[F, 496] packed upper triangle of [A1 A2 b]^T [A1 A2 b] by the kernel's 3x3 block algebra; R [F, 225] column-major."""
    F = rec.shape[0]
    out = np.full((F, 496), np.nan)
    lib().hs_hessian(model, C.c_long(F), dp(rec), dp(xi), dp(xj), dp(np.ascontiguousarray(R)), dp(out))
    return out


def predict(model, rec, xi):
    xj = np.zeros_like(xi)
    lib().hs_predict(model, C.c_long(rec.shape[0]), dp(rec), dp(xi), dp(xj))
    return xj


def forster(kn, lin):
    W, n1, _ = kn.shape
    raw = np.zeros((W, 308))
    lib().hs_forster(C.c_long(W), n1 - 1, dp(kn), dp(lin), dp(SIG), dp(raw))
    return raw
