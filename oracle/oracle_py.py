"""ctypes loader for oracle/liboracle.so (C restatement) and oracle/_ref/libcpi_ref.so
(the reference's own CpiV1/CpiV2 headers, compiled unchanged).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product (cpi_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class Params(C.Structure):
    _fields_ = [("sigma_w", C.c_double), ("sigma_wb", C.c_double), ("sigma_a", C.c_double),
                ("sigma_ab", C.c_double), ("grav", C.c_double * 3), ("model", C.c_int),
                ("imu_avg", C.c_int), ("state_transition_jacobians", C.c_int)]


OUT_FIELDS = [("DT", 1), ("alpha", 3), ("beta", 3), ("q", 4), ("R", 9), ("J_q", 9), ("J_a", 9),
              ("J_b", 9), ("H_a", 9), ("H_b", 9), ("O_a", 9), ("O_b", 9), ("P", 225)]
OUT_DOUBLES = sum(n for _, n in OUT_FIELDS)  # 308

FACTOR_FIELDS = [("alpha", 3), ("beta", 3), ("q_KtoK1", 4), ("ba_lin", 3), ("bg_lin", 3),
                 ("J_q", 9), ("J_beta", 9), ("J_alpha", 9), ("H_beta", 9), ("H_alpha", 9),
                 ("deltatime", 1), ("grav", 3), ("q_K_lin", 4), ("O_beta", 9), ("O_alpha", 9)]
FACTOR_DOUBLES = sum(n for _, n in FACTOR_FIELDS)  # 87

# ADIS16448 values of the reference's launch file (cpi_compare/launch/synthetic_test.launch:13-17)
DEFAULT_SIGMAS = dict(sigma_w=0.005, sigma_wb=4e-6, sigma_a=0.01, sigma_ab=2e-4)
DEFAULT_GRAV = (0.0, 0.0, 9.8)


def build(force=False):
    """(Re)build liboracle.so and, when /root/reference is present, _ref/libcpi_ref.so."""
    lib = os.path.join(_HERE, "liboracle.so")
    if force or not os.path.exists(lib) or \
            os.path.getmtime(lib) < max(os.path.getmtime(os.path.join(_HERE, f))
                                        for f in ("cpi_oracle.c", "forster_oracle.c", "cpi_oracle.h")):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    ref = os.path.join(_HERE, "_ref", "libcpi_ref.so")
    if os.path.isdir("/root/reference/cpi_compare/src/cpi") and (force or not os.path.exists(ref)):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


def make_params(model=1, imu_avg=0, stj=1, sigmas=None, grav=DEFAULT_GRAV):
    s = dict(DEFAULT_SIGMAS)
    if sigmas:
        s.update(sigmas)
    p = Params()
    p.sigma_w, p.sigma_wb, p.sigma_a, p.sigma_ab = s["sigma_w"], s["sigma_wb"], s["sigma_a"], s["sigma_ab"]
    p.grav[:] = grav
    p.model, p.imu_avg, p.state_transition_jacobians = model, imu_avg, stj
    return p


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def split_out(raw):
    """raw [W, 308] -> dict of arrays (matrices stay flat, column-major)."""
    out, o = {}, 0
    for name, n in OUT_FIELDS:
        out[name] = raw[:, o] if n == 1 else raw[:, o:o + n]
        o += n
    return out


class _Lib:
    def __init__(self, path, prefix):
        self.lib = C.CDLL(path)
        self.prefix = prefix
        self.batch = getattr(self.lib, prefix + "_batch")
        self.batch_mt = getattr(self.lib, prefix + "_batch_mt")
        for f in (self.batch, self.batch_mt):
            f.restype = None

    def run(self, prm, knots, lin, q_k_lin=None, nthreads=1, raw=None):
        """raw: optional preallocated (and already touched) [W, 308] output buffer -- timing loops reuse it so that
        page-faulting a fresh 25 MB array under hundreds of threads is not what gets measured."""
        knots = np.ascontiguousarray(knots, dtype=np.float64)
        lin = np.ascontiguousarray(lin, dtype=np.float64)
        W, n1, seven = knots.shape
        assert seven == 7 and lin.shape == (W, 6)
        if q_k_lin is not None:
            q_k_lin = np.ascontiguousarray(q_k_lin, dtype=np.float64)
            assert q_k_lin.shape == (W, 4)
        if raw is None:
            raw = np.zeros((W, OUT_DOUBLES), dtype=np.float64)
        assert raw.shape == (W, OUT_DOUBLES) and raw.dtype == np.float64 and raw.flags.c_contiguous
        args = [C.byref(prm), C.c_long(W), C.c_int(n1 - 1), _dp(knots), _dp(lin), _dp(q_k_lin), _dp(raw)]
        if nthreads > 1:
            self.batch_mt(*args, C.c_int(nthreads))
        else:
            self.batch(*args)
        return split_out(raw)


_cache = {}


def oracle():
    if "o" not in _cache:
        build()
        # CPI_ORACLE_LIB: an instrumented build of the same sources (tests/tools/sanitize.sh)
        _cache["o"] = _OracleLib(os.environ.get("CPI_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so"))
    return _cache["o"]


def reference():
    """The compiled reference (None when oracle/_ref/libcpi_ref.so is not available)."""
    if "r" not in _cache:
        path = os.path.join(_HERE, "_ref", "libcpi_ref.so")
        if not os.path.exists(path) and os.path.isdir("/root/reference/cpi_compare/src/cpi"):
            build()
        _cache["r"] = _Lib(path, "cpi_ref") if os.path.exists(path) else None
    return _cache["r"]


class _OracleLib(_Lib):
    def __init__(self, path):
        super().__init__(path, "cpi_oracle")
        for name in ("cpi_oracle_window_trace", "cpi_oracle_stream", "cpi_oracle_factor_v1", "cpi_oracle_factor_v2",
                     "cpi_oracle_predict", "cpi_oracle_retract", "cpi_oracle_local", "cpi_oracle_forster_window",
                     "cpi_oracle_navstate_retract", "cpi_oracle_navstate_update"):
            getattr(self.lib, name).restype = None

    def navstate_retract(self, state15, xi9):
        """gtsam::NavState::retract restated; state15 = [nRb row-major 9, n_t 3, n_v 3]."""
        s = np.ascontiguousarray(state15, dtype=np.float64)
        xi = np.ascontiguousarray(xi9, dtype=np.float64)
        out = np.zeros(15)
        self.lib.cpi_oracle_navstate_retract(_dp(s), _dp(xi), _dp(out))
        return out

    def navstate_update(self, state15, acc, om, dt):
        """gtsam::NavState::update restated -> (new state15, A 9x9, B 9x3, C 9x3)."""
        s = np.ascontiguousarray(state15, dtype=np.float64)
        acc = np.ascontiguousarray(acc, dtype=np.float64)
        om = np.ascontiguousarray(om, dtype=np.float64)
        out, A, B, Cm = np.zeros(15), np.zeros((9, 9)), np.zeros((9, 3)), np.zeros((9, 3))
        self.lib.cpi_oracle_navstate_update(_dp(s), _dp(acc), _dp(om), C.c_double(dt), _dp(out), _dp(A), _dp(B), _dp(Cm))
        return out, A, B, Cm

    def trace(self, prm, knots1, lin1, q1=None):
        knots1 = np.ascontiguousarray(knots1, dtype=np.float64)
        n = knots1.shape[0] - 1
        raw = np.zeros((n, OUT_DOUBLES))
        lin1 = np.ascontiguousarray(lin1, dtype=np.float64)
        q1 = None if q1 is None else np.ascontiguousarray(q1, dtype=np.float64)
        self.lib.cpi_oracle_window_trace(C.byref(prm), C.c_int(n), _dp(knots1), _dp(lin1), _dp(q1), _dp(raw))
        return split_out(raw)

    def stream(self, prm, stream_knots, update_times, lin, q_k_lin=None):
        """GraphSolver_IMU.cpp:50-69 window assembly + preintegration over one IMU stream."""
        st = np.ascontiguousarray(stream_knots, dtype=np.float64)
        ut = np.ascontiguousarray(update_times, dtype=np.float64)
        lin = np.ascontiguousarray(lin, dtype=np.float64)
        q = None if q_k_lin is None else np.ascontiguousarray(q_k_lin, dtype=np.float64)
        raw = np.zeros((ut.shape[0], OUT_DOUBLES))
        self.lib.cpi_oracle_stream(C.byref(prm), C.c_long(st.shape[0]), _dp(st), C.c_long(ut.shape[0]), _dp(ut), _dp(lin),
                                   _dp(q), _dp(raw))
        return split_out(raw)

    def factor(self, model, frec, xi, xj, want_H=True):
        """frec [F,87], xi/xj [F,16] -> err [F,15], H1 [F,225], H2 [F,225] (column-major)."""
        frec = np.ascontiguousarray(frec, dtype=np.float64)
        xi = np.ascontiguousarray(xi, dtype=np.float64)
        xj = np.ascontiguousarray(xj, dtype=np.float64)
        F = frec.shape[0]
        err = np.zeros((F, 15)); H1 = np.zeros((F, 225)); H2 = np.zeros((F, 225))
        fn = self.lib.cpi_oracle_factor_v1 if model == 1 else self.lib.cpi_oracle_factor_v2
        for k in range(F):
            fn(_dp(frec[k]), _dp(xi[k]), _dp(xj[k]), _dp(err[k]),
               _dp(H1[k]) if want_H else None, _dp(H2[k]) if want_H else None)
        return err, H1, H2

    def factor_batch(self, model, frec, xi, xj, nthreads=1, out=None):
        """Batched, multi-threaded evaluateError restatement (timing leg of bench.py).  out = (err, H1, H2) to reuse."""
        frec = np.ascontiguousarray(frec, dtype=np.float64)
        xi = np.ascontiguousarray(xi, dtype=np.float64)
        xj = np.ascontiguousarray(xj, dtype=np.float64)
        F = frec.shape[0]
        assert frec.shape[1] == FACTOR_DOUBLES
        err, H1, H2 = out if out is not None else (np.zeros((F, 15)), np.zeros((F, 225)), np.zeros((F, 225)))
        self.lib.cpi_oracle_factor_batch_mt.restype = None
        self.lib.cpi_oracle_factor_batch_mt(C.c_int(model), C.c_long(F), _dp(frec), _dp(xi), _dp(xj), _dp(err), _dp(H1), _dp(H2),
                                            C.c_int(nthreads))
        return err, H1, H2

    def predict(self, model, frec, xi):
        frec = np.ascontiguousarray(frec, dtype=np.float64)
        xi = np.ascontiguousarray(xi, dtype=np.float64)
        xj = np.zeros_like(xi)
        for k in range(frec.shape[0]):
            self.lib.cpi_oracle_predict(C.c_int(model), _dp(frec[k]), _dp(xi[k]), _dp(xj[k]))
        return xj

    def retract(self, x, xi15):
        x = np.ascontiguousarray(x, dtype=np.float64); xi15 = np.ascontiguousarray(xi15, dtype=np.float64)
        out = np.zeros(16)
        self.lib.cpi_oracle_retract(_dp(x), _dp(xi15), _dp(out))
        return out

    def local(self, x, other):
        x = np.ascontiguousarray(x, dtype=np.float64); other = np.ascontiguousarray(other, dtype=np.float64)
        out = np.zeros(15)
        self.lib.cpi_oracle_local(_dp(x), _dp(other), _dp(out))
        return out


def factor_records(out, lin, q_k_lin, grav=DEFAULT_GRAV):
    """Pack preintegration outputs into factor records [W,87] using the field->ctor mapping of
    GraphSolver_IMU.cpp:74-75,129-130 (J_b->J_beta, J_a->J_alpha, H_b->H_beta, H_a->H_alpha,
    O_b->O_beta, O_a->O_alpha)."""
    W = lin.shape[0]
    rec = np.zeros((W, FACTOR_DOUBLES))
    cols = {}
    o = 0
    for name, n in FACTOR_FIELDS:
        cols[name] = slice(o, o + n)
        o += n
    rec[:, cols["alpha"]] = out["alpha"]; rec[:, cols["beta"]] = out["beta"]
    rec[:, cols["q_KtoK1"]] = out["q"]
    rec[:, cols["ba_lin"]] = lin[:, 3:6]; rec[:, cols["bg_lin"]] = lin[:, 0:3]
    rec[:, cols["J_q"]] = out["J_q"]; rec[:, cols["J_beta"]] = out["J_b"]; rec[:, cols["J_alpha"]] = out["J_a"]
    rec[:, cols["H_beta"]] = out["H_b"]; rec[:, cols["H_alpha"]] = out["H_a"]
    rec[:, cols["deltatime"]] = out["DT"][:, None]
    rec[:, cols["grav"]] = np.asarray(grav)[None, :]
    if q_k_lin is not None:
        rec[:, cols["q_K_lin"]] = q_k_lin
    else:
        rec[:, cols["q_K_lin"]] = np.array([0, 0, 0, 1.0])
    if "O_b" in out:
        rec[:, cols["O_beta"]] = out["O_b"]; rec[:, cols["O_alpha"]] = out["O_a"]
    return rec
